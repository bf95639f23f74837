"""Synthetic inputs for the LP-relaxation hot path.

PRNG: splitmix64(seed) -> u = (x >> 11) * 2**-53, reproducible in C/Go/Python.
Dense LP(m, seed): G[i][j] = u (row-major fill order), h[i] = 1 + u, c[j] = -u, then the
standard form A = [G | I_m], c = [c, 0], b = h is assembled exactly like
convertToEqualities (/root/reference/subproblem.go:81-139) with A = nil.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64_uniform(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """`count` uniforms in [0,1) from the splitmix64 stream of `seed`, starting at draw `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + count + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def dense_lp_inequality_form(m: int, seed: int, nv: int | None = None):
    """(c, G, h) with nv structural variables and m rows: minimise c.x s.t. G x <= h, x >= 0."""
    nv = m if nv is None else nv
    u = splitmix64_uniform(seed, m * nv + m + nv)
    G = u[: m * nv].reshape(m, nv).copy()
    h = 1.0 + u[m * nv : m * nv + m]
    c = -u[m * nv + m :]
    return c, G, h


def dense_lp_standard_form(m: int, seed: int, nv: int | None = None):
    """(c, A, b): A = [G | I_m] (m x (nv+m)), c = [c, 0], b = h — what lp.Simplex receives."""
    c, G, h = dense_lp_inequality_form(m, seed, nv)
    nv = G.shape[1]
    A = np.zeros((m, nv + m))
    A[:, :nv] = G
    A[np.arange(m), nv + np.arange(m)] = 1.0
    return np.concatenate([c, np.zeros(m)]), A, h.copy()


def integrality_mask(nv: int, m: int) -> np.ndarray:
    """C3/C5: structural variables with j % 4 == 0 are integer (25 %), slacks are not."""
    mask = np.zeros(nv + m, dtype=bool)
    mask[:nv:4] = True
    return mask


# BASELINE.json configs: name -> (m, seed)
CONFIGS = {"C2": (1024, 1), "M": (2048, 2), "C4": (4096, 4), "C3": (512, 3), "C5": (512, 3)}


def frontier_children(root_x, integrality, nvars: int = 8):
    """C5: the `nvars` highest-index integer-constrained variables with a fractional root value;
    every sign pattern of {x_j <= floor(x_j*), -x_j <= -(floor(x_j*)+1)} is one independent child, described as
    [(var, sign, rhs), ...] exactly like the bnbConstraints of /root/reference/subproblem.go:193-259."""
    import math
    picks = [j for j in range(len(integrality) - 1, -1, -1)
             if integrality[j] and root_x[j] != math.floor(root_x[j])][:nvars]
    if len(picks) < nvars:   # a wider frontier than the root has fractional variables: integer-valued ones behind them (x_j <= v | x_j >= v + 1)
        picks += [j for j in range(len(integrality) - 1, -1, -1) if integrality[j] and j not in picks][: nvars - len(picks)]
    children = []
    for pattern in range(1 << len(picks)):
        cons = []
        for k, j in enumerate(picks):
            fl = float(math.floor(root_x[j]))
            if (pattern >> k) & 1:
                cons.append((j, -1, -(fl + 1.0)))
            else:
                cons.append((j, 1, fl))
        children.append(cons)
    return children


def degenerate_integer_milp(seed):
    """Small integer-data MILP in inequality form (x = 0 feasible, every column bounded): duplicate rows, integer right-hand
    sides and the reference's habit of branching on the same variable again and again stack identical rows — vertices
    where several basic variables sit at level zero, the place where the rounding noise of the reference's fresh x_B decides."""
    rng = np.random.default_rng(7000 + seed)
    nv, m = int(rng.integers(3, 8)), int(rng.integers(2, 6))
    G = rng.integers(0, 4, (m, nv)).astype(float)
    G[int(rng.integers(0, m))] = np.maximum(G[int(rng.integers(0, m))], 1.0)
    G[0] = np.maximum(G[0], 1.0)                      # every variable bounded by row 0
    h = rng.integers(1, 9, m).astype(float)
    if m >= 3 and rng.random() < 0.6:
        G[m - 1], h[m - 1] = G[1], h[1]                # a duplicate row
    c = -rng.integers(0, 5, nv).astype(float)
    integ = [bool(v) for v in rng.integers(0, 2, nv)]
    integ[int(rng.integers(0, nv))] = True
    return c, G, h, integ
