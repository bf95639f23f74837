"""Multi-GPU sharding of one branch-and-bound wave.

The frontier of an enumeration tree is a set of independent LP relaxations that share the root data
(/root/reference/subproblem.go:20-29); the reference spreads them over goroutine workers
(/root/reference/tree.go:98-100,196-205).  Here: one process per GPU, the children of the wave are dealt round-robin over a fixed shuffle
(equal batch sizes; results do not depend on the GPU count), every rank
solves its shard on its own GPU through gomilp_frontier_solve, and the only exchange is the incumbent bound:
one all-reduce(min) per wave (RCCL over xGMI on the GPU box, gloo in the CPU tests).  The bound is used exactly
like /root/reference/tree.go:228-230 uses it — after the solves, for pruning — never inside a solve.
"""
from __future__ import annotations

import functools
import math
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

BIG_INDEX = 1 << 52   # == GOMILP_NO_INCUMBENT (include/gomilp_lp.h): exact in a double


def _mix(i: int) -> int:
    """Fixed 32-bit mixing of the child index (deterministic, same on every rank)."""
    x = (i * 2654435761) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 2246822519) & 0xFFFFFFFF
    return x ^ (x >> 13)


@functools.lru_cache(maxsize=64)
def _shard_indices_cached(count: int, rank: int, world: int) -> tuple:
    order = sorted(range(count), key=lambda i: (_mix(i), i))
    return tuple(sorted(order[rank::world]))


def shard_indices(count: int, rank: int, world: int) -> List[int]:
    """Children owned by `rank`: round-robin over a fixed pseudo-random order of the wave.

    Plain `i % world` piles up the expensive children of a sign-pattern frontier (the few feasible ones are the
    patterns 0, 1, 2, 4, 8, ... — mostly multiples of the GPU count) on rank 0; a fixed shuffle spreads them.  The
    per-child results do not depend on the assignment, and incumbent ties are resolved by the child index."""
    return list(_shard_indices_cached(count, rank, world))


def is_all_integer(v: float) -> bool:
    """tree.go:290-297: k == math.Trunc(k) (true for +-Inf, false for NaN)."""
    if math.isnan(v):
        return False
    if math.isinf(v):
        return True
    return v == math.trunc(v)


def feasible_for_ip(integrality: Sequence[bool], x: Sequence[float]) -> bool:
    """tree.go:276-288 (vectorised: k == math.Trunc(k) for every integer-constrained coordinate; +-Inf passes, NaN does not)."""
    mask = np.asarray(integrality, dtype=bool)
    v = np.asarray(x, dtype=np.float64)[: len(mask)][mask]
    if v.size == 0:
        return True
    with np.errstate(invalid="ignore"):
        return bool(np.all((v == np.trunc(v)) | np.isinf(v)))


def local_incumbent(indices: Sequence[int], status, z, x, has_x, integrality) -> Tuple[float, int]:
    """Best integer-feasible relaxation of this rank's shard: (z, global child index), (+inf, BIG_INDEX) if none.
    Ties on z go to the smaller child index = the one the reference's FIFO order would have met first."""
    best_z, best_i = math.inf, BIG_INDEX
    for local, gi in enumerate(indices):
        if status[local] != 0 or not has_x[local]:
            continue
        if not feasible_for_ip(integrality, x[local]):
            continue
        if z[local] < best_z or (z[local] == best_z and gi < best_i):
            best_z, best_i = float(z[local]), gi
    return best_z, best_i


def allreduce_incumbent(z_local: float, idx_local: int, dist=None, device=None, comm=None) -> Tuple[float, int]:
    """Global (min z, then min index).

    `comm` (gomilp_amd.lp.Comm): the product path — gomilp_incumbent_allreduce of the C-ABI, ONE RCCL all-reduce(min) of a
    2 * world table per wave.  Without it (CPU tests under gloo): the same table through torch.distributed, picked by the
    same C function (gomilp_incumbent_pick)."""
    if comm is not None:
        return comm.incumbent_allreduce(z_local, idx_local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return z_local, idx_local
    if device is None or str(device) == "cpu":
        import torch
        from . import lp as _lp
        w, r = dist.get_world_size(), dist.get_rank()
        tab = torch.full((w, 2), math.inf, dtype=torch.float64)
        if z_local < math.inf and idx_local < BIG_INDEX:
            tab[r, 0], tab[r, 1] = z_local, float(idx_local)
        dist.all_reduce(tab, op=dist.ReduceOp.MIN)
        return _lp.incumbent_pick(tab.numpy())
    raise ValueError("incumbent exchange on a device goes through the C-ABI (lp.Comm -> gomilp_incumbent_allreduce); "
                     "torch.distributed is only the CPU rehearsal path (gloo)")


def make_shard(children: List[list], rank: int = 0, world: int = 1) -> Tuple[List[int], List[list]]:
    """This rank's part of a wave: (global child indices, the children themselves).  A caller that solves the SAME frontier again
    (a bench, a re-solve with other knobs) builds it once and hands it to solve_wave; the library keeps no cache of its own — a
    frontier list that its owner updates in place must never meet an older wave's shard."""
    mine = shard_indices(len(children), rank, world)
    return mine, [children[i] for i in mine]


def solve_wave(solve_shard: Callable[[List[list]], tuple], children: List[list], integrality: Sequence[bool],
               rank: int = 0, world: int = 1, dist=None, device=None, comm=None, shard: Optional[Tuple[List[int], List[list]]] = None) -> dict:
    """One wave: shard -> solve the shard -> incumbent all-reduce.

    `solve_shard(list_of_children)` returns (status, z, x, has_x) arrays for that list (FrontierPool.solve on
    the GPU box; a stub in the gloo tests).  `shard`: make_shard(children, rank, world) of exactly this wave, built by the caller
    (optional: built here otherwise, every call)."""
    import time
    mine, mine_children = shard if shard is not None else make_shard(children, rank, world)
    t0 = time.perf_counter()
    status, z, x, has_x = solve_shard(mine_children)
    t1 = time.perf_counter()
    zl, il = local_incumbent(mine, status, z, x, has_x, integrality)
    t2 = time.perf_counter()
    zg, ig = allreduce_incumbent(zl, il, dist, device, comm)
    t3 = time.perf_counter()
    # (seconds: this rank's solve, its host scan for the local incumbent, the exchange — which also absorbs the wait for the slowest rank)
    return {"indices": mine, "status": status, "z": z, "x": x, "has_x": has_x, "incumbent_z": zg, "incumbent_index": ig,
            "local_incumbent_z": zl, "seconds_solve": t1 - t0, "seconds_scan": t2 - t1, "seconds_exchange": t3 - t2}
