"""CPU tests of the oracle's building blocks (no GPU): fast-path equivalence, LU blocking, generator."""
import ctypes as C
import math

import numpy as np
import pytest

from gomilp_amd import synth
from oracle import oracle as O


def test_splitmix64_reference_values():
    # splitmix64(seed=0): first outputs of the published reference implementation
    u = synth.splitmix64_uniform(0, 3)
    first = [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F]
    assert [int(x * 2 ** 53) for x in u] == [v >> 11 for v in first]


def test_generator_shapes_and_structure():
    c, A, b = synth.dense_lp_standard_form(8, 3)
    assert A.shape == (8, 16) and c.shape == (16,) and b.shape == (8,)
    assert np.array_equal(A[:, 8:], np.eye(8)) and np.all(c[8:] == 0) and np.all(c[:8] <= 0)
    assert np.all((A[:, :8] >= 0) & (A[:, :8] < 1)) and np.all((b >= 1) & (b < 2))


@pytest.mark.parametrize("m,seed", [(5, 1), (12, 2), (40, 3), (96, 4)])
def test_unit_column_fast_path_is_equivalent(m, seed):
    """findLinearlyIndependent's cond test (simplex.go:630) is == 1 on distinct unit columns, so skipping it
    must not change anything: same basis, same pivots, same bits."""
    c, A, b = synth.dense_lp_standard_form(m, seed)
    full = O.simplex(c, A, b, 0.0, None, trace=True)
    fast = O.simplex(c, A, b, 0.0, None, trace=True, fast_initial_basis=True)
    assert full.cond_evaluations == m - 1 and fast.cond_evaluations == 0
    assert full.status == fast.status == O.OK
    assert full.pivots == fast.pivots
    assert np.array_equal(full.x, fast.x) and full.z == fast.z
    assert O.find_linearly_independent(A) == O.find_linearly_independent(A, fast=True) == list(range(2 * m - 1, m - 1, -1))


def test_unit_columns_have_cond_one():
    rng = np.random.default_rng(0)
    m = 30
    perm = rng.permutation(m)
    for k in (2, 7, 30):
        cols = np.zeros((m, k))
        cols[perm[:k], np.arange(k)] = 1.0
        assert O.cond1(cols) == 1.0


def test_blocked_lu_equals_unblocked_bitwise():
    """dgetrf.go (nb = 64 panels + Dtrsm + Dgemm) and dgetf2.go give the same bits."""
    L = O.lib()
    rng = np.random.default_rng(7)
    n = 150
    A = rng.standard_normal((n, n))
    a1, a2 = A.copy(), A.copy()
    p1, p2 = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int64)
    L.g_dgetrf(n, n, a1.ctypes.data_as(dp), n, p1.ctypes.data_as(ip))
    L.g_dgetf2(n, n, a2.ctypes.data_as(dp), n, p2.ctypes.data_as(ip))
    assert np.array_equal(p1, p2) and np.array_equal(a1, a2)


def test_threads_do_not_change_results():
    c, A, b = synth.dense_lp_standard_form(150, 9)
    O.set_threads(1)
    r1 = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True)
    O.set_threads(4)
    r4 = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True)
    O.set_threads(1)
    assert np.array_equal(r1.x, r4.x) and r1.z == r4.z and r1.pivots_phase2 == r4.pivots_phase2


def test_solve_matches_numpy():
    L = O.lib()
    rng = np.random.default_rng(3)
    dp = C.POINTER(C.c_double)
    for n in (1, 2, 17, 80):
        A = rng.standard_normal((n, n)) + n * np.eye(n)
        b = rng.standard_normal(n)
        x = b.copy()
        assert L.g_solve_vec(n, A.ctypes.data_as(dp), n, x.ctypes.data_as(dp)) == 0
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-10, atol=1e-12)
        xt = b.copy()
        assert L.g_solve_vec_trans(n, A.ctypes.data_as(dp), n, xt.ctypes.data_as(dp)) == 0
        assert np.allclose(xt, np.linalg.solve(A.T, b), rtol=1e-10, atol=1e-12)


def test_cond_estimates_are_sane():
    rng = np.random.default_rng(5)
    for shape in ((20, 20), (40, 13), (300, 200)):   # square -> LU path; tall -> QR (300x200 takes the blocked Dgeqrf)
        A = rng.standard_normal(shape)
        est = O.cond1(A)
        if shape[0] == shape[1]:
            true = np.linalg.cond(A, 1)
        else:
            R = np.linalg.qr(A, mode="r")
            true = np.linalg.cond(R, 1)
        assert true / 10 <= est <= true * 1.0001, (shape, est, true)
    A = rng.standard_normal((30, 5))
    A[:, 4] = A[:, 0] + A[:, 1]
    assert O.cond1(A) > 1e12


def test_error_returns_follow_reference_conventions():
    # unbounded via verifyInputs: zero column with negative cost -> (-Inf, nil, ErrUnbounded)  simplex.go:96-98
    r = O.simplex([-1.0, 0.0, 0.0], [[0.0, 1.0, 1.0]], [1.0])
    assert r.status == O.ERR_UNBOUNDED and r.z == -math.inf and r.x is None
    r = O.simplex([1.0, 0.0, 0.0], [[0.0, 1.0, 1.0]], [1.0])
    assert r.status == O.ERR_ZERO_COLUMN and math.isnan(r.z) and r.x is None
    r = O.simplex([1.0, 0.0], [[0.0, 0.0], [1.0, 1.0]], [0.0, 1.0])
    assert r.status == O.ERR_ZERO_ROW
    r = O.simplex([1.0, 0.0], [[0.0, 0.0], [1.0, 1.0]], [1.0, 1.0])
    assert r.status == O.ERR_INFEASIBLE
    # size mismatch panics in the reference (simplex.go:387-398)
    r = O.simplex([1.0], [[1.0, 1.0]], [1.0])
    assert r.status == O.ERR_BAD_SHAPE


def test_bnb_child_layout_matches_convert_to_equalities():
    """subproblem_test.go:296-357 pins [[A,0],[G,I]]; children add ±e_j rows (subproblem.go:245-255)."""
    c0 = np.array([1.0, 2.0, 0.0])
    A0 = np.array([[1.0, 1.0, 1.0]])
    b0 = np.array([4.0])
    c, A, b = O.child_standard_form(c0, A0, b0, [(1, 1, 2.0), (0, -1, -3.0)])
    assert np.array_equal(c, [1, 2, 0, 0, 0])
    assert np.array_equal(A, [[1, 1, 1, 0, 0], [0, 1, 0, 1, 0], [-1, 0, 0, 0, 1]])
    assert np.array_equal(b, [4, 2, -3])
