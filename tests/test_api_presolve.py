"""Host-side mirror of the reference's builder API and presolve (gomilp_amd/api.py, gomilp_amd/presolve.py)
against the reference's own test data: api_conversion_test.go (toSolveable, 7 cases), api_test.go:13-32 (checkExpression),
api_test.go:86-139 (K6: the end-to-end solve), presolve_test.go (filterFixedVars) — plus the documented quirks of presolve.go.
The CPU tests drive the tree search with the oracle (host logic only); the GPU test runs the same problems with every
relaxation on the device."""
import math

import numpy as np
import pytest

from gomilp_amd import api
from gomilp_amd.presolve import PreProcessor, remove_duplicate_constraints


def _abc(maximize=False, ints=(False, True, True), bounds=False):
    prob = api.Problem()
    v1 = prob.add_variable("v1").set_coeff(-1)
    v2 = prob.add_variable("v2").set_coeff(-2)
    v3 = prob.add_variable("v3").set_coeff(1)
    for v, i in zip((v1, v2, v3), ints):
        if i:
            v.is_integer()
    if bounds:
        v1.upper_bound(4).lower_bound(2)
        v3.lower_bound(1)
    if maximize:
        prob.maximize()
    return prob, v1, v2, v3


def _k6():
    prob = api.Problem()
    v1 = prob.add_variable("v1").set_coeff(-1)
    v2 = prob.add_variable("v2").set_coeff(-2)
    v3 = prob.add_variable("v3").set_coeff(1)
    v4 = prob.add_variable("v4").set_coeff(3)
    prob.add_constraint().add_expression(1, v1).equal_to(5)
    prob.add_constraint().add_expression(3, v2).equal_to(2)
    prob.add_constraint().add_expression(1, v3).equal_to(2)
    prob.add_constraint().add_expression(1, v4).smaller_than_or_equal_to(2)
    return prob


def _same(got, c, A, b, G, h, integ):
    assert np.array_equal(got.c, np.array(c, float))
    for g, w in ((got.A, A), (got.b, b), (got.G, G), (got.h, h)):
        assert (g is None) == (w is None)
        if w is not None:
            assert np.array_equal(g, np.array(w, float))
    assert got.integrality == list(integ)


def test_to_solveable_A_one_inequality():            # api_conversion_test.go:12-48, api_test.go:86-117
    _same(_k6().to_solveable(), [-1, -2, 1, 3], [[1, 0, 0, 0], [0, 3, 0, 0], [0, 0, 1, 0]], [5, 2, 2], [[0, 0, 0, 1]], [2], [False] * 4)


def test_to_solveable_B_C_D_equalities():            # api_conversion_test.go:50-152
    for maximize, c in ((False, [-1, -2, 1]), (True, [1, 2, -1])):
        prob, v1, v2, v3 = _abc(maximize)
        prob.add_constraint().add_expression(1, v1).equal_to(5)
        prob.add_constraint().add_expression(3, v2).equal_to(2)
        prob.add_constraint().add_expression(1, v3).equal_to(2)
        _same(prob.to_solveable(), c, [[1, 0, 0], [0, 3, 0], [0, 0, 1]], [5, 2, 2], None, None, [False, True, True])
    prob, v1, v2, v3 = _abc(True)
    prob.add_constraint().add_expression(1, v1).add_expression(1, v2).equal_to(5)
    prob.add_constraint().add_expression(3, v2).equal_to(2)
    prob.add_constraint().add_expression(1, v3).equal_to(2)
    _same(prob.to_solveable(), [1, 2, -1], [[1, 1, 0], [0, 3, 0], [0, 0, 1]], [5, 2, 2], None, None, [False, True, True])


def test_to_solveable_E_F_G_inequalities_and_bounds():   # api_conversion_test.go:154-279
    prob, v1, v2, v3 = _abc(True)
    prob.add_constraint().add_expression(1, v1).add_expression(1, v2).equal_to(5)
    prob.add_constraint().add_expression(3, v2).equal_to(2)
    prob.add_constraint().add_expression(1, v3).equal_to(2)
    prob.add_constraint().add_expression(1, v3).add_expression(1, v1).smaller_than_or_equal_to(2)
    _same(prob.to_solveable(), [1, 2, -1], [[1, 1, 0], [0, 3, 0], [0, 0, 1]], [5, 2, 2], [[1, 0, 1]], [2], [False, True, True])
    for bounds in (False, True):
        prob, v1, v2, v3 = _abc(True, bounds=bounds)
        prob.add_constraint().add_expression(1, v1).add_expression(1, v2).smaller_than_or_equal_to(5)
        prob.add_constraint().add_expression(3, v2).smaller_than_or_equal_to(2)
        prob.add_constraint().add_expression(1, v3).smaller_than_or_equal_to(2)
        prob.add_constraint().add_expression(1, v3).add_expression(1, v1).smaller_than_or_equal_to(2)
        G = [[1, 1, 0], [0, 3, 0], [0, 0, 1], [1, 0, 1]]
        h = [5, 2, 2, 2]
        if bounds:   # upper row, then lower row, variable by variable; lower bounds <= 0 give no row (api.go:245-272)
            G += [[1, 0, 0], [-1, 0, 0], [0, 0, -1]]
            h += [4, -2, -1]
        _same(prob.to_solveable(), [1, 2, -1], None, None, G, h, [False, True, True])


def test_check_expression():                          # api_test.go:13-32
    prob = api.Problem()
    v = prob.add_variable("v1").set_coeff(1)
    assert prob.check_expression(api.Expression(2, v))
    assert not prob.check_expression(api.Expression(1, api.Variable("other")))
    with pytest.raises(RuntimeError):
        prob.add_constraint().add_expression(1, api.Variable("undeclared"))


def test_filter_fixed_vars():                         # presolve_test.go:8-66 + the rhs update and the undoer of presolve.go:139-176
    prob = api.Problem()
    ok = prob.add_variable("okayvar").lower_bound(1).upper_bound(3)
    bad = prob.add_variable("notokayvar").lower_bound(1).upper_bound(1).set_coeff(4)
    con = prob.add_constraint().add_expression(2, ok).add_expression(5, bad).smaller_than_or_equal_to(10)
    pre = PreProcessor()
    out = pre.filter_fixed_vars(prob)
    assert out.variables == [ok] and len(prob.variables) == 2        # the caller's Problem keeps its slice (passed by value)
    assert [e.variable for e in con.expressions] == [ok]
    assert con.rhs == 10 - 4 * 1                                      # (sic) coefficient * lower, not a_ij * lower (presolve.go:152)
    assert len(pre.undoers) == 1
    sol = pre.post_solve({"okayvar": 2.0})
    assert sol.by_name == {"okayvar": 2.0, "notokayvar": 4.0}         # (sic) objective contribution, not the value (presolve.go:141)
    assert sol.objective == 6.0                                       # (sic) sum of the values (presolve.go:92-95)
    with pytest.raises(RuntimeError):
        pre.post_solve({"notokayvar": 1.0})


def test_implicit_zero_empty_and_duplicate_constraints():   # presolve.go:188-295
    prob = api.Problem()
    x, y, z = (prob.add_variable(n) for n in "xyz")
    prob.add_constraint().add_expression(1, x).add_expression(2, y).smaller_than_or_equal_to(0)   # x = y = 0 implied
    prob.add_constraint().add_expression(1, z).add_expression(0, x).smaller_than_or_equal_to(7)    # zero coefficient: sanitised away
    prob.add_constraint().add_expression(1, z).smaller_than_or_equal_to(5)                         # duplicate of the row above: smaller rhs wins
    pre = PreProcessor()
    out = pre.pre_solve(prob)
    assert (x.lower, x.upper, y.lower, y.upper) == (0, 0, 0, 0)       # the caller's variables were rewritten (:226-229)
    # no undoer was registered in this pass, so the loop of presolve.go:59-69 ends: the implicitly fixed variables stay in the
    # problem with bounds [0, 0] (they would only be removed by a further pass)
    assert [v.name for v in out.variables] == ["x", "y", "z"]
    assert [(len(c.expressions), c.rhs) for c in out.constraints] == [(2, 0.0), (1, 5.0)]
    # equal right-hand sides: BOTH copies go (presolve.go:283-287)
    p2 = api.Problem()
    w = p2.add_variable("w")
    p2.add_constraint().add_expression(1, w).smaller_than_or_equal_to(3)
    p2.add_constraint().add_expression(1, w).smaller_than_or_equal_to(3)
    assert remove_duplicate_constraints(p2).constraints == []


def _oracle_solver(c, A, b, G, h, integ, max_nodes=255):
    from oracle import oracle as O
    return O.solve_milp(c, A, b, G, h, integ, max_nodes=max_nodes)


def _check_k6(sol):
    assert sol.get_value_for("v1") == 5.0 and sol.get_value_for("v2") == 0.6666666666666666
    assert sol.get_value_for("v3") == 2.0 and sol.get_value_for("v4") == 0.0
    assert sol.objective == 5.0 + 0.6666666666666666 + 2.0 + 0.0      # (sic)
    with pytest.raises(KeyError):
        sol.get_value_for("nope")


def test_k6_end_to_end_host_logic_with_the_oracle():   # api_test.go:119-139
    _check_k6(_k6().solve(milp_solver=_oracle_solver))


def _bounded_milp():
    prob = api.Problem()
    a = prob.add_variable("a").set_coeff(3).is_integer().upper_bound(4)
    b = prob.add_variable("b").set_coeff(2).upper_bound(2.5)
    f = prob.add_variable("f").set_coeff(1.5).lower_bound(2).upper_bound(2)        # fixed: removed by presolve
    prob.add_constraint().add_expression(2, a).add_expression(1, b).add_expression(1, f).smaller_than_or_equal_to(11.0)
    prob.add_constraint().add_expression(1, a).add_expression(3, b).smaller_than_or_equal_to(8)
    prob.maximize()
    return prob


def test_presolved_milp_matches_between_oracle_and_reference_quirks():
    sol = _bounded_milp().solve(milp_solver=_oracle_solver, max_nodes=63)
    assert sol.by_name["f"] == 3.0                                     # (sic) 1.5 * 2: the objective contribution
    assert sol.by_name == {"a": 3.0, "b": 1.666666666666667, "f": 3.0}
    assert sol.objective == 3.0 + 1.666666666666667 + 3.0                # (sic)


@pytest.mark.gpu
def test_builder_presolve_tree_postsolve_on_the_gpu():
    """The same two problems with every relaxation on the device (gomilp_amd.bnb -> gomilp_frontier_solve): identical to the
    oracle-driven run, value by value."""
    _check_k6(_k6().solve())
    want = _bounded_milp().solve(milp_solver=_oracle_solver, max_nodes=63)
    got = _bounded_milp().solve(max_nodes=63)
    assert got.by_name == want.by_name and got.objective == want.objective
