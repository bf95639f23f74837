"""Pins the CPU oracle (oracle/) to the reference's own known-answer vectors K1-K9
(/root/reference/ilp_test.go, api_test.go, subproblem_test.go).
Expected values are the literals the reference asserts with reflect.DeepEqual / ==, so the
comparison is bit-for-bit (np.array_equal on float64, == on z)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def _arr(v):
    return None if v is None else np.array(v, dtype=np.float64)


@pytest.mark.parametrize("kat", KATS["milp"], ids=[k["id"] for k in KATS["milp"]])
def test_milp_golden(kat):
    res = O.solve_milp(_arr(kat["c"]), _arr(kat["A"]), _arr(kat["b"]), _arr(kat["G"]), _arr(kat["h"]),
                       kat["integrality"], max_nodes=200)
    assert res.error == kat["want_err"]
    if kat["want_err"] is None:
        assert np.array_equal(res.x, _arr(kat["want_x"])), (res.x.tolist(), kat["want_x"])
        if kat["want_z"] is not None:
            assert res.z == kat["want_z"]
    else:
        # the reference returns the zero-value solution{} (ilp_test.go:54,97)
        assert res.x is None and res.z == 0


@pytest.mark.parametrize("kat", KATS["lp"], ids=[k["id"] for k in KATS["lp"]])
def test_lp_golden(kat):
    r = O.simplex(_arr(kat["c"]), _arr(kat["A"]), _arr(kat["b"]), 0.0, None)
    assert O.STATUS_NAMES[r.status] == kat["want_status"]
    if kat["want_x"] is not None:
        assert np.array_equal(r.x, _arr(kat["want_x"]))
        assert r.z == kat["want_z"]
    else:
        assert r.x is None and math.isnan(r.z)


def test_tree_pins():
    pins = KATS["tree_pins"]
    by_id = {k["id"]: k for k in KATS["milp"]}
    k2 = by_id["K2"]
    res = O.solve_milp(_arr(k2["c"]), _arr(k2["A"]), _arr(k2["b"]), None, None, k2["integrality"])
    root, le, ge = res.nodes[0], res.nodes[1], res.nodes[2]
    assert root.z == pins["K2_root_z"] and root.decision == "BETTER_THAN_INCUMBENT_BRANCHING"
    assert le.constraints == [(1, 1, 2.0)] and le.decision == "BETTER_THAN_INCUMBENT_FEASIBLE"
    assert ge.constraints == [(1, -1, -3.0)] and ge.status == O.ERR_INFEASIBLE
    k7 = by_id["K7"]
    res = O.solve_milp(_arr(k7["c"]), _arr(k7["A"]), _arr(k7["b"]), _arr(k7["G"]), _arr(k7["h"]), k7["integrality"])
    assert res.nodes[0].z == pins["K7_root_z"]
    assert [n.constraints[-1][0] for n in res.nodes[1:]] == [3, 3]
    assert all(n.status == O.ERR_INFEASIBLE for n in res.nodes[1:])
    k8 = by_id["K8"]
    res = O.solve_milp(_arr(k8["c"]), _arr(k8["A"]), _arr(k8["b"]), _arr(k8["G"]), _arr(k8["h"]), k8["integrality"],
                       max_nodes=40)
    zs = []
    for n in res.nodes:
        if n.status == O.OK and (not zs or n.z != zs[-1]):
            zs.append(n.z)
    assert zs[:3] == pins["K8_z_sequence"]
    assert all(n.constraints[-1][0] == 2 for n in res.nodes[1:])


def test_c1_plumbing_case_fixture_is_reproducible():
    """BASELINE config 1 (10 variables / 5 constraints, CPU oracle only): the committed tree fixtures are what the
    oracle produces today (tools/gen_golden.py c1)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    from gen_golden import DECISIONS, c1_problem
    for single, name in ((False, "milp_C1.npz"), (True, "milp_C1_single.npz")):
        fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
        c, G, h, integ = c1_problem(single)
        res = O.solve_milp(c, None, None, G, h, integ, max_nodes=int(fx["budget"]))
        solved = [nd for nd in res.nodes if nd.status != -1]
        assert [nd.status for nd in solved] == list(fx["status"])
        assert [DECISIONS.index(nd.decision) for nd in solved] == list(fx["decision"])
        assert np.array_equal(np.array([nd.z for nd in solved]), fx["z"], equal_nan=True)
        assert (res.error or "") == str(fx["error"])
