"""CPU: the branch-and-bound harness with the oracle as node solver reproduces the MISDP optimum of example_small
(check/testset/short.solu:1: -8.0) and proves example_inf infeasible (short.solu:4).  Pins reader + presolve restatement +
oracle on the reference's end-to-end answers; the GPU test runs the same harness over the HIP backend on more instances."""
import os
import pytest
import bnb
import sdpa_io
from conftest import GOLDEN


def run(name):
    inst = sdpa_io.read_sdpa(os.path.join(GOLDEN, "instances", name))
    prob = bnb.instance_to_sdpi(inst)
    return bnb.branch_and_bound(prob, inst.intvars, bnb.oracle_node_solver())


def test_example_small_optimum():
    best, y, nodes, failed = run("example_small.dat-s")
    assert failed == 0
    assert best == pytest.approx(-8.0, abs=1e-5)


def test_example_inf_is_infeasible():
    best, y, nodes, failed = run("example_inf.dat-s")
    assert best is None and failed == 0
