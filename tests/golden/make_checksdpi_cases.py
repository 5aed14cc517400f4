"""Writes tests/golden/checksdpi_cases.json: the inputs and pinned outputs of the reference's SDPI known-answer tests that
reach the solver backend (unittests/src/checksdpi.c tests 1-4 and 9-11; tolerance EPS = 1e-6, checksdpi.c:49) plus
check1dsdp.c test5 with its one-variable shortcut bypassed (optimum 1.541381, check1dsdp.c:341-411).  Pure data
transcription; run once, output committed."""
import json
import os

INF = 1e20
cases = []

lp4 = [[-1.0, INF, {"0": 1.0}], [-1.0, INF, {"0": -1.0}], [-1.0, INF, {"1": 1.0}], [-1.0, INF, {"1": -1.0}]]

cases.append(dict(name="test1", ref="checksdpi.c:537", obj=[-3, -1], lb=[0, 0], ub=[INF, INF], blocks=[],
                  lp=[[-INF, 10, {"0": 2, "1": 1}], [-INF, 15, {"0": 1, "1": 3}]],
                  expect=dict(primal="feas", dual="feas", dualsol=[5, 0], lbvals=[0, 0.5], rhsvals=[1.5, 0], objval=-15.0)))
cases.append(dict(name="test2", ref="checksdpi.c:577", obj=[-3, -1], lb=[-INF, -INF], ub=[INF, INF], blocks=[],
                  lp=[[-INF, 10, {"0": 2, "1": 1}], [-INF, 15, {"0": 1, "1": 3}]],
                  expect=dict(primal="infeas", dual="unbounded")))
cases.append(dict(name="test3", ref="checksdpi.c:616", obj=[10, 15], lb=[0, 0], ub=[INF, INF], blocks=[],
                  lp=[[3, 3, {"0": 2, "1": 1}], [1, 1, {"0": 1, "1": 3}]],
                  expect=dict(primal="unbounded", dual="infeas")))
cases.append(dict(name="test4", ref="checksdpi.c:656", obj=[-1, -1], lb=[-INF, -INF], ub=[INF, INF], blocks=[],
                  lp=[[-INF, 0, {"0": 1, "1": -1}], [-INF, -1, {"0": -1, "1": 1}]],
                  expect=dict(primal="infeas", dual="infeas")))
cases.append(dict(name="test9", ref="checksdpi.c:921", obj=[-1, 0], lb=[-INF, -INF], ub=[INF, INF],
                  blocks=[dict(n=2, vars={"0": [[0, 0, 1.0]], "1": [[1, 1, 0.75]]}, const=[[1, 0, -1.0]])], lp=lp4,
                  expect=dict(primal="unbounded", dual="infeas")))
cases.append(dict(name="test10", ref="checksdpi.c:1022", obj=[-1, -1], lb=[-INF, -INF], ub=[INF, INF],
                  blocks=[dict(n=2, vars={"0": [[0, 0, 1.0]], "1": [[1, 1, 1.0]]}, const=[])], lp=lp4,
                  expect=dict(primal="feas", dual="feas", dualsol=[1, 1], lhsvals=[0, 1, 0, 1], rhsvals=[0, 0, 0, 0],
                              X=[[0, 0], [0, 0]], objval=-2.0)))
cases.append(dict(name="test11", ref="checksdpi.c:1113", obj=[1], lb=[-INF], ub=[INF],
                  blocks=[dict(n=2, vars={"0": [[0, 0, 1.0], [1, 1, 1.0]]}, const=[[0, 0, 1.0], [1, 0, 2.0], [1, 1, 4.0]])],
                  lp=[], expect=dict(primal="feas", dual="feas", dualsol=[5], X=[[0.2, 0.4], [0.4, 0.8]], objval=5.0)))
cases.append(dict(name="check1dsdp_test5", ref="check1dsdp.c:341", obj=[1], lb=[0], ub=[2],
                  blocks=[dict(n=2, vars={"0": [[0, 0, 1.0], [1, 0, -2.0], [1, 1, 5.0]]},
                               const=[[0, 0, -2.0], [1, 0, 1.0], [1, 1, 3.0]])],
                  lp=[], expect=dict(primal="feas", dual="feas", dualsol=[1.541381], objval=1.541381, tol=1e-6)))
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "checksdpi_cases.json"), "w") as f:
    json.dump(dict(eps=1e-6, cases=cases), f, indent=1)
print("wrote", len(cases), "cases")
