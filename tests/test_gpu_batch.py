"""The held batch regions of the B&B-sized regime (csrc/kernels.hip: hs_red_batch_hold; csrc/ipm.hip: BatchRegion): a whole Newton
direction, and the step of the iterate with the residual pass behind it, are recorded and executed by one single-workgroup launch
each.  Every recorded operation does the arithmetic of its own launch in the same order, so a solve must not depend on the switch
(HIPSDP_BATCH=0: every kernel its own launch) - compared here bit for bit, in two processes because the switch is read once."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys
ROOT = sys.argv[1]
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'harness')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import importlib.util
import numpy as np
spec = importlib.util.spec_from_file_location('hipsdp_binding', os.path.join(ROOT, 'scip-sdp_amd', 'binding.py'))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
from stress_cases import rand_core
out = []
# the random family of tests/devtools/stress_gpu.py: 1-3 blocks of 2-100 rows, m = 1-200 (both sides of the one-block factor of M),
# 0-150 LP rows; feasible, infeasible and unbounded problems - whatever path a problem takes, recorded or not
for seed in range(int(sys.argv[2])):
    core, kind = rand_core(np.random.default_rng(seed))
    s = hb.Solver(0); s.load_core(core)
    info = s.solve(gaptol=1e-6, feastol=1e-6, pabstol=1e-5)
    y = s.y(); X = [s.X(k) for k in range(len(core.blocks))]
    s.close()
    out.append({"status": int(info.status), "iterations": int(info.iterations), "y": [float(v).hex() for v in y],
                "X": [float(v).hex() for M in X for v in np.asarray(M).ravel()[:64]], "dobj": float(info.dobj).hex()})
print("RESULT " + json.dumps(out))
"""


COUNT = 60


def run(batch):
    env = dict(os.environ)
    env["HIPSDP_BATCH"] = batch
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(COUNT)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_recorded_regions_do_not_change_a_single_bit(gpu):
    off = run("0")
    on = run("1")
    assert len(off) == len(on) == COUNT
    assert sum(1 for a in off if a["status"] == 0) >= COUNT // 3          # the family is not all failures
    for k, (a, b) in enumerate(zip(off, on)):
        assert a == b, "seed %d differs: %s / %s" % (k, {x: a[x] for x in ("status", "iterations", "dobj")},
                                                     {x: b[x] for x in ("status", "iterations", "dobj")})
