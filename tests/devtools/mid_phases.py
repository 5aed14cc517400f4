#!/usr/bin/env python3
"""mid_phases.py N M [SOLVES] - developer tool: wall time of every solve of a planted instance and, with the engine's phase events on,
the device time of each phase per iteration - to see where a slow solve of the mid-size regime loses its time."""
import ctypes as C
import importlib.util
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

hb = bench.load_binding()
lib = hb.lib()
n, m = int(sys.argv[1]), int(sys.argv[2])
solves = int(sys.argv[3]) if len(sys.argv) > 3 else 8
prof = int(os.environ.get("MID_PROF", "1"))
s = hb.Solver(0)
s.set_shape(m, [n], 0)
Xs, Zs, ys = bench.planted_pair(n, m, 20240)
b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
lib.hipsdp_phase_name.restype = C.c_char_p
for k in range(solves):
    if prof:
        lib.hipsdp_set_profiling(s.h, 1)
    t0 = time.perf_counter()
    info = s.solve(gaptol=1e-5, feastol=1e-5)
    el = time.perf_counter() - t0
    msg = "solve %d: %.3f ms wall, %d iterations, %.3f ms/iter" % (k, 1e3 * el, info.iterations, 1e3 * el / max(1, info.iterations))
    if prof:
        ms = (C.c_double * 8)()
        lib.hipsdp_get_phase_times(s.h, ms)
        its = max(1, info.iterations)
        msg += " | " + " ".join("%s %.3f" % (lib.hipsdp_phase_name(p).decode(), ms[p] / its) for p in range(7))
    print(msg, flush=True)
s.close()
