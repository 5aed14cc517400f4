#!/usr/bin/env python3
"""c4_dbg.py [n m] - developer tool: one verbose solve of the planted bench instance at a given size (default 2000 4000)"""
import os, sys, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
hb = bench.load_binding()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
s = hb.Solver(0)
s.set_shape(m, [n], 0)
Xs, Zs, ys = bench.planted_pair(n, m, 20240)
b = s.gen_planted(n, m, 20240, Xs, Zs, ys)
info = s.solve(gaptol=1e-5, feastol=1e-5, verbose=1)
print("status", info.status, "iterations", info.iterations, "dobj", info.dobj, "planted", float(b @ ys), "chol_fail", info.chol_fail, "solve s", info.solve_seconds)
s.close()
