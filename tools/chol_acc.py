#!/usr/bin/env python3
"""chol_acc.py - developer tool: accuracy of potrf / potrs / trtri on ill-conditioned SPD matrices (HIPSDP_LIB selects the library)"""
import os, sys, importlib.util
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
if os.environ.get("HIPSDP_LIB"):
    hb.LIBPATH = os.environ["HIPSDP_LIB"]
rng = np.random.default_rng(5)
for n in (16, 40, 64, 100, 200, 500):
    for cond in (1e2, 1e8, 1e12):
        Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        ev = np.logspace(0, -np.log10(cond), n)
        A = (Q * ev) @ Q.T; A = 0.5 * (A + A.T)
        L, fail = hb.potrf(A)
        fres = np.abs(L @ L.T - A).max() / np.abs(A).max()
        Li = hb.trtri(A)
        ires = np.abs(L @ Li - np.eye(n)).max()
        b = rng.standard_normal(n)
        x = hb.potrs(A, b)
        sres = np.abs(A @ x - b).max() / (np.abs(A).max() * np.abs(x).max() + np.abs(b).max())
        xr = np.linalg.solve(A, b)
        print("n %4d cond %.0e fail %d  |LL'-A| %.2e  |L Linv - I| %.2e  solve resid %.2e  x err %.2e" % (n, cond, fail, fres, ires, sres, np.abs(x - xr).max() / np.abs(xr).max()))
