import importlib.util, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
np.set_printoptions(linewidth=200, precision=2)
n = 8
rng = np.random.default_rng(100 + n)
Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
for _ in range(3): rng.standard_normal((n, n))
W = (Q * np.where(np.arange(n) < n // 2, -1.0, 2.0)) @ Q.T
W = 0.5 * (W + W.T)
lam, V = hb.syev(W)
print(lam)
print(np.abs(V @ V.T - np.eye(n)))
print("resid", np.abs(V @ W @ V.T - np.diag(lam)).max())
