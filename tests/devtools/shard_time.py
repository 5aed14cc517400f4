#!/usr/bin/env python3
"""shard_time.py [n m] | var n m ranks cw [rank ...] - developer tool: per-rank time of the row-sharded Schur assembly at 1/2/4/8 ranks, measured on ONE
device (each rank's share is run alone), next to the time of the unsharded assembly of the bench."""
import ctypes as C, importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec); spec.loader.exec_module(hb)
var = len(sys.argv) > 1 and sys.argv[1] == "var"
n = int(sys.argv[1]) if len(sys.argv) > 1 and not var else 500
m = int(sys.argv[2]) if len(sys.argv) > 2 and not var else 1000
lib = hb.ulib()
if len(sys.argv) > 1 and sys.argv[1] == "var":
    # one rank's share of the variable-sharded assembly (matrices sharded by variable): only that rank's rows of A are allocated
    n, m, G, cw = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    ranks = [int(a) for a in sys.argv[6:]] or list(range(G))
    flop = 4.0 * (m + 1) * n ** 3 + float(m + 1) ** 2 * n ** 2
    print("matrices sharded by variable: n=%d m=%d, %d ranks, column slices of %d" % (n, m, G, cw), flush=True)
    for r in ranks:
        ms, by = C.c_double(), C.c_double()
        rc = lib.hipsdp_schur_var_share_time(0, m + 1, n, G, r, cw, 1 if n >= 2000 else 5, C.byref(ms), C.byref(by))
        assert rc == 0, (rc, lib.hipsdp_last_error())
        print("  rank %d: %.1f ms per assembly -> %.1f TF (algorithmic share 1/%d of 4 m n^3 + m^2 n^2); all-to-all %.2f GB sent and "
              "received per assembly (%.0f ms at 7 x 50 GB/s)" % (r, ms.value, flop / G / ms.value * 1e-9, G, by.value * 1e-9,
                                                                   by.value / 350e9 * 1e3), flush=True)
    sys.exit(0)
flop = 4.0 * (m + 1) * n ** 3 + float(m + 1) ** 2 * n ** 2
for bycol in (1, 0):
    print("column slices of the W formulation + all-reduce" if bycol else "row chunks of the U formulation + all-gather")
    for G in (1, 2, 4, 8):
        ts = []
        for r in range(G):
            ms = C.c_double()
            rc = lib.hipsdp_schur_shard_time(0, m + 1, n, G, r, bycol, 5, C.c_double(0.0), C.byref(ms))
            assert rc == 0, (rc, lib.hipsdp_last_error())
            ts.append(ms.value)
        if G == 1:
            base = max(ts)
        print("  ranks %d: per-rank ms min %.3f max %.3f  -> %.1f TF/rank (algorithmic) at the slowest, speed-up of the assembly %.2f"
              % (G, min(ts), max(ts), flop / G / max(ts) * 1e-9, base / max(ts)))
        if len(sys.argv) > 3:
            print("     ", " ".join("%.3f" % t for t in ts))
