"""Where the time of one block-column launch of the blocked Cholesky goes (development tool, GPU box).

Needs the timing build of chol.hip (see README.md in this directory):
  hipcc ... -DPD_TIMING -c csrc/chol.hip -o build/var/chol_pdt.o   and link lib/variants/libhipsdp_pdt.so
  HIPSDP_LIB=scip-sdp_amd/lib/variants/libhipsdp_pdt.so python tests/devtools/potrf_phase_time.py [n]
Prints the 100 MHz time stamps of the second step workgroup of launch kb (differences in microseconds)."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
spec = importlib.util.spec_from_file_location("hipsdp_binding", os.path.join(ROOT, "scip-sdp_amd", "binding.py"))
hb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(hb)

NAMES = {0: "start", 1: "tiles of column kb-1 in LDS", 2: "P P^T products, updated diagonal block in LDS", 4: "rows in registers",
         3: "factorization: column 32", 5: "factorization done", 6: "L stored", 7: "16 x 16 inverses", 8: "off-diagonal inverse blocks",
         9: "inv(L) stored (owner) / panel starts", 10: "panel block stored",
         11: "  panel 2: its tile column updated, loaded", 12: "  panel 2: 16 pivot steps", 13: "  panel 2: L panel stored",
         15: "  panel 2: barrier"}
ORDER = [0, 1, 2, 4, 3, 11, 12, 13, 15, 5, 6, 7, 8, 9, 10]


def show(t, head):
    print(head)
    prev = None
    for i in ORDER:
        if t[i] == 0:
            continue
        if prev is not None and t[i] < prev[1]:
            continue                      # stale stamp of a path this launch did not take
        d = 0.0 if prev is None else (t[i] - prev[1]) / 100.0
        print("  %-52s +%6.2f us" % (NAMES[i], d))
        prev = (i, t[i])


def in_solve(fn):
    """the stamps of the last launch with first column 64 inside a solve of the bench's n = 500, m = 1000 instance: the clocks a
    launch sees between the assemblies, not those of an idle device"""
    sys.path.insert(0, ROOT)
    import bench
    n, m = 500, 1000
    s = hb.Solver(0)
    s.set_shape(m, [n], 0)
    Xs, Zs, ys = bench.planted_pair(n, m, 20240)
    s.gen_planted(n, m, 20240, Xs, Zs, ys)
    out = (C.c_longlong * 16)()
    assert fn(64, None) == 0
    for rep in range(2):
        s.solve(gaptol=1e-5, feastol=1e-5)
    assert fn(-1, out) == 0
    show([out[i] for i in range(16)], "last launch kb = 1 of a solve (n = %d, m = %d)" % (n, m))
    s.close()


def main():
    lib = hb.lib()
    fn = lib.hipsdp_debug_pd_timing
    fn.argtypes = [C.c_int, C.POINTER(C.c_longlong)]
    if len(sys.argv) > 1 and sys.argv[1] == "solve":
        return in_solve(fn)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1001
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n + 10))
    A = G @ G.T + n * np.eye(n)
    out = (C.c_longlong * 16)()
    for kb in (0, 1, (n // 64) // 2):
        assert fn(64 * kb, None) == 0
        for rep in range(3):
            hb.potrf_ex(A)
        assert fn(-1, out) == 0
        t = [out[i] for i in range(16)]
        print("launch kb = %d (n = %d)" % (kb, n))
        prev = None
        for i in ORDER:
            if t[i] == 0:
                continue
            if prev is not None and t[i] < prev[1]:
                continue                      # stale stamp of a path this launch did not take
            d = 0.0 if prev is None else (t[i] - prev[1]) / 100.0
            print("  %-52s +%6.2f us" % (NAMES[i], d))
            prev = (i, t[i])
        for i in range(16):
            out[i] = 0


if __name__ == "__main__":
    main()
