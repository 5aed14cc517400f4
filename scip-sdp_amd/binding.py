"""binding.py - ctypes view of libhipsdp.so's C ABI (include/hipsdp.h) for the test-suite, bench.py and
__graft_entry__.py.  This is plumbing only: no arithmetic happens here, and nothing in this file imports oracle/.
If the shared library is missing or there is no GPU, calls fail loudly (RuntimeError) - there is no fallback path."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.environ.get("HIPSDP_LIB", os.path.join(_HERE, "lib", "libhipsdp.so"))     # HIPSDP_LIB: developer experiments with kernel variants

STATUS_NAMES = {0: "optimal", 1: "dual_infeasible", 2: "dual_unbounded", 3: "both_infeasible", 4: "iterlimit",
                5: "numeric", 6: "timelimit", 7: "objlimit", -1: "unsolved"}


class Params(C.Structure):
    _fields_ = [("gaptol", C.c_double), ("feastol", C.c_double), ("infeastol", C.c_double), ("objlimit", C.c_double),
                ("timelimit", C.c_double), ("gamma", C.c_double), ("ws_gbytes", C.c_double), ("maxiter", C.c_int),
                ("verbose", C.c_int), ("lanczos_steps", C.c_int), ("settings", C.c_int), ("pabstol", C.c_double), ("preoptgap", C.c_double)]


class Info(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("pobj", C.c_double), ("dobj", C.c_double),
                ("pinf", C.c_double), ("dinf", C.c_double), ("dabs", C.c_double), ("gap", C.c_double), ("mu", C.c_double),
                ("tau", C.c_double), ("kappa", C.c_double), ("solve_seconds", C.c_double), ("schur_seconds", C.c_double),
                ("schur_flops", C.c_double), ("schur_calls", C.c_int), ("chol_fail", C.c_int), ("warm_started", C.c_int),
                ("settings_used", C.c_int), ("schur_flops_executed", C.c_double)]


_lib = None


def lib():
    """loads libhipsdp.so (raises if it was not built: run `python -c 'import __graft_entry__ as g; g.build()'`)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise RuntimeError("libhipsdp.so not built (%s); run __graft_entry__.build()" % LIBPATH)
        eng = C.CDLL(LIBPATH, mode=C.RTLD_GLOBAL)
        # the solver-interface backend (SCIPsdpiSolver*, SCIPlapack*) is a library of its own that needs the engine: symbol lookups
        # on its handle also find the engine's (dlsym searches the dependencies)
        sdpi = os.path.join(os.path.dirname(LIBPATH), os.path.basename(LIBPATH).replace("libhipsdp", "libhipsdp_sdpi", 1))
        _lib = C.CDLL(sdpi, mode=C.RTLD_GLOBAL) if os.path.exists(sdpi) else eng
        _lib.hipsdp_last_error.restype = C.c_char_p
        _lib.hipsdp_version.restype = C.c_char_p
    return _lib


_ulib = None


def ulib():
    """libhipsdp_units.so: the engine's objects plus the TEST / BENCH entry points of csrc/units.hip (include/hipsdp_units.h) - a library
    of its own, loaded beside the product libraries with local symbol scope (its copy of the engine is self-contained)"""
    global _ulib
    if _ulib is None:
        path = os.path.join(os.path.dirname(LIBPATH), "libhipsdp_units.so")
        if not os.path.exists(path):
            raise RuntimeError("libhipsdp_units.so not built (%s); run __graft_entry__.build()" % path)
        _ulib = C.CDLL(path, mode=C.RTLD_LOCAL)
        _ulib.hipsdp_last_error.restype = C.c_char_p
    return _ulib


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: rc=%d (%s)" % (what, rc, lib().hipsdp_last_error().decode()))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def device_count():
    return lib().hipsdp_device_count()


class Solver:
    """thin RAII wrapper of hipsdp_solver"""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(lib().hipsdp_create(C.byref(self.h), device), "hipsdp_create")
        self.m = 0
        self.ns = []
        self.q = 0

    def close(self):
        if self.h:
            lib().hipsdp_free(C.byref(self.h))
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_shape(self, m, blocksizes, q, nnz=None):
        """nnz: lower-triangular triplets the caller is going to add per block (hipsdp_set_shape2: a block whose count makes the
        pair formula the cheaper Schur assembly is kept as nonzeros)"""
        bs = np.asarray(blocksizes, dtype=np.int32)
        if nnz is None:
            _chk(lib().hipsdp_set_shape(self.h, m, len(bs), _ip(bs), q), "hipsdp_set_shape")
        else:
            cnt = np.ascontiguousarray(nnz, dtype=np.int64)
            assert len(cnt) == len(bs)
            _chk(lib().hipsdp_set_shape2(self.h, m, len(bs), _ip(bs), q, cnt.ctypes.data_as(C.POINTER(C.c_longlong))), "hipsdp_set_shape2")
        self.m, self.ns, self.q = m, [int(v) for v in bs], q

    def sparse_policy(self, mode):
        _chk(lib().hipsdp_sparse_policy(self.h, mode), "hipsdp_sparse_policy")

    def is_sparse(self, k):
        return bool(lib().hipsdp_block_is_sparse(self.h, k))

    def load_sparse(self, m, n, b, coo, A0):
        """one block given as triplets of the variables' matrices (var 1 .. m, row >= col) and a dense constant matrix"""
        var, row, col, val = coo
        il = np.tril_indices(n)
        c0 = A0[il]
        keep = c0 != 0.0
        self.set_shape(m, [n], 0, nnz=[len(val) + int(keep.sum())])
        self.set_obj(b)
        self.add_entries(0, np.concatenate([np.zeros(int(keep.sum()), dtype=np.int32), var]),
                         np.concatenate([il[0][keep].astype(np.int32), row]), np.concatenate([il[1][keep].astype(np.int32), col]),
                         np.concatenate([c0[keep], val]))

    def set_obj(self, b):
        b = _f64(b)
        _chk(lib().hipsdp_set_obj(self.h, _dp(b)), "hipsdp_set_obj")

    def set_block_dense(self, k, A):
        A = _f64(A)
        assert A.size == (self.m + 1) * self.ns[k] ** 2
        _chk(lib().hipsdp_set_block_dense(self.h, k, _dp(A)), "hipsdp_set_block_dense")

    def add_entries(self, k, var, row, col, val):
        var = np.ascontiguousarray(var, dtype=np.int32)
        row = np.ascontiguousarray(row, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = _f64(val)
        _chk(lib().hipsdp_add_entries(self.h, k, C.c_longlong(len(val)), _ip(var), _ip(row), _ip(col), _dp(val)),
             "hipsdp_add_entries")

    def set_lp(self, Dext):
        Dext = _f64(Dext)
        _chk(lib().hipsdp_set_lp(self.h, _dp(Dext)), "hipsdp_set_lp")

    def block_device_ptr(self, k):
        p = C.POINTER(C.c_double)()
        _chk(lib().hipsdp_block_device_ptr(self.h, k, C.byref(p)), "hipsdp_block_device_ptr")
        return C.cast(p, C.c_void_p).value

    def gen_planted(self, n, m, seed, Xstar, Zstar, ystar):
        """device-side synthetic instance (see hipsdp_gen_planted); returns b"""
        Xstar, Zstar, ystar = _f64(Xstar), _f64(Zstar), _f64(ystar)
        b = np.zeros(m)
        _chk(lib().hipsdp_gen_planted(self.h, n, m, C.c_longlong(seed), _dp(Xstar), _dp(Zstar), _dp(ystar), _dp(b)),
             "hipsdp_gen_planted")
        return b

    def gen_planted_density(self, n, m, seed, density, Xstar, Zstar, ystar):
        Xstar, Zstar, ystar = _f64(Xstar), _f64(Zstar), _f64(ystar)
        b = np.zeros(m)
        _chk(lib().hipsdp_gen_planted_density(self.h, n, m, C.c_longlong(seed), C.c_double(density), _dp(Xstar), _dp(Zstar), _dp(ystar),
                                              _dp(b)), "hipsdp_gen_planted_density")
        return b

    def get_block_dense(self, k):
        A = np.zeros((self.m + 1, self.ns[k], self.ns[k]))
        _chk(lib().hipsdp_get_block_dense(self.h, k, _dp(A)), "hipsdp_get_block_dense")
        return A

    def load_core(self, prob):
        """prob: object with m, b, blocks (list of [m+1, n, n]), D [q, m], c [q] (the oracle's CoreProblem layout)"""
        self.set_shape(prob.m, [A.shape[1] for A in prob.blocks], prob.q)
        self.set_obj(prob.b)
        for k, A in enumerate(prob.blocks):
            self.set_block_dense(k, A)
        if prob.q:
            self.set_lp(np.concatenate([np.asarray(prob.c).reshape(-1, 1), prob.D], axis=1))

    def set_start(self, y, X, Z, x=None, z=None):
        """start point of the next solve (used by the engine only if strictly interior: Info.warm_started)"""
        y = _f64(y)
        Xs = [_f64(M) for M in X]
        Zs = [_f64(M) for M in Z]
        PX = (C.POINTER(C.c_double) * max(len(Xs), 1))(*[_dp(M) for M in Xs])
        PZ = (C.POINTER(C.c_double) * max(len(Zs), 1))(*[_dp(M) for M in Zs])
        xx = _f64(x if x is not None else np.zeros(max(self.q, 1)))
        zz = _f64(z if z is not None else np.zeros(max(self.q, 1)))
        _chk(lib().hipsdp_set_start(self.h, _dp(y), PX, PZ, _dp(xx), _dp(zz)), "hipsdp_set_start")

    def solve(self, **kw):
        p = Params()
        lib().hipsdp_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        info = Info()
        _chk(lib().hipsdp_solve(self.h, C.byref(p), C.byref(info)), "hipsdp_solve")
        return info

    def solve_path(self):
        """1: the last solve ran in the one-launch kernel of csrc/solve1.hip, 0: the general path"""
        return lib().hipsdp_solve_path(self.h)

    def solve1_trace(self, rows=0):
        out = np.zeros(64)
        hist = np.zeros((max(rows, 1), 16))
        _chk(lib().hipsdp_solve1_trace(self.h, _dp(out), rows, _dp(hist) if rows else None), "hipsdp_solve1_trace")
        return out, hist[:rows]

    def y(self):
        out = np.zeros(self.m)
        _chk(lib().hipsdp_get_y(self.h, _dp(out)), "hipsdp_get_y")
        return out

    def X(self, k):
        out = np.zeros((self.ns[k], self.ns[k]))
        _chk(lib().hipsdp_get_X(self.h, k, _dp(out)), "hipsdp_get_X")
        return out

    def Z(self, k):
        out = np.zeros((self.ns[k], self.ns[k]))
        _chk(lib().hipsdp_get_Z(self.h, k, _dp(out)), "hipsdp_get_Z")
        return out

    def lp(self):
        x = np.zeros(self.q)
        z = np.zeros(self.q)
        _chk(lib().hipsdp_get_lp(self.h, _dp(x), _dp(z)), "hipsdp_get_lp")
        return x, z

    def preoptimal(self):
        """(y, [X_k], x) of the preoptimal iterate of the last solve (params preoptgap > 0) or None"""
        avail = C.c_int(0)
        y = np.zeros(max(1, self.m))
        x = np.zeros(max(1, self.q))
        _chk(lib().hipsdp_get_preoptimal(self.h, C.byref(avail), _dp(y), _dp(x)), "hipsdp_get_preoptimal")
        if not avail.value:
            return None
        Xs = []
        for k, n in enumerate(self.ns):
            X = np.zeros((n, n))
            _chk(lib().hipsdp_get_preoptimal_X(self.h, k, _dp(X)), "hipsdp_get_preoptimal_X")
            Xs.append(X)
        return y[:self.m], Xs, x[:self.q]

    def check_y(self, y, tol=0.0):
        """lambda_min of Z(y) per block (tol > 0: blocks above 64 rows report the certified bound -0.999 tol when it holds)"""
        y = _f64(y)
        lmin = np.zeros(max(1, len(self.ns)))
        viol = C.c_double(0.0)
        if tol > 0.0:
            _chk(lib().hipsdp_check_y_tol(self.h, _dp(y), C.c_double(tol), _dp(lmin), C.byref(viol)), "hipsdp_check_y_tol")
        else:
            _chk(lib().hipsdp_check_y(self.h, _dp(y), _dp(lmin), C.byref(viol)), "hipsdp_check_y")
        return lmin[:len(self.ns)], viol.value

    def eigencuts(self, block, y, tol, maxcuts):
        """eigenvector cuts of one block at y: (eigvals[k], coefs[k, m], lhs[k], vecs[k, n]); cut c: coefs[c] @ y >= lhs[c]"""
        y = _f64(y)
        m, n = len(y), self.ns[block]
        k = C.c_int(0)
        ev = np.zeros(max(1, maxcuts))
        co = np.zeros((max(1, maxcuts), max(1, m)))
        lh = np.zeros(max(1, maxcuts))
        ve = np.zeros((max(1, maxcuts), n))
        _chk(lib().hipsdp_eigencuts(self.h, block, _dp(y), C.c_double(tol), maxcuts, C.byref(k), _dp(ev), _dp(co), _dp(lh), _dp(ve)),
             "hipsdp_eigencuts")
        return ev[:k.value], co[:k.value, :m], lh[:k.value], ve[:k.value]

# ---- unit-level host-buffer kernels ---------------------------------------------------------------------------------

def dgemm(A, B, layA=0, layB=1, alpha=1.0, beta=0.0, Cin=None, lower_only=False, splitk=0, device=0):
    """row-major C = alpha op(A) op(B) + beta C.  layA = 0: A is [M, K]; 1: A is [K, M].  layB = 0: B is [N, K]; 1: [K, N]."""
    A = _f64(A)
    B = _f64(B)
    M, K = (A.shape if layA == 0 else A.shape[::-1])
    N = B.shape[0] if layB == 0 else B.shape[1]
    Cout = np.zeros((M, N)) if Cin is None else _f64(Cin).copy()
    _chk(lib().hipsdp_dgemm(device, layA, layB, M, N, K, C.c_double(alpha), _dp(A), C.c_longlong(A.shape[1]), _dp(B),
                            C.c_longlong(B.shape[1]), C.c_double(beta), _dp(Cout), C.c_longlong(N), int(lower_only), splitk),
         "hipsdp_dgemm")
    return Cout


def schur_dense(A, X, Zinv, ws_gbytes=0.0, device=0):
    A = _f64(A)
    m1, n = A.shape[0], A.shape[1]
    X = _f64(X)
    Zinv = _f64(Zinv)
    Mx = np.zeros((m1, m1))
    _chk(ulib().hipsdp_schur_dense(device, m1, n, _dp(A), _dp(X), _dp(Zinv), _dp(Mx), C.c_double(ws_gbytes)),
         "hipsdp_schur_dense")
    return Mx


def schur_sparse_unit(n, m, coo, X, Zinv, device=0):
    """(m + 1) x (m + 1) array whose lower triangle of rows / columns 1 .. m holds tr(A_i X A_j Zinv) as csrc/sparse.hip assembles it"""
    var, row, col, val = coo
    var = np.ascontiguousarray(var, dtype=np.int32); row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32); val = _f64(val)
    X = _f64(X); Zinv = _f64(Zinv)
    Mx = np.zeros((m + 1, m + 1))
    _chk(ulib().hipsdp_schur_sparse_unit(device, n, m, C.c_longlong(len(val)), _ip(var), _ip(row), _ip(col), _dp(val), _dp(X), _dp(Zinv), _dp(Mx)),
         "hipsdp_schur_sparse_unit")
    return Mx


def schur_w(A, X, Z, device=0):
    A = _f64(A)
    m1, n = A.shape[0], A.shape[1]
    X = _f64(X)
    Z = _f64(Z)
    Mx = np.zeros((m1, m1))
    _chk(ulib().hipsdp_schur_w(device, m1, n, _dp(A), _dp(X), _dp(Z), _dp(Mx)), "hipsdp_schur_w")
    return Mx


def dgemm_selfcheck(M, N, K, layB=1, batch=1, splitk=1, flags=0, beta=0.0, device=0):
    """both GEMM kernels on the same device-generated operands -> (used_v2, number of elements of C differing in any bit)"""
    used = C.c_int(0)
    nd = C.c_longlong(0)
    _chk(ulib().hipsdp_dgemm_selfcheck(device, M, N, K, layB, batch, splitk, flags, C.c_double(beta), C.byref(used), C.byref(nd)),
         "hipsdp_dgemm_selfcheck")
    return used.value, nd.value


def dgemm_selfcheck2(M, N, K, layB=1, batch=1, splitk=1, flags=0, alpha=1.0, beta=0.0, reps=0, device=0):
    """the tile kernel alone against the default dispatch (persistent tile kernel, strip kernel) on the same device-generated
    operands -> (used bits: 1 persistent tile kernel, 2 strip kernel; differing elements; ms tile; ms default)"""
    used = C.c_int(0)
    nd = C.c_longlong(0)
    t0, t1 = C.c_double(0.0), C.c_double(0.0)
    _chk(ulib().hipsdp_dgemm_selfcheck2(device, M, N, K, layB, batch, splitk, flags, C.c_double(alpha), C.c_double(beta), reps,
                                       C.byref(used), C.byref(nd), C.byref(t0), C.byref(t1)), "hipsdp_dgemm_selfcheck2")
    return used.value, nd.value, t0.value, t1.value


def dgemm_selfcheck3(M, N, K, layB=1, batch=1, splitk=1, flags=0, alpha=1.0, beta=0.0, reps=0, device=0):
    """as dgemm_selfcheck2, with the largest absolute difference of the two results and the number of elements that differ between
    two runs of the default dispatch -> (used bits, differing elements, max |difference|, not reproduced, ms tile, ms default)"""
    used = C.c_int(0)
    nd, nr = C.c_longlong(0), C.c_longlong(0)
    md, t0, t1 = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    _chk(ulib().hipsdp_dgemm_selfcheck3(device, M, N, K, layB, batch, splitk, flags, C.c_double(alpha), C.c_double(beta), reps,
                                       C.byref(used), C.byref(nd), C.byref(md), C.byref(nr), C.byref(t0), C.byref(t1)), "hipsdp_dgemm_selfcheck3")
    return used.value, nd.value, md.value, nr.value, t0.value, t1.value


def gram_selfcheck(M, K, reps=0, device=0):
    """W W^T on the lower triangle through the K-sliced tile kernels and through the Gram kernel (csrc/gram.hip)
    -> (Gram kernel took it, max |difference| over the lower triangle, elements not reproduced by a second run, ms tile path, ms Gram kernel)"""
    used = C.c_int(0)
    nr = C.c_longlong(0)
    md, t0, t1 = C.c_double(0.0), C.c_double(0.0), C.c_double(0.0)
    _chk(ulib().hipsdp_gram_selfcheck(device, M, C.c_longlong(K), reps, C.byref(used), C.byref(md), C.byref(nr), C.byref(t0), C.byref(t1)),
         "hipsdp_gram_selfcheck")
    return used.value, md.value, nr.value, t0.value, t1.value


def potrf(A, device=0):
    L = _f64(A).copy()
    fail = C.c_int(0)
    _chk(ulib().hipsdp_potrf(device, L.shape[0], _dp(L), C.byref(fail)), "hipsdp_potrf")
    return np.tril(L), fail.value


def potrf_ex(A, psd=False, v1=False, device=0):
    """blocked Cholesky in either form -> (L with the untouched upper part as stored, dinv, regmask, fail)"""
    L = _f64(A).copy()
    n = L.shape[0]
    dinv = np.zeros(((n + 63) // 64) * 4096)
    mask = np.zeros(n, dtype=np.int32)
    fail = C.c_int(0)
    _chk(ulib().hipsdp_potrf_ex(device, n, _dp(L), int(psd), int(v1), _dp(dinv), _ip(mask), C.byref(fail)), "hipsdp_potrf_ex")
    return L, dinv, mask, fail.value


def potrs(A, rhs, device=0):
    A = _f64(A)
    r = _f64(rhs).copy()
    r2 = r.reshape(-1, A.shape[0])
    _chk(ulib().hipsdp_potrs(device, A.shape[0], _dp(A), r2.shape[0], _dp(r2)), "hipsdp_potrs")
    return r2.reshape(r.shape)


def trtri(A, device=0):
    A = _f64(A)
    Li = np.zeros_like(A)
    _chk(ulib().hipsdp_trtri(device, A.shape[0], _dp(A), _dp(Li)), "hipsdp_trtri")
    return Li


def lambda_min(W, steps=0, device=0):
    W = _f64(W)
    th = C.c_double(0.0)
    rs = C.c_double(0.0)
    _chk(ulib().hipsdp_lambda_min(device, W.shape[0], _dp(W), steps, C.byref(th), C.byref(rs)), "hipsdp_lambda_min")
    return th.value, rs.value


def syev(A, device=0):
    A = _f64(A)
    n = A.shape[0]
    lam = np.zeros(n)
    V = np.zeros((n, n))
    _chk(lib().hipsdp_syev(device, n, _dp(A), _dp(lam), _dp(V)), "hipsdp_syev")
    return lam, V


def gemv_n(A, V, device=0):
    A = _f64(A)
    V = _f64(V).reshape(-1, A.shape[1])
    out = np.zeros((V.shape[0], A.shape[0]))
    _chk(lib().hipsdp_gemv_n(device, A.shape[0], C.c_longlong(A.shape[1]), _dp(A), V.shape[0], _dp(V), _dp(out)),
         "hipsdp_gemv_n")
    return out


def gemv_t(A, coef, device=0):
    A = _f64(A)
    coef = _f64(coef)
    out = np.zeros(A.shape[1])
    _chk(lib().hipsdp_gemv_t(device, A.shape[0], C.c_longlong(A.shape[1]), _dp(A), _dp(coef), _dp(out)), "hipsdp_gemv_t")
    return out


def psd_project(n, row, col, val, minev, epsilon=1e-9, mode=0, device=0):
    """fused PSD projection chain (hipsdp_psd_project): returns (row, col, val) of the upper triangle, row-major order"""
    row = np.ascontiguousarray(row, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = _f64(val)
    cap = n * (n + 1) // 2
    ro = np.zeros(cap, dtype=np.int32)
    co = np.zeros(cap, dtype=np.int32)
    vo = np.zeros(cap)
    k = C.c_int(0)
    _chk(lib().hipsdp_psd_project(device, n, len(val), _ip(row), _ip(col), _dp(val), C.c_double(minev), C.c_double(epsilon), mode,
                                  cap, C.byref(k), _ip(ro), _ip(co), _dp(vo)), "hipsdp_psd_project")
    return ro[:k.value], co[:k.value], vo[:k.value]
