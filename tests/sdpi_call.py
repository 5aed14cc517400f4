"""sdpi_call.py - ctypes marshalling of the 48-argument SCIPsdpiSolverLoadAndSolveWithPenalty call
(include/sdpisolver_hip.h; reference signature src/sdpi/sdpisolver.h:258-322) for the tests.  The arguments come from
tests/harness/sdpi_prepare.prepare(), i.e. they look like what sdpi.c:3399-3405 passes."""
import ctypes as C
import time
import numpy as np

SCIP_OKAY = 1
SCIP_LPERROR = -6
SCIP_PARAMETERUNKNOWN = -12
UNSOLVED, PENALTY, FAST, MEDIUM, STABLE = -1, 0, 1, 2, 3        # SCIP_SDPSOLVERSETTING (type_sdpi.h:69-77)

PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)
PPD = C.POINTER(PD)
PPI = C.POINTER(PI)


def _pi(a):
    return a.ctypes.data_as(PI)


def _pd(a):
    return a.ctypes.data_as(PD)


class SdpiSolver:
    def __init__(self, lib):
        self.lib = lib
        self.h = C.c_void_p()
        lib.SCIPsdpiSolverGetSolverName.restype = C.c_char_p
        lib.SCIPsdpiSolverGetSolverDesc.restype = C.c_char_p
        lib.SCIPsdpiSolverInfinity.restype = C.c_double
        lib.SCIPsdpiSolverGetMaxPrimalEntry.restype = C.c_double
        lib.SCIPsdpiSolverGetSolverPointer.restype = C.c_void_p
        rc = lib.SCIPsdpiSolverCreate(C.byref(self.h), None, None, None)
        assert rc == SCIP_OKAY
        self._keep = []
        self._blockcache = {}

    def free(self):
        if self.h:
            assert self.lib.SCIPsdpiSolverFree(C.byref(self.h)) == SCIP_OKAY

    def set_real(self, par, val):
        return self.lib.SCIPsdpiSolverSetRealpar(self.h, par, C.c_double(val))

    def get_real(self, par):
        v = C.c_double(0.0)
        rc = self.lib.SCIPsdpiSolverGetRealpar(self.h, par, C.byref(v))
        return rc, v.value

    def set_int(self, par, val):
        return self.lib.SCIPsdpiSolverSetIntpar(self.h, par, val)

    def solve(self, P, penaltyparam=0.0, withobj=True, rbound=True, timelimit=1e20, clock=None, start=None, startsettings=UNSOLVED):
        """P: oracle.sdpi_prepare.Prepared.  Returns (retcode, feasorig, penaltybound).
        start: optional dict(y=[nvars], Z=[(rows, cols, vals)] * (nblocks + 1), X=likewise) in ORIGINAL indices, the last
        entry being the diagonal LP block (sdpisolver.h:160-173)."""
        prob = P.prob
        keep = []
        nb = len(prob.blocks)
        obj = np.ascontiguousarray(prob.obj, dtype=np.float64)
        lb = np.ascontiguousarray(P.lb, dtype=np.float64)
        ub = np.ascontiguousarray(P.ub, dtype=np.float64)
        sizes = np.array([b['n'] for b in prob.blocks] or [0], dtype=np.int32)
        nblockvars = np.array([len(b['vars']) for b in prob.blocks] or [0], dtype=np.int32)
        keep += [obj, lb, ub, sizes, nblockvars]

        constn = np.array([len(c) for c in P.sdpconst] or [0], dtype=np.int32)
        crow = (PI * max(nb, 1))()
        ccol = (PI * max(nb, 1))()
        cval = (PD * max(nb, 1))()
        nvarnonz = (PI * max(nb, 1))()
        sdpvar = (PI * max(nb, 1))()
        srow = (PPI * max(nb, 1))()
        scol = (PPI * max(nb, 1))()
        sval = (PPD * max(nb, 1))()
        indch = (PI * max(nb, 1))()
        sdpnnonz = 0
        for b, blk in enumerate(prob.blocks):
            ce = P.sdpconst[b]
            r = np.array([e[0] for e in ce] or [0], dtype=np.int32)
            c = np.array([e[1] for e in ce] or [0], dtype=np.int32)
            v = np.array([e[2] for e in ce] or [0.0], dtype=np.float64)
            keep += [r, c, v]
            crow[b], ccol[b], cval[b] = _pi(r), _pi(c), _pd(v)
            # the arrays of a block are built once per block object and re-used (same addresses on every call, like sdpi.c)
            ck = id(blk['vars'])
            if ck not in self._blockcache:
                vars_ = sorted(blk['vars'].keys())
                nn = np.array([len(blk['vars'][x]) for x in vars_] or [0], dtype=np.int32)
                vv = np.array(vars_ or [0], dtype=np.int32)
                k = max(len(vars_), 1)
                rr = (PI * k)()
                cc = (PI * k)()
                va = (PD * k)()
                held = [blk['vars'], nn, vv]
                cnt = 0
                for j, x in enumerate(vars_):
                    ents = blk['vars'][x]
                    er = np.array([e[0] for e in ents], dtype=np.int32)
                    ec = np.array([e[1] for e in ents], dtype=np.int32)
                    ev = np.array([e[2] for e in ents], dtype=np.float64)
                    held += [er, ec, ev]
                    rr[j], cc[j], va[j] = _pi(er), _pi(ec), _pd(ev)
                    cnt += len(ents)
                self._blockcache[ck] = (nn, vv, rr, cc, va, cnt, held)
            nn, vv, rr, cc, va, cnt, held = self._blockcache[ck]
            sdpnnonz += cnt
            nvarnonz[b], sdpvar[b] = _pi(nn), _pi(vv)
            srow[b], scol[b], sval[b] = rr, cc, va
            indch[b] = _pi(P.indchanges[b])
        nrem = np.array(P.nremovedinds or [0], dtype=np.int32)
        bic = np.array(P.blockindchanges or [0], dtype=np.int32)
        keep += [constn, nrem, bic, crow, ccol, cval, nvarnonz, sdpvar, srow, scol, sval, indch]
        feasorig = C.c_uint(0)
        penaltybound = C.c_uint(0)
        startargs = [None] * 9
        if start is not None:
            sy = np.ascontiguousarray(start['y'], dtype=np.float64)
            keep.append(sy)
            packed = []
            for key in ('Z', 'X'):
                nnz = np.array([len(t[0]) for t in start[key]], dtype=np.int32)
                rows = (PI * (nb + 1))()
                cols = (PI * (nb + 1))()
                vals = (PD * (nb + 1))()
                for k, (r, c, v) in enumerate(start[key]):
                    ra = np.array(list(r) or [0], dtype=np.int32)
                    ca = np.array(list(c) or [0], dtype=np.int32)
                    va = np.array(list(v) or [0.0], dtype=np.float64)
                    keep.extend([ra, ca, va])
                    rows[k], cols[k], vals[k] = _pi(ra), _pi(ca), _pd(va)
                keep.extend([nnz, rows, cols, vals])
                packed += [_pi(nnz), rows, cols, vals]
            startargs = [_pd(sy)] + packed
        self._keep = keep
        args = (
            self.h, C.c_double(penaltyparam), C.c_uint(1 if withobj else 0), C.c_uint(1 if rbound else 0),
            C.c_int(prob.nvars), _pd(obj), _pd(lb), _pd(ub),
            C.c_int(nb), _pi(sizes), _pi(nblockvars),
            C.c_int(int(constn.sum()) if nb else 0), _pi(constn), crow, ccol, cval,
            C.c_int(sdpnnonz), nvarnonz, sdpvar, srow, scol, sval,
            indch, _pi(nrem), _pi(bic), C.c_int(P.nremovedblocks),
            C.c_int(P.nlpcons), _pi(P.lpindchanges), _pd(P.lplhs), _pd(P.lprhs),
            C.c_int(P.lpnnonz), _pi(P.lpbeg), _pi(P.lpind), _pd(P.lpval),
            *startargs,
            C.c_int(startsettings), C.c_double(timelimit), clock,
            C.byref(feasorig), C.byref(penaltybound))
        # wall time of the call itself (what bench.py's node solves per second divide by): the numpy / ctypes marshalling above is
        # this harness's work, not the library's
        t0 = time.perf_counter()
        rc = self.lib.SCIPsdpiSolverLoadAndSolveWithPenalty(*args)
        self.last_call_seconds = time.perf_counter() - t0
        self._P = P
        return rc, bool(feasorig.value), bool(penaltybound.value)

    # ---- predicates / getters
    def flag(self, name):
        return bool(getattr(self.lib, "SCIPsdpiSolver" + name)(self.h))

    def internal_status(self):
        return self.lib.SCIPsdpiSolverGetInternalStatus(self.h)

    def sol_feasibility(self):
        p = C.c_uint(0)
        d = C.c_uint(0)
        rc = self.lib.SCIPsdpiSolverGetSolFeasibility(self.h, C.byref(p), C.byref(d))
        return rc, bool(p.value), bool(d.value)

    def dual_sol(self):
        n = self._P.prob.nvars
        y = np.zeros(n)
        o = C.c_double(0.0)
        rc = self.lib.SCIPsdpiSolverGetDualSol(self.h, C.byref(o), _pd(y))
        return rc, o.value, y

    def objval(self):
        o = C.c_double(0.0)
        rc = self.lib.SCIPsdpiSolverGetObjval(self.h, C.byref(o))
        assert rc == SCIP_OKAY, rc
        return o.value

    def bound_vars(self):
        n = self._P.prob.nvars
        l = np.zeros(n)
        u = np.zeros(n)
        rc = self.lib.SCIPsdpiSolverGetPrimalBoundVars(self.h, _pd(l), _pd(u))
        return rc, l, u

    def lp_sides(self):
        n = self._P.nlpcons
        l = np.zeros(max(n, 1))
        u = np.zeros(max(n, 1))
        P = self._P
        rc = self.lib.SCIPsdpiSolverGetPrimalLPSides(self.h, n, _pi(P.lpindchanges), _pd(P.lplhs), _pd(P.lprhs), _pd(l), _pd(u))
        return rc, l[:n], u[:n]

    def primal_solution_matrix(self):
        P = self._P
        prob = P.prob
        nb = len(prob.blocks)
        sizes = np.array([b['n'] for b in prob.blocks], dtype=np.int32)
        mats = [np.zeros(b['n'] * b['n']) for b in prob.blocks]
        pm = (PD * nb)()
        ic = (PI * nb)()
        for b in range(nb):
            pm[b] = _pd(mats[b])
            ic[b] = _pi(P.indchanges[b])
        nrem = np.array(P.nremovedinds, dtype=np.int32)
        bic = np.array(P.blockindchanges, dtype=np.int32)
        rc = self.lib.SCIPsdpiSolverGetPrimalSolutionMatrix(self.h, nb, _pi(sizes), ic, _pi(nrem), _pi(bic), pm)
        return rc, [m.reshape(b['n'], b['n']) for m, b in zip(mats, prob.blocks)]

    def primal_matrix_sparse(self):
        nb = len(self._P.prob.blocks) + 1
        cnt = np.zeros(nb, dtype=np.int32)
        rc = self.lib.SCIPsdpiSolverGetPrimalNonzeros(self.h, nb, _pi(cnt))
        if rc != SCIP_OKAY:
            return rc, None
        rows = [np.zeros(max(int(c), 1), dtype=np.int32) for c in cnt]
        cols = [np.zeros(max(int(c), 1), dtype=np.int32) for c in cnt]
        vals = [np.zeros(max(int(c), 1)) for c in cnt]
        pr = (PI * nb)()
        pc = (PI * nb)()
        pv = (PD * nb)()
        for b in range(nb):
            pr[b], pc[b], pv[b] = _pi(rows[b]), _pi(cols[b]), _pd(vals[b])
        cnt2 = cnt.copy()
        rc = self.lib.SCIPsdpiSolverGetPrimalMatrix(self.h, nb, _pi(cnt2), pr, pc, pv)
        return rc, [(rows[b][:cnt2[b]], cols[b][:cnt2[b]], vals[b][:cnt2[b]]) for b in range(nb)]

    def max_primal_entry(self):
        return float(self.lib.SCIPsdpiSolverGetMaxPrimalEntry(self.h))

    def iterations(self):
        v = C.c_int(0)
        self.lib.SCIPsdpiSolverGetIterations(self.h, C.byref(v))
        return v.value

    def sdpcalls(self):
        v = C.c_int(0)
        self.lib.SCIPsdpiSolverGetSdpCalls(self.h, C.byref(v))
        return v.value

    def opttime(self):
        v = C.c_double(0)
        self.lib.SCIPsdpiSolverGetTime(self.h, C.byref(v))
        return v.value

    def settings_used(self):
        v = C.c_int(0)
        self.lib.SCIPsdpiSolverSettingsUsed(self.h, C.byref(v))
        return v.value
