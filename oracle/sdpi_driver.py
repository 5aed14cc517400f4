"""sdpi_driver.py - TEST INFRASTRUCTURE.  Restatement of the solve driver of the reference's solver-independent SDPI,
SCIPsdpiSolve (src/sdpi/sdpi.c:3123-3640), i.e. the immediate caller of the backend entry points this repository replaces:

   sdpi.c:3190-3275   preparation, all-fixed / bound-conflict exits                      (sdpi_prepare.prepare)
   sdpi.c:3301-3381   one-variable shortcut        -> solve_one_var_sdp  (src/sdpi/solveonevarsdp.c:160-366)
   sdpi.c:3389-3396   optional Slater check        -> slater_check       (sdpi.c:1518-1869)
   sdpi.c:3399-3420   SCIPsdpiSolverLoadAndSolve, statistics
   sdpi.c:3437-3620   penalty fallback: feasibility problem (Gamma = 1, no objective, r free), then the loop that raises
                      Gamma (factor (maxpenaltyparam / penaltyparam)^(1 / npenaltyincr)) or lowers the gap tolerance until the
                      penalty solution is feasible for the original problem

It drives any object with the interface of tests/sdpi_call.SdpiSolver (the ctypes view of libhipsdp.so's SCIPsdpiSolver*
functions) - that is what the gpu-marked tests do - or OracleBackend below (numpy IPM) on the CPU.  Only tests may import
this module."""
import copy
import math
import numpy as np

import sdpi_prepare

INF = 1e20
MIN_GAPTOL = 1e-10                  # sdpi.c:197
DEFAULT_PENALTYPARAM = 1e5          # sdpi.c:201
DEFAULT_MAXPENALTYPARAM = 1e10      # sdpi.c:202
DEFAULT_NPENALTYINCR = 8            # sdpi.c:203
DEFAULT_PENINFEASADJUST = 1.1       # relax_sdp.c:96
SLATER_INF, SLATER_NOINFO, SLATER_NOT, SLATER_HOLDS = -2, -1, 0, 1      # type_sdpi.h:104-110
PAR_GAPTOL = 1                      # type_sdpi.h:47-67


class SdpiResult:
    def __init__(self):
        self.solved = False          # sdpi->solved
        self.infeasible = False      # sdpi->infeasible
        self.allfixed = False
        self.penalty = False         # a penalty formulation was used
        self.onevar = None           # 'optimal' | 'infeasible' when the one-variable shortcut decided
        self.objval = None
        self.y = None
        self.dualslater = SLATER_NOINFO
        self.primalslater = SLATER_NOINFO
        self.niterations = 0
        self.nsdpcalls = 0
        self.npenaltysolves = 0
        self.penaltyparam_used = None
        self.bestbound = -INF


# ---- one-variable SDPs (solveonevarsdp.c:160-366) ---------------------------------------------------------------------
def _dense(n, entries):
    M = np.zeros((n, n))
    for (r, c, v) in entries:
        M[r, c] = v
        M[c, r] = v
    return M


def lmin_numpy(M):
    """smallest eigenvalue and its eigenvector: SCIPlapackComputeIthEigenvalue(.., i = 1) (lapack_interface.c:178-288)"""
    w, V = np.linalg.eigh(M)
    return float(w[0]), V[:, 0].copy()


def solve_one_var_sdp(obj, lb, ub, n, const_entries, var_entries, feastol, lmin=lmin_numpy, infinity=INF):
    """min obj * y  s.t.  y A - A_0 >= 0 (psd), lb <= y <= ub.  Returns (objval, optval, certificate) with objval None when the
    routine declines (infinite bound or negative objective: solveonevarsdp.c:205-213) and objval = infinity when infeasible.
    The minimal eigenvalue is concave in y; v^T A v at the eigenvector v is a supergradient."""
    if lb <= -infinity or ub >= infinity or obj < 0.0:
        return None, None, None
    A = _dense(n, var_entries)
    A0 = _dense(n, const_entries)

    def f(alpha):
        ev, vec = lmin(alpha * A - A0)
        return ev, vec, float(vec @ A @ vec)

    ev, vec, g = f(ub)
    if ev < -feastol and g > 0.0:
        return infinity, ub, g                     # increasing and still not psd at the upper bound
    ev, vec, g = f(lb)
    if ev >= -feastol:
        return obj * lb, lb, 0.0
    if g <= 0.0:
        return infinity, lb, g
    mu = lb
    while ev < -feastol and g > 0.0:
        mu = mu - (feastol / 2.0 + ev) / g         # where the supergradient inequality reaches -feastol / 2
        if mu > ub:
            break
        ev, vec, g = f(mu)
    if ev < -feastol:
        return infinity, mu, g
    return obj * mu, mu, g


# ---- Slater check (sdpi.c:1518-1869) ------------------------------------------------------------------------------------
def primal_slater_arguments(P, eps=sdpi_prepare.EPS):
    """the argument set of the second backend call of checkSlaterCondition: constant part dropped, every finite variable bound
    and LP side set to 0, one extra row  sum_j tr(A_j) y_j >= 1  over the non-fixed variables with a nonzero trace
    (sdpi.c:1650-1789).  Fixedness is judged on the ORIGINAL prepared bounds (isFixed(sdpi, v))."""
    prob = P.prob
    Q = copy.copy(P)
    fixed = [P.ub[v] - P.lb[v] <= eps for v in range(prob.nvars)]
    tr = np.zeros(prob.nvars)
    for blk in prob.blocks:
        for var, ents in blk['vars'].items():
            if not fixed[var]:
                tr[var] += sum(v for (r, c, v) in ents if r == c)
    extra = [(v, tr[v]) for v in range(prob.nvars) if not fixed[v] and abs(tr[v]) > eps]
    nlp = P.nlpcons
    lhs = [(P.lplhs[i] if P.lplhs[i] <= -INF else 0.0) if P.lpindchanges[i] >= 0 else 0.0 for i in range(nlp)]
    rhs = [(P.lprhs[i] if P.lprhs[i] >= INF else 0.0) if P.lpindchanges[i] >= 0 else 0.0 for i in range(nlp)]
    nremoved = sum(1 for i in range(nlp) if P.lpindchanges[i] < 0)
    Q.lpindchanges = np.array(list(P.lpindchanges[:nlp]) + [nremoved], dtype=np.int32)
    Q.lplhs = np.array(lhs + [1.0], dtype=np.float64)
    Q.lprhs = np.array(rhs + [INF], dtype=np.float64)
    Q.lpbeg = np.array(list(P.lpbeg[:nlp]) + [P.lpnnonz], dtype=np.int32)
    Q.lpind = np.array(list(P.lpind[:P.lpnnonz]) + [v for v, _ in extra] + [0], dtype=np.int32)
    Q.lpval = np.array(list(P.lpval[:P.lpnnonz]) + [t for _, t in extra] + [0.0], dtype=np.float64)
    Q.lpnnonz = P.lpnnonz + len(extra)
    Q.nlpcons = nlp + 1 if extra else nlp
    lb = np.array(P.lb, dtype=np.float64)
    ub = np.array(P.ub, dtype=np.float64)
    nfinite = 0
    for v in range(prob.nvars):
        if lb[v] > -INF:
            lb[v] = 0.0
            nfinite += 1
        if ub[v] < INF:
            ub[v] = 0.0
            nfinite += 1
    Q.lb, Q.ub = lb, ub
    Q.sdpconst = [[] for _ in prob.blocks]
    return Q, nfinite == 2 * prob.nvars


def slater_check(backend, P, feastol):
    """returns (dualslater, primalslater, calls).  Dual: min r s.t. sum A_j y_j - A_0 + r I >= 0, rows + r >= lhs, r free
    (penalty call (1.0, FALSE, FALSE)); an interior point exists iff the optimum is < -feastol."""
    calls = 0
    rc, feasorig, pb = backend.solve(P, penaltyparam=1.0, withobj=False, rbound=False)
    assert rc == 1, rc
    calls += 1
    if backend.flag("IsOptimal"):
        objval = backend.objval()
        dual = SLATER_HOLDS if objval < -feastol else (SLATER_NOT if objval < feastol else SLATER_INF)
    elif backend.flag("IsDualUnbounded"):
        dual = SLATER_HOLDS
    elif backend.flag("IsDualInfeasible"):
        dual = SLATER_NOT
    else:
        dual = SLATER_NOINFO
    Q, allbounded = primal_slater_arguments(P)
    if allbounded:
        return dual, SLATER_HOLDS, calls           # sdpi.c:1771-1781
    rc, _, _ = backend.solve(Q)
    assert rc == 1, rc
    calls += 1
    if backend.flag("IsOptimal"):
        primal = SLATER_NOT if backend.objval() > -feastol else SLATER_HOLDS
    elif backend.flag("IsDualUnbounded"):
        primal = SLATER_NOT
    elif backend.flag("IsPrimalUnbounded"):
        primal = SLATER_HOLDS
    else:
        primal = SLATER_NOINFO
    return dual, primal, calls


# ---- SCIPsdpiSolve ------------------------------------------------------------------------------------------------------
def sdpi_solve(backend, prob, feastol=1e-6, gaptol=1e-5, penaltyparam=DEFAULT_PENALTYPARAM, maxpenaltyparam=DEFAULT_MAXPENALTYPARAM,
               npenaltyincr=DEFAULT_NPENALTYINCR, peninfeasadjust=DEFAULT_PENINFEASADJUST, slatercheck=False,
               enforceslatercheck=False, start=None, lmin=lmin_numpy, onevar_shortcut=True, force_penalty=False, eps=sdpi_prepare.EPS,
               prepared=None):
    """force_penalty: treat the first solve as unacceptable (lets tests walk the fallback on well-behaved problems)."""
    R = SdpiResult()
    P = prepared if prepared is not None else sdpi_prepare.prepare(prob, eps)
    R.prepared = P
    if P.status == 'infeasible':
        R.solved, R.infeasible = True, True
        return R
    if P.status == 'allfixed':
        R.solved, R.allfixed = True, True
        R.y = np.array(P.lb, dtype=np.float64)
        R.objval = float(prob.obj @ R.y)
        return R
    active = [v for v in range(prob.nvars) if P.ub[v] - P.lb[v] > eps]
    fixedobj = float(sum(prob.obj[v] * P.lb[v] for v in range(prob.nvars) if v not in active))

    # one active variable and at most one block (sdpi.c:3301-3381)
    nactivelp = sum(1 for i in range(P.nlpcons) if P.lpindchanges[i] >= 0)
    if onevar_shortcut and len(active) == 1 and len(prob.blocks) <= 1 and nactivelp == 0:
        v = active[0]
        if not prob.blocks:
            if P.lb[v] > -INF and P.ub[v] < INF:
                opt = P.lb[v] if prob.obj[v] >= 0.0 else P.ub[v]
                R.solved, R.onevar = True, 'optimal'
                R.y = np.array(P.lb, dtype=np.float64)
                R.y[v] = opt
                R.objval = prob.obj[v] * opt + fixedobj
                return R
        else:
            blk = prob.blocks[0]
            objval, optval, cert = solve_one_var_sdp(prob.obj[v], P.lb[v], P.ub[v], blk['n'], P.sdpconst[0], blk['vars'].get(v, []),
                                                     feastol, lmin)
            if objval is not None:
                R.solved = True
                R.y = np.array(P.lb, dtype=np.float64)
                R.y[v] = optval
                if objval >= INF:
                    R.onevar, R.infeasible = 'infeasible', True
                else:
                    R.onevar, R.objval = 'optimal', objval + fixedobj
                return R

    def account():
        R.niterations += backend.iterations()
        R.nsdpcalls += backend.sdpcalls()

    if slatercheck:
        R.dualslater, R.primalslater, _ = slater_check(backend, P, feastol)

    rc, _, _ = backend.solve(P, start=start)
    assert rc == 1, rc
    R.solved = True
    account()

    acceptable = backend.flag("IsAcceptable") and not force_penalty
    if not acceptable and not backend.flag("IsTimelimExc"):
        # feasibility problem first (sdpi.c:3452-3489)
        rc, feasorig, _ = backend.solve(P, penaltyparam=1.0, withobj=False, rbound=False, start=start)
        assert rc == 1, rc
        account()
        R.npenaltysolves += 1
        objval = backend.objval() if backend.flag("WasSolved") else -INF
        if (backend.flag("IsOptimal") and objval > peninfeasadjust * max(feastol, gaptol)) or \
           (backend.flag("WasSolved") and backend.flag("IsDualInfeasible")):
            R.penalty, R.infeasible = True, True
        else:
            pen = penaltyparam
            gt = gaptol
            if npenaltyincr > 0:
                penfact = (maxpenaltyparam / penaltyparam) ** (1.0 / npenaltyincr)
                gapfact = (MIN_GAPTOL / gaptol) ** (1.0 / npenaltyincr)
            else:
                penfact = 2.0 * maxpenaltyparam / penaltyparam
                gapfact = 0.5 * MIN_GAPTOL / gaptol
            feasorig = False
            while (not backend.flag("IsAcceptable") or not feasorig) and pen < maxpenaltyparam + eps and gt > 0.99 * MIN_GAPTOL \
                    and not backend.flag("IsTimelimExc"):
                rc, feasorig, penaltybound = backend.solve(P, penaltyparam=pen, withobj=True, rbound=True, start=start)
                assert rc == 1, rc
                account()
                R.npenaltysolves += 1
                R.penaltyparam_used = pen
                if not backend.flag("IsAcceptable"):
                    pen *= penfact
                    continue
                bound = backend.objval()
                if bound > R.bestbound + gaptol:
                    R.bestbound = bound
                if not feasorig:
                    if penaltybound:
                        pen *= penfact
                    else:
                        gt *= gapfact
                        backend.set_real(PAR_GAPTOL, gt)
            if gt < gaptol:
                backend.set_real(PAR_GAPTOL, gaptol)
            R.penalty = True
            R.solved = bool(backend.flag("IsAcceptable") and feasorig)
            if not R.solved and enforceslatercheck:
                R.dualslater, R.primalslater, _ = slater_check(backend, P, feastol)
            if not R.solved:
                return R
    elif not acceptable:
        R.solved = False
        return R

    if R.infeasible:
        return R
    if backend.flag("IsDualInfeasible"):
        R.infeasible = True
        return R
    if backend.flag("IsOptimal") or (R.penalty and backend.flag("IsAcceptable")):
        rc, R.objval, y = backend.dual_sol()
        assert rc == 1, rc
        R.y = np.array(y, dtype=np.float64)
    return R


# ---- a CPU backend with the same observable interface (numpy IPM of oracle/ipm_ref.py) -----------------------------------
class OracleBackend:
    """the subset of tests/sdpi_call.SdpiSolver the driver uses, served by ipm_ref.hsd_solve on the problem
    sdpi_prepare.to_core builds.  Penalty post-processing as sdpisolver_dsdp.c:1655-1734 / sdpisolver_sdpa.cpp:1846-1857.
    ladder=True additionally restates what sdpisolver_hip.c does around the engine: the settings ladder of the backend
    (sdpisolver_sdpa.cpp:1415-1449 start settings, :1698-1795 retries: not acceptable and no penalty formulation -> solve again
    with medium, then stable settings) and, on every rung, the tolerance re-solve loop (sdpisolver_dsdp.c:1527-1606: y checked
    against feastol with exact eigenvalues as sdpsolchecker.c:201-257, |pobj - dobj| against gaptol; a violation tightens the
    solver's own tolerance by 0.1 down to 1e-10)."""

    UNSOLVED, PENALTY, FAST, MEDIUM, STABLE = -1, 0, 1, 2, 3

    def __init__(self, feastol=1e-6, gaptol=1e-6, ladder=False, solverfeastol=None):
        self.feastol, self.gaptol = feastol, gaptol
        self.solverfeastol = feastol if solverfeastol is None else solverfeastol
        self.ladder = ladder
        self.res = None
        self._calls = 0
        self._iters = 0
        self.usedsetting = self.UNSOLVED

    def set_real(self, par, val):
        if par == PAR_GAPTOL:
            self.gaptol = val
        elif par == 2:
            self.feastol = val
        elif par == 3:
            self.solverfeastol = val
        return 1

    def _check_y(self, core, y):
        """min over blocks of lambda_min(sum_i A_i y_i - A_0), largest LP row violation"""
        lmin = np.inf
        for A in core.blocks:
            Z = np.tensordot(np.concatenate([[-1.0], y]), A, axes=(0, 0))
            lmin = min(lmin, float(np.linalg.eigvalsh(0.5 * (Z + Z.T))[0]))
        viol = float(max(0.0, -np.min(core.D @ y - core.c))) if core.q else 0.0
        return lmin, viol

    def _rung(self, ipm_ref, core, level):
        feast, gapt = self.solverfeastol, (self.gaptol if self.ladder else min(self.gaptol, 1e-6))
        pab = self.feastol if self.feastol > feast else 0.0
        res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=gapt, feastol=feast, settings=level, pabstol=pab if self.ladder else 0.0))
        self._calls += 1
        self._iters += res.iterations
        while self.ladder and res.status == ipm_ref.STATUS_OPTIMAL and not self.penalty:
            lmin, viol = self._check_y(core, res.y)
            infeasible = lmin < -self.feastol or viol > self.feastol
            again = False
            if infeasible:
                feast *= 0.1
                again = feast >= 1e-10
            if abs(res.pobj - res.dobj) >= self.gaptol:
                infeasible = True
                gapt *= 0.1
                again = again or gapt >= 1e-10
            if not again:
                if infeasible:
                    res.status = ipm_ref.STATUS_NUMERIC
                break
            pab = self.feastol if self.feastol > feast else 0.0
            res = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=gapt, feastol=feast, settings=level, pabstol=pab))
            self._calls += 1
            self._iters += res.iterations
        return res

    def solve(self, P, penaltyparam=0.0, withobj=True, rbound=True, timelimit=1e20, clock=None, start=None, startsettings=-1):
        import ipm_ref
        b, blocks, D, c, maps = sdpi_prepare.to_core(P, penaltyparam, withobj, rbound)
        self.P, self.maps = P, maps
        self.penalty = penaltyparam > sdpi_prepare.EPS
        self.withobj = withobj
        core = ipm_ref.CoreProblem(b, blocks, D, c)
        self.ipm = ipm_ref
        self._calls = 0
        self._iters = 0
        acceptable = (ipm_ref.STATUS_OPTIMAL, ipm_ref.STATUS_DINF, ipm_ref.STATUS_DUNB, ipm_ref.STATUS_PDINF)
        if not self.ladder:
            level = 0
        elif self.penalty or startsettings in (self.STABLE, self.PENALTY):
            level = 2
        elif startsettings == self.MEDIUM:
            level = 1
        else:
            level = 0
        while True:
            self.res = self._rung(ipm_ref, core, level)
            self.usedsetting = self.PENALTY if self.penalty else (self.FAST, self.MEDIUM, self.STABLE)[level]
            if not self.ladder or self.penalty or level >= 2 or self.res.status in acceptable:
                break
            level += 1
        feasorig = penaltybound = False
        self.feasorig = False
        if self.penalty and self.res.status == ipm_ref.STATUS_OPTIMAL:
            r = self.res.y[-1]
            feasorig = bool(r < self.feastol)
            if withobj:
                self.feasorig = feasorig
            if not feasorig:
                nrows = len(c) - sum(1 for v in maps["active"] for s in (P.lb[v] > -INF, P.ub[v] < INF) if s) - (1 if rbound else 0)
                trace = sum(float(np.trace(X)) for X in self.res.X) + float(np.sum(self.res.x[:nrows]))
                penaltybound = bool((penaltyparam - trace) / penaltyparam < 1e-3)
        return 1, feasorig, penaltybound

    def flag(self, name):
        s, I = self.res.status, self.ipm
        return {
            "WasSolved": True,
            "IsTimelimExc": False,
            "IsAcceptable": s in (I.STATUS_OPTIMAL, I.STATUS_DINF, I.STATUS_DUNB, I.STATUS_PDINF),
            "IsOptimal": s == I.STATUS_OPTIMAL,
            "IsDualInfeasible": s in (I.STATUS_DINF, I.STATUS_PDINF),
            "IsDualUnbounded": s == I.STATUS_DUNB,
            "IsPrimalUnbounded": s == I.STATUS_DINF,
            "IsPrimalInfeasible": s in (I.STATUS_DUNB, I.STATUS_PDINF),
        }[name]

    def settings_used(self):
        return self.usedsetting if self.flag("IsAcceptable") else self.UNSOLVED

    def _y(self):
        y = np.array(self.P.lb, dtype=np.float64)
        for k, v in enumerate(self.maps["active"]):
            y[v] = self.res.y[k]
        return y

    def dual_sol(self):
        return 1, self.objval(), self._y()

    def objval(self):
        if self.penalty and not self.feasorig:
            return float(self.res.dobj)           # the solver's own objective (sdpisolver_dsdp.c:2148-2190)
        return float(self.P.prob.obj @ self._y())

    def max_primal_entry(self):
        """largest entry of X over the SDP blocks and the LP multipliers (sdpisolver_sdpa.cpp:3090-3125)"""
        m = 0.0
        for X in self.res.X:
            m = max(m, float(np.max(X))) if X.size else m
        if self.res.x.size:
            m = max(m, float(np.max(self.res.x)))
        return m

    def iterations(self):
        return int(self._iters)

    def sdpcalls(self):
        return self._calls
