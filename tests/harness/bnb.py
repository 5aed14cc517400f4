"""bnb.py - TEST INFRASTRUCTURE.  Minimal best-bound-first branch-and-bound over node SDP relaxations, enough to reproduce the
MISDP optima of check/testset/short.solu on the reference's example instances WITHOUT SCIP (SURVEY.md section 8(f)-2,
BASELINE configs 3 and 5).  The node relaxation solver is pluggable: the HIP backend through the SCIPsdpiSolver* boundary
(tests/test_gpu_bnb.py) or the numpy oracle (CPU test).  Branching: most fractional integer variable, child with the
rounded-down bound first or second depending on the fraction.  No cuts, no heuristics, no presolve beyond
tests/harness/sdpi_prepare.py - this is a parity harness, not a solver."""
import math
import numpy as np
import sdpi_prepare
import sdpa_io

INF = 1e20


def instance_to_sdpi(inst, integrality=False):
    """SdpaInstance -> SdpiProblem (variables free; bounds come from single-variable LP rows, as in reader_sdpa.c).
    integrality=True hands the integrality flags on, which switches the coefficient tightening of the LP rows on
    (sdpi.c:812-1129) - tightened rows make more nodes lose their interior, so this belongs with the full SCIPsdpiSolve driver
    (penalty fallback), not with the plain single-call harness."""
    blocks = []
    for blk in inst.sdpblocks:
        vars_ = {}
        const = []
        for var, ents in blk['entries'].items():
            if var == 0:
                const = [(i, j, v) for (i, j, v) in ents]
            else:
                vars_[var - 1] = [(i, j, v) for (i, j, v) in ents]
        blocks.append(dict(n=blk['n'], vars=vars_, const=const))
    lp = []
    for row in inst.lprows:
        coefs = {var - 1: v for var, v in row.items() if var != 0}
        lp.append((row.get(0, 0.0), INF, coefs))          # sum d_ri y_i - d_r0 >= 0
    m = inst.m
    isint = [False] * m
    if integrality:
        for v in getattr(inst, 'intvars', []):
            isint[v] = True
    return sdpi_prepare.SdpiProblem(inst.obj, [-INF] * m, [INF] * m, blocks, lp, isintegral=isint)


class NodeResult:
    def __init__(self, status, obj=None, y=None, aux=None):
        self.status = status      # 'optimal' | 'infeasible' | 'unbounded' | 'failed'
        self.obj = obj
        self.y = y
        self.aux = aux            # whatever the node solver wants its children to see (P.parent_aux): warm-start data


def check_fixed_point(prob, y, tol=1e-6):
    """all variables fixed: feasibility of the point itself (sdpi.c:1377-1510 does this with lambda_min)"""
    for (lhs, rhs, row) in prob.lp:
        act = sum(c * y[v] for v, c in row.items())
        if act < lhs - tol or act > rhs + tol:
            return False
    for blk in prob.blocks:
        n = blk['n']
        Z = np.zeros((n, n))
        for var, ents in blk['vars'].items():
            for (r, c, v) in ents:
                Z[r, c] += v * y[var]
                if r != c:
                    Z[c, r] += v * y[var]
        for (r, c, v) in blk['const']:
            Z[r, c] -= v
            if r != c:
                Z[c, r] -= v
        if np.linalg.eigvalsh(Z)[0] < -tol:
            return False
    return True


def branch_and_bound(prob, intvars, solve_node, inttol=1e-5, maxnodes=20000, verbose=False):
    """returns (best objective or None if infeasible, best y, number of nodes, number of failed node solves)"""
    import heapq
    best = [math.inf, None]
    nnodes = 0
    nfailed = 0
    counter = [0]
    heap = []                       # best-bound-first: (parent bound, tie-break, lb, ub)

    class _Stack:
        def append(self, item):
            counter[0] += 1
            heapq.heappush(heap, (self.bound, counter[0], item[0], item[1], self.aux))
    stack = _Stack()
    stack.bound = -math.inf
    stack.aux = None
    stack.append((prob.lb.copy(), prob.ub.copy()))
    while heap and nnodes < maxnodes:
        pbound, _, lb, ub, paux = heapq.heappop(heap)
        if pbound >= best[0] - 1e-6 * max(1.0, abs(best[0])):
            continue                                       # the parent's bound already prunes this node
        nnodes += 1
        node = sdpi_prepare.SdpiProblem(prob.obj, lb, ub, prob.blocks, prob.lp, isintegral=prob.isintegral)
        P = sdpi_prepare.prepare(node)
        if P.status == 'infeasible':
            continue
        if P.status == 'allfixed':
            y = np.array(P.lb, dtype=float)
            if check_fixed_point(prob, y):
                val = float(prob.obj @ y)
                if val < best[0] - 1e-9:
                    best = [val, y]
            continue
        P.parent_aux = paux
        P.cutoff = best[0]            # incumbent value: a node solver may stop as soon as its lower bound exceeds it
        res = solve_node(P)
        stack.bound = res.obj if res.status == 'optimal' else pbound
        stack.aux = res.aux if res.status == 'optimal' else paux
        if res.status in ('infeasible', 'cutoff'):
            continue
        if res.status != 'optimal':
            nfailed += 1
            # cannot bound this node: branch anyway on the first unfixed integer variable
            cand = [v for v in intvars if P.ub[v] - P.lb[v] > 0.5]
            if not cand:
                continue
            v = cand[0]
            mid = math.floor(0.5 * (max(P.lb[v], -1e6) + min(P.ub[v], 1e6)))
            l1, u1 = np.array(P.lb), np.array(P.ub); u1[v] = mid
            l2, u2 = np.array(P.lb), np.array(P.ub); l2[v] = mid + 1
            stack.append((l1, u1)); stack.append((l2, u2))
            continue
        if res.obj >= best[0] - 1e-6 * max(1.0, abs(best[0])):
            continue                                       # bound
        frac = [(abs(res.y[v] - round(res.y[v])), v) for v in intvars]
        f, v = max(frac) if frac else (0.0, -1)
        if f <= inttol:
            y = res.y.copy()
            for w in intvars:
                y[w] = round(y[w])
            best = [res.obj, y]
            if verbose:
                print("  new incumbent %.8g at node %d" % (res.obj, nnodes))
            continue
        fl = math.floor(res.y[v])
        lo_l, lo_u = np.array(P.lb), np.array(P.ub); lo_u[v] = fl
        hi_l, hi_u = np.array(P.lb), np.array(P.ub); hi_l[v] = fl + 1
        # explore the nearer child first (it is pushed last)
        if res.y[v] - fl > 0.5:
            stack.append((lo_l, lo_u)); stack.append((hi_l, hi_u))
        else:
            stack.append((hi_l, hi_u)); stack.append((lo_l, lo_u))
    return (None if best[1] is None else best[0]), best[1], nnodes, nfailed


def oracle_node_solver(tol=1e-6):
    import ipm_ref

    def solve(P):
        b, blk, D, c, maps = sdpi_prepare.to_core(P)
        core = ipm_ref.CoreProblem(b, blk, D, c)
        r = ipm_ref.hsd_solve(core, ipm_ref.Params(gaptol=tol, feastol=tol))
        if r.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
            return NodeResult('infeasible')
        if r.status == ipm_ref.STATUS_DUNB:
            return NodeResult('unbounded')
        if r.status != ipm_ref.STATUS_OPTIMAL:
            return NodeResult('failed')
        y = np.array(P.lb, dtype=float)
        for k, v in enumerate(maps["active"]):
            y[v] = r.y[k]
        return NodeResult('optimal', float(P.prob.obj @ y), y)
    return solve


def cpu_c_node_solver(tol=1e-6, stats=None):
    """the same with the plain-C restatement of the iteration (oracle/cpu_ref.c) as node solver: the compiled CPU figure beside the
    device numbers (bench.py); stats (dict) collects calls / iterations / seconds inside the C solve"""
    import time
    import ipm_ref
    import cpu_ref

    def solve(P):
        b, blk, D, c, maps = sdpi_prepare.to_core(P)
        core = ipm_ref.CoreProblem(b, blk, D, c)
        t0 = time.perf_counter()
        info, yv = cpu_ref.solve(core, gaptol=tol, feastol=tol)
        if stats is not None:
            stats["seconds"] = stats.get("seconds", 0.0) + time.perf_counter() - t0
            stats["calls"] = stats.get("calls", 0) + 1
            stats["iters"] = stats.get("iters", 0) + info.iterations
        if info.status in (ipm_ref.STATUS_DINF, ipm_ref.STATUS_PDINF):
            return NodeResult('infeasible')
        if info.status == ipm_ref.STATUS_DUNB:
            return NodeResult('unbounded')
        if info.status != ipm_ref.STATUS_OPTIMAL:
            return NodeResult('failed')
        y = np.array(P.lb, dtype=float)
        for k, v in enumerate(maps["active"]):
            y[v] = yv[k]
        return NodeResult('optimal', float(P.prob.obj @ y), y)
    return solve
