"""Warm-started node solver for the B&B harness (test side): the child of a node starts from its parent's optimal
(y, Z(y), X), pushed into the interior by a convex combination with a scaled identity - the simple variant of what
relax_sdp.c does with "warmstartipfactor" (relax_sdp.c:2649-3825) - handed to the backend through starty / startZ* / startX* of
SCIPsdpiSolverLoadAndSolve in ORIGINAL indices (sdpisolver.h:160-173), LP block last with the 2 * row (+1) /
2 * nlpcons + 2 * var (+1) convention."""
import numpy as np
import bnb
import sdpi_call

INF = 1e20


def dense_Z(prob, y):
    Zs = []
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
        Zs.append(Z)
    return Zs


def sparse_lower(M):
    n = M.shape[0]
    r, c = np.tril_indices(n)
    return (r.astype(np.int32), c.astype(np.int32), M[r, c])


def make_start(prob, P, aux, lam):
    """aux = dict(y, X=[dense], lhs, rhs, lb, ub) of the parent; the child's bounds may have changed: slacks are recomputed"""
    y = aux["y"]
    nlp = len(prob.lp)
    Zb, Xb = [], []
    for Z, X in zip(dense_Z(prob, y), aux["X"]):
        n = Z.shape[0]
        sz = max(1.0, np.trace(Z) / n)
        sx = max(1.0, np.trace(X) / n)
        Zb.append(sparse_lower((1.0 - lam) * Z + lam * sz * np.eye(n)))
        Xb.append(sparse_lower((1.0 - lam) * X + lam * sx * np.eye(n)))
    # LP block (diagonal): positions 2 * row (+1 for rhs), then 2 * nlp + 2 * var (+1 for ub); Z entries are the slacks
    zi, zv, xi, xv = [], [], [], []
    for r, (lhs, rhs, row) in enumerate(prob.lp):
        act = sum(c * y[v] for v, c in row.items())
        for side, (bound, mult) in enumerate(((lhs, aux["lhs"][r]), (rhs, aux["rhs"][r]))):
            if abs(bound) >= INF:
                continue
            slack = (act - bound) if side == 0 else (bound - act)
            zi.append(2 * r + side); zv.append((1.0 - lam) * max(slack, 0.0) + lam)
            xi.append(2 * r + side); xv.append((1.0 - lam) * max(mult, 0.0) + lam)
    for v in range(prob.nvars):
        for side, (bound, mult) in enumerate(((P.lb[v], aux["lb"][v]), (P.ub[v], aux["ub"][v]))):
            if abs(bound) >= INF:
                continue
            slack = (y[v] - bound) if side == 0 else (bound - y[v])
            pos = 2 * nlp + 2 * v + side
            zi.append(pos); zv.append((1.0 - lam) * max(slack, 0.0) + lam)
            xi.append(pos); xv.append((1.0 - lam) * max(mult, 0.0) + lam)
    Zb.append((np.array(zi, dtype=np.int32), np.array(zi, dtype=np.int32), np.array(zv)))
    Xb.append((np.array(xi, dtype=np.int32), np.array(xi, dtype=np.int32), np.array(xv)))
    return dict(y=y, Z=Zb, X=Xb)


def warm_node_solver(lib, tol, lam, use_objlimit=False):
    """lam <= 0: cold starts (same bookkeeping); use_objlimit: hand the incumbent to the backend as SCIP_SDPPAR_OBJLIMIT
    (relaxing/SDP/objlimit of the reference, relax_sdp.c:4074-4079), nodes that exceed it stop early"""
    s = sdpi_call.SdpiSolver(lib)
    for par in (1, 2, 3):
        assert s.set_real(par, tol) == sdpi_call.SCIP_OKAY
    stats = dict(calls=0, iters=0, warm=0, time=0.0, cutoff=0, wall=0.0)

    def solve(P):
        prob = P.prob
        start = None
        if lam > 0.0 and getattr(P, "parent_aux", None) is not None:
            start = make_start(prob, P, P.parent_aux, lam)
            stats["warm"] += 1
        if use_objlimit:
            cut = getattr(P, "cutoff", np.inf)
            assert s.set_real(4, cut if np.isfinite(cut) else INF) == sdpi_call.SCIP_OKAY
        rc, _, _ = s.solve(P, start=start)
        stats["wall"] += s.last_call_seconds
        assert rc == sdpi_call.SCIP_OKAY
        stats["calls"] += 1
        stats["iters"] += s.iterations()
        stats["time"] += s.opttime()
        if s.flag("IsDualInfeasible"):
            return bnb.NodeResult('infeasible')
        if use_objlimit and s.flag("IsObjlimExc"):
            stats["cutoff"] += 1
            return bnb.NodeResult('cutoff')
        if s.flag("IsDualUnbounded"):
            return bnb.NodeResult('unbounded')
        if not s.flag("IsOptimal"):
            return bnb.NodeResult('failed')
        rc, obj, y = s.dual_sol()
        aux = None
        if lam > 0.0:
            rc1, X = s.primal_solution_matrix()
            rc2, lhs, rhs = s.lp_sides()
            rc3, lbm, ubm = s.bound_vars()
            assert rc1 == rc2 == rc3 == sdpi_call.SCIP_OKAY
            aux = dict(y=y.copy(), X=X, lhs=lhs, rhs=rhs, lb=lbm, ub=ubm)
        return bnb.NodeResult('optimal', obj, y, aux)
    return s, solve, stats
