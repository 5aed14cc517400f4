"""checker.py - TEST INFRASTRUCTURE.  Independent acceptance check of a solution, restating the reference's own checkers
on the marshalled problem (ipm_ref.CoreProblem):

   y-side ("dual"), src/sdpi/sdpsolchecker.c:58-265:  every row  D y - c >= -feastol  (bounds :135-144 and LP rows :149-176
       are rows of (D, c) after marshalling);  lambda_min(sum_i A_i^k y_i - A_0^k) >= -feastol per block (:201-257).
   X-side ("primal"), src/sdpi/sdpsolchecker.c:553-993: multipliers x >= -feastol (:723-741);
       sum_k <A_i^k, X_k> + (D^T x)_i = b_i within feastol (:775-931); lambda_min(X_k) >= -feastol (:945-983).
   gap: |b^T y - (<A_0, X> + c^T x)| < gaptol, absolute (sdpisolver_dsdp.c:1558-1571).

A point that passes all three is optimal to tolerance whatever algorithm produced it, so this pins results without any
reference arithmetic.  Eigenvalues come from LAPACK (scipy), not from the code under test."""
import numpy as np
import scipy.linalg as sla


def check_dual(prob, y, feastol):
    y = np.asarray(y, dtype=np.float64)
    rows = prob.D @ y - prob.c if prob.q else np.zeros(0)
    lmins = []
    for A in prob.blocks:
        Z = np.tensordot(y, A[1:], axes=(0, 0)) - A[0]
        lmins.append(float(sla.eigh(0.5 * (Z + Z.T), eigvals_only=True, subset_by_index=[0, 0])[0]))
    lpviol = float(max(0.0, -rows.min())) if prob.q else 0.0
    ok = lpviol <= feastol and all(l >= -feastol for l in lmins)
    return dict(feasible=ok, lpviol=lpviol, lmin=lmins)


def check_primal(prob, X, x, feastol):
    AX = np.zeros(prob.m)
    for A, Xk in zip(prob.blocks, X):
        AX += A[1:].reshape(prob.m, -1) @ np.asarray(Xk).reshape(-1)
    if prob.q:
        AX += prob.D.T @ x
    res = float(np.max(np.abs(AX - prob.b))) if prob.m else 0.0
    lmins = [float(sla.eigh(0.5 * (np.asarray(Xk) + np.asarray(Xk).T), eigvals_only=True, subset_by_index=[0, 0])[0]) for Xk in X]
    xmin = float(x.min()) if prob.q else 0.0
    ok = res <= feastol and all(l >= -feastol for l in lmins) and xmin >= -feastol
    return dict(feasible=ok, residual=res, lmin=lmins, xmin=xmin)


def objective_values(prob, y, X, x):
    dobj = float(prob.b @ y)
    pobj = float(sum(np.sum(A[0] * np.asarray(Xk)) for A, Xk in zip(prob.blocks, X)) + (prob.c @ x if prob.q else 0.0))
    return pobj, dobj


def certificate(prob, y, X, x, gaptol, feastol):
    """optimality certificate; returns (ok, details)"""
    d = check_dual(prob, y, feastol)
    p = check_primal(prob, X, x, feastol)
    pobj, dobj = objective_values(prob, y, X, x)
    gap = abs(pobj - dobj)
    return (d['feasible'] and p['feasible'] and gap < gaptol), dict(dual=d, primal=p, pobj=pobj, dobj=dobj, gap=gap)


def farkas_dual_infeasible(prob, X, x, tol):
    """X-ray certificate of infeasibility of the y-problem: A(X, x) = 0, X psd, x >= 0, <A_0, X> + c^T x > 0"""
    AX = np.zeros(prob.m)
    for A, Xk in zip(prob.blocks, X):
        AX += A[1:].reshape(prob.m, -1) @ np.asarray(Xk).reshape(-1)
    if prob.q:
        AX += prob.D.T @ x
    val = float(sum(np.sum(A[0] * np.asarray(Xk)) for A, Xk in zip(prob.blocks, X)) + (prob.c @ x if prob.q else 0.0))
    lmins = [float(np.linalg.eigvalsh(0.5 * (np.asarray(Xk) + np.asarray(Xk).T))[0]) for Xk in X]
    ok = val > 0 and np.max(np.abs(AX)) <= tol * val and all(l >= -tol * val for l in lmins) and (not prob.q or x.min() >= -tol * val)
    return ok, dict(value=val, residual=float(np.max(np.abs(AX))) if prob.m else 0.0)


def farkas_dual_unbounded(prob, y, tol):
    """y-ray: b^T y < 0, A^T y psd, D y >= 0 (proves the X-problem infeasible)"""
    y = np.asarray(y)
    val = float(prob.b @ y)
    rows = prob.D @ y if prob.q else np.zeros(0)
    lmins = [float(np.linalg.eigvalsh(np.tensordot(y, A[1:], axes=(0, 0)))[0]) for A in prob.blocks]
    ok = val < 0 and all(l >= -tol * (-val) for l in lmins) and (not prob.q or rows.min() >= -tol * (-val))
    return ok, dict(value=val)
