"""stress_cases.py - random engine problems for the stress tool (tests/devtools/stress_gpu.py) and the parity tests: dense / sparse
blocks across the 64 and 128 size boundaries, LP rows, planted optima, strictly feasible, infeasible and random instances.  The
generator is seeded by the caller; STRESS_BIG selects the mid-size / near-bench-size families."""
import os
import numpy as np
import ipm_ref


def rand_core(rng):
    kind = rng.integers(0, 5)
    K = int(rng.integers(1, 4))
    if os.environ.get("STRESS_BIG") == "2":
        # towards the bench size: several blocks through the persistent GEMM paths (the oracle needs seconds per iteration)
        K = int(rng.integers(1, 3))
        ns = [int(rng.choice([300, 385, 500])) for _ in range(K)]
        m = int(rng.choice([300, 600, 1000]))
    elif os.environ.get("STRESS_BIG"):
        # the regime between the B&B-sized and the bench-sized problems: general kernels, MFMA tile paths, K-sliced Gram product
        K = int(rng.integers(1, 3))
        ns = [int(rng.choice([100, 129, 160, 200, 257])) for _ in range(K)]
        m = int(rng.choice([129, 200, 257, 300, 400]))
        if m * sum(n * n for n in ns) > 3e7:
            ns = ns[:1]
    else:
        ns = [int(rng.choice([2, 3, 5, 9, 16, 17, 32, 33, 63, 64, 65, 70, 100])) for _ in range(K)]
        if sum(n * n for n in ns) > 12000:
            ns = ns[:1]
        m = int(rng.choice([1, 2, 5, 13, 40, 64, 65, 129, 140, 200]))
        if m * sum(n * n for n in ns) > 2.5e6:
            m = max(1, int(2.5e6 / sum(n * n for n in ns)))
    q = int(rng.choice([0, 0, 3, 17, 64, 150]))
    blocks = []
    for n in ns:
        A = np.zeros((m + 1, n, n))
        dens = rng.choice([1.0, 0.3, 0.05])
        for i in range(1, m + 1):
            G = rng.standard_normal((n, n)) * (rng.random((n, n)) < dens)
            A[i] = (G + G.T) / np.sqrt(2 * n)
        blocks.append(A)
    D = rng.standard_normal((q, m)) * (rng.random((q, m)) < 0.3) if q else np.zeros((0, m))
    # kinds: 0/1 planted optimum, 2 strictly feasible both (A0 = -dI, random b with bounded set via box rows), 3 y-infeasible, 4 random
    y0 = rng.uniform(-1, 1, m)
    if kind in (0, 1):
        b = np.zeros(m)
        for A in blocks:
            n = A.shape[1]
            Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            r = max(1, n // 3)
            ev = rng.uniform(1, 2, n)
            Xs = (Q * np.where(np.arange(n) < r, ev, 0)) @ Q.T
            Zs = (Q * np.where(np.arange(n) < r, 0, ev)) @ Q.T
            A[0] = np.tensordot(y0, A[1:], axes=(0, 0)) - Zs
            b += A[1:].reshape(m, -1) @ Xs.reshape(-1)
        c = np.zeros(q)
        if q:
            xs = rng.uniform(0, 1, q) * (rng.random(q) < 0.5)
            zs = rng.uniform(0.5, 1.5, q) * (xs == 0)
            c = D @ y0 - zs
            b += D.T @ xs
    elif kind == 2:
        for A in blocks:
            A[0] = -rng.uniform(1, 3) * np.eye(A.shape[1])
        c = D @ y0 - rng.uniform(0.5, 2, q) if q else np.zeros(0)
        b = rng.standard_normal(m)
    elif kind == 3:
        # infeasible y-problem: block 0 demands sum A y - A0 psd with A0 = +dI and A_i negative semidefinite
        for k, A in enumerate(blocks):
            n = A.shape[1]
            if k == 0:
                for i in range(1, m + 1):
                    v = rng.standard_normal(n)
                    A[i] = -np.outer(v, v) / n
                A[0] = np.eye(n)
            else:
                A[0] = -np.eye(n)
        # y >= 0 rows make it truly infeasible
        D = np.concatenate([D, np.eye(m)], axis=0)
        c = np.concatenate([(D[:q] @ y0 - 1.0) if q else np.zeros(0), np.zeros(m)])
        b = rng.standard_normal(m)
    else:
        for A in blocks:
            G = rng.standard_normal(A.shape[1:]); A[0] = (G + G.T) / 2 - 2 * np.eye(A.shape[1])
        c = rng.standard_normal(D.shape[0])
        b = rng.standard_normal(m)
    return ipm_ref.CoreProblem(b, blocks, D, c), kind

