"""core_of: the synthetic sparse problems of the solve1_sizes / solve1_mbig developer scripts"""
import numpy as np
import ipm_ref


def core_of(sizes, m, q, seed):
    rng = np.random.default_rng(seed)
    ystar = rng.standard_normal(m)
    blocks = []
    for n in sizes:
        A = np.zeros((m + 1, n, n))
        for i in range(1, m + 1):
            for _ in range(3):
                r, c = rng.integers(0, n, 2)
                v = rng.standard_normal()
                A[i, r, c] += v
                if r != c:
                    A[i, c, r] += v
        Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
        A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
        blocks.append(A)
    D = rng.standard_normal((q, m)) * (rng.random((q, m)) < 0.3)
    c = D @ ystar - rng.random(q) - 0.1
    b = sum(np.array([np.trace(A[i]) for i in range(1, m + 1)]) for A in blocks) + (D.T @ np.ones(q) if q else 0.0)
    return ipm_ref.CoreProblem(b, blocks, D, c)


