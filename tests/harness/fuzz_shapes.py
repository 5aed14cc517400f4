"""Random problem shapes over everything the one-launch kernel (csrc/solve1_body.h) is offered: 1-5 blocks of 1-30 rows, up to 110
variables with 0-4 nonzeros per matrix (variables without entries: M close to singular or singular), up to 200 LP rows of density
0.02-0.6, a third of the constant matrices diagonal.  Used by tests/test_gpu_solve1.py (the slice that runs in the driver's suite) and
by tests/devtools/solve1_fuzz.py / solve1_dump.py."""
import numpy as np
import ipm_ref


def problem(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(1, 6))
    sizes = [int(rng.integers(1, 31 if K == 1 else (22 if K == 2 else 14))) for _ in range(K)]
    dims = sum(n * (n + 1) // 2 for n in sizes)
    m = int(rng.integers(1, max(2, min(110, dims))))
    q = int(rng.integers(0, 200))
    dens = float(rng.uniform(0.02, 0.6))
    ystar = rng.standard_normal(m)
    blocks = []
    for n in sizes:
        A = np.zeros((m + 1, n, n))
        for i in range(1, m + 1):
            for _ in range(int(rng.integers(0, 5))):
                r, c = rng.integers(0, n, 2)
                v = rng.standard_normal()
                A[i, r, c] += v
                if r != c:
                    A[i, c, r] += v
        Zs = rng.standard_normal((n, n)); Zs = Zs @ Zs.T + 0.5 * np.eye(n)
        A[0] = np.tensordot(ystar, A[1:], axes=(0, 0)) - Zs
        if rng.random() < 0.3:
            A[0] = np.diag(np.diag(A[0])) - 0.0          # sparse constant matrix in some blocks
        blocks.append(A)
    D = rng.standard_normal((q, m)) * (rng.random((q, m)) < dens)
    c = D @ ystar - rng.random(q) - 0.1
    b = sum(np.array([np.trace(A[i]) for i in range(1, m + 1)]) for A in blocks) + (D.T @ np.ones(q) if q else 0.0)
    return ipm_ref.CoreProblem(b, blocks, D, c), "sizes %s m %d q %d density %.2f" % (sizes, m, q, dens)
