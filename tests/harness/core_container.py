"""The marshalled problem a node solve is given - b[m], one array A[m + 1, n, n] per block (A[0] the constant matrix), LP rows
D y - c >= 0 - as a plain container for code that only DRIVES the device path (bench.py's timed B&B-sized solves, tools): no
arithmetic, nothing of oracle/ is imported.  The attribute layout is the one scip-sdp_amd/binding.py: Solver.load_core reads (and the
one oracle/ipm_ref.CoreProblem has, so that the same object can be handed to the oracle by a test)."""
import numpy as np


class CoreContainer:
    def __init__(self, b, blocks, D=None, c=None):
        self.b = np.asarray(b, dtype=np.float64)
        self.m = self.b.shape[0]
        self.blocks = [np.ascontiguousarray(A, dtype=np.float64) for A in blocks]
        for A in self.blocks:
            assert A.shape[0] == self.m + 1 and A.shape[1] == A.shape[2]
        if D is None:
            D = np.zeros((0, self.m))
            c = np.zeros(0)
        self.D = np.ascontiguousarray(D, dtype=np.float64).reshape(-1, self.m)
        self.c = np.asarray(c, dtype=np.float64).reshape(-1)
        self.q = self.D.shape[0]
        assert self.c.shape[0] == self.q
