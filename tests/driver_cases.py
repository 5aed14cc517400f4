"""Problems for the tests of the SCIPsdpiSolve restatement (oracle/sdpi_driver.py), shared by the CPU test (numpy backend) and
the gpu-marked test (libhipsdp.so backend).  Known answers are derived by hand in the comments."""
import numpy as np
import sdpi_prepare

INF = 1e20


def maxcut_like(n=5, seed=3):
    """min sum_i y_i  s.t.  Diag(y) - C >= 0, 0 <= y <= 50; strictly feasible (y large), bounded: a regular node problem"""
    rng = np.random.default_rng(seed)
    Cm = rng.standard_normal((n, n))
    Cm = (Cm + Cm.T) / 2
    blocks = [dict(n=n, vars={i: [(i, i, 1.0)] for i in range(n)},
                   const=[(r, c, float(Cm[r, c])) for r in range(n) for c in range(r + 1)])]
    return sdpi_prepare.SdpiProblem(np.ones(n), np.zeros(n), 50.0 * np.ones(n), blocks, []), Cm


def infeasible_block():
    """y [[1,0],[0,-1]] - [[1,0],[0,1]] >= 0 has no solution (needs y >= 1 and -y >= 1); second variable keeps it off the
    one-variable shortcut: entries on a separate 1x1-like diagonal position"""
    blocks = [dict(n=3, vars={0: [(0, 0, 1.0), (1, 1, -1.0)], 1: [(2, 2, 1.0)]}, const=[(0, 0, 1.0), (1, 1, 1.0), (2, 2, 1.0)])]
    return sdpi_prepare.SdpiProblem([1.0, 1.0], [-10.0, -10.0], [10.0, 10.0], blocks, [])


def no_interior():
    """[[y0, y1], [y1, 0]] >= 0 forces y1 = 0: feasible, but no positive definite point -> the dual Slater condition fails;
    min y0 + y1 with y0 >= 1 (bound): optimum 1 at (1, 0)"""
    blocks = [dict(n=2, vars={0: [(0, 0, 1.0)], 1: [(1, 0, 1.0)]}, const=[])]
    return sdpi_prepare.SdpiProblem([1.0, 1.0], [1.0, -5.0], [5.0, 5.0], blocks, [])


def free_variable_with_rows():
    """2 variables, one free: y0 I_2 + y1 [[1,0],[0,-1]] - [[0,1],[1,0]] >= 0, row y0 - y1 >= 0.5 (two nonzeros: stays an LP
    row), y0 in [0, 10], y1 free; min y0.  psd iff y0 >= sqrt(y1^2 + 1) -> optimum y0 = 1 at y1 = 0 (row: 1 - 0 >= 0.5 ok)"""
    blocks = [dict(n=2, vars={0: [(0, 0, 1.0), (1, 1, 1.0)], 1: [(0, 0, 1.0), (1, 1, -1.0)]}, const=[(1, 0, 1.0)])]
    lp = [(0.5, INF, {0: 1.0, 1: -1.0})]
    return sdpi_prepare.SdpiProblem([1.0, 0.0], [0.0, -INF], [10.0, INF], blocks, lp)
