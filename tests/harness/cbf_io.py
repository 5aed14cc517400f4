"""cbf_io.py - TEST INFRASTRUCTURE (harness: host-side driver code, no arithmetic of the path).  Reader of the subset of the Conic Benchmark Format that SCIP-SDP's
dual-form examples use (format: /root/reference/src/scipsdp/reader_cbf.c:33-80 and the CBF 1 specification it cites):

   VER, OBJSENSE MIN|MAX, VAR (cones F, L+, L-, L=), INT, CON (cones L+, L-, L=), PSDCON,
   OBJACOORD, OBJBCOORD, ACOORD, BCOORD, HCOORD, DCOORD

meaning   min/max  sum_j c_j x_j + c_0   s.t.  sum_j a_ij x_j + b_i  in  K_i   (L+: >= 0, L-: <= 0, L=: = 0),
                                               sum_j x_j H_j^k + D^k  psd,   x_j in its VAR cone.

PSDVAR / FCOORD / OBJFCOORD (primal-form matrix variables), quadratic cones and rank-1 sections are not read: the reference
reformulates those inside its reader, which is outside the solver-interface path.  read_cbf returns an SdpiProblem
(tests/harness/sdpi_prepare.py) plus the integer variables and the objective constant / sense, i.e. what reader_cbf.c hands to
cons_sdp + the LP rows."""
import numpy as np
import sdpi_prepare

INF = 1e20


def read_cbf(path, integrality=False):
    with open(path, 'rt') as f:
        lines = [l.split('#')[0].strip() for l in f.read().splitlines()]
    lines = [l for l in lines if l]
    pos = 0

    def take():
        nonlocal pos
        l = lines[pos]
        pos += 1
        return l

    sense = 1.0
    nvars = 0
    varcones = []
    concones = []
    intvars = []
    psdsizes = []
    obj = {}
    objconst = 0.0
    acoord = []
    bcoord = {}
    hcoord = []
    dcoord = []
    while pos < len(lines):
        key = take()
        if key == 'VER':
            take()
        elif key == 'OBJSENSE':
            sense = 1.0 if take().upper() == 'MIN' else -1.0
        elif key == 'VAR':
            nvars, k = (int(x) for x in take().split())
            for _ in range(k):
                cone, cnt = take().split()
                varcones += [cone] * int(cnt)
        elif key == 'INT':
            for _ in range(int(take())):
                intvars.append(int(take()))
        elif key == 'CON':
            ncon, k = (int(x) for x in take().split())
            for _ in range(k):
                cone, cnt = take().split()
                concones += [cone] * int(cnt)
        elif key == 'PSDCON':
            for _ in range(int(take())):
                psdsizes.append(int(take()))
        elif key == 'OBJACOORD':
            for _ in range(int(take())):
                j, v = take().split()
                obj[int(j)] = float(v)
        elif key == 'OBJBCOORD':
            objconst = float(take())
        elif key == 'ACOORD':
            for _ in range(int(take())):
                i, j, v = take().split()
                acoord.append((int(i), int(j), float(v)))
        elif key == 'BCOORD':
            for _ in range(int(take())):
                i, v = take().split()
                bcoord[int(i)] = float(v)
        elif key == 'HCOORD':
            for _ in range(int(take())):
                k, j, r, c, v = take().split()
                hcoord.append((int(k), int(j), int(r), int(c), float(v)))
        elif key == 'DCOORD':
            for _ in range(int(take())):
                k, r, c, v = take().split()
                dcoord.append((int(k), int(r), int(c), float(v)))
        elif key in ('PSDVAR', 'FCOORD', 'OBJFCOORD', 'PSDVARRANK1', 'PSDCONRANK1'):
            raise NotImplementedError("CBF section %s (primal-form / rank-1 data) is outside this reader" % key)
        else:
            raise ValueError("unknown CBF section %r" % key)
    assert len(varcones) == nvars
    lb = [-INF] * nvars
    ub = [INF] * nvars
    for j, cone in enumerate(varcones):
        if cone == 'L+':
            lb[j] = 0.0
        elif cone == 'L-':
            ub[j] = 0.0
        elif cone == 'L=':
            lb[j] = ub[j] = 0.0
        elif cone != 'F':
            raise NotImplementedError("variable cone %s" % cone)
    rows = [dict() for _ in concones]
    for i, j, v in acoord:
        rows[i][j] = rows[i].get(j, 0.0) + v
    lp = []
    for i, cone in enumerate(concones):
        b = bcoord.get(i, 0.0)
        if cone == 'L+':
            lp.append((-b, INF, rows[i]))
        elif cone == 'L-':
            lp.append((-INF, -b, rows[i]))
        elif cone == 'L=':
            lp.append((-b, -b, rows[i]))
        else:
            raise NotImplementedError("constraint cone %s" % cone)
    blocks = [dict(n=n, vars={}, const=[]) for n in psdsizes]
    for k, j, r, c, v in hcoord:
        if r < c:
            r, c = c, r
        blocks[k]['vars'].setdefault(j, []).append((r, c, v))
    for k, r, c, v in dcoord:
        if r < c:
            r, c = c, r
        blocks[k]['const'].append((r, c, -v))          # sum_j x_j H_j + D psd  <=>  sum_j x_j A_j - A_0 psd with A_0 = -D
    c = np.zeros(nvars)
    for j, v in obj.items():
        c[j] = sense * v
    prob = sdpi_prepare.SdpiProblem(c, lb, ub, blocks, lp, isintegral=[integrality and v in set(intvars) for v in range(len(c))])
    return prob, sorted(intvars), sense, objconst
