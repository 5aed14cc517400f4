"""sdpa_io.py - TEST INFRASTRUCTURE (harness: host-side driver code, no arithmetic of the path).  Minimal reader of the SDPA sparse format with SCIP-SDP's '*INTEGER'
extension (format description: /root/reference/sdpa_format.txt:22-61; reference reader: src/scipsdp/reader_sdpa.c).

   min b^T y  s.t.  sum_i A_i^k y_i - A_0^k psd (k = SDP blocks),   sum_i d_ri y_i - d_r0 >= 0 (rows of the LP block)

Only what the parity tests need: dense assembly of the blocks, the diagonal (negative-size) block as LP rows, the list of
integer variables.  No '*RANK1', no CBF.
"""
import gzip
import re
import numpy as np


class SdpaInstance:
    def __init__(self):
        self.m = 0
        self.obj = None
        self.blocksizes = []       # signed, as in the file
        self.sdpblocks = []        # list of dict: n, entries {var: [(i, j, v)]} 0-based, lower (i >= j), var 0 = constant
        self.lprows = []           # list of dict {var: coef} (var 0 = constant d_r0), one per diagonal position
        self.intvars = []          # 0-based


def _tokens(line):
    line = line.split('*')[0] if not line.startswith('*') else ''
    line = re.sub(r'[{}(),]', ' ', line)
    return line.split()


def read_sdpa(path):
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'rt') as f:
        raw = f.read().splitlines()
    inst = SdpaInstance()
    stage = 0
    intsection = False
    blockmap = []   # file block index -> ('sdp', idx) / ('lp', offset)
    for line in raw:
        s = line.strip()
        if s.startswith('*INTEGER'):
            intsection = True
            continue
        if s.startswith('*RANK1'):
            intsection = False
            continue
        if intsection and s.startswith('*'):
            t = s[1:].split()
            if t and t[0].isdigit():
                inst.intvars.append(int(t[0]) - 1)
            continue
        if not s or s.startswith('*') or s.startswith('"'):
            continue
        if stage == 0:
            inst.m = int(re.split(r'[\s=]+', s)[0])
            stage = 1
        elif stage == 1:
            nblocks = int(re.split(r'[\s=]+', s)[0])
            stage = 2
        elif stage == 2:
            t = [x for x in _tokens(s.split('=')[0])]
            inst.blocksizes = [int(float(x)) for x in t[:nblocks]]
            nlp = 0
            for bs in inst.blocksizes:
                if bs < 0:
                    blockmap.append(('lp', nlp))
                    nlp += -bs
                else:
                    blockmap.append(('sdp', len(inst.sdpblocks)))
                    inst.sdpblocks.append({'n': bs, 'entries': {}})
            inst.lprows = [dict() for _ in range(nlp)]
            stage = 3
        elif stage == 3:
            t = _tokens(s)
            inst.obj = np.array([float(x) for x in t[:inst.m]])
            stage = 4
        else:
            t = _tokens(s)
            if len(t) < 5:
                continue
            var, blk, i, j, v = int(t[0]), int(t[1]) - 1, int(t[2]) - 1, int(t[3]) - 1, float(t[4])
            kind, idx = blockmap[blk]
            if kind == 'lp':
                assert i == j
                inst.lprows[idx + i][var] = inst.lprows[idx + i].get(var, 0.0) + v
            else:
                if i < j:
                    i, j = j, i
                inst.sdpblocks[idx]['entries'].setdefault(var, []).append((i, j, v))
    return inst


def dense_blocks(inst):
    """list of arrays A[m+1, n, n] (index 0 = constant matrix)"""
    out = []
    for blk in inst.sdpblocks:
        n = blk['n']
        A = np.zeros((inst.m + 1, n, n))
        for var, ents in blk['entries'].items():
            for (i, j, v) in ents:
                A[var, i, j] = v
                A[var, j, i] = v
        out.append(A)
    return out


def lp_dense(inst):
    """(D[q, m], c[q]) with rows D y - c >= 0"""
    q = len(inst.lprows)
    D = np.zeros((q, inst.m))
    c = np.zeros(q)
    for r, row in enumerate(inst.lprows):
        for var, v in row.items():
            if var == 0:
                c[r] = v
            else:
                D[r, var - 1] = v
    return D, c
