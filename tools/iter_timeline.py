#!/usr/bin/env python3
"""iter_timeline.py KERNEL_TRACE_CSV - developer tool: what the device does between two Schur assemblies of the bench solve.
Takes the kernel trace of a bench run (rocprofv3 --kernel-trace --output-format csv), finds the Gram kernels (one per iteration) and,
for the iterations of the last solves, prints: the span from the end of one assembly to the start of the next first product, the time
in which no kernel runs at all (host turn-arounds, launch gaps), and the kernels that make up the rest."""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")))
rows.sort()
gram = [i for i, r in enumerate(rows) if r[2].startswith("hs_gram_kernel") or r[2].startswith("hs_dgemm2_kernel<0, 0>")]
first = [i for i, r in enumerate(rows) if r[2].startswith("hs_dgemm5_kernel<1, 1>") or r[2].startswith("hs_dgemm2_kernel<1, 1>")]
if len(gram) < 10 or len(first) < 10:
    print("no assemblies found"); sys.exit(1)
# pairs: end of gram g -> start of the next first product (skip solve boundaries: gaps above 6 ms)
spans = []
for g in gram:
    nxt = [f for f in first if f > g]
    if not nxt:
        continue
    f = nxt[0]
    t0, t1 = rows[g][1], rows[f][0]
    if t1 - t0 > 6e6:
        continue
    spans.append((g, f, t0, t1))
spans = spans[-18:]
tot = collections.defaultdict(float); cnt = collections.Counter()
idle_sum = 0.0; span_sum = 0.0; crit = collections.defaultdict(float)
for (g, f, t0, t1) in spans:
    span_sum += t1 - t0
    ks = [r for r in rows[g + 1:f] if r[1] > t0 and r[0] < t1]
    # union of busy intervals
    cur = t0; busy = 0.0
    for (a, b, nm) in sorted(ks):
        a = max(a, t0); b = min(b, t1)
        if b <= cur:
            continue
        # the part of this kernel during which nothing earlier was still running: attributed to it
        crit[nm] += b - max(a, cur)
        busy += b - max(a, cur)
        cur = b
    idle_sum += (t1 - t0) - busy
    for (a, b, nm) in ks:
        tot[nm] += b - a; cnt[nm] += 1
n = len(spans)
print("%d iterations: span between assemblies %.3f ms, of which no kernel running %.3f ms" % (n, span_sum / n / 1e6, idle_sum / n / 1e6))
print("%-60s %8s %10s %12s" % ("kernel", "calls/it", "us/it", "alone us/it"))
for nm in sorted(tot, key=lambda k: -crit[k])[:28]:
    print("%-60s %8.1f %10.1f %12.1f" % (nm[:60], cnt[nm] / n, tot[nm] / n / 1e3, crit[nm] / n / 1e3))
