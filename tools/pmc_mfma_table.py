#!/usr/bin/env python3
"""pmc_mfma_table.py DIR - per-kernel FP64-MFMA utilisation from the two rocprofv3 passes of tools/pmc_mfma.sh (developer tool).

 MFMA busy  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   busy share of the matrix pipes
 flops      = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 (the counter advances by one per 512 flops: a v_mfma_f64_16x16x4 is 2048 flops = 4 MOPS)
 clock      = GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS)
"""
import csv, collections, glob, sys
root = sys.argv[1]
def find(sub, suffix):
    g = glob.glob("%s/%s/**/*%s" % (root, sub, suffix), recursive=True)
    return g[0] if g else None
def counters(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(find(sub, "counter_collection.csv"))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in cnt.items()}
def durations(sub):
    d = collections.defaultdict(float)
    for r in csv.DictReader(open(find(sub, "kernel_trace.csv"))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return d
sq, n = counters("sq"); gr, _ = counters("grbm"); dur_sq = durations("sq"); dur_gr = durations("grbm")
print("%-34s %5s %9s %9s %10s %10s %9s %9s" % ("kernel", "calls", "ms/call", "GHz", "MFMA busy", "TFLOP/s", "wait_inst", "wait_any"))
rows = sorted(sq, key=lambda k: -dur_sq.get(k, 0))[:8]
for k in rows:
    c = sq[k]; calls = n[k]; t = dur_sq[k] * 1e-9
    gui = gr.get(k, {}).get("GRBM_GUI_ACTIVE", 0.0); tg = dur_gr.get(k, 0.0) * 1e-9
    ghz = gui / 8.0 / tg / 1e9 if tg > 0 else float("nan")
    # cycles available to the matrix pipes during the profiled (SQ) pass, at the clock measured in the GRBM pass
    avail = ghz * 1e9 * t * 256 * 4
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / avail if avail > 0 else float("nan")
    tf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512 / t / 1e12 if t > 0 else 0.0
    wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    print("%-34s %5d %9.3f %9.2f %9.1f%% %10.1f %8.1f%% %8.1f%%" % (k[:34], calls, 1e3 * t / calls, ghz, 100 * busy, tf,
        100 * c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 100 * c.get("SQ_WAIT_ANY", 0.0) / wc))
