import csv
rows=list(csv.DictReader(open("gpurun_out/bnb_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6, "calls", sum(int(r["Calls"]) for r in rows))
for r in rows[:45]:
    print("%-70s %7d %8.2f ms %7.1f us %5.1f%%" % (r["Name"][:70], int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
