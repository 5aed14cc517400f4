cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hip -o h -- python3 $GRAFT_REPO_ROOT/tests/devtools/mid_phases.py 300 100 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_hip.txt 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/prof_hip.txt | cut -c1-120
ls gpurun_out/prof_hip
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_hip/*hip_api_trace.csv')
rows=list(csv.DictReader(open(f[0])))
print(rows[0].keys())
rows.sort(key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']), reverse=True)
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in rows[:25]:
    print("%-40s %10.3f ms at %10.3f ms" % (r['Function'], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6, (int(r['Start_Timestamp'])-t0)/1e6))
PY
rm -rf gpurun_out/prof_hip
