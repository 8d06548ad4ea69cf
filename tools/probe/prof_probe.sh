# usage: bash tools/probe/prof_probe.sh <python probe script>: per-kernel durations (rocprofv3 kernel trace) of a probe
R=$GRAFT_REPO_ROOT
P=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_probe
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_probe -- python3 $R/$P > $R/gpurun_out/prof_probe.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/prof_probe/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for r in rows:
    agg[(r['Kernel_Name'][:48], r.get('Grid_Size_X', '?'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in sorted(agg.items()):
    if len(d) >= 5: print(f"{k[0]:50s} grid {k[1]:>9s} n={len(d):5d} median {sorted(d)[len(d)//2]/1e3:8.1f} us  min {min(d)/1e3:8.1f} max {max(d)/1e3:8.1f}")
# gaps between consecutive kernels of the last 12 launches
print("--- last launches: start offset / duration (us)")
t0 = int(rows[-12]['Start_Timestamp'])
for r in rows[-12:]:
    print(f"{r['Kernel_Name'][:44]:46s} start +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f}  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f}")
PY
