# per-kernel durations of the sharded C5 path (rocprofv3 kernel trace of tools/probe/c5_shard_ab.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c5 -- python3 $R/tools/probe/c5_shard_ab.py > $R/gpurun_out/prof_c5.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/prof_c5/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'][:70]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, d in agg.items():
    print(f"{k:72s} n={len(d):4d} median {sorted(d)[len(d)//2]/1e3:8.1f} us  min {min(d)/1e3:8.1f} max {max(d)/1e3:8.1f}")
print("--- last 16 launches, in order")
for r in rows[-16:]:
    print(f"{r['Kernel_Name'][:60]:62s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))}")
PY
