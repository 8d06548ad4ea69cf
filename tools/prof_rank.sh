# usage (GPU box): bash tools/prof_rank.sh <tag> <C4|C5> <rank> <world>  -- rocprofv3 kernel-trace stats of one sharded rank, one frame at a
# time: per-kernel median durations -> gpurun_out/<tag>_rank_<scene>_<rank>of<world>.txt (copy to profiles/)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/$1_rank_$2_$3of$4
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/probe/shard_rank_trace.py $2 $3 $4 > $d/log.txt 2>&1
cd $R
python3 - "$d" > $d.txt <<'PY'
import csv, glob, collections, sys
d = sys.argv[1]
print(open(d + "/log.txt").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]  # the steady-state half (sharded frames)
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'].split('(')[0][:64]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"{k:66s} n={len(v):4d} median {v[len(v)//2]/1e3:8.1f} us  min {v[0]/1e3:8.1f}  max {v[-1]/1e3:8.1f}")
PY
cat $d.txt
