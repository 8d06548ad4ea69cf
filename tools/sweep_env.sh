# usage (GPU box): bash tools/sweep_env.sh "VAR=a VAR=b ..." : headline bench under each environment setting, two alternating repetitions
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for kv in $1; do
  env $(echo $kv | tr "," " ") python bench.py --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$kv]', d['ms_per_step'], d['roofline']['stage_ms_serial']['geom'], d['roofline']['stage_ms_serial']['tile'], d['latency']['ms_per_frame_latency'])"
done; done
