# usage (GPU box): bash tools/sweep_tile_run.sh "<runs>" : bins per run dealt to the XCDs in turn by the tile kernels (MTR_TILE_RUN), headline bench
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for r in $1; do
  MTR_TILE_RUN=$r python bench.py --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[run $r]', d['ms_per_step'], d['roofline']['stage_ms_serial']['geom'], d['roofline']['stage_ms_serial']['tile'], d['latency']['ms_per_frame_latency'])"
done; done
