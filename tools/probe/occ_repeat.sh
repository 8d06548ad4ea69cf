# headline bench (pipelined ms/frame) with the resident k_geom workgroups per CU capped at N (0 = no cap), three rounds
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for n in 0 5 6 7; do
    export MTR_GEOM_WGS_PER_CU=$n
    echo "rep $rep WGS_PER_CU=$n $(python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_min_max'], d['roofline']['stage_ms'], d['latency']['ms_per_frame_latency'])")"
done; done
