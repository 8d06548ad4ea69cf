# usage (GPU box): bash tools/probe/ab_files.sh <old.hip> <new.hip> <object>  -- paired A/B of two versions of one kernel FILE (headline bench, 3 x 2 runs)
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces -Wno-pass-failed"
for rep in 1 2 3; do for v in $1 $2; do
  /opt/rocm/bin/hipcc $FL -c $v -o $3 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  (cd ../.. && python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['roofline']['stage_ms_serial']['geom'], d['roofline']['stage_ms_serial']['tile'], d['latency']['ms_per_frame_latency'])")
done; done
