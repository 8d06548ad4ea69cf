# usage (GPU box): bash tools/probe/ab_files_cfg.sh <old.hip> <new.hip> <object>  -- like ab_files.sh, plus every config of bench_configs.py once per version
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces -Wno-pass-failed"
for rep in 1 2; do for v in $1 $2; do
  /opt/rocm/bin/hipcc $FL -c $v -o $3 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  (cd ../.. && python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['roofline']['stage_ms_serial']['geom'], d['roofline']['stage_ms_serial']['tile'], d['latency']['ms_per_frame_latency'])")
  if [ $rep = 1 ]; then (cd ../.. && python tools/bench_configs.py C 2>&1 | grep -v amdgpu | sed 's/tris_in.*stages_ms/stages_ms/' | cut -c1-140); fi
done; done
