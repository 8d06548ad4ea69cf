set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
for v in 4 5 6 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DGEOM_OCC=$v -c k_geom.hip -o k_geom.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o
  cd ../..
  echo "GEOM_OCC=$v" >> gpurun_out/sweep5.log
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms'])" >> gpurun_out/sweep5.log
  cd mt_renderer_amd/csrc
done
