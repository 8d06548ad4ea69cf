set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
for v in 5 6 7 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DVIS_OCC=$v -c k_tile_vis.hip -o k_tile_vis.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o
  cd ../..
  echo "VIS_OCC=$v" >> gpurun_out/sweep4.log
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms'])" >> gpurun_out/sweep4.log
  cd mt_renderer_amd/csrc
done
