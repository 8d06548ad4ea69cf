set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
for v in 4 6 5 4 6 5; do
  /opt/rocm/bin/hipcc $FL -DGEOM_OCC=$v -c k_geom.hip -o k_geom.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o
  for r in 1 2; do (cd ../.. && python bench.py --steps 600 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GEOM_OCC=$v', d['ms_per_step'])") >> ../../gpurun_out/sweep_overlap.log; done
done
