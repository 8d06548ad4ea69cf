# usage (here): bash tools/sweep_vis_occ.sh build   -> mt_renderer_amd/libmtr_vocc<N>.so, N in 4 5 6 7 8 (k_tile_vis with VIS_OCC=N)
# usage (GPU box): bash tools/sweep_vis_occ.sh run  -> headline bench twice + C3 / C5 with each
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd mt_renderer_amd/csrc
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-missing-braces -Wno-pass-failed"
  for n in 4 5 6 7 8; do /opt/rocm/bin/hipcc $FL -DVIS_OCC=$n -c k_tile_vis.hip -o /tmp/k_tv_occ$n.o & done; wait
  for n in 4 5 6 7 8; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr_vocc$n.so k_geom.o k_bin.o k_tile.o /tmp/k_tv_occ$n.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz; done
  ls ../libmtr_vocc*.so
else
  for rep in 1 2; do for n in 4 5 6 7 8; do
    export MTR_LIB_PATH=$PWD/mt_renderer_amd/libmtr_vocc$n.so
    echo "rep $rep VIS_OCC=$n $(python bench.py --steps 2000 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms_serial'], d['latency']['ms_per_frame_latency'])")"
  done; done
  for n in 4 5 6 7 8; do export MTR_LIB_PATH=$PWD/mt_renderer_amd/libmtr_vocc$n.so; echo "VIS_OCC=$n"; python tools/bench_configs.py "C" 2>&1 | grep -v amdgpu | cut -c1-70; done
fi
