# usage (here): bash tools/sweep_geom_occ.sh build   -> mt_renderer_amd/libmtr_gocc<N>.so, N in 4 5 6 7 8 (k_geom with GEOM_OCC=N)
# usage (GPU box): bash tools/sweep_geom_occ.sh run  -> headline bench (pipelined ms/frame, stage times) and C3 / C5 ms/frame with each
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd mt_renderer_amd/csrc
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-missing-braces"
  for n in 4 5 6 7 8; do /opt/rocm/bin/hipcc $FL -DGEOM_OCC=$n -c k_geom.hip -o /tmp/k_geom_occ$n.o & done; wait
  for n in 4 5 6 7 8; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr_gocc$n.so /tmp/k_geom_occ$n.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz; done
  ls ../libmtr_gocc*.so
else
  for n in 4 5 6 7 8; do
    export MTR_LIB_PATH=$PWD/mt_renderer_amd/libmtr_gocc$n.so
    echo "GEOM_OCC=$n $(python bench.py --steps 500 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['stage_ms_serial'], d['latency']['ms_per_frame_latency'])")"
    python tools/bench_configs.py "C3" 2>&1 | grep -v amdgpu | cut -c1-60; python tools/bench_configs.py "opaque" 2>&1 | grep -v amdgpu | cut -c1-60
  done
fi
