# usage (GPU box): bash tools/sweep_overlap.sh  -- launch-parameter sweep under frames-in-flight (headline bench)
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
run() {
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  (cd ../.. && python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['roofline']['stage_ms_serial'])") >> ../../gpurun_out/sweep_overlap.log
}
for v in 3 4 5 6; do /opt/rocm/bin/hipcc $FL -DGEOM_OCC=$v -c k_geom.hip -o k_geom.o; run "GEOM_OCC=$v"; done
/opt/rocm/bin/hipcc $FL -c k_geom.hip -o k_geom.o
for v in "VIS_WAVES=1" "VIS_WAVES=4" "VIS_OCC=4" "VIS_OCC=8"; do /opt/rocm/bin/hipcc $FL -D$v -c k_tile_vis.hip -o k_tile_vis.o; run "$v"; done
/opt/rocm/bin/hipcc $FL -c k_tile_vis.hip -o k_tile_vis.o
run "default"
