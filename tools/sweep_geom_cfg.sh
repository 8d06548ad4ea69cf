# usage (GPU box): bash tools/sweep_geom_cfg.sh "<flags>" ...   -- k_geom variants on the instanced configs (C3, C5 opaque)
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL $v -c k_geom.hip -o k_geom.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  echo "== $v"; (cd ../.. && python tools/bench_configs.py "C3" 2>&1 | grep "C3" | sed 's/tris_in.*stages_ms/stages_ms/'; python tools/bench_configs.py "BC7 opaque" 2>&1 | grep "C5" | sed 's/tris_in.*stages_ms/stages_ms/')
done
/opt/rocm/bin/hipcc $FL -c k_geom.hip -o k_geom.o
