# usage (here, no GPU needed): bash tools/abl_build.sh  -- builds mt_renderer_amd/libmtr_abl<N>.so, N = 1..6: libmtr.so with one part of
# k_geom cut out (MTR_ABL in csrc/k_geom.hip); tools/abl_geom.sh then counts instructions per variant on the GPU box
set -e
cd "$(dirname "$0")/../mt_renderer_amd/csrc"
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-unused-function -Wno-missing-braces"
for n in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc $FL -DMTR_ABL=$n -c k_geom.hip -o /tmp/k_geom_abl$n.o &
done
wait
for n in 1 2 3 4 5 6; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr_abl$n.so /tmp/k_geom_abl$n.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
done
ls -la ../libmtr_abl*.so
