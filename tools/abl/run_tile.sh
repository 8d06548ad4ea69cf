set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
make -s k_geom.o  # the other kernel: the stock build (an earlier ablation may have left a variant behind)
python ../../tools/abl/make_tile_abl.py
cp ../../tools/abl/k_tile_vis_abl.hip ./k_tile_vis_abl.hip
for v in "ABL_NONE" "ABL_T_NOFLAT" "ABL_T_NOCOOP" "ABL_T_NOSETUP" "ABL_T_NOWINNER"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces -D$v -c k_tile_vis_abl.hip -o k_tile_vis.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o -lz
  cd ../..
  echo "variant=$v" >> gpurun_out/abl_tile.log
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HL', d['ms_per_step'], d['roofline']['stage_ms_serial'])" >> gpurun_out/abl_tile.log
  cd mt_renderer_amd/csrc
done
rm -f k_tile_vis_abl.hip
