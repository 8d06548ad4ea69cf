set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
make -s k_tile_vis.o  # the other kernel: the stock build (an earlier ablation may have left a variant behind)
python ../../tools/abl/make_geom_abl.py
cp ../../tools/abl/k_geom_abl.hip ./k_geom_abl.hip
for v in "ABL_NONE" "ABL_NOBIN" "ABL_NOBIN -DABL_NOREC" "ABL_EARLY" "ABL_EARLY -DABL_NOSKIN"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces -D$v -c k_geom_abl.hip -o k_geom.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o -lz
  cd ../..
  echo "variant=$v" >> gpurun_out/abl_geom.log
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HL', d['ms_per_step'], d['roofline']['stage_ms_serial'])" >> gpurun_out/abl_geom.log
  cd mt_renderer_amd/csrc
done
rm -f k_geom_abl.hip
