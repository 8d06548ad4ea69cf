# usage (GPU box): bash tools/ab_configs.sh "<flags A>" "<flags B>" [file=k_geom]  -- two builds of one kernel file: headline bench + every config of tools/bench_configs.py for each
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
F=${3:-k_geom}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
OUT=../../gpurun_out/ab_configs.log
for rep in 1 2; do for v in "$1" "$2"; do
  /opt/rocm/bin/hipcc $FL $v -c $F.hip -o $F.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  echo "== [$v] rep $rep" >> $OUT
  (cd ../.. && python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['value'], d['ms_per_step'], d['roofline']['stage_ms_serial'], d['latency']['ms_per_frame_latency'], {k: d['frame_stats'][k] for k in ('tris_setup', 'bin_entries')})") >> $OUT
  if [ $rep = 1 ]; then (cd ../.. && python tools/bench_configs.py C) >> $OUT 2>&1; fi
done; done
/opt/rocm/bin/hipcc $FL -c $F.hip -o $F.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
