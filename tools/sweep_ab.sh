# usage (GPU box): bash tools/sweep_ab.sh "<flags A>" "<flags B>" [file=k_geom] [object it replaces=file]  -- paired A/B of two builds of one kernel file
set -e
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
F=${3:-k_geom}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
for rep in 1 2 3; do for v in "$1" "$2"; do
  /opt/rocm/bin/hipcc $FL $v -c $F.hip -o ${4:-$F}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
  (cd ../.. && python bench.py --steps 400 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['ms_per_step'], d['roofline']['stage_ms_serial']['geom'], d['roofline']['stage_ms_serial']['tile'])") >> ../../gpurun_out/sweep_ab.log
done; done
/opt/rocm/bin/hipcc $FL -c ${4:-$F}.hip -o ${4:-$F}.o
