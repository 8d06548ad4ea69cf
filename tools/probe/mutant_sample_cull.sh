# usage (GPU box): bash tools/probe/mutant_sample_cull.sh  -- the test of the test: builds k_geom with the top-left bias of sample_cull broken (every edge
# exclusive) and runs tests/test_gpu_small_triangles.py, which must FAIL (31 of 31 did).  Leaves the box copy mutated; never run it in the tree you ship.
cd $GRAFT_REPO_ROOT/mt_renderer_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wno-missing-braces"
sed -i 's|const int32_t bias = (((dy << 12) + ndx) - 1) >> 31;|const int32_t bias = -1; /* MUTANT: every edge exclusive */|' k_geom.hip
grep -c MUTANT k_geom.hip
/opt/rocm/bin/hipcc $FL -c k_geom.hip -o k_geom.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtr.so k_geom.o k_bin.o k_tile.o k_tile_vis.o k_texture.o k_shard.o mtr_api.o mtr_files.o mtr_group.o -lz
cd ../..
timeout -k 10 600 python -m pytest tests/test_gpu_small_triangles.py -q -m gpu 2>&1 | tail -4
