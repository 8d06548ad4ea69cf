# usage (GPU box): bash tools/probe/cull_keepall.sh : the culled geometry path with culling that keeps everything (MTR_CULL_DEBUG
# 3: chunk tests run, all kept; 4: no chunk tests, every instance "inside") against the unsharded kernel, C5, kernel trace
R=$GRAFT_REPO_ROOT
for dbg in 3 4; do
  export MTR_CULL_DEBUG=$dbg
  echo "== MTR_CULL_DEBUG=$dbg"
  bash $R/tools/probe/prof_probe.sh tools/probe/c5_shard_ab.py 2>&1 | grep -E "k_cull|k_geom" | head -6 | cut -c1-170
done
