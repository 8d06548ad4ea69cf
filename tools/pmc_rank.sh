# usage (GPU box): bash tools/pmc_rank.sh <tag> <C4|C5> <rank> <world>  -- instruction mix / stalls of one sharded rank's kernels, one frame at a
# time (separate --pmc passes, kernel-trace only; summary in gpurun_out/pmcr_<tag>_<scene>_<rank>of<world>.csv)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU"; do
  d=$R/gpurun_out/pmcr_$1_$2_$3of$4/$(echo $grp | tr ' ' '_')
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe/shard_rank_trace.py $2 $3 $4 6 > $d/log.txt 2>&1 || echo "pass failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcr_$1_$2_$3of$4/* > gpurun_out/pmcr_$1_$2_$3of$4.csv; grep "k_geom\|k_cull" gpurun_out/pmcr_$1_$2_$3of$4.csv
