# usage (GPU box): bash tools/pmc_insts.sh <tag>   -- dynamic instruction mix of the headline frame's kernels
# (separate --pmc passes, kernel-trace only; summaries land in gpurun_out/pmc_<tag>.csv)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  d=$R/gpurun_out/pmc_$1/$(echo $grp | tr ' ' '_')
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $d/log.txt 2>&1 || echo "pass failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc_$1/* > gpurun_out/pmc_$1.csv; cat gpurun_out/pmc_$1.csv
