# usage (GPU box): bash tools/pmc_scene.sh <tag> <HL|C2|C3|C4|C5|C5T> -- dynamic instruction mix / stalls of one scene's kernels, one frame
# at a time (separate --pmc passes, kernel-trace only; summary lands in gpurun_out/pmcs_<tag>_<scene>.csv)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES_EQ_64"; do
  d=$R/gpurun_out/pmcs_$1_$2/$(echo $grp | tr ' ' '_')
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe/render_scene.py $2 6 > $d/log.txt 2>&1 || echo "pass failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcs_$1_$2/* > gpurun_out/pmcs_$1_$2.csv; cat gpurun_out/pmcs_$1_$2.csv
