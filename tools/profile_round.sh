# usage (GPU box): bash tools/profile_round.sh <tag>   e.g. r01_d
# rocprofv3 evidence for bench.py's numbers: kernel-trace stats of the default bench command, then separate --pmc
# passes (never combined with other tracing).  Results are condensed into gpurun_out/<tag>_*; copy them to profiles/.
set -e
T=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --steps 200 --warmup 20 > $R/gpurun_out/${T}_bench_line.json 2> $R/gpurun_out/${T}_bench_err.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_stats -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/${T}_bench_line_under_rocprof.json 2> $R/gpurun_out/${T}_stats.log
cp $(ls $R/gpurun_out/${T}_stats/*/*kernel_stats.csv | head -1) $R/gpurun_out/${T}_kernel_stats_headline.csv
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  d=$R/gpurun_out/${T}_pmc/$(echo $grp | tr ' ' '_')
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $d/log.txt 2>&1 || echo "pass failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/${T}_pmc/* > gpurun_out/${T}_pmc_headline.csv
python tools/pmc_traffic_json.py gpurun_out/${T}_pmc ${T} gpurun_out/${T}_pmc_traffic.json  # -> profiles/pmc_traffic.json (bench.py reads it)
cat gpurun_out/${T}_bench_line.json; head -6 gpurun_out/${T}_kernel_stats_headline.csv; cat gpurun_out/${T}_pmc_headline.csv
