# usage (GPU box): bash tools/pmc_quick.sh <tag>: FETCH_SIZE / WRITE_SIZE per kernel of the headline bench (separate passes)
T=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  d=$R/gpurun_out/${T}_pmc/$grp
  mkdir -p $d
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $d/log.txt 2>&1 || echo "pass failed: $grp"
done
cd $R && python tools/pmc_summary.py gpurun_out/${T}_pmc/* | grep -E "k_geom|k_tile"
