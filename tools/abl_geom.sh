# usage (GPU box): bash tools/abl_geom.sh <scene>  -- SQ_INSTS_VALU / SALU / LDS and time of k_geom, full kernel and each ablation
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-C5}
for n in 0 1 2 3 4 5 6; do
  if [ $n = 0 ]; then unset MTR_LIB_PATH; else export MTR_LIB_PATH=$R/mt_renderer_amd/libmtr_abl$n.so; fi
  d=$R/gpurun_out/abl_${S}_$n
  rm -rf $d; mkdir -p $d
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe/render_scene.py $S 4 > $d/log.txt 2>&1 || echo "pass failed: $n"
  echo "ABL=$n $(grep -h "^$S" $d/log.txt | cut -c1-80) $(cd $R && python tools/pmc_summary.py $d | grep 'k_geom<2, false' | cut -d, -f3-)"
done
