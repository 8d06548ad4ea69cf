# usage (GPU box): bash tools/probe/c5_texres_pmc.sh : FETCH_SIZE of the tile kernels on C5, textures decoded at upload vs kept as BC7 blocks
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in decoded blocks; do
  d=$R/gpurun_out/c5_texres_pmc/$mode
  rm -rf $d; mkdir -p $d
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $d -- python3 $R/tools/probe/c5_texres.py $mode 6 > $d/log.txt 2>&1 || echo "pass failed: $mode"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for mode in ("decoded", "blocks"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/c5_texres_pmc/{mode}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "k_tile" in r["Kernel_Name"] and int(r["Grid_Size"]) > 1000000:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        v.sort()
        print(f"{mode:8s} {k:48s} launches {len(v):3d}  FETCH_SIZE median {v[len(v)//2]/1024:8.1f} MB x2 (gfx950: 128-B requests tallied at 64 B) = {2*v[len(v)//2]/1024:8.1f} MB")
PY
