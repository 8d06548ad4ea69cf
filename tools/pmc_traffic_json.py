"""profiles/pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh:
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B,
MI355X_MICROARCH.md "HBM"), per kernel, tagged with the hash of the kernel sources the run was made on (bench.py prints
`traffic: null` when the sources it runs hash differently).   usage: python tools/pmc_traffic_json.py <pmc dir> <tag> <out.json>"""
import collections, csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

root, tag, out = sys.argv[1], sys.argv[2], sys.argv[3]
vals = collections.defaultdict(dict)
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(root, ctr, "*", "*counter_collection.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "mtr::" in r["Kernel_Name"]:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            name = re.sub(r"^void ", "", k).split("(")[0].replace("mtr::", "")
            vals[name][ctr] = sum(v) / len(v)
            vals[name]["launches_" + ctr] = len(v)
res = {"kernel_source_hash": bench.kernel_source_hash(), "tag": tag, "command": "bench.py --steps 10 --warmup 2 --no-cpu-baseline under rocprofv3 --pmc <counter> --kernel-trace (one pass per counter)",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KB; bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
       "raw_kb_per_launch": vals,
       "bytes_per_launch": {k: int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) for k, v in vals.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res["bytes_per_launch"], indent=1))
