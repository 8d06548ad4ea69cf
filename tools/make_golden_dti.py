"""Extracts (class name, hash) known-answer pairs for the MT crc32 from the reference's DTI table
(/root/reference/src/dti.txt, the data file behind the reference's `test_dti_hashes`,
src/dti.rs:169-193: hash == crc32(name, 0xFFFFFFFF) & 0x7fffffff).  Every 8th entry plus the three
classes on the draw path are kept.  Output: tests/golden/dti_crc32_kat.json"""
import json
import os

SRC = "/root/reference/src/dti.txt"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dti_crc32_kat.json")

rows = [json.loads(l) for l in open(SRC) if l.strip()]
keep = {"rModel", "rTexture", "rMaterial", "rArchive", "rShader2", "rScheduler"}
out = [{"name": r["name"], "hash": r["hash"]} for i, r in enumerate(rows) if i % 8 == 0 or r["name"] in keep]
json.dump({"source": "src/dti.txt (reference data file), rule src/dti.rs:174", "pairs": out}, open(OUT, "w"), indent=0)
print(len(out), "pairs ->", OUT)
