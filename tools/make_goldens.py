"""Generates the committed golden vectors under tests/golden/ from the CPU oracle.

The reference holds no pixel / depth / vertex golden of any kind: these vectors pin THIS
BUILD's normative semantics (SPEC.md) so that neither the oracle nor the HIP path can drift silently.
Small scenes only (KBs).  Run: python tools/make_goldens.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mt_renderer_amd import scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from tests.golden_scenes import SCENES  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def main():
    index = {}
    for name, build in SCENES.items():
        w, h, draws = build()
        from tests.helpers import render_oracle
        col, dep, st = render_oracle(w, h, draws)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), color=col, depth=dep)
        index[name] = dict(width=w, height=h, stats=st, color_sha256=hashlib.sha256(col.tobytes()).hexdigest(),
                           depth_sha256=hashlib.sha256(dep.tobytes()).hexdigest())
        print(name, st)
    # full-size scenes: hashes only
    for name, (w, h, md, M, pal) in {
        "c2_mesh50k_1080p": (1920, 1080, scene.mesh50k(), scene.headline_transform(1920, 1080), scene.bone_palette()),
        "headline_1m_1080p": (1920, 1080, scene.headline_model(), scene.headline_transform(1920, 1080), scene.bone_palette()),
    }.items():
        f = orc.OracleFrame(w, h)
        f.draw(orc.OracleModel(md), scene.to_f32_colmajor(M), pal, nthreads=8)
        index[name] = dict(width=w, height=h, stats=f.stats(), color_sha256=hashlib.sha256(f.color().tobytes()).hexdigest(),
                           depth_sha256=hashlib.sha256(f.depth().tobytes()).hexdigest())
        print(name, f.stats())
    json.dump(index, open(os.path.join(OUT, "index.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
