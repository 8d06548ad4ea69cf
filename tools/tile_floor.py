"""Tuning aid (GPU): tile-kernel time of nearly empty frames (fixed per-frame floor)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mt_renderer_amd import api, scene

dev = api.Device(0)
dev.set_profiling(True)
W, H = 1920, 1080


def t(name, fn, n=20):
    acc = {}
    for i in range(n + 3):
        fr = api.Frame(dev, W, H)
        fn(fr)
        fr.end()
        if i >= 3:
            for k, v in fr.timings_ms().items():
                acc[k] = acc.get(k, 0) + v / n
        st = fr.stats()
        fr.close()
    print(name, {k: round(v * 1e3, 1) for k, v in acc.items()}, "us; entries", st["bin_entries"], "kernel", st["tile_kernel"], flush=True)


t("empty frame", lambda fr: None)
cube = api.Model.new(dev, scene.cube_model(1))
Mc = scene.to_f32_colmajor(scene.cube_transform(W, H))
t("cube (12 tris)", lambda fr: cube.render(fr, Mc))
m = api.Model.new(dev, scene.mesh50k())
m.set_palette(scene.bone_palette())
M = scene.to_f32_colmajor(scene.headline_transform(W, H))
t("C2 mesh50k", lambda fr: m.render(fr, M))
for mode in (1,):
    dev.set_tile_mode(mode)
    t("C2 mesh50k ordered kernel", lambda fr: m.render(fr, M))
    t("empty frame ordered kernel", lambda fr: None)
