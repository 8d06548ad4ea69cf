"""Tuning aid (GPU): per-bin queue statistics of the headline scene."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mt_renderer_amd import api, scene

W, H = 1920, 1080
dev = api.Device(0)
md = scene.headline_model()
m = api.Model.new(dev, md)
m.set_palette(scene.bone_palette())
fr = api.Frame(dev, W, H)
m.render(fr, scene.to_f32_colmajor(scene.headline_transform(W, H)))
fr.end()
e, s = fr.bin_counts()
print("stats", fr.stats())
for name, a in (("entries", e), ("segments", s)):
    nz = a[a > 0]
    print(name, "bins>0:", len(nz), "mean", nz.mean().round(1), "p50", np.percentile(nz, 50), "p90", np.percentile(nz, 90),
          "p99", np.percentile(nz, 99), "max", nz.max(), "sum", a.sum())
print("top entries:", np.sort(e)[-12:])
print("bins with >64 segs:", int((s > 64).sum()))
