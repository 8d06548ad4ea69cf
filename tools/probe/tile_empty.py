"""how long does the tile kernel take when every bin is empty (pure dispatch + clear), at several frame sizes?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mt_renderer_amd import api, scene
dev = api.Device(0)
md = scene.mesh50k(rows=12, cols=20)
model = api.Model.new(dev, md); model.set_palette(scene.bone_palette())
big = api.Model.new(dev, scene.headline_model()); big.set_palette(scene.bone_palette())
M_off = scene.to_f32_colmajor(scene.reference_view_proj(1920, 1080) @ scene.mat_translate(500.0, 0.0, 0.0))  # far off screen: every triangle rejected
loop = api.FrameLoop(dev, 1920, 1080, model=big, view_proj=scene.to_f32_colmajor(scene.headline_transform(1920, 1080)))
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: loop.run(50)
torch.cuda.synchronize()
dev.set_profiling(True)
for (w, h) in ((640, 360), (1280, 720), (1920, 1080), (3840, 2160)):
    acc = {}
    for it in range(30):
        loop.run(2)
        fr = api.Frame(dev, w, h); model.render(fr, M_off); fr.end()
        if it >= 5:
            for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 25
        st = fr.stats(); fr.close()
    print(f"{w}x{h}: bins {st['nbins']} setup {st['tris_setup']}: geom {acc['geom']*1e3:.1f} us, tile (all bins empty) {acc['tile']*1e3:.1f} us", flush=True)
