"""k_geom duration against the number of workgroups: what one wave's latency chain costs (unsharded, 1080p, debug shader)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene
W, H = 1920, 1080
pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0)
big = api.Model.new(dev, scene.headline_model()); big.set_palette(pal)
loop = api.FrameLoop(dev, W, H, model=big, view_proj=M)
fr = api.Frame(dev, W, H); big.render(fr, M); fr.end(); fr.close()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: loop.run(50)
torch.cuda.synchronize()
dev.set_profiling(True)
for rows, cols in ((1, 100), (2, 200), (8, 200), (30, 200), (125, 200)):
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=rows, cols=cols)
    m = api.Model.new(dev, md); m.set_palette(pal)
    acc = {}
    for it in range(40):
        loop.run(2)  # keep the clocks up
        fr = api.Frame(dev, W, H); m.render(fr, M); fr.end()
        if it >= 10:
            for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 30
        st = fr.stats(); fr.close()
    print(f"rows={rows} cols={cols}: chunks {st['chunks']} (WGs {(st['chunks']+3)//4}) tris {st['tris_in']}: geom {acc['geom']*1e3:.1f} us tile {acc['tile']*1e3:.1f} us", flush=True)
    m.close()
