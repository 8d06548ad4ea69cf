"""Headline scene with a NEW bone palette before every frame (an animated model): frames must still overlap."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene
W, H = 1920, 1080
md = scene.headline_model(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
pals = [scene.bone_palette(t=0.01 * k) for k in range(64)]
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pals[0])
def one(k, animate):
    if animate: model.set_palette(pals[k % 64])
    fr = api.Frame(dev, W, H); model.render(fr, M); fr.submit(); fr.close()
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); fr.close()
for animate in (False, True, False, True):
    t_end = time.perf_counter() + 0.3
    k = 0
    while time.perf_counter() < t_end:
        one(k, animate); k += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2000): one(k, animate)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2000
    print(f"palette per frame={animate}: {dt*1e6:.1f} us/frame, {1e-3/dt:.2f} Gtri/s")
