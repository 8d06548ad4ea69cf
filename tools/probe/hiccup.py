import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pal)
def one():
    fr = api.Frame(dev, W, H); model.render(fr, M); fr.submit(); fr.close()
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); fr.close()
torch.cuda.synchronize()
ts = [time.perf_counter()]
for i in range(4000):
    one()
    if i % 100 == 99: ts.append(time.perf_counter())
torch.cuda.synchronize()
print("per-100-frame us/frame:", " ".join(f"{(b-a)*1e4:.0f}" for a, b in zip(ts, ts[1:])))
