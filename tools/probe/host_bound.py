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
for _ in range(30): one()
torch.cuda.synchronize()
for n in (200, 200, 1000, 200, 3000, 200):
    t0 = time.perf_counter()
    for _ in range(n): one()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host submit {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame, drain {1e6*(t2-t1):.0f} us")
# split of the host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): one()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumtime").print_stats(12)
