"""is a sharded rank's frame rate bound by the host's submissions or by the GPU?  headline scene, BANDS rank 3 of N"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mt_renderer_amd import api, scene, sharding
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pal)
for world in (1, 2, 8):
    rank = min(3, world - 1)
    loop = api.FrameLoop(dev, W, H, model=model, view_proj=M, shard=(rank, world, sharding.BANDS) if world > 1 else None)
    for _ in range(3):
        fr = api.Frame(dev, W, H)
        if world > 1: fr.set_shard(rank, world, sharding.BANDS)
        model.render(fr, M); fr.end(); fr.close()
    loop.run(3000); torch.cuda.synchronize()
    for n in (8, 200, 2000):
        t0 = time.perf_counter(); loop.run(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"world={world} n={n}: host {1e6*(t1-t0)/n:.1f} us/frame, total {1e6*(t2-t0)/n:.1f} us/frame, drain {1e6*(t2-t1):.0f} us", flush=True)
