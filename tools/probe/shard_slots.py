"""frames-in-flight sweep for one sharded configuration: headline scene, BANDS (balanced), rank r of N, MTR_NSLOTS from the environment"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from mt_renderer_amd import api, scene, sharding
W, H = 1920, 1080
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pal)
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); e, _ = fr.bin_counts(); fr.close()
nbx, nby, _ = sharding.grid(W, H)
bands = sharding.balanced_bands(e.reshape(nby, nbx).sum(axis=1) + 8.0 * nbx, world)
out = []
for rank in range(world):
    loop = api.FrameLoop(dev, W, H, model=model, view_proj=M, shard=(rank, world, sharding.BANDS, 0, bands))
    def one():
        loop.run(1)
    for _ in range(3):
        fr = api.Frame(dev, W, H); fr.set_shard(rank, world, sharding.BANDS, 0, bands); model.render(fr, M); fr.end(); fr.close()
    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        for _ in range(50): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.run(600)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 600 * 1e6)
print(f"NSLOTS={os.environ.get('MTR_NSLOTS','3')} VIS_WAVES={os.environ.get('MTR_VIS_WAVES','auto')} world={world}: per-rank us/frame {[round(x,1) for x in out]} worst {max(out):.1f}", flush=True)
