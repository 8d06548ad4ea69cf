"""Per-rank cost of a sharded headline frame on ONE GPU (no exchange): what N-way bin sharding leaves each rank to do."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pal)
for world in (1, 2, 4, 8):
    def one():
        fr = api.Frame(dev, W, H); fr.set_shard(0, world); model.render(fr, M); fr.submit(); fr.close()
    fr = api.Frame(dev, W, H); fr.set_shard(0, world); model.render(fr, M); fr.end(); fr.close()
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(50): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(400): one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 400
    dev.set_profiling(True)
    acc = {}
    for _ in range(10):
        fr = api.Frame(dev, W, H); fr.set_shard(0, world); model.render(fr, M); fr.end()
        for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 10
        st = fr.stats(); fr.close()
    dev.set_profiling(False)
    print(f"world={world}: {dt*1e6:.1f} us/frame per rank (-> {world/dt/1e3:.1f} Gtri/s if N ranks ran in parallel, no exchange), serial geom {acc['geom']*1e3:.1f} tile {acc['tile']*1e3:.1f} us, setup={st['tris_setup']} entries={st['bin_entries']}")
