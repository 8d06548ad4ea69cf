"""timing ablation of the sharded geometry kernel (MTR_CULL_DEBUG: 0 = no test (unsharded kernel variant), 1 = normal,
2 = every chunk leaves at once, 3 = test but keep all); clocks warmed up first; serial kernel times (one frame at a time)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene, sharding
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pal)
loop = api.FrameLoop(dev, W, H, model=model, view_proj=M)
fr = api.Frame(dev, W, H); model.render(fr, M); fr.end(); fr.close()
t_end = time.perf_counter() + 0.5
while time.perf_counter() < t_end: loop.run(50)
torch.cuda.synchronize()
dev.set_profiling(True)
for world, own_map, rank in ((1, 0, 0), (8, 0, 3), (8, 1, 3), (8, 1, 0), (2, 1, 0), (8, 2, 3)):
    acc = {}
    for it in range(40):
        fr = api.Frame(dev, W, H)
        if world > 1: fr.set_shard(rank, world, own_map, 3 if own_map == 2 else 0)
        model.render(fr, M); fr.end()
        if it >= 10:
            for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 30
        st = fr.stats(); fr.close()
    print(f"MTR_CULL_DEBUG={os.environ.get('MTR_CULL_DEBUG')} world={world} map={own_map} rank={rank}: geom {acc['geom']*1e3:.1f} us tile {acc['tile']*1e3:.1f} us culled {st['chunks_culled']}/{st['chunks']} setup {st['tris_setup']}", flush=True)
