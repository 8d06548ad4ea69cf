"""C5 (1024 instances, BC7, 4K) sharded BANDS: serial kernel times for rank r of N (env knobs in the caller)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene, sharding
W, H = 3840, 2160
dev = api.Device(0)
vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
mats, pals = scene.instance_lattice(32, 32)
texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=True) for i in range(64)]
m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs)); batch = api.Batch(dev, m, mats, pals, [i // 16 for i in range(1024)])
for _ in range(4):
    fr = api.Frame(dev, W, H); fr.draw_batch(batch, vp); fr.end(); fr.close()
dev.set_profiling(True)
for world, rank in ((1, 0), (2, 0), (8, 3)):
    acc = {}
    for it in range(12):
        fr = api.Frame(dev, W, H)
        if world > 1: fr.set_shard(rank, world, sharding.BANDS)
        fr.draw_batch(batch, vp); fr.end()
        if it >= 4:
            for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0) + v / 8
        st = fr.stats(); fr.close()
    print(f"NOLOOP={os.environ.get('MTR_GEOM_NOLOOP','0')} world={world} rank={rank}: geom {acc['geom']*1e3:.1f} us tile {acc['tile']*1e3:.1f} us setup {st['tris_setup']} culled {st['chunks_culled']}/{st['chunks']}", flush=True)
