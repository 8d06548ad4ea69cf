"""One sharded rank of C4 or C5 (bands balanced as bench.py cuts them), one frame at a time: for a rocprofv3 kernel trace.
usage: python tools/probe/shard_rank_trace.py <C4|C5> <rank> <world> [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mt_renderer_amd import api, scene, sharding

name, rank, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 20
W, H = 3840, 2160
dev = api.Device(0)
vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
if name == "C4":
    mats, pals = scene.instance_lattice(16, 8)
    m = api.Model.new(dev, scene.mesh50k()); batch = api.Batch(dev, m, mats, pals)
else:
    mats, pals = scene.instance_lattice(32, 32)
    texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=True) for i in range(64)]
    m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs)); batch = api.Batch(dev, m, mats, pals, [i // 16 for i in range(1024)])
for _ in range(4):
    fr = api.Frame(dev, W, H); fr.draw_batch(batch, vp); fr.end()
    e, _ = fr.bin_counts(); fr.close()
nbx, nby, _ = sharding.grid(W, H)
bands = sharding.balanced_bands(e.reshape(nby, nbx).sum(axis=1).astype(np.float64) + 8.0 * nbx, world)
dev.set_profiling(True)
acc = {}
for it in range(frames + 4):
    fr = api.Frame(dev, W, H); fr.set_shard(rank, world, sharding.BANDS, 0, bands); fr.draw_batch(batch, vp); fr.end()
    if it >= 4:
        for k, v in fr.timings_ms().items():
            acc[k] = acc.get(k, 0.0) + v / frames
    st = fr.stats(); fr.close()
print(f"{name} rank {rank} of {world}, bands {bands}: geom stage {acc['geom']*1e3:.1f} us, tile {acc['tile']*1e3:.1f} us, kept {st['chunks'] - st['chunks_culled']} of {st['chunks']} chunks, set up {st['tris_setup']}", flush=True)
