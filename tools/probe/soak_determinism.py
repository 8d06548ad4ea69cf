"""Determinism soak: the headline frame (and a C5-like batch frame, and a sharded frame) rendered many times with frames in
flight; every k-th frame is read back and must equal the first, bit for bit.  Catches rare races (LDS hand-offs, queue
reservations, frame-slot reuse).   python tools/probe/soak_determinism.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mt_renderer_amd import api, scene, sharding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
dev = api.Device(0)
def soak(name, make, frames, every):
    fr = make(); fr.end(); ref = (fr.color().copy(), fr.depth().view(np.uint32).copy(), fr.stats()["tris_setup"], fr.stats()["bin_entries"]); fr.close()
    t0 = time.perf_counter(); bad = 0
    for i in range(frames):
        fr = make(); fr.submit()
        if i % every == every - 1:
            fr.wait(); st = fr.stats()
            if not ((fr.color() == ref[0]).all() and (fr.depth().view(np.uint32) == ref[1]).all() and st["tris_setup"] == ref[2] and st["bin_entries"] == ref[3]):
                bad += 1; print(name, "MISMATCH at frame", i, flush=True)
        fr.close()
    dev.synchronize()
    print(f"{name}: {frames} frames, {frames // every} compared, {bad} mismatches, {(time.perf_counter() - t0) / frames * 1e6:.1f} us per frame", flush=True)
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
model = api.Model.new(dev, md); model.set_palette(pal)
def headline():
    fr = api.Frame(dev, W, H); model.render(fr, M); return fr
soak("headline", headline, n, 2000)
def headline_rank3():
    fr = api.Frame(dev, W, H); fr.set_shard(3, 8, sharding.BANDS); model.render(fr, M); return fr
soak("headline rank 3 of 8", headline_rank3, n, 2000)
model.close()
W4, H4 = 3840, 2160
vp = scene.to_f32_colmajor(scene.reference_view_proj(W4, H4))
mats, pals = scene.instance_lattice(16, 8)
texs = [scene.random_bc7_texture(256, 256, seed=300 + i, opaque_modes_only=(i % 2 == 0)) for i in range(4)]
m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs)); batch = api.Batch(dev, m, mats, pals, [i % 4 for i in range(128)])
def mixed():
    fr = api.Frame(dev, W4, H4); fr.draw_batch(batch, vp); return fr
soak("128 instances, mixed opaque / translucent BC7, 4K", mixed, n // 10, 500)
def mixed_rank():
    fr = api.Frame(dev, W4, H4); fr.set_shard(1, 4, sharding.BANDS); fr.draw_batch(batch, vp); return fr
soak("the same as rank 1 of 4", mixed_rank, n // 10, 500)
print("soak done")
