"""Per-rank cost of the LARGE configurations under N-way bin sharding, on ONE GPU (shard 0 of N, no exchange): C4 = 128
instances of mesh50k at 4K (6.4 M triangles), C5 = 1024 instances with 64 BC7 textures at 4K (51 M triangles).  What a
rank has left to do is what bounds the multi-GPU frame rate before the all-gather (33 MB per 4K frame) costs anything."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mt_renderer_amd import api, scene

dev = api.Device(0)
W, H = 3840, 2160
vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))


def run(name, md, mats, pals, tex_override, nframes):
    m = api.Model.new(dev, md)
    batch = api.Batch(dev, m, mats, pals, tex_override)
    base = None
    for world in (1, 2, 4, 8):
        def one(wait=False):
            fr = api.Frame(dev, W, H); fr.set_shard(0, world); fr.draw_batch(batch, vp); fr.submit()
            if wait:
                fr.wait()
            return fr
        fr = one(True); st = fr.stats(); fr.close()
        t_end = time.perf_counter() + 0.4
        while time.perf_counter() < t_end:
            one().close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nframes):
            one().close()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nframes
        base = base or dt
        print(f"{name} world={world}: {dt*1e3:.3f} ms/frame per rank = {base/dt:.2f}x of one GPU (ideal {world}x), "
              f"kept {st['tris_setup']} triangles, {st['bin_entries']} queue entries, gather {W*H*4*(world-1)/world/1e6:.1f} MB in per rank", flush=True)
    batch.close(); m.close()


mats, pals = scene.instance_lattice(16, 8)
run("C4 128 inst 4K", scene.mesh50k(), mats, pals, None, 200)
mats, pals = scene.instance_lattice(32, 32)
texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=True) for i in range(64)]
run("C5 1024 inst BC7 4K", scene.mesh50k(textured=True, textures=texs), mats, pals, [i // 16 for i in range(1024)], 40)
