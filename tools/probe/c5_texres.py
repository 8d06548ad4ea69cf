"""C5 (1024 instanced mesh50k, 64 BC7 1024x1024 textures, 3840x2160) with the textures decoded at upload (RGBA8 in HBM)
against the textures kept as BC7 blocks and decoded per fetch.
    python tools/probe/c5_texres.py [decoded|blocks|both] [frames]   (one mode per process under rocprofv3 --pmc)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mt_renderer_amd import api, scene

which = sys.argv[1] if len(sys.argv) > 1 else "both"
nframes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
W, H = 3840, 2160
dev = api.Device(0)
vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
mats, pals = scene.instance_lattice(32, 32)
for kind in ("opaque", "translucent"):
    texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=(kind == "opaque")) for i in range(64)]
    for mode, name in ((api.TEXRES_DECODED, "decoded"), (api.TEXRES_BLOCKS, "blocks")):
        if which not in ("both", name):
            continue
        dev.set_texture_residency(mode)
        free0 = torch.cuda.mem_get_info()[0]
        m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs))
        tex_bytes = free0 - torch.cuda.mem_get_info()[0]
        batch = api.Batch(dev, m, mats, pals, [i // 16 for i in range(1024)])

        def frame():
            fr = api.Frame(dev, W, H); fr.draw_batch(batch, vp); return fr
        for _ in range(3):
            fr = frame(); fr.submit(); fr.wait(); st = fr.stats(); fr.close()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nframes):
            fr = frame(); fr.submit(); fr.close()
        dev.synchronize()
        dt = (time.perf_counter() - t0) / nframes
        dev.set_profiling(True)
        acc = {}
        for _ in range(4):
            fr = frame(); fr.end()
            for k, v in fr.timings_ms().items():
                acc[k] = acc.get(k, 0.0) + v / 4
            fr.close()
        dev.set_profiling(False)
        print(f"C5 {kind:11s} textures {name:8s}: {dt*1e3:.3f} ms/frame pipelined; serial geom {acc['geom']*1e3:.0f} us tile {acc['tile']*1e3:.0f} us; "
              f"tile kernel {st['tile_kernel']}; model + textures {tex_bytes / 2**20:.0f} MiB of HBM", flush=True)
        batch.close(); m.close()
dev.close()
