"""GPU timings of the other BASELINE.json configs (C2, C3, C4 on one GPU, C5) -- supplementary to bench.py, which
times the headline scene.  Prints one line per config: ms/frame (hipEvent stage sum), Mtris/s, stage split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mt_renderer_amd import api, scene

dev = api.Device(0)
KERNEL_NAMES = {1: 'ordered', 2: 'vis', 3: 'mixed'}


ONLY = sys.argv[1] if len(sys.argv) > 1 else ""  # optional substring filter on the config name


def run(name, w, h, md, draw, nframes=30):
    if ONLY not in name:
        return
    m = api.Model.new(dev, md)
    batch = None
    if "model_mats" in draw:
        batch = api.Batch(dev, m, draw["model_mats"], draw.get("palettes"), draw.get("tex_override"))
    else:
        m.set_palette(draw.get("palette"))

    def frame(wait):
        fr = api.Frame(dev, w, h)
        if batch:
            fr.draw_batch(batch, draw["vp"])
        else:
            m.render(fr, draw["M"])
        fr.submit()
        if wait:
            fr.wait()
        return fr
    for _ in range(3):
        frame(True).close()
    dev.set_profiling(True)
    acc, st = {}, None
    for _ in range(nframes):
        fr = frame(True)
        for k, v in fr.timings_ms().items():
            acc[k] = acc.get(k, 0.0) + v / nframes
        st = fr.stats()
        fr.close()
    dev.set_profiling(False)
    t0 = time.perf_counter()
    frs = [frame(False) for _ in range(nframes)]
    frs[-1].wait()
    dt = (time.perf_counter() - t0) / nframes
    for fr in frs:
        fr.close()
    print(f"{name}: {dt*1e3:.3f} ms/frame  {st['tris_in']/dt/1e6:.0f} Mtris/s  tris_in={st['tris_in']} setup={st['tris_setup']} "
          f"entries={st['bin_entries']} kernel={KERNEL_NAMES[st['tile_kernel']]} binning={st['binning']} "
          f"stages_ms={ {k: round(v, 4) for k, v in acc.items()} }", flush=True)
    if batch:
        batch.close()
    m.close()


W, H = 1920, 1080
run("C2  mesh50k 1080p", W, H, scene.mesh50k(), dict(M=scene.to_f32_colmajor(scene.headline_transform(W, H)), palette=scene.bone_palette()))
run("HL  1M tris 1080p", W, H, scene.headline_model(), dict(M=scene.to_f32_colmajor(scene.headline_transform(W, H)), palette=scene.bone_palette()))
mats, pals = scene.instance_lattice(16, 8)
run("C3  128 inst 1080p", W, H, scene.mesh50k(), dict(vp=scene.to_f32_colmajor(scene.reference_view_proj(W, H)), model_mats=mats, palettes=pals))
W, H = 3840, 2160
run("C4  128 inst 4K (1 GPU)", W, H, scene.mesh50k(), dict(vp=scene.to_f32_colmajor(scene.reference_view_proj(W, H)), model_mats=mats, palettes=pals))
mats, pals = scene.instance_lattice(32, 32)
for kind in ("opaque", "translucent", "8 of 64 translucent"):
    texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=(kind == "opaque" or (kind != "translucent" and i % 8 != 0)))
            for i in range(64)]
    md = scene.mesh50k(textured=True, textures=texs)
    run(f"C5  1024 inst BC7 {kind} 4K (1 GPU)", W, H, md,
        dict(vp=scene.to_f32_colmajor(scene.reference_view_proj(W, H)), model_mats=mats, palettes=pals,
             tex_override=[i // 16 for i in range(1024)]), nframes=10)
