"""Renders N frames of one named scene, one frame at a time (for rocprofv3 --pmc / --kernel-trace passes of a single config).
usage: python tools/probe/render_scene.py <HL|C2|C3|C4|C5|C5T> [frames]      C5T = C5 with every texture translucent"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mt_renderer_amd import api, scene

name = sys.argv[1] if len(sys.argv) > 1 else "HL"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = api.Device(0)
batch = None
if name in ("HL", "C2"):
    W, H = 1920, 1080
    m = api.Model.new(dev, scene.headline_model() if name == "HL" else scene.mesh50k())
    m.set_palette(scene.bone_palette())
    view = scene.to_f32_colmajor(scene.headline_transform(W, H))
else:
    W, H = (1920, 1080) if name == "C3" else (3840, 2160)
    view = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
    if name in ("C3", "C4"):
        mats, pals = scene.instance_lattice(16, 8)
        m = api.Model.new(dev, scene.mesh50k())
        batch = api.Batch(dev, m, mats, pals)
    else:
        mats, pals = scene.instance_lattice(32, 32)
        texs = [scene.random_bc7_texture(1024, 1024, seed=200 + i, opaque_modes_only=(name == "C5")) for i in range(64)]
        m = api.Model.new(dev, scene.mesh50k(textured=True, textures=texs))
        batch = api.Batch(dev, m, mats, pals, [i // 16 for i in range(1024)])
dev.set_profiling(True)
acc = {}
for it in range(frames + 3):
    fr = api.Frame(dev, W, H)
    if batch is not None:
        fr.draw_batch(batch, view)
    else:
        m.render(fr, view)
    fr.end()
    if it >= 3:
        for k, v in fr.timings_ms().items():
            acc[k] = acc.get(k, 0.0) + v / frames
    st = fr.stats()
    fr.close()
print(name, {k: round(v * 1e3, 1) for k, v in acc.items()}, "us;", {k: st[k] for k in ("tris_in", "tris_setup", "bin_entries", "chunks", "binning", "tile_kernel")}, flush=True)
