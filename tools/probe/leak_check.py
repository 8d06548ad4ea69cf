"""Host and device memory after many frames (static, animated palette, per-frame batches): must be flat."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import psutil, torch
from mt_renderer_amd import api, scene
W, H = 640, 360
md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=20, cols=30)
M = scene.to_f32_colmajor(scene.headline_transform(W, H)); vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
pals = [scene.bone_palette(t=0.01 * k) for k in range(16)]
mats, ipal = scene.instance_lattice(2, 2)
dev = api.Device(0); model = api.Model.new(dev, md); model.set_palette(pals[0])
proc = psutil.Process()
def mem():
    free, total = torch.cuda.mem_get_info()
    return proc.memory_info().rss / 2**20, (total - free) / 2**20
def loop(n):
    for k in range(n):
        model.set_palette(pals[k % 16])
        fr = api.Frame(dev, W, H); model.render(fr, M)
        if k % 3 == 0: fr.draw_instances(model, vp, mats, ipal)
        fr.submit()
        if k % 50 == 0: fr.wait(); fr.stats()
        fr.close()
    torch.cuda.synchronize()
loop(2000)
r0, d0 = mem()
loop(20000)
r1, d1 = mem()
loop(20000)
r2, d2 = mem()
print(f"host RSS MiB: {r0:.1f} -> {r1:.1f} -> {r2:.1f}; device used MiB: {d0:.1f} -> {d1:.1f} -> {d2:.1f}")
ok = (r2 - r1) < 8 and (d2 - d1) < 8
print("flat" if ok else "GROWING")
sys.exit(0 if ok else 1)
