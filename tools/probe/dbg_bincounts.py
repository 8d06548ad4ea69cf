"""bin_counts() of a frame after frames of another size ran on the device: are they this frame's?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mt_renderer_amd import api, scene, sharding
dev = api.Device(0)
W, H = 1920, 1080
md = scene.headline_model(); pal = scene.bone_palette(); M = scene.to_f32_colmajor(scene.headline_transform(W, H))
model = api.Model.new(dev, md); model.set_palette(pal)
for i in range(50):
    fr = api.Frame(dev, W, H); model.render(fr, M); fr.submit(); fr.close()
dev.synchronize()
if len(sys.argv) > 1:  # a sharded phase first
    which = {"b": [(sharding.BANDS, 0)], "s": [(sharding.SUPERTILES, 3)], "i": [(sharding.INTERLEAVED, 0)]}.get(os.environ.get("DBG_MAPS", ""), [(sharding.BANDS, 0), (sharding.SUPERTILES, 3), (sharding.INTERLEAVED, 0)])
    if os.environ.get("DBG_NOCULL"): dev.set_culling(False)
    for own_map, param in which:
        for rank in range(8):
            for i in range(30):
                fr = api.Frame(dev, W, H); fr.set_shard(rank, 8, own_map, param); model.render(fr, M); fr.submit(); fr.close()
    dev.synchronize()
    if sys.argv[1] == "close": model.close()
W, H = 3840, 2160
vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
mats, pals = scene.instance_lattice(16, 8)
m = api.Model.new(dev, scene.mesh50k()); batch = api.Batch(dev, m, mats, pals, None)
for nwarm in (0, 50):
    for i in range(nwarm):
        fr = api.Frame(dev, W, H); fr.draw_batch(batch, vp); fr.submit(); fr.close()
    for k in range(3):
        fr = api.Frame(dev, W, H); fr.draw_batch(batch, vp); fr.submit(); fr.wait()
        e, sg = fr.bin_counts(); st = fr.stats(); fr.close()
        if int(e.sum()) != st["bin_entries"]:
            bad = np.nonzero(e > 100000)[0]
            print("   first bad bins", bad[:6].tolist(), "count", len(bad), "last", bad[-3:].tolist(), "entries hex", [hex(int(v)) for v in e[bad[:6]]], "segs hex", [hex(int(v)) for v in sg[bad[:6]]])
        nbx, nby, _ = sharding.grid(W, H)
        rows = e.reshape(nby, nbx).sum(axis=1)
        print(f"after {nwarm} unwaited frames, calibration {k}: sum {int(e.sum())} (stats entries {st['bin_entries']}), rows 0-3 {rows[:4].tolist()}, rows 60-63 {rows[60:64].tolist()}", flush=True)
