"""Experiment (REJECTED, needs tools/abl/r03_heaviest_bins_first_rejected.diff applied): tile kernels take the heaviest bins first (order from a census frame).
usage: python tools/probe/bin_order.py [HL|C4]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mt_renderer_amd import api, scene

lib = api.lib
lib.mtr_device_set_bin_order.restype = C.c_int32
lib.mtr_device_set_bin_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
what = sys.argv[1] if len(sys.argv) > 1 else "HL"
dev = api.Device(0)
if what == "HL":
    W, H = 1920, 1080
    m = api.Model.new(dev, scene.headline_model()); m.set_palette(scene.bone_palette()); batch = None
    M = scene.to_f32_colmajor(scene.headline_transform(W, H))
else:
    W, H = 3840, 2160
    mats, pals = scene.instance_lattice(16, 8)
    m = api.Model.new(dev, scene.mesh50k()); batch = api.Batch(dev, m, mats, pals)
    M = scene.to_f32_colmajor(scene.reference_view_proj(W, H))


def frame(wait):
    fr = api.Frame(dev, W, H)
    if batch: fr.draw_batch(batch, M)
    else: m.render(fr, M)
    fr.submit()
    if wait: fr.wait()
    return fr


def measure(tag):
    for _ in range(5): frame(True).close()
    dev.set_profiling(True)
    acc = {}
    n = 40
    for _ in range(n):
        fr = frame(True)
        for k, v in fr.timings_ms().items(): acc[k] = acc.get(k, 0.0) + v / n
        fr.close()
    dev.set_profiling(False)
    best = 1e9
    for rep in range(5):
        nf = 400
        t0 = time.perf_counter()
        frs = [frame(False) for _ in range(nf)]
        frs[-1].wait()
        best = min(best, (time.perf_counter() - t0) / nf)
        for fr in frs: fr.close()
    print(f"{tag}: pipelined {best*1e6:.1f} us/frame, stand-alone geom {acc['geom']*1e3:.1f} tile {acc['tile']*1e3:.1f} us", flush=True)


measure("default order")
fr = frame(True); e, _ = fr.bin_counts(); ref = fr.color().copy(); fr.close()
print("bins", e.size, "non-empty", int((e > 0).sum()), "max", int(e.max()), "mean of non-empty", float(e[e > 0].mean()), "p99", float(np.percentile(e[e > 0], 99)))
for name, order in (("heaviest first", np.argsort(-e.astype(np.int64), kind="stable")),
                    ("heavy (> 4 x mean) first, the rest in row order", None)):
    if order is None:
        thr = 4 * e[e > 0].mean()
        heavy = np.flatnonzero(e > thr); heavy = heavy[np.argsort(-e[heavy].astype(np.int64), kind="stable")]
        order = np.concatenate([heavy, np.flatnonzero(e <= thr)])
    order = np.ascontiguousarray(order, dtype=np.uint32)
    rc = lib.mtr_device_set_bin_order(dev._h, W, H, order.ctypes.data_as(C.c_void_p), order.size)
    assert rc == 0, rc
    fr = frame(True); assert (fr.color() == ref).all(); fr.close()
    measure(name)
lib.mtr_device_set_bin_order(dev._h, W, H, None, 0)
measure("default order again")
