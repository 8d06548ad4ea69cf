"""Shared helpers of the parity tests: render one scene description through the oracle and through
libmtr.so and compare bit for bit."""
from __future__ import annotations

import numpy as np

from mt_renderer_amd import scene
from oracle import oracle as orc


def render_oracle(w, h, draws, clear=(1.0, 1.0, 1.0, 1.0), clear_depth=1.0, nthreads=1):
    """draws: list of dicts {md, M | (vp, model_mats, palettes, tex_override), palette}"""
    f = orc.OracleFrame(w, h, clear, clear_depth)
    cache = {}
    for d in draws:
        md = d["md"]
        om = cache.setdefault(id(md), orc.OracleModel(md))
        if "overlay" in d:
            f.draw_overlay_cubes(d["vp"], d["overlay"])
        elif "model_mats" in d:
            f.draw_instances(om, d["vp"], d["model_mats"], d.get("palettes"), d.get("tex_override"), nthreads)
        else:
            f.draw(om, d["M"], d.get("palette"), nthreads=nthreads)
    out = f.color(), f.depth(), f.stats()
    f.close()
    return out


def render_gpu(dev, w, h, draws, clear=(1.0, 1.0, 1.0, 1.0), clear_depth=1.0, shard=None, tile_mode=None):
    """tile_mode None: render the scene with BOTH tile kernels (ordered, then auto = visibility-key when every
    material is opaque), require identical pixels, return the auto result."""
    from mt_renderer_amd import api
    if tile_mode is None:
        # both tile kernels x both bin-queue builders (exact two-pass, single-pass bounded) must agree bit for bit
        dev.set_binning(False)
        a = render_gpu(dev, w, h, draws, clear, clear_depth, shard, api.TILE_ORDERED)
        dev.set_binning(True)
        a2 = render_gpu(dev, w, h, draws, clear, clear_depth, shard, api.TILE_ORDERED)
        b = render_gpu(dev, w, h, draws, clear, clear_depth, shard, api.TILE_AUTO)
        assert a[2]["tile_kernel"] == api.TILE_ORDERED and a[2]["binning"] == 2
        for other in (a2, b):
            assert (a[0] == other[0]).all() and (a[1].view(np.uint32) == other[1].view(np.uint32)).all(), "kernel variants disagree"
            assert a[2]["bin_entries"] == other[2]["bin_entries"] and a[2]["tris_setup"] == other[2]["tris_setup"]
        return b
    dev.set_tile_mode(tile_mode)
    fr = api.Frame(dev, w, h, clear, clear_depth)
    if shard:
        fr.set_shard(*shard)
    models, batches = {}, []
    try:
        for d in draws:
            md = d["md"]
            if "overlay" in d:
                fr.draw_overlay_cubes(d["vp"], d["overlay"])
                continue
            if id(md) not in models:
                # "make_model": build the api.Model some other way (e.g. from resource files) for the same md
                models[id(md)] = d["make_model"](dev) if "make_model" in d else api.Model.new(dev, md)
            m = models[id(md)]
            if "model_mats" in d:
                b = api.Batch(dev, m, d["model_mats"], d.get("palettes"), d.get("tex_override"))
                batches.append(b)
                fr.draw_batch(b, d["vp"])
            else:
                m.set_palette(d.get("palette"))
                m.render(fr, d["M"])
        fr.end()
        return fr.color(), fr.depth(), fr.stats()
    finally:
        dev.set_tile_mode(api.TILE_AUTO)
        fr.close()
        for b in batches:
            b.close()
        for m in models.values():
            m.close()


def assert_same(gpu, ref, what=""):
    gc, gd, gs = gpu
    rc, rd, rs = ref
    nd = int((gd.view(np.uint32) != rd.view(np.uint32)).sum())
    ncol = int((gc != rc).any(axis=-1).sum())
    assert gs["tris_in"] == rs["tris_in"], (what, gs, rs)
    assert gs["tris_setup"] == rs["tris_setup"], (what, gs, rs)
    assert nd == 0 and ncol == 0, f"{what}: {nd} depth pixels and {ncol} colour pixels differ"
