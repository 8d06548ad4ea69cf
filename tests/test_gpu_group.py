"""-m gpu: mtr_group_* (include/mtr.h, "one host thread, N devices"; SURVEY 8b's mtr_group_create) -- one host thread drives
every rank, a group frame is one sharded frame per rank, mtr_group_frame_end gathers the colour on rank 0's device.  A GPU
box has one card, so every rank of these groups is a device on card 0 (the header allows an index to repeat): the ranks
are separate mtr_devices with their own streams, models, textures and queues, and the gather is a device-to-device copy
where two cards would use a peer copy.  The gathered image must be the ORACLE's frame, bit for bit."""
import numpy as np
import pytest

from mt_renderer_amd import scene, sharding
from tests.helpers import render_oracle

pytestmark = pytest.mark.gpu


def _scene(w, h):
    """a skinned mesh drawn on its own, an instanced batch of it with per-instance palettes, overlay cubes on top"""
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=40)
    mats, pals = scene.instance_lattice(4, 3)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    cubes = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5.0 + 0.1 * i, 0.05 * i, -1.3) @ scene.mat_scale(0.05, 0.05, 0.05)) for i in range(4)])
    return [dict(md=md, M=M, palette=scene.bone_palette()), dict(md=md, vp=vp, model_mats=mats, palettes=pals),
            dict(md=md, vp=vp, overlay=cubes)]


def _render_group(group, w, h, draws, own_map=0, param=0, bands=None):
    from mt_renderer_amd import api
    gf = api.GroupFrame(group, w, h, own_map=own_map, param=param, band_rows=bands)
    keep = []
    try:
        for r in range(len(group)):
            dev, part, models = group.device(r), gf.part(r), {}
            for d in draws:
                if "overlay" in d:
                    part.draw_overlay_cubes(d["vp"], d["overlay"])
                    continue
                if id(d["md"]) not in models:  # a rank's resources live on ITS device
                    models[id(d["md"])] = api.Model.new(dev, d["md"])
                    keep.append(models[id(d["md"])])
                m = models[id(d["md"])]
                if "model_mats" in d:
                    b = api.Batch(dev, m, d["model_mats"], d.get("palettes"), d.get("tex_override"))
                    keep.append(b)
                    part.draw_batch(b, d["vp"])
                else:
                    m.set_palette(d.get("palette"))
                    m.render(part, d["M"])
        gf.end()
        stats = [gf.part(r).stats() for r in range(len(group))]
        return gf.color(), stats
    finally:
        gf.close()
        for o in reversed(keep):
            o.close()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_group_frame_is_the_oracle_frame_under_every_ownership_map(world):
    from mt_renderer_amd import api
    w, h = 640, 360
    draws = _scene(w, h)
    oc, _, ost = render_oracle(w, h, draws, nthreads=8)
    with api.Group([0] * world) as g:
        assert len(g) == world
        maps = [(sharding.INTERLEAVED, 0, None), (sharding.SUPERTILES, 1, None), (sharding.BANDS, 0, None)]
        if world == 3:
            maps.append((sharding.BANDS, 0, [0, 5, 9, (h + 15) // 16]))  # uneven bands
        for own_map, param, bands in maps:
            color, stats = _render_group(g, w, h, draws, own_map, param, bands)
            assert (color == oc).all(), (world, own_map, int((color != oc).any(axis=-1).sum()))
            assert all(st["tris_in"] == ost["tris_in"] and st["shard_map"] == own_map for st in stats)
            if world > 1:
                assert sum(st["tris_setup"] for st in stats) >= ost["tris_setup"]


def test_group_of_two_renders_the_headline_scene():
    """1 M triangles, 1920x1080, two ranks with bands balanced by the unsharded frame's queue lengths: the gathered image
    equals the unsharded frame of a plain device (which tests/test_gpu_parity.py pins to the oracle)."""
    from mt_renderer_amd import api
    w, h = 1920, 1080
    md = scene.headline_model()
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    with api.Device(0) as dev:
        model = api.Model.new(dev, md)
        model.set_palette(scene.bone_palette())
        fr = api.Frame(dev, w, h)
        model.render(fr, M)
        fr.end()
        full = fr.color()
        entries, _ = fr.bin_counts()
        fr.close()
        model.close()
    nbx, nby, _ = sharding.grid(w, h)
    bands = sharding.balanced_bands(entries.reshape(nby, nbx).sum(axis=1).astype(np.float64) + 8.0 * nbx, 2)
    with api.Group([0, 0]) as g:
        for _ in range(2):  # the second frame reuses the group's send / gathered / image buffers
            color, stats = _render_group(g, w, h, [dict(md=md, M=M, palette=scene.bone_palette())], sharding.BANDS, 0, bands)
            assert (color == full).all()
            assert stats[0]["chunks_culled"] > 0 and stats[1]["chunks_culled"] > 0  # each rank skipped the other's band


def test_a_part_that_overflows_its_bin_queues_is_rerun_before_the_gather():
    from mt_renderer_amd import api
    w, h = 640, 360
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=100, cols=160)  # 32 000 small triangles
    draws = [dict(md=md, M=scene.to_f32_colmajor(scene.headline_transform(w, h)), palette=scene.bone_palette())]
    oc, _, _ = render_oracle(w, h, draws, nthreads=8)
    with api.Group([0, 0]) as g:
        g.device(1).set_binning(True, 64)  # rank 1: bounded queues of 64 entries per bin overflow on this scene
        color, stats = _render_group(g, w, h, draws, sharding.BANDS)
        assert stats[1]["binning"] == 2 and stats[0]["binning"] == 1  # rank 1's part went through the exact two-pass queues
        assert (color == oc).all()


def test_group_errors_are_reported_not_swallowed():
    from mt_renderer_amd import api
    for bad in ([], [0] * 65, [99], [-1]):
        with pytest.raises(api.MtrError) as e:
            api.Group(bad)
        assert e.value.code == api.MTR_E_INVALID
    with api.Group([0, 0]) as g:
        with pytest.raises(api.MtrError):  # bands that do not cover the frame
            api.GroupFrame(g, 64, 64, own_map=sharding.BANDS, band_rows=[0, 1, 2])
        with pytest.raises(api.MtrError):
            api.GroupFrame(g, 0, 64)
        a = api.GroupFrame(g, 64, 48, clear_rgba=(0.0, 1.0, 0.0, 1.0))
        with pytest.raises(api.MtrError):
            a.color()  # not ended
        assert a.color_devptr() == 0
        a.end()
        assert (a.color() == np.array([0, 255, 0, 255], dtype=np.uint8)).all()  # nothing drawn: the clear colour, every bin
        with pytest.raises(api.MtrError):
            a.end()  # ended twice
        b = api.GroupFrame(g, 64, 48, clear_rgba=(1.0, 0.0, 0.0, 1.0))
        b.end()
        with pytest.raises(api.MtrError) as e:
            a.color()  # the group's image is b's now
        assert "later group frame" in str(e.value)
        assert a.color_devptr() == 0 and b.color_devptr() != 0
        assert (b.color() == np.array([255, 0, 0, 255], dtype=np.uint8)).all()
        a.close()
        b.close()
