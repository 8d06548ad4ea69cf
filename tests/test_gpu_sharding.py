"""-m gpu: multi-GPU v2 -- ownership maps (interleaved bins, bands of bin rows, super-tiles) and the geometry culling
of sharded frames.  One GPU stands in for every rank, one after the other.  Everything is a property of exact
equality: the pixels a rank owns are those of the unsharded frame (which the other tests pin to the oracle), the
gathered frame equals the unsharded one, and culling changes neither the pixels nor the triangles a rank sets up."""
import numpy as np
import pytest

from mt_renderer_amd import scene, sharding
from tests.helpers import render_gpu

pytestmark = pytest.mark.gpu

MAPS = [("interleaved", sharding.INTERLEAVED, 0, None), ("bands", sharding.BANDS, 0, None), ("bands-uneven", sharding.BANDS, 0, "uneven"),
        ("supertiles1", sharding.SUPERTILES, 0, None), ("supertiles4", sharding.SUPERTILES, 2, None)]


def _bands(kind, h, world):
    if kind != "uneven":
        return None
    nby = (h + 15) // 16
    cuts = sorted(int(v) for v in np.random.default_rng(world).integers(0, nby + 1, size=world - 1))
    b = [0] + cuts + [nby]
    if world >= 3:
        b[2] = b[1]  # an empty band: that rank owns no bin at all
    return b


def _scene(w, h):
    """a skinned model filling most of the frame + an instanced, skinned batch + overlay cubes"""
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=40, cols=64)
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    small = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=12, cols=20)
    mats, pals = scene.instance_lattice(6, 4)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    cubes = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5.0 + 0.5 * i - 1.0, 0.5 * (i % 3) - 0.5, -1.0) @ scene.mat_scale(0.08, 0.08, 0.08))
                      for i in range(5)])
    return [dict(md=md, M=M, palette=scene.bone_palette()),
            dict(md=small, vp=vp, model_mats=mats, palettes=pals),
            dict(md=small, vp=vp, overlay=cubes)]


@pytest.mark.parametrize("name,own_map,param,bands_kind", MAPS, ids=[m[0] for m in MAPS])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_every_ownership_map_rebuilds_the_unsharded_frame(gpu_device, name, own_map, param, bands_kind, world):
    import torch
    from mt_renderer_amd import api
    w, h = 333, 171  # 21 x 11 bins, ragged right and bottom edges
    draws = _scene(w, h)
    full = render_gpu(gpu_device, w, h, draws, tile_mode=api.TILE_AUTO)
    bands = _bands(bands_kind, h, world)
    owner = sharding.owner_map(w, h, world, own_map, param, bands)
    nbytes = api.shard_bytes_map(w, h, world, own_map, param, bands)
    assert nbytes == sharding.shard_bytes(w, h, world, own_map, param, bands)
    shards = []
    for rank in range(world):
        own = owner == rank
        setups = {}
        for cull in (True, False):
            gpu_device.set_culling(cull)
            try:
                part = render_gpu(gpu_device, w, h, draws, shard=(rank, world, own_map, param, bands))  # every kernel variant
            finally:
                gpu_device.set_culling(True)
            assert (part[0][own] == full[0][own]).all() and (part[1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), (name, world, rank, cull)
            assert part[2]["shard_bins"] * 256 >= int(own.sum())
            setups[cull] = (part[2]["tris_setup"], part[2]["bin_entries"])
            if not cull:
                assert part[2]["chunks_culled"] == 0
        assert setups[True] == setups[False], "culling must not drop a triangle that reaches one of the rank's bins"
        # pack this rank's bins (a fresh frame: render_gpu closed its own)
        fr = api.Frame(gpu_device, w, h)
        fr.set_shard(rank, world, own_map, param, bands)
        models = []
        for d in draws:
            if "overlay" in d:
                fr.draw_overlay_cubes(d["vp"], d["overlay"])
                continue
            m = api.Model.new(gpu_device, d["md"])
            models.append(m)
            if "model_mats" in d:
                fr.draw_instances(m, d["vp"], d["model_mats"], d.get("palettes"))
            else:
                m.set_palette(d.get("palette"))
                m.render(fr, d["M"])
        fr.end()
        assert fr.shard_bytes() == nbytes
        buf = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        fr.pack_color_shard(buf.data_ptr(), nbytes)
        torch.cuda.synchronize()
        mine = np.where(own[..., None], full[0], 0).astype(np.uint8)
        assert (buf.cpu().numpy().reshape(-1, 16, 16, 4) == sharding.pack_shard(mine, rank, world, own_map, param, bands)).all(), (name, world, rank)
        shards.append(buf)
        if rank == world - 1:
            gathered = torch.cat(shards)
            out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # torch's stream made them; the unpack runs on the device's own (non-blocking) stream
            fr.unpack_color_shards(gathered.data_ptr(), out.data_ptr())
            torch.cuda.synchronize()
            assert (out.cpu().numpy().reshape(h, w, 4) == full[0]).all(), (name, world)
        fr.close()
        for m in models:
            m.close()


def test_band_sharding_culls_most_of_the_geometry(gpu_device):
    """the point of v2: with bands a rank of 8 skips most chunks of a screen-filling model and most instances of a batch"""
    from mt_renderer_amd import api
    w, h = 1920, 1080
    md = scene.mesh50k()
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    draws = [dict(md=md, M=M, palette=scene.bone_palette())]
    full = render_gpu(gpu_device, w, h, draws, tile_mode=api.TILE_AUTO)
    owner = sharding.owner_map(w, h, 8, sharding.BANDS)
    kept = []
    for rank in range(8):
        part = render_gpu(gpu_device, w, h, draws, shard=(rank, 8, sharding.BANDS), tile_mode=api.TILE_AUTO)
        own = owner == rank
        assert (part[0][own] == full[0][own]).all() and (part[1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), rank
        kept.append(1.0 - part[2]["chunks_culled"] / part[2]["chunks"])
    assert max(kept) < 0.45 and sum(kept) < 2.2, kept  # 8 ranks together touch each chunk about 1.3 times
    # a batch: 8 x 8 instances at 1080p
    mats, pals = scene.instance_lattice(8, 8)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    small = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=40)
    draws = [dict(md=small, vp=vp, model_mats=mats, palettes=pals)]
    full = render_gpu(gpu_device, w, h, draws, tile_mode=api.TILE_AUTO)
    for rank in (0, 3, 7):
        part = render_gpu(gpu_device, w, h, draws, shard=(rank, 8, sharding.BANDS), tile_mode=api.TILE_AUTO)
        own = owner == rank
        assert (part[0][own] == full[0][own]).all() and (part[1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), rank


def _random_model(rng, normalised, many_joints):
    """strips of random quads with random joints / weights: nothing like the tidy capsule"""
    nverts, nquads = 600, 260
    pos = rng.uniform(-1, 1, size=(nverts, 3))
    pos[:, 2] *= 0.2
    stride = 24
    vb = np.zeros((nverts, stride), dtype=np.uint8)
    p16 = np.clip(np.round(pos * 32767), -32767, 32767).astype(np.int16)
    vb[:, 0:6] = p16.view(np.uint8).reshape(nverts, 6)
    vb[:, 12:16] = np.float16(rng.uniform(0, 1, size=(nverts, 2))).view(np.uint8).reshape(nverts, 4)
    joints = rng.integers(0, 40 if many_joints else 6, size=(nverts, 4)).astype(np.uint8)
    wts = rng.integers(0, 256, size=(nverts, 4)).astype(np.int64)
    wts[rng.uniform(size=nverts) < 0.3, 2:] = 0  # some vertices use two joints only
    if normalised:
        wts = np.maximum(wts, 0)
        tot = wts.sum(axis=1, keepdims=True)
        tot[tot == 0] = 1
        wts = wts * 255 // tot
        wts[:, 0] += 255 - wts.sum(axis=1)
    vb[:, 16:20] = joints
    vb[:, 20:24] = np.clip(wts, 0, 255).astype(np.uint8)
    idx = []
    for q in range(nquads):  # short strips of nearby vertices so triangles stay small
        a = int(rng.integers(0, nverts - 8))
        idx += [a, a + 1, a + 2, a + 3, a + 4, 0xFFFF]
    # make triangles local: sort vertices along x so neighbours in index are neighbours in space
    order = np.argsort(pos[:, 0] + 0.05 * pos[:, 1])
    vb = vb[order]
    index_buf = np.array(idx, dtype=np.uint16)
    prim = scene.pack_primitive(vertex_num=nverts, vertex_stride=stride, topology=scene.TOPO_STRIP, vertex_base=0, index_ofs=0,
                                index_num=len(idx), index_base=0)
    lay = [(scene.SEM_POSITION, scene.IEF_S16N, 3, 0), (scene.SEM_TEXCOORD, scene.IEF_F16, 2, 12), (scene.SEM_JOINT, scene.IEF_U8, 4, 16),
           (scene.SEM_WEIGHT, scene.IEF_U8N, 4, 20)]
    return scene.ModelData(vertex_buf=vb.reshape(-1), index_buf=index_buf, prims=np.stack([prim]), layouts=[lay],
                           prim_to_texture=np.array([-1], dtype=np.int32), prim_debug_id=np.array([5], dtype=np.uint32),
                           parts_disp=np.ones(1, dtype=np.uint8))


@pytest.mark.parametrize("seed", range(6))
def test_culling_is_conservative_on_hostile_inputs(gpu_device, seed):
    """random joints / weights (normalised or not, few or many joints per chunk), random palettes with rotation, shear,
    scale and translation, cameras that put the near plane through the model: for every rank of every map the owned
    pixels and the set-up counts are identical with and without culling."""
    from mt_renderer_amd import api
    rng = np.random.default_rng(100 + seed)
    md = _random_model(rng, normalised=seed % 3 != 2, many_joints=seed % 2 == 1)
    npal = 40
    pal = np.zeros((npal, 16), dtype=np.float32)
    for j in range(npal):
        A = np.eye(4)
        A[:3, :3] = np.linalg.qr(rng.normal(size=(3, 3)))[0] @ np.diag(rng.uniform(0.6, 1.5, size=3))
        A[:3, 3] = rng.uniform(-0.4, 0.4, size=3)
        pal[j] = scene.to_f32_colmajor(A)
    w, h = 320, 200
    dist = [2.3, 0.9, 0.3][seed % 3]  # the last ones push geometry through the near plane / behind the camera
    M = scene.to_f32_colmajor(scene.reference_view_proj(w, h) @ scene.mat_translate(-5.0, 0.0, 1.0 - dist))
    mats = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5.0 + dx, dy, 1.0 - dist - 0.5) @ scene.mat_scale(0.3, 0.3, 0.3))
                     for dx, dy in [(-0.8, -0.4), (0.0, 0.3), (0.9, -0.1), (4.0, 0.0), (0.2, 3.0)]])
    pals = np.stack([np.roll(pal, k, axis=0) for k in range(len(mats))])
    draws = [dict(md=md, M=M, palette=pal),
             dict(md=md, vp=scene.to_f32_colmajor(scene.reference_view_proj(w, h)), model_mats=mats, palettes=pals),
             dict(md=md, M=M, palette=None)]  # unskinned draw of the same model
    full = render_gpu(gpu_device, w, h, draws, tile_mode=api.TILE_AUTO)
    for own_map, param in ((sharding.BANDS, 0), (sharding.SUPERTILES, 1), (sharding.INTERLEAVED, 0)):
        world = 5
        owner = sharding.owner_map(w, h, world, own_map, param)
        for rank in range(world):
            res = {}
            for cull in (True, False):
                gpu_device.set_culling(cull)
                try:
                    res[cull] = render_gpu(gpu_device, w, h, draws, shard=(rank, world, own_map, param), tile_mode=api.TILE_AUTO)
                finally:
                    gpu_device.set_culling(True)
            own = owner == rank
            for cull in (True, False):
                assert (res[cull][0][own] == full[0][own]).all() and (res[cull][1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), (seed, own_map, rank, cull)
            assert res[True][2]["tris_setup"] == res[False][2]["tris_setup"] and res[True][2]["bin_entries"] == res[False][2]["bin_entries"]


def test_bin_counts_of_empty_bins_are_zero_whatever_the_buffer_held(gpu_device):
    """mtr_frame_read_bin_counts is what bench.py balances bands by.  A frame after a larger model was freed and the target
    grew gets its per-bin words from memory that held other data: every bin, empty ones included, must report its own
    count (the empty-bin fast path of the visibility kernel used to leave the word alone)."""
    from mt_renderer_amd import api
    big = api.Model.new(gpu_device, scene.headline_model())
    big.set_palette(scene.bone_palette())
    M = scene.to_f32_colmajor(scene.headline_transform(640, 360))
    for _ in range(6):
        fr = api.Frame(gpu_device, 640, 360); big.render(fr, M); fr.end(); fr.close()
    big.close()  # 14 MB of vertex / index data back to the allocator
    w, h = 2560, 1440
    small = api.Model.new(gpu_device, scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=12, cols=20))
    mats, pals = scene.instance_lattice(4, 2)
    batch = api.Batch(gpu_device, small, mats, pals, None)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    try:
        for k in range(4):  # every frame slot once
            fr = api.Frame(gpu_device, w, h); fr.draw_batch(batch, vp); fr.end()
            e, s = fr.bin_counts(); st = fr.stats(); fr.close()
            assert int(e.astype(np.uint64).sum()) == st["bin_entries"], k
            assert (e == 0).sum() > len(e) // 2  # most of the frame is empty
    finally:
        batch.close(); small.close()


@pytest.mark.parametrize("slots", ["1", "5"])
def test_instances_beyond_the_full_rate_launch_take_the_slow_exact_path(gpu_device, slots, monkeypatch):
    """the geometry launch of a sharded batch covers twice the rank's fair share of instance slots at full rate and the
    rest through k_geom_rest (one workgroup per slot); MTR_GEOM_SLOTS (read once, at device creation) shrinks the first
    part so that most of the kept instances take the second: same pixels, same counts, every queue builder and tile kernel"""
    from mt_renderer_amd import api
    w, h = 640, 360
    small = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=40)
    mats, pals = scene.instance_lattice(6, 4)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    draws = [dict(md=small, vp=vp, model_mats=mats, palettes=pals)]
    full = render_gpu(gpu_device, w, h, draws, tile_mode=api.TILE_AUTO)
    owner = sharding.owner_map(w, h, 3, sharding.BANDS)
    ref = {r: render_gpu(gpu_device, w, h, draws, shard=(r, 3, sharding.BANDS)) for r in range(3)}
    monkeypatch.setenv("MTR_GEOM_SLOTS", slots)
    dev = api.Device(0)  # a device of its own: the hook is not read in the submit path
    monkeypatch.delenv("MTR_GEOM_SLOTS")
    try:
        for rank in range(3):
            part = render_gpu(dev, w, h, draws, shard=(rank, 3, sharding.BANDS))  # all builders x both tile kernels inside
            own = owner == rank
            assert (part[0][own] == full[0][own]).all() and (part[1].view(np.uint32)[own] == full[1].view(np.uint32)[own]).all(), rank
            for key in ("tris_setup", "bin_entries", "chunks_culled"):
                assert part[2][key] == ref[rank][2][key], (rank, key)
    finally:
        dev.close()


def test_unsharded_frames_can_cull_what_is_off_the_target(gpu_device):
    """MTR_GEOM_CULL_ALL_FRAMES: the culling of sharded frames applied to an unsharded one = frustum culling.  A lattice of
    instances seen by a camera that has most of it out of view: same pixels and counts as without, most chunks skipped;
    a single model hanging over the edge of the target too."""
    from mt_renderer_amd import api
    w, h = 640, 360
    small = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=40)
    mats, pals = scene.instance_lattice(8, 6)
    base = scene.reference_view_proj(w, h)
    views = [scene.to_f32_colmajor(base @ scene.mat_translate(dx, dy, dz)) for dx, dy, dz in ((0.0, 0.0, 0.0), (2.2, 0.9, 0.0), (-1.5, 0.0, 1.2), (0.0, 0.0, 2.1))]
    big = scene.mesh50k()
    Mbig = scene.to_f32_colmajor(base @ scene.mat_translate(-5.0 + 1.4, 0.6, 1.0 - 2.0))  # half of it past the right / top edge
    try:
        for vp in views:
            draws = [dict(md=small, vp=vp, model_mats=mats, palettes=pals), dict(md=big, M=Mbig, palette=scene.bone_palette())]
            gpu_device.set_culling(api.GEOM_CULL_SHARDED)
            ref = render_gpu(gpu_device, w, h, draws)
            gpu_device.set_culling(api.GEOM_CULL_ALL_FRAMES)
            got = render_gpu(gpu_device, w, h, draws)  # every queue builder x both tile kernels inside
            assert (got[0] == ref[0]).all() and (got[1].view(np.uint32) == ref[1].view(np.uint32)).all()
            for key in ("tris_setup", "bin_entries", "chunks"):
                assert got[2][key] == ref[2][key], key
            assert ref[2]["chunks_culled"] == 0 and got[2]["chunks_culled"] > 0
        assert got[2]["chunks_culled"] > got[2]["chunks"] // 2  # the last view: behind the camera / out of sight, mostly
    finally:
        gpu_device.set_culling(api.GEOM_CULL_SHARDED)
    with pytest.raises(api.MtrError):
        gpu_device.set_culling(7)


def test_launch_sizes_from_reported_counts_survive_a_moving_camera_and_a_recycled_slot():
    """A sharded batch draw sizes its launches by the instance counts its last frames reported (pinned host words written by
    the tile kernel; mtr_api.cpp: hint_host).  The count is a hint: frames whose rank keeps MANY more instances than the hint
    says -- the camera jumps from a view where a few instances reach the band to one where most of them do -- must come out
    exactly as without it (the remainder launch takes what the hint misses), and a hint slot that changes hands when a batch
    is destroyed must not leak a frame's worth of wrong geometry into the next batch."""
    from mt_renderer_amd import api
    w, h, world, rank = 640, 360, 4, 1
    md = scene.mesh50k()  # 813 chunks: 51 mask groups per instance, so the slots a low hint leaves out are worth a second launch
    mats, pals = scene.instance_lattice(16, 12)
    bands = [0, 6, 12, 18, (h + 15) // 16]
    owner = sharding.owner_map(w, h, world, sharding.BANDS, 0, bands)
    own = owner == rank
    cams = [scene.to_f32_colmajor(scene.reference_view_proj(w, h)),                                   # the lattice fills the frame
            scene.to_f32_colmajor(scene.reference_view_proj(w, h, position=(-5.0, 3.5, 1.0))),      # ... slides down: few instances in band 1
            scene.to_f32_colmajor(scene.reference_view_proj(w, h, position=(-5.0, 0.0, 6.0)))]      # ... far away: every instance near the middle rows
    with api.Device(0) as dev:
        model = api.Model.new(dev, md)

        def frame(batch, vp, shard):
            fr = api.Frame(dev, w, h)
            if shard:
                fr.set_shard(rank, world, sharding.BANDS, 0, bands)
            fr.draw_batch(batch, vp)
            fr.end()
            out = fr.color(), fr.depth().view(np.uint32), fr.stats()
            fr.close()
            return out

        batch = api.Batch(dev, model, mats, pals)
        refs = [frame(batch, vp, False) for vp in cams]
        kept = []
        for k in (1, 1, 1, 1, 2, 2, 0, 0, 1, 2, 0, 1):  # four frames settle the hint low, then the jumps
            c, d, st = frame(batch, cams[k], True)
            assert (c[own] == refs[k][0][own]).all() and (d[own] == refs[k][1][own]).all(), k
            kept.append(st["chunks"] - st["chunks_culled"])
        assert max(kept) > 3 * min(kept), kept  # the views really differ in what the rank keeps
        # a second batch takes over the first one's hint slot: its first sharded frames run with whatever the slot holds
        batch.close()
        mats2, pals2 = scene.instance_lattice(5, 40)
        batch2 = api.Batch(dev, model, mats2, pals2)
        ref2 = frame(batch2, cams[0], False)
        for _ in range(4):
            c, d, _ = frame(batch2, cams[0], True)
            assert (c[own] == ref2[0][own]).all() and (d[own] == ref2[1][own]).all()
        batch2.close()
        model.close()
