"""-m gpu: edge cases of the draw path, HIP (through the C ABI) vs the CPU oracle, bit for bit:
fill rule, strips / restart, depth ties, clipping, big triangles (64-bit edge path), every vertex
format of the reference's table, textures (RGBA8 / BC1 / BC7, linear + nearest, blending), instancing,
the debug overlay, many-segment bins, sharded bins and error behaviour."""
import numpy as np
import pytest

from mt_renderer_amd import scene, sharding
from oracle import oracle as orc
from tests.helpers import assert_same, render_gpu, render_oracle
from tests.pixel_scenes import pixel_model, pixel_to_ndc_matrix

pytestmark = pytest.mark.gpu


def _both(dev, w, h, draws, **kw):
    g = render_gpu(dev, w, h, draws, **kw)
    assert_same(g, render_oracle(w, h, draws, **kw), "case")
    return g


def _px(dev, prims, w=64, h=64, textures=None, parts_disp=None, **kw):
    md = pixel_model(prims, textures, parts_disp)
    return _both(dev, w, h, [dict(md=md, M=pixel_to_ndc_matrix(w, h))], **kw)


def _quad(x0, y0, x1, y1, z, u0=0.0, v0=0.0, u1=1.0, v1=1.0, tex=0, did=0):
    v = [(x0, y0, z, u0, v0), (x0, y1, z, u0, v1), (x1, y1, z, u1, v1), (x1, y0, z, u1, v0)]
    return dict(verts=v, indices=[0, 1, 2, 0, 2, 3], texture=tex, debug_id=did)


def test_empty_frame_is_clear(gpu_device):
    from mt_renderer_amd import api
    fr = api.Frame(gpu_device, 70, 33, (0.25, 0.5, 0.75, 1.0), 0.5)
    fr.end()
    c, d = fr.color(), fr.depth()
    fr.close()
    assert (c == np.array([64, 128, 191, 255], dtype=np.uint8)).all() and (d == np.float32(0.5)).all()


def test_fill_rule_cases(gpu_device):
    _px(gpu_device, [dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2])])
    v = [(2.5, 2.5, .5), (2.5, 10.5, .5), (10.5, 10.5, .5), (10.5, 2.5, .5)]
    g = _px(gpu_device, [dict(verts=v, indices=[0, 1, 2, 0, 2, 3])])
    assert (g[1] < 1).sum() == 64
    # a fan around an off-grid centre with vertices on pixel centres and pixel corners
    c = (17.3, 19.8, .4)
    ring = [(30.5, 20.5, .4), (25, 31, .4), (12.5, 30.5, .4), (5, 17, .4), (11.5, 6.5, .4), (24, 5, .4)]
    idx = []
    for k in range(6):
        idx += [0, 1 + (k + 1) % 6, 1 + k]
    _px(gpu_device, [dict(verts=[c] + ring, indices=idx)])


def test_strips_restart_parity_and_index_base(gpu_device):
    v = [(0, 0, .5), (0, 8, .5), (8, 0, .5), (8, 8, .5), (16, 0, .5), (16, 8, .5), (24, 0, .3), (24, 8, .3)]
    S = scene.TOPO_STRIP
    _px(gpu_device, [dict(verts=v, indices=[0, 1, 2, 3, 4, 5, 6, 7], topology=S)])
    _px(gpu_device, [dict(verts=v, indices=[0, 1, 2, 0xFFFF, 2, 3, 4, 5, 0xFFFF, 0xFFFF, 4, 5, 6, 7, 0xFFFF], topology=S)])
    _px(gpu_device, [dict(verts=v, indices=[0, 1, 0xFFFF, 2, 3, 0xFFFF, 4], topology=S)])
    _px(gpu_device, [dict(verts=v, indices=[0, 1, 2, 3, 4, 5], topology=S, index_base=2)])
    # out-of-range vertices drop their triangles only
    _px(gpu_device, [dict(verts=v[:4], indices=[0, 1, 2, 0, 1, 9, 1, 3, 2])])
    # a long strip with restarts at awkward places relative to the 62-position geometry chunks
    rng = np.random.default_rng(5)
    n = 400
    vv = [(float(4 + (k // 2) * 0.3), float(8 + 20 * (k & 1)), 0.5) for k in range(n)]
    idx = list(range(n))
    for pos in sorted(rng.choice(np.arange(3, n - 3), size=25, replace=False), reverse=True):
        idx.insert(int(pos), 0xFFFF)
    _px(gpu_device, [dict(verts=vv, indices=idx, topology=S)], w=128, h=64)


def test_depth_ties_order_and_clip(gpu_device):
    a = dict(verts=[(0, 0, .5), (0, 16, .5), (16, 0, .5)], indices=[0, 1, 2], debug_id=1)
    b = dict(verts=[(0, 0, .5), (0, 16, .5), (16, 0, .5)], indices=[0, 1, 2], debug_id=2)
    c = dict(verts=[(0, 0, .75), (0, 16, .75), (16, 0, .75)], indices=[0, 1, 2], debug_id=3)
    _px(gpu_device, [a, b, c])
    _px(gpu_device, [c, b, a])
    _px(gpu_device, [dict(verts=[(0, 0, -0.5), (0, 32, -0.5), (64, 0, 1.5)], indices=[0, 1, 2])])
    _px(gpu_device, [dict(verts=[(0, 0, 0.0), (0, 32, 1.0), (64, 0, 1.0)], indices=[0, 1, 2])])
    _px(gpu_device, [a], clear=(0.1, 0.2, 0.3, 0.4), clear_depth=0.5)  # depth cleared to exactly the triangle's z


def test_parts_disp_and_errors(gpu_device):
    from mt_renderer_amd import api
    a = dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2], parts_no=1)
    b = dict(verts=[(20, 0, .5), (20, 8, .5), (28, 0, .5)], indices=[0, 1, 2], parts_no=0)
    _px(gpu_device, [a, b], parts_disp=[1, 0])
    _px(gpu_device, [a, b], parts_disp=[0, 1])
    md = pixel_model([dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2], parts_no=5)])
    with pytest.raises(api.MtrError) as e:
        render_gpu(gpu_device, 64, 64, [dict(md=md, M=pixel_to_ndc_matrix(64, 64))])
    assert e.value.code == api.MTR_E_INVALID
    bad = scene.cube_model()
    bad.layouts = [[(scene.SEM_POSITION, scene.IEF_U16, 2, 0)]]
    with pytest.raises(api.MtrError) as e:
        api.Model.new(gpu_device, bad)
    assert e.value.code == api.MTR_E_UNSUPPORTED
    with pytest.raises(api.MtrError) as e:
        api.Texture.new(gpu_device, scene.TextureData(4, 4, 99, bytes(64)))
    assert e.value.code == api.MTR_E_UNSUPPORTED
    with pytest.raises(api.MtrError) as e:
        api.Texture.new(gpu_device, scene.TextureData(8, 8, scene.TEX_BC7, bytes(16)))
    assert e.value.code == api.MTR_E_INVALID


def test_big_triangles_use_the_64bit_edge_path(gpu_device):
    tris = [dict(verts=[(-3000.25, -2000.5, .2), (-1500, 9000.75, .9), (7000.5, -3000, .6)], indices=[0, 1, 2], debug_id=4),
            dict(verts=[(-100000, 10, .5), (50, 90000, .5), (120000, -60000, .45)], indices=[0, 1, 2], debug_id=5),
            dict(verts=[(-900000.0, -900000.0, .7), (-900000.0, 900000.0, .7), (900000.0, 0.0, .3)], indices=[0, 1, 2], debug_id=6),
            dict(verts=[(10.5, 10.5, .1), (10.5, 100.25, .1), (100.75, 55.5, .1)], indices=[0, 1, 2], debug_id=7)]
    _px(gpu_device, tris, w=256, h=128)
    # beyond the guard band (2^20 px): dropped by both
    _px(gpu_device, [dict(verts=[(-2e6, 0, .5), (0, 3e6, .5), (2e6, 0, .5)], indices=[0, 1, 2])], w=256, h=128)


def test_near_plane_clip(gpu_device):
    w, h = 320, 200
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=36)
    vp = scene.reference_view_proj(w, h)
    for dz in (0.05, 0.2, 0.34, 0.5):  # the capsule straddles the near plane (z_eye = -0.01) / encloses the camera
        M = scene.to_f32_colmajor(vp @ scene.mat_translate(-5.0, 0.05, 1.0 - dz) @ scene.mat_rot_x(0.7))
        g = _both(gpu_device, w, h, [dict(md=md, M=M, palette=scene.bone_palette())])
        assert g[2]["tris_setup"] > 0


VERTEX_FORMATS = [
    # (position element, texcoord element, stride): every float-class arm of src/rshader2.rs:516-564
    ((scene.SEM_POSITION, scene.IEF_F32, 3, 0), (scene.SEM_TEXCOORD, scene.IEF_F16, 2, 12), 16),
    ((scene.SEM_POSITION, scene.IEF_S16N, 3, 0), (scene.SEM_TEXCOORD, scene.IEF_U8N, 4, 8), 12),
    ((scene.SEM_POSITION, scene.IEF_S16N, 1, 0), (scene.SEM_TEXCOORD, scene.IEF_S8N, 1, 4), 6),
    ((scene.SEM_POSITION, scene.IEF_S8N, 3, 1), (scene.SEM_TEXCOORD, scene.IEF_U8N, 1, 5), 7),      # unaligned
    ((scene.SEM_POSITION, scene.IEF_S8N, 4, 0), (scene.SEM_TEXCOORD, scene.IEF_U8NL, 3, 4), 8),
    ((scene.SEM_POSITION, scene.IEF_F16, 2, 2), (scene.SEM_TEXCOORD, scene.IEF_S16N, 1, 6), 10),    # 2-byte aligned
    ((scene.SEM_POSITION, scene.IEF_U8N, 4, 3), (scene.SEM_TEXCOORD, scene.IEF_F32, 3, 7), 19),     # unaligned F32
    ((scene.SEM_POSITION, scene.IEF_F32, 3, 0), (scene.SEM_TEXCOORD, scene.IEF_S16N, 3, 12), 20),
]


@pytest.mark.parametrize("fmt", VERTEX_FORMATS, ids=lambda f: f"pos{f[0][1]}x{f[0][2]}_uv{f[1][1]}x{f[1][2]}_s{f[2]}")
def test_vertex_formats(gpu_device, fmt):
    pos_el, uv_el, stride = fmt
    rng = np.random.default_rng(stride)
    nv = 300
    vb = rng.integers(0, 256, size=(nv, stride), dtype=np.uint8)
    if pos_el[1] == scene.IEF_F32:
        vb[:, 0:12] = rng.uniform(-0.9, 0.9, size=(nv, 3)).astype(np.float32).view(np.uint8).reshape(nv, 12)
    if pos_el[1] == scene.IEF_F16:
        vb[:, 2:6] = rng.uniform(-0.9, 0.9, size=(nv, 2)).astype("<f2").view(np.uint8).reshape(nv, 4)
    if uv_el[1] == scene.IEF_F32:
        vb[:, 7:19] = rng.uniform(0, 1, size=(nv, 3)).astype(np.float32).view(np.uint8).reshape(nv, 12)
    if uv_el[1] == scene.IEF_F16:
        vb[:, 12:16] = rng.uniform(0, 1, size=(nv, 2)).astype("<f2").view(np.uint8).reshape(nv, 4)
    idx = rng.integers(0, nv, size=900).astype(np.uint16)
    tex = scene.checker_rgba8_texture(32, 32, cell=4, alpha=(255, 120))
    base = 3 if stride in (7, 19) else 0  # unaligned vertex_base as well
    md = scene.ModelData(
        vertex_buf=np.concatenate([np.zeros(base, np.uint8), vb.reshape(-1)]), index_buf=idx,
        prims=scene.pack_primitive(vertex_num=nv, vertex_stride=stride, topology=scene.TOPO_LIST, index_num=len(idx),
                                   vertex_base=base)[None, :],
        layouts=[[pos_el, uv_el, (77, scene.IEF_F32, 3, 0), (scene.SEM_POSITION, scene.IEF_SCMP3N, 1, 0)][:3]],
        prim_to_texture=np.array([0], np.int32), prim_debug_id=np.array([3], np.uint32), parts_disp=np.ones(1, np.uint8),
        textures=[tex])
    w, h = 160, 96
    M = scene.to_f32_colmajor(scene.mat_translate(0.05, -0.03, 0.5) @ scene.mat_scale(0.9, 0.9, 0.4))
    _both(gpu_device, w, h, [dict(md=md, M=M)])
    # and the vertex stage alone, bit for bit
    from mt_renderer_amd import api
    m = api.Model.new(gpu_device, md)
    try:
        gc, gu = m.vertex_stage(0, M)
    finally:
        m.close()
    oc, ou = orc.OracleModel(md).vertex_stage(0, M)
    assert (gc.view(np.uint32) == oc.view(np.uint32)).all() and (gu.view(np.uint32) == ou.view(np.uint32)).all()


def test_vertex_stage_skinned_bitwise(gpu_device):
    from mt_renderer_amd import api
    md = scene.skinned_capsule_model([((0.1, -0.2, 0.05), 0.3, 1.2)], rows=40, cols=60)
    M = scene.to_f32_colmajor(scene.headline_transform(1920, 1080))
    pal = scene.bone_palette(64, t=1.7)
    m = api.Model.new(gpu_device, md)
    try:
        for p in (None, pal, pal[:5]):  # no palette / full / short palette (joint index clamped)
            m.set_palette(p)
            gc, gu = m.vertex_stage(0, M)
            oc, ou = orc.OracleModel(md).vertex_stage(0, M, p)
            assert (gc.view(np.uint32) == oc.view(np.uint32)).all() and (gu.view(np.uint32) == ou.view(np.uint32)).all()
    finally:
        m.close()


@pytest.mark.parametrize("kind", ["rgba8", "bc1", "bc7"])
def test_texture_decode_parity(gpu_device, kind):
    from mt_renderer_amd import api
    for (w, h) in [(64, 64), (52, 36), (7, 5), (4, 4)]:
        t = {"rgba8": scene.checker_rgba8_texture(w, h, 3, (255, 40)), "bc1": scene.random_bc1_texture(w, h, 9),
             "bc7": scene.random_bc7_texture(w, h, 11)}[kind]
        tex = api.Texture.new(gpu_device, t)
        try:
            got = tex.read_rgba8()
        finally:
            tex.close()
        assert (got == orc.decode_texture(t.fmt, w, h, t.data)).all(), (kind, w, h)


def test_textured_sampling_and_blending(gpu_device):
    texs = [scene.checker_rgba8_texture(64, 64, cell=4), scene.random_bc7_texture(32, 32, 3),
            scene.random_bc1_texture(16, 16, 4), scene.checker_rgba8_texture(4, 4, cell=1, alpha=(90, 200))]
    prims = [_quad(0, 0, 8, 8, .5, tex=0),                       # minified -> nearest
             _quad(10, 0, 74, 64, .5, tex=3),                    # magnified -> linear, translucent
             _quad(0, 10, 40, 50, .6, tex=1), _quad(20, 30, 60, 60, .4, tex=1),   # BC7 with alpha, overlapping
             _quad(30, 5, 62, 37, .45, tex=2, u0=-0.5, v0=-0.25, u1=1.5, v1=1.25),  # BC1, uv outside [0,1] -> clamp
             _quad(5, 40, 30, 63, .3, tex=-1, did=9)]             # untextured among textured
    _px(gpu_device, prims, w=80, h=64, textures=texs)
    # rho right at the mag/min switch: one texel per pixel, and slightly more / less
    for scale in (1.0, 1.0000001, 0.9999999, 1.5, 0.75):
        _px(gpu_device, [_quad(0, 0, 32, 32, .5, u1=scale, v1=scale, tex=0)], w=32, h=32,
            textures=[scene.checker_rgba8_texture(32, 32, cell=1, alpha=(255, 77))])


def test_textured_perspective_skinned(gpu_device):
    w, h = 640, 360
    tex = scene.random_bc7_texture(128, 128, seed=21)
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=30, cols=48, textured=True, textures=[tex])
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    _both(gpu_device, w, h, [dict(md=md, M=M, palette=scene.bone_palette())])
    # close-up: magnified, perspective-correct texcoords, near-plane clipping of textured triangles
    vp = scene.reference_view_proj(w, h)
    M2 = scene.to_f32_colmajor(vp @ scene.mat_translate(-5.0, 0.0, 1.0 - 0.3) @ scene.mat_rot_x(0.5))
    _both(gpu_device, w, h, [dict(md=md, M=M2, palette=scene.bone_palette())])


def test_instances_batches_and_draw_order(gpu_device):
    w, h = 480, 270
    texs = [scene.checker_rgba8_texture(16, 16, 2, (255, 128)), scene.random_bc7_texture(16, 16, 8)]
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=16, cols=20, textured=True, textures=texs)
    mats, pals = scene.instance_lattice(4, 3)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    cube = scene.cube_model(6)
    Mc = scene.to_f32_colmajor(scene.reference_view_proj(w, h) @ scene.mat_translate(-5, 0, -1.2) @ scene.mat_scale(.6, .6, .6))
    draws = [dict(md=cube, M=Mc),
             dict(md=md, vp=vp, model_mats=mats, palettes=pals, tex_override=[0, 1, -1, 1, 0, 0, 1, 1, -1, 0, 1, 0]),
             dict(md=md, vp=vp, model_mats=mats[:5] @ np.eye(16, dtype=np.float32), palettes=None),
             dict(md=cube, M=Mc)]
    g = _both(gpu_device, w, h, draws)
    assert g[2]["ndraws"] == 4


def test_debug_overlay_cubes(gpu_device):
    w, h = 256, 256
    cam = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    inst = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5 + 0.3 * i - 0.6, 0.2 * (i % 3) - 0.2, -1.0 - 0.1 * i) @
                                          scene.mat_scale(0.1, 0.1, 0.1) @ scene.mat_rot_y(0.3 * i)) for i in range(7)])
    md = scene.cube_model(2)
    M = scene.to_f32_colmajor(scene.cube_transform(w, h))
    _both(gpu_device, w, h, [dict(md=md, M=M), dict(md=md, vp=cam, overlay=inst)])


def test_many_segments_in_one_bin(gpu_device):
    """1200 translucent quads (one strip each, 0xFFFF-separated) stacked at equal depth inside a couple of
    16x16 bins: ~100 geometry chunks feed each bin, far more than the 64 segments the tile kernel sorts per
    key range, and every fragment blends, so any ordering slip changes the colour."""
    rng = np.random.default_rng(42)
    nq = 1200
    verts, idx = [], []
    for q in range(nq):
        x0, y0 = rng.uniform(1, 26), rng.uniform(1, 12)
        sz = rng.uniform(1.5, 4.0)
        u, v = rng.uniform(0, 1, 2)
        b = 4 * q
        verts += [(x0, y0, .5, u, v), (x0, y0 + sz, .5, u, v), (x0 + sz, y0, .5, u, v), (x0 + sz, y0 + sz, .5, u, v)]
        idx += [b, b + 1, b + 2, b + 3, 0xFFFF]
    tex_img = rng.integers(0, 256, size=(32, 32, 4), dtype=np.uint8)
    tex_img[..., 3] = rng.integers(60, 200, size=(32, 32))
    tex = scene.TextureData(32, 32, scene.TEX_RGBA8, tex_img.tobytes())
    g = _px(gpu_device, [dict(verts=verts, indices=idx, topology=scene.TOPO_STRIP, texture=0)], w=64, h=32, textures=[tex])
    assert g[2]["tris_setup"] == 2 * nq and g[2]["segments"] > 2 * 64 + 20, g[2]


def test_sharded_bins_and_pack_unpack(gpu_device):
    import torch
    from mt_renderer_amd import api
    w, h = 333, 171
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=24, cols=36)
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    draws = [dict(md=md, M=M, palette=scene.bone_palette())]
    full = render_oracle(w, h, draws)
    for world in (2, 3, 8):
        shards = []
        for rank in range(world):
            m = api.Model.new(gpu_device, md)
            m.set_palette(scene.bone_palette())
            fr = api.Frame(gpu_device, w, h)
            fr.set_shard(rank, world)
            m.render(fr, M)
            fr.end()
            own = sharding.owner_map(w, h, world) == rank
            c, d = fr.color(), fr.depth()
            assert (c[own] == full[0][own]).all() and (d[own] == full[1][own]).all()
            buf = torch.zeros(sharding.shard_bytes(w, h, world), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()  # torch's stream and the library's stream are different streams
            fr.pack_color_shard(buf.data_ptr(), buf.numel())
            fr.wait()
            torch.cuda.synchronize()
            got = buf.cpu().numpy().reshape(-1, sharding.BIN, sharding.BIN, 4)
            mine = np.where(own[..., None], c, 0).astype(np.uint8)
            assert (got == sharding.pack_shard(mine, rank, world)).all()
            shards.append(buf)
            fr.close()
            m.close()
        gathered = torch.cat(shards)
        out = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        gpu_device.unpack_color_shards(gathered.data_ptr(), world, w, h, out.data_ptr())
        # the unpack runs on the library's stream
        fr = api.Frame(gpu_device, 16, 16)
        fr.end()
        fr.close()
        torch.cuda.synchronize()
        assert (out.cpu().numpy().reshape(h, w, 4) == full[0]).all()


@pytest.mark.parametrize("seed", range(6))
def test_random_clip_space_soup(gpu_device, seed):
    """Seeded fuzz: random triangles straight in clip space through a random perspective-like matrix --
    negative / zero w, z outside the volume, sub-pixel slivers, huge extents."""
    rng = np.random.default_rng(seed)
    nv, nt = 120, 200
    v = rng.normal(0, 1.2, size=(nv, 3)).astype(np.float32)
    v[rng.integers(0, nv, 10)] *= 50.0
    idx = rng.integers(0, nv, size=3 * nt).astype(np.uint16)
    md = scene.ModelData(
        vertex_buf=v.view(np.uint8).reshape(-1).copy(), index_buf=idx,
        prims=scene.pack_primitive(vertex_num=nv, vertex_stride=12, topology=scene.TOPO_LIST, index_num=len(idx))[None, :],
        layouts=[[(scene.SEM_POSITION, scene.IEF_F32, 3, 0)]], prim_to_texture=np.array([-1], np.int32),
        prim_debug_id=np.array([seed], np.uint32), parts_disp=np.ones(1, np.uint8))
    M = rng.normal(0, 1, size=(4, 4))
    M[3] = (0.1 * rng.normal(), 0.1 * rng.normal(), -1.0, 0.8)  # w depends on z: some vertices behind the eye
    _both(gpu_device, 200, 120, [dict(md=md, M=scene.to_f32_colmajor(M))])


def test_tile_kernel_selection(gpu_device):
    """AUTO picks the visibility-key kernel alone exactly when every material is opaque; a frame with alpha-blended
    translucent materials goes through its order-list variant first (TILE_MIXED: the ordered kernel then takes the bins
    that variant flags -- none here)."""
    from mt_renderer_amd import api
    w, h = 96, 64
    opaque = scene.random_bc7_texture(32, 32, 5, opaque_modes_only=True)
    translucent = scene.checker_rgba8_texture(8, 8, 1, alpha=(255, 254))
    q = [_quad(4, 4, 60, 60, .5, tex=0)]
    for texs, want in (([opaque], api.TILE_VISIBILITY), ([translucent], api.TILE_MIXED)):
        md = pixel_model(q, texs)
        g = render_gpu(gpu_device, w, h, [dict(md=md, M=pixel_to_ndc_matrix(w, h))], tile_mode=api.TILE_AUTO)
        assert g[2]["tile_kernel"] == want
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=8, cols=10)
    g = render_gpu(gpu_device, w, h, [dict(md=md, M=scene.to_f32_colmajor(scene.headline_transform(w, h)))], tile_mode=api.TILE_AUTO)
    assert g[2]["tile_kernel"] == api.TILE_VISIBILITY
    # forcing VISIBILITY on an ineligible frame falls back to the ordered kernel
    md = pixel_model(q, [translucent])
    g = render_gpu(gpu_device, w, h, [dict(md=md, M=pixel_to_ndc_matrix(w, h))], tile_mode=api.TILE_VISIBILITY)
    assert g[2]["tile_kernel"] == api.TILE_ORDERED


def test_opaque_textures_deferred_shading(gpu_device):
    """Opaque textures go through the visibility kernel's deferred shading: magnified, minified, the rho = 1
    switch, perspective, near clip, equal-depth ties and many layers -- identical to the ordered kernel and the oracle."""
    texs = [scene.random_bc7_texture(64, 64, 31, opaque_modes_only=True), scene.checker_rgba8_texture(4, 4, 1),
            scene.checker_rgba8_texture(32, 32, 1)]
    prims = [_quad(0, 0, 8, 8, .5, tex=0), _quad(10, 0, 74, 64, .5, tex=1), _quad(0, 10, 40, 50, .6, tex=0),
             _quad(20, 30, 60, 60, .6, tex=2), _quad(20, 30, 60, 60, .6, tex=1),      # same depth: the later one wins
             _quad(30, 5, 62, 37, .45, tex=0, u0=-0.5, v0=-0.25, u1=1.5, v1=1.25), _quad(5, 40, 30, 63, .3, tex=-1, did=9)]
    g = _px(gpu_device, prims, w=80, h=64, textures=texs)
    assert g[2]["tile_kernel"] == 2
    for scale in (1.0, 1.0000001, 0.9999999):
        _px(gpu_device, [_quad(0, 0, 32, 32, .5, u1=scale, v1=scale, tex=2)], w=32, h=32, textures=texs)
    w, h = 640, 360
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=30, cols=48, textured=True, textures=[texs[0]])
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    g = _both(gpu_device, w, h, [dict(md=md, M=M, palette=scene.bone_palette())])
    assert g[2]["tile_kernel"] == 2
    vp = scene.reference_view_proj(w, h)
    M2 = scene.to_f32_colmajor(vp @ scene.mat_translate(-5.0, 0.0, 1.0 - 0.3) @ scene.mat_rot_x(0.5))
    _both(gpu_device, w, h, [dict(md=md, M=M2, palette=scene.bone_palette())])


def test_single_pass_queue_overflow_falls_back(gpu_device):
    """A bounded per-bin queue that fills up must not lose triangles: the frame is re-run with the exact two-pass
    queues, and the bound doubles for the next frame."""
    from mt_renderer_amd import api
    rng = np.random.default_rng(7)
    verts, idx = [], []
    for q in range(400):  # 800 triangles inside one 16x16 bin
        x0, y0, sz = rng.uniform(1, 10), rng.uniform(1, 10), rng.uniform(1.5, 4.0)
        b = 4 * q
        verts += [(x0, y0, .5 - q * 1e-4), (x0, y0 + sz, .5 - q * 1e-4), (x0 + sz, y0, .5 - q * 1e-4), (x0 + sz, y0 + sz, .5 - q * 1e-4)]
        idx += [b, b + 1, b + 2, b + 3, 0xFFFF]
    md = pixel_model([dict(verts=verts, indices=idx, topology=scene.TOPO_STRIP, debug_id=4)])
    draws = [dict(md=md, M=pixel_to_ndc_matrix(32, 32))]
    ref = render_oracle(32, 32, draws)
    gpu_device.set_binning(True, 64)
    try:
        g = render_gpu(gpu_device, 32, 32, draws, tile_mode=api.TILE_AUTO)
        assert g[2]["binning"] == 2  # fell back
        assert_same(g, ref, "overflow fallback")
        for _ in range(5):  # the bound doubles until the scene fits the single-pass queues
            g = render_gpu(gpu_device, 32, 32, draws, tile_mode=api.TILE_AUTO)
            assert_same(g, ref, "after growth")
        assert g[2]["binning"] == 1
    finally:
        gpu_device.set_binning(True, 1024)


@pytest.mark.gpu
def test_visibility_kernel_many_bin_filling_triangles(gpu_device):
    """150 opaque quads of 20..60 px (i32 edge class) stacked over a 64x64 target at random depths, some of them
    equal: every bin holds hundreds of triangles that each cover all 256 of its pixels, so a pass of the
    visibility kernel's flattened walk spans many 4096-pair rounds and 64-pair batches, and depth ties are
    decided by submission order."""
    rng = np.random.default_rng(7)
    prims = []
    for q in range(150):
        sz = rng.uniform(20, 60)
        x0, y0 = rng.uniform(-10, 50), rng.uniform(-10, 50)
        z = float(rng.integers(1, 9)) / 16.0
        prims.append(dict(verts=[(x0, y0, z), (x0, y0 + sz, z), (x0 + sz, y0 + sz, z), (x0 + sz, y0, z)],
                          indices=[0, 1, 2, 0, 2, 3], debug_id=q))
    g = _px(gpu_device, prims, w=64, h=64)
    assert g[2]["tile_kernel"] == 2 and g[2]["bin_entries"] > 16 * 100, g[2]


def test_unordered_binning_mixed_triangle_sizes(gpu_device):
    """One opaque strip per row whose triangles alternate between sub-pixel slivers, 1-4 bin triangles, 5-16 bin
    triangles (group loop of the unordered binner) and > 16 bin triangles (cooperative wide path), so single chunks mix
    all binning paths and the 8x8 bin window test fails for some rounds and holds for others."""
    rng = np.random.default_rng(11)
    prims = []
    for r in range(6):
        y0 = 8 + r * 40
        xs = np.cumsum(rng.choice([0.4, 3.0, 17.0, 45.0, 90.0], size=90, p=[0.35, 0.3, 0.2, 0.1, 0.05]))
        xs = xs[xs < 500]
        verts, idx = [], []
        for i, x in enumerate(xs):
            h = float(rng.choice([0.6, 6.0, 30.0, 70.0]))
            z = float(rng.integers(1, 15)) / 16.0
            verts += [(float(x), y0, z), (float(x), y0 + h, z)]
            idx += [2 * i, 2 * i + 1]
        prims.append(dict(verts=verts, indices=idx, topology=scene.TOPO_STRIP, debug_id=r))
    g = _px(gpu_device, prims, w=512, h=256)
    assert g[2]["tile_kernel"] == 2 and g[2]["binning"] == 1 and g[2]["tris_setup"] > 100, g[2]


def test_mixed_frame_uses_both_tile_kernels(gpu_device):
    """Opaque and translucent materials in one frame: AUTO renders the bins that hold only opaque triangles with the
    visibility-key kernel and the others with the ordered kernel; the pixels must equal the all-ordered render (and the
    oracle), including bins where an opaque triangle is submitted AFTER a nearer / farther translucent one."""
    from mt_renderer_amd import api
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(40, 220, size=(16, 16))
    translucent = scene.TextureData(16, 16, scene.TEX_RGBA8, img.tobytes())
    opaque = scene.random_bc7_texture(16, 16, 8, opaque_modes_only=True)
    prims = []
    for i in range(40):  # opaque: debug-id and opaque-textured quads all over a 160x96 target
        x0, y0 = rng.uniform(0, 140), rng.uniform(0, 80)
        sz = rng.uniform(4, 30)
        if i % 3 == 0:
            prims.append(_quad(x0, y0, x0 + sz, y0 + sz, float(rng.integers(2, 14)) / 16, tex=1, did=i))
        else:
            v = [(x0, y0, .5), (x0, y0 + sz, .5), (x0 + sz, y0 + sz, .5), (x0 + sz, y0, .5)]
            prims.append(dict(verts=[(a, b, float(rng.integers(2, 14)) / 16) for (a, b, _) in v], indices=[0, 1, 2, 0, 2, 3], debug_id=i))
    for i in range(12):  # translucent quads confined to the left third, interleaved in submission order
        x0, y0 = rng.uniform(0, 40), rng.uniform(0, 80)
        sz = rng.uniform(6, 20)
        prims.insert(int(rng.integers(0, len(prims))), _quad(x0, y0, x0 + sz, y0 + sz, float(rng.integers(2, 14)) / 16, tex=0))
    g = _px(gpu_device, prims, w=160, h=96, textures=[translucent, opaque])
    assert g[2]["tile_kernel"] == api.TILE_MIXED, g[2]


def test_frames_in_flight_keep_their_own_pixels(gpu_device):
    """Ten frames with different transforms (and two target sizes) are submitted back to back without waiting --
    they overlap on the library's internal streams, share its slots and recycle colour / depth sets -- then read in
    a scrambled order; each must equal the oracle's render of ITS transform.  Twice, so the second round runs on
    recycled framebuffers whose previous frames are still in flight when it starts."""
    from mt_renderer_amd import api
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=16, cols=24)
    pal = scene.bone_palette()
    m = api.Model.new(gpu_device, md)
    m.set_palette(pal)
    try:
        for rnd in range(2):
            frames, want = [], []
            for k in range(10):
                w, h = (240, 136) if k % 3 else (200, 120)
                T = scene.headline_transform(w, h) @ scene.mat_rot_y(0.3 * k + rnd) @ scene.mat_translate(0.05 * k, -0.02 * k, 0.0)
                M = scene.to_f32_colmajor(T)
                fr = api.Frame(gpu_device, w, h)
                m.render(fr, M)
                fr.submit()
                frames.append(fr)
                want.append((w, h, M))
            for k in (7, 0, 9, 3, 1, 8, 2, 6, 4, 5):
                w, h, M = want[k]
                ref = render_oracle(w, h, [dict(md=md, M=M, palette=pal)])
                fr = frames[k]
                fr.wait()
                assert_same((fr.color(), fr.depth(), fr.stats()), ref, f"frame {k} of round {rnd}")
            for fr in frames:
                fr.close()
    finally:
        m.close()


def test_palette_changes_between_frames_in_flight(gpu_device):
    """An animated model: a new bone palette before every frame, forty frames submitted without waiting (more palette
    changes than the library's ring of palette buffers holds, so buffers are reused while frames are in flight), then
    read in reverse; every frame must show ITS palette.  A second model gets many palette changes and few frames."""
    from mt_renderer_amd import api
    w, h = 200, 120
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=14, cols=20)
    M = scene.to_f32_colmajor(scene.headline_transform(w, h))
    m = api.Model.new(gpu_device, md)
    m2 = api.Model.new(gpu_device, md)
    try:
        frames, pals = [], []
        for k in range(40):
            pal = scene.bone_palette(t=0.05 * k)
            m.set_palette(pal)
            fr = api.Frame(gpu_device, w, h)
            m.render(fr, M)
            fr.submit()
            frames.append(fr)
            pals.append(pal)
        for k in reversed(range(40)):
            ref = render_oracle(w, h, [dict(md=md, M=M, palette=pals[k])]) if k % 4 == 0 else None
            frames[k].wait()
            if ref is not None:
                assert_same((frames[k].color(), frames[k].depth(), frames[k].stats()), ref, f"animated frame {k}")
            frames[k].close()
        # many palette changes per frame: a buffer comes round while the one frame that read it may still run
        got = []
        for k in range(3):
            pal = scene.bone_palette(t=0.3 + 0.2 * k)
            m2.set_palette(pal)
            fr = api.Frame(gpu_device, w, h)
            m2.render(fr, M)
            fr.submit()
            for j in range(25):
                m2.set_palette(scene.bone_palette(t=1.0 + 0.01 * j))  # overwrites the whole ring
            got.append((fr, pal))
        for fr, pal in got:
            fr.wait()
            assert_same((fr.color(), fr.depth(), fr.stats()), render_oracle(w, h, [dict(md=md, M=M, palette=pal)]), "frame under palette churn")
            fr.close()
    finally:
        m.close()
        m2.close()


def test_dynamic_instances_between_frames_in_flight(gpu_device):
    """Per-frame instance matrices: twenty-four frames, each drawing its own freshly uploaded instances -- odd frames
    through a Batch that is destroyed right after submit (its buffers must outlive the frame that is still in flight),
    even frames through draw_instances (a batch owned by the frame, which is destroyed in flight too) -- submitted
    without waiting; then every fourth one is compared with the oracle."""
    from mt_renderer_amd import api
    w, h = 240, 136
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=10, cols=16)
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    m = api.Model.new(gpu_device, md)
    try:
        frames, args = [], []
        for k in range(24):
            mats, pals = scene.instance_lattice(3, 2, seed=100 + k)
            fr = api.Frame(gpu_device, w, h)
            if k % 2:
                b = api.Batch(gpu_device, m, mats, pals)
                fr.draw_batch(b, vp)
                fr.submit()
                b.close()
            else:
                fr.draw_instances(m, vp, mats, pals)
                fr.submit()
            frames.append(fr)
            args.append((mats, pals))
        colors = {}
        for k in reversed(range(24)):
            frames[k].wait()
            if k % 4 < 2:
                colors[k] = (frames[k].color(), frames[k].depth(), frames[k].stats())
            frames[k].close()
        for k, got in colors.items():
            mats, pals = args[k]
            assert_same(got, render_oracle(w, h, [dict(md=md, vp=vp, model_mats=mats, palettes=pals)]), f"dynamic instances, frame {k}")
    finally:
        m.close()


def test_fragment_lists_small_translucent_triangles_odd_viewport(gpu_device):
    """The ordered kernel's fragment-list path: a finely tessellated skinned mesh with a TRANSLUCENT texture (thousands of
    2-10 pixel triangles, several layers deep where the capsule's back shows through... culled, so 1-2 layers), drawn
    twice with different transforms so fragments of two draws interleave per pixel, plus overlay cubes (constant
    colour, no blend) in the same frame, on a 203x117 target whose last bins are partial."""
    from mt_renderer_amd import api
    w, h = 203, 117
    rng = np.random.default_rng(21)
    img = rng.integers(0, 256, size=(32, 32, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(30, 230, size=(32, 32))
    tex = scene.TextureData(32, 32, scene.TEX_RGBA8, img.tobytes())
    md = scene.mesh50k(textured=True, textures=[tex], rows=40, cols=64)
    pal = scene.bone_palette()
    M1 = scene.to_f32_colmajor(scene.headline_transform(w, h))
    M2 = scene.to_f32_colmajor(scene.headline_transform(w, h) @ scene.mat_rot_y(0.4) @ scene.mat_translate(0.1, 0.05, 0.0))
    cubes = np.stack([scene.to_f32_colmajor(scene.mat_translate(-5.0 + 0.1 * i, 0.05 * i, -1.3) @ scene.mat_scale(0.05, 0.05, 0.05)) for i in range(4)])
    vp = scene.to_f32_colmajor(scene.reference_view_proj(w, h))
    draws = [dict(md=md, M=M1, palette=pal), dict(md=md, overlay=cubes, vp=vp), dict(md=md, M=M2, palette=pal)]
    g = render_gpu(gpu_device, w, h, draws)
    assert g[2]["tile_kernel"] in (api.TILE_ORDERED, api.TILE_MIXED) and g[2]["tris_setup"] > 2000, g[2]
    assert_same(g, render_oracle(w, h, draws), "fragment lists")


def test_translucent_layers_resolve_through_prefix_minima(gpu_device):
    """Alpha-blended translucent fragments in the default depth state: a pixel's result depends on submission order only
    through the fragments that pass LessEqual, i.e. the prefix minima of z -- what the visibility kernel's order lists hold
    (k_tile_vis.hip, STAIR).  Layered quads over the same pixels: depths falling in submission order (every layer passes
    and blends), rising (only the first passes), shuffled, exact ties (a later equal z passes too), an opaque layer in
    the middle (replaces what is below), more passing layers than a list holds (the bin falls back to the ordered
    kernel), and a hard order-dependent material next to them (additive blend: its bins are the ordered kernel's).
    AUTO == ORDERED == oracle, bit for bit (render through _px: every queue builder x both tile paths)."""
    from mt_renderer_amd import api
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, size=(16, 16, 4), dtype=np.uint8)
    img[..., 3] = rng.integers(30, 230, size=(16, 16))
    translucent = scene.TextureData(16, 16, scene.TEX_RGBA8, img.tobytes())
    opaque = scene.random_bc7_texture(16, 16, 9, opaque_modes_only=True)

    def layers(zs, x0=3.0, y0=2.0, size=26.0, step=0.7, texs=None):
        out = []
        for i, z in enumerate(zs):
            t = 0 if texs is None else texs[i]
            out.append(_quad(x0 + step * i, y0 + 0.4 * i, x0 + step * i + size, y0 + 0.4 * i + size, z, tex=t, did=i,
                             u0=0.1 * (i % 3), v0=0.07 * (i % 4), u1=1.0 + 0.2 * (i % 2), v1=0.9))
        return out

    falling = [0.9 - 0.05 * i for i in range(7)]
    shuffled = [float(v) for v in rng.permutation(np.linspace(0.2, 0.8, 8))]
    cases = {
        "falling": layers(falling),
        "rising": layers(falling[::-1]),
        "shuffled": layers(shuffled),
        "ties": layers([0.5, 0.5, 0.6, 0.5, 0.4, 0.4, 0.7]),
        "opaque in the middle": layers([0.8, 0.7, 0.6, 0.5, 0.4], texs=[0, 0, 1, 0, 0]) + [dict(verts=[(8, 8, .45), (8, 30, .45), (30, 30, .45)], indices=[0, 1, 2], debug_id=4)],
        "twelve passing layers": layers([0.95 - 0.05 * i for i in range(12)], step=0.3),
        "two bins, one deep": layers(shuffled, x0=2.0) + layers([0.95 - 0.04 * i for i in range(11)], x0=40.0, step=0.2),
    }
    for name, prims in cases.items():
        g = _px(gpu_device, prims, w=80, h=48, textures=[translucent, opaque])
        assert g[2]["tile_kernel"] == api.TILE_MIXED, (name, g[2])
    # a hard order-dependent material (additive blend) among alpha-blended ones: per primitive state
    prims = layers(shuffled[:5]) + layers([0.3, 0.6, 0.2], x0=44.0)
    md = pixel_model(prims, [translucent, opaque])
    st = np.tile(np.array([[api_blend("alpha"), 1, 1, 0]], dtype=np.uint8), (md.nprims, 1))
    st[6, 0] = api_blend("add")
    md.prim_states = st
    M = pixel_to_ndc_matrix(80, 48)
    _both(gpu_device, 80, 48, [dict(md=md, M=M)])


def api_blend(name):
    return {"alpha": 0, "off": 1, "add": 2}[name]  # include/mtr.h: MTR_BLEND_*
