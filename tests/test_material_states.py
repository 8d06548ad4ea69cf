"""CPU: known answers of the oracle's material-state extension (SPEC.md section 10): blend off /
additive, depth write off, depth test off, cull none / front, mip-level selection, SCMP3N positions -- each computed by
hand for a scene whose pixels are obvious."""
import numpy as np
import pytest

from mt_renderer_amd import scene
from oracle import oracle as orc
from tests.pixel_scenes import PALETTE, pixel_model, pixel_to_ndc_matrix

BLEND_ALPHA, BLEND_OFF, BLEND_ADD = 0, 1, 2
CULL_BACK, CULL_NONE, CULL_FRONT = 0, 1, 2


def _quad(x0, y0, x1, y1, z, tex=-1, did=0, flip=False):
    v = [(x0, y0, z, 0.0, 0.0), (x0, y1, z, 0.0, 1.0), (x1, y1, z, 1.0, 1.0), (x1, y0, z, 1.0, 0.0)]
    idx = [0, 1, 2, 0, 2, 3] if not flip else [0, 2, 1, 0, 3, 2]
    return dict(verts=v, indices=idx, texture=tex, debug_id=did)


def _render(prims, states, textures=(), w=32, h=32):
    md = pixel_model(prims, list(textures))
    md.prim_states = None if states is None else np.array(states, dtype=np.uint8)
    f = orc.OracleFrame(w, h)
    f.draw(orc.OracleModel(md), pixel_to_ndc_matrix(w, h))
    out = f.color(), f.depth()
    f.close()
    return out


def _solid_tex(rgba):
    return scene.TextureData(2, 2, scene.TEX_RGBA8, bytes(rgba) * 4)


def test_default_state_is_the_reference_pipeline():
    prims = [_quad(4, 4, 20, 20, .5, did=1), _quad(10, 10, 28, 28, .25, did=2)]
    a = _render(prims, None)
    b = _render(prims, [(BLEND_ALPHA, 1, 1, CULL_BACK)] * 2)
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


def test_blend_off_replaces_and_additive_adds():
    half = _solid_tex((200, 100, 50, 128))
    # over white: alpha blend mixes, OFF stores the texel, ADD saturates
    c_alpha, _ = _render([_quad(0, 0, 16, 16, .5, tex=0)], [(BLEND_ALPHA, 1, 1, CULL_BACK)], [half])
    c_off, _ = _render([_quad(0, 0, 16, 16, .5, tex=0)], [(BLEND_OFF, 1, 1, CULL_BACK)], [half])
    c_add, _ = _render([_quad(0, 0, 16, 16, .5, tex=0)], [(BLEND_ADD, 1, 1, CULL_BACK)], [half])
    assert tuple(c_off[8, 8]) == (200, 100, 50, 128)
    a = np.float32(128) / np.float32(255)
    exp = [int(np.rint(np.clip(np.float32(c) / np.float32(255) * a + np.float32(1.0) * (np.float32(1) - a), 0, 1) * 255)) for c in (200, 100, 50)]
    assert tuple(c_alpha[8, 8][:3]) == tuple(exp) and c_alpha[8, 8][3] == 128
    assert tuple(c_add[8, 8]) == (255, 255, 255, 128)  # white + anything saturates
    # additive over a dark layer: dst = (10, 20, 30) stored by a blend-off quad, then + src * a
    dark = _solid_tex((10, 20, 30, 255))
    c, _ = _render([_quad(0, 0, 16, 16, .5, tex=0), _quad(0, 0, 16, 16, .4, tex=1)],
                   [(BLEND_OFF, 1, 1, CULL_BACK), (BLEND_ADD, 1, 1, CULL_BACK)], [dark, half])
    exp = [int(np.rint(np.clip(np.float32(np.float32(s) / np.float32(255)) * a + np.float32(d) / np.float32(255), 0, 1) * 255))
           for s, d in ((200, 10), (100, 20), (50, 30))]
    assert tuple(c[8, 8][:3]) == tuple(exp)


def test_depth_write_off_and_depth_test_off():
    # near quad first with depth write off, then a farther quad: the far one still passes (the depth buffer kept 1.0)
    c, d = _render([_quad(0, 0, 16, 16, .25, did=1), _quad(0, 0, 16, 16, .75, did=2)],
                   [(BLEND_ALPHA, 0, 1, CULL_BACK), (BLEND_ALPHA, 1, 1, CULL_BACK)])
    assert tuple(c[8, 8][:3]) == tuple(PALETTE[2]) and d[8, 8] == np.float32(.75)
    c, d = _render([_quad(0, 0, 16, 16, .25, did=1), _quad(0, 0, 16, 16, .75, did=2)], None)
    assert tuple(c[8, 8][:3]) == tuple(PALETTE[1]) and d[8, 8] == np.float32(.25)
    # depth test off: the far quad drawn last wins although the near one wrote depth; it writes its own depth too
    c, d = _render([_quad(0, 0, 16, 16, .25, did=1), _quad(0, 0, 16, 16, .75, did=2)],
                   [(BLEND_ALPHA, 1, 1, CULL_BACK), (BLEND_ALPHA, 1, 0, CULL_BACK)])
    assert tuple(c[8, 8][:3]) == tuple(PALETTE[2]) and d[8, 8] == np.float32(.75)
    # ... but near / far clipping still applies without the test
    c, _ = _render([_quad(0, 0, 16, 16, 1.5, did=3)], [(BLEND_ALPHA, 1, 0, CULL_BACK)])
    assert tuple(c[8, 8]) == (255, 255, 255, 255)


def test_cull_modes():
    front, back = _quad(0, 0, 16, 16, .5, did=1), _quad(16, 0, 32, 16, .5, did=2, flip=True)
    white = (255, 255, 255)
    for cull, exp in ((CULL_BACK, (tuple(PALETTE[1]), white)), (CULL_NONE, (tuple(PALETTE[1]), tuple(PALETTE[2]))),
                      (CULL_FRONT, (white, tuple(PALETTE[2])))):
        c, _ = _render([front, back], [(BLEND_ALPHA, 1, 1, cull)] * 2)
        assert (tuple(c[8, 8][:3]), tuple(c[8, 24][:3])) == exp, cull
    # a kept back face covers exactly the pixels of the same quad wound the other way (top-left rule included)
    a, _ = _render([_quad(3, 5, 19, 27, .5, did=4)], [(BLEND_ALPHA, 1, 1, CULL_BACK)])
    b, _ = _render([_quad(3, 5, 19, 27, .5, did=4, flip=True)], [(BLEND_ALPHA, 1, 1, CULL_NONE)])
    assert (a == b).all()


def _mip_chain(w, h, colours):
    data, lw, lh = b"", w, h
    for c in colours:
        data += bytes(c) * (lw * lh)
        lw, lh = max(1, lw >> 1), max(1, lh >> 1)
    return scene.TextureData(w, h, scene.TEX_RGBA8, data, levels=len(colours))


def test_mip_level_selection():
    cols = [(255, 0, 0, 255), (0, 255, 0, 255), (0, 0, 255, 255), (255, 255, 0, 255), (0, 255, 255, 255)]
    t = _mip_chain(64, 64, cols)
    # 64 texels over n pixels: m = 64 / n; level l while m > 2^(l - 1/2)
    for n, level in ((64, None), (32, 1), (16, 2), (8, 3), (4, 4), (2, 4), (46, 0), (45, 1), (23, 1), (22, 2)):
        c, _ = _render([_quad(0, 0, n, n, .5, tex=0)], None, [t], w=64, h=64)
        exp = cols[0] if level is None else cols[level]
        assert tuple(c[n // 2, n // 2]) == exp, (n, tuple(c[n // 2, n // 2]))
    # one level: always level 0 (the reference)
    t1 = _mip_chain(64, 64, cols[:1])
    c, _ = _render([_quad(0, 0, 4, 4, .5, tex=0)], None, [t1], w=64, h=64)
    assert tuple(c[2, 2]) == cols[0]


def test_scmp3n_position_decode_is_opt_in():
    # three signed 10-bit fields: (511, -511, 0) -> (1, -1, 0); a triangle in NDC through the identity matrix
    def pack(x, y, z):
        return np.uint32((x & 0x3ff) | ((y & 0x3ff) << 10) | ((z & 0x3ff) << 20))
    verts = np.array([pack(-511, -511, 256), pack(-511, 511, 256), pack(511, 511, 256)], dtype=np.uint32)  # CCW in NDC (y up)?
    for order in ([0, 1, 2], [0, 2, 1]):
        md = scene.ModelData(vertex_buf=verts.view(np.uint8).copy(), index_buf=np.array(order, dtype=np.uint16),
                             prims=np.stack([scene.pack_primitive(vertex_num=3, vertex_stride=4, topology=scene.TOPO_LIST, index_num=3)]),
                             layouts=[[(scene.SEM_POSITION, scene.IEF_SCMP3N, 1, 0, 1)]], prim_to_texture=np.array([-1], dtype=np.int32),
                             prim_debug_id=np.array([6], dtype=np.uint32), parts_disp=np.ones(1, dtype=np.uint8))
        f = orc.OracleFrame(16, 16)
        f.draw(orc.OracleModel(md), np.eye(4, dtype=np.float32).reshape(16))
        col, dep = f.color(), f.depth()
        f.close()
        if (col[..., :3] != 255).any():
            assert np.isclose(dep[dep < 1].max(), 256 / 511, atol=1e-6)
            hit = True
    assert hit
    md.layouts = [[(scene.SEM_POSITION, scene.IEF_SCMP3N, 1, 0)]]  # not opted in: skipped like the reference, so no Position at all
    f = orc.OracleFrame(16, 16)
    with pytest.raises(orc.OracleError):
        f.draw(orc.OracleModel(md), np.eye(4, dtype=np.float32).reshape(16))
    f.close()
