"""CPU: pins the oracle against everything the reference's own files hold for this path:
crc32 KATs, PrimitiveInfo bit-fields / struct sizes, cube geometry + cull state, the 20-colour
palette -- and against hand-computable raster cases for the rules the reference delegates to hardware."""
import json
import os

import numpy as np
import pytest

from mt_renderer_amd import scene
from oracle import oracle as orc
from tests.pixel_scenes import PALETTE, pixel_model, pixel_to_ndc_matrix

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_crc32_kat():
    # src/util/crc.rs:52-63
    assert orc.crc32(b"MtObject") == 0x2EA10CEB
    assert orc.crc32(b"MtObject\0") == 0x2EA10CEB
    assert orc.crc32(b"MtObject\0garbage") == 0x2EA10CEB


def test_crc32_dti_table():
    # src/dti.rs:169-193: hash == crc32(name, u32::MAX) & 0x7fffffff for the DTI table
    pairs = json.load(open(os.path.join(GOLDEN, "dti_crc32_kat.json")))["pairs"]
    assert len(pairs) > 250
    for p in pairs:
        assert orc.crc32(p["name"].encode()) & 0x7FFFFFFF == p["hash"], p["name"]
    by_name = {p["name"]: p["hash"] for p in pairs}
    # the three resource classes on the draw path (component 13)
    assert by_name["rModel"] == 1486968918 and by_name["rTexture"] == 606035435 and by_name["rMaterial"] == 659146920


def test_primitive_info_bitfields():
    # src/rmodel.rs:173-225 accessors, :489 size
    rec = scene.pack_primitive(vertex_num=0xBEEF, parts_no=0xABC, material_no=0x123, weight_num=0x15, vertex_stride=0x24,
                               topology=4, vertex_ofs=7, vertex_base=0x11223344, inputlayout=0xDEADB000, index_ofs=0x01020304,
                               index_num=0x0A0B0C0D, index_base=0x55, boundary_num=0xEE)
    assert rec.size == 0x38
    exp = dict(vertex_num=0xBEEF, parts_no=0xABC, material_no=0x123, weight_num=0x15, vertex_stride=0x24, topology=4,
               vertex_base=0x11223344, inputlayout=0xDEADB000, index_ofs=0x01020304, index_num=0x0A0B0C0D, index_base=0x55,
               boundary_num=0xEE)
    for k, v in exp.items():
        assert orc.prim_field(k, rec) == v, k


def test_cube_all_faces_front_from_outside():
    """The reference cube (src/debug_overlay.rs:10-35) is wound for cull_mode Back / front CCW
    (src/debug_overlay.rs:165-168): seen from outside exactly the camera-facing half survives, and from
    inside the cube nothing does."""
    w = h = 256
    om = orc.OracleModel(scene.cube_model(0))
    f = orc.OracleFrame(w, h)
    f.draw(om, scene.to_f32_colmajor(scene.cube_transform(w, h)))
    st = f.stats()
    assert st["tris_in"] == 12 and st["tris_setup"] == 6
    col = f.color()
    covered = (f.depth() < 1.0)
    assert covered.sum() > 10000
    assert (col[covered] == np.array([*PALETTE[0], 255], dtype=np.uint8)).all()  # palette[0 % 20], alpha 1
    assert (col[~covered] == 255).all()  # cleared to white, src/bin/modelviewer.rs:196
    # camera at the cube centre: every face is seen from behind -> all culled
    vp = scene.reference_view_proj(w, h)
    g = orc.OracleFrame(w, h)
    g.draw(om, scene.to_f32_colmajor(vp @ scene.mat_translate(-5.0, 0.0, 1.0) @ scene.mat_scale(3, 3, 3)))
    assert g.stats()["tris_setup"] == 0


def test_debug_palette_all_ids():
    w = h = 64
    M = pixel_to_ndc_matrix(w, h)
    for i in (0, 7, 19, 20, 39, 0xFFFFFFFF):
        md = pixel_model([dict(verts=[(0, 0, .5), (0, 32, .5), (32, 0, .5)], indices=[0, 1, 2], debug_id=i)])
        f = orc.OracleFrame(w, h)
        f.draw(orc.OracleModel(md), M)
        assert tuple(f.color()[2, 2]) == (*PALETTE[i % 20], 255)


def _frags(prims, w=64, h=64, **kw):
    f = orc.OracleFrame(w, h, **kw)
    f.draw(orc.OracleModel(pixel_model(prims)), pixel_to_ndc_matrix(w, h))
    return f


def test_right_triangle_top_left_rule():
    # legs on x=0 and y=0 (left + top edges: inclusive), hypotenuse x+y=8 (exclusive): centres with i+j+1 < 8
    f = _frags([dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2])])
    cov = f.depth() < 1.0
    exp = np.zeros((64, 64), bool)
    for j in range(8):
        for i in range(8):
            exp[j, i] = i + j + 1 < 8
    assert (cov == exp).all() and f.stats()["frags"] == 28


def test_shared_edge_no_double_hit_no_gap():
    # a square split along its diagonal, vertices ON pixel centres: every centre inside is hit exactly once
    v = [(2.5, 2.5, .5), (2.5, 10.5, .5), (10.5, 10.5, .5), (10.5, 2.5, .5)]
    f = _frags([dict(verts=v, indices=[0, 1, 2, 0, 2, 3])])
    cov = f.depth() < 1.0
    assert f.stats()["tris_setup"] == 2
    assert f.stats()["frags"] == cov.sum() == 64  # [2.5,10.5) x [2.5,10.5): left/top inclusive, right/bottom exclusive
    assert cov[2:10, 2:10].all()


def test_back_face_and_degenerate_culled():
    assert _frags([dict(verts=[(0, 0, .5), (8, 0, .5), (0, 8, .5)], indices=[0, 1, 2])]).stats()["tris_setup"] == 0
    assert _frags([dict(verts=[(0, 0, .5), (4, 4, .5), (8, 8, .5)], indices=[0, 1, 2])]).stats()["tris_setup"] == 0


def test_strip_parity_and_restart():
    # strip over a 2x1 quad ribbon: (row b, row a) interleave; every triangle front-facing
    v = [(0, 0, .5), (0, 8, .5), (8, 0, .5), (8, 8, .5), (16, 0, .5), (16, 8, .5)]
    f = _frags([dict(verts=v, indices=[0, 1, 2, 3, 4, 5], topology=scene.TOPO_STRIP)])
    assert f.stats()["tris_in"] == 4 and f.stats()["tris_setup"] == 4 and f.stats()["frags"] == 16 * 8
    # 0xFFFF restarts the strip (src/model.rs:251): two separate triangles, parity restarts at even
    g = _frags([dict(verts=v, indices=[0, 1, 2, 0xFFFF, 2, 3, 4], topology=scene.TOPO_STRIP)])
    assert g.stats()["tris_in"] == 2 and g.stats()["tris_setup"] == 2
    # a restart leaving fewer than 3 indices yields nothing
    k = _frags([dict(verts=v, indices=[0, 1, 0xFFFF, 2, 3, 0xFFFF, 4], topology=scene.TOPO_STRIP)])
    assert k.stats()["tris_in"] == 0


def test_depth_less_equal_tie_and_order():
    a = dict(verts=[(0, 0, .5), (0, 16, .5), (16, 0, .5)], indices=[0, 1, 2], debug_id=1)
    b = dict(verts=[(0, 0, .5), (0, 16, .5), (16, 0, .5)], indices=[0, 1, 2], debug_id=2)
    c = dict(verts=[(0, 0, .75), (0, 16, .75), (16, 0, .75)], indices=[0, 1, 2], debug_id=3)
    f = _frags([a, b, c])
    assert tuple(f.color()[1, 1][:3]) == tuple(PALETTE[2])  # equal depth: the later primitive wins (LessEqual)
    assert f.depth()[1, 1] == np.float32(0.5)
    g = _frags([c, a])
    assert tuple(g.color()[1, 1][:3]) == tuple(PALETTE[1])  # nearer wins regardless of order


def test_depth_range_clip():
    # z outside [0,1] is clipped (unclipped_depth off): a triangle spanning z -0.5 .. 1.5 keeps only the middle
    f = _frags([dict(verts=[(0, 0, -0.5), (0, 32, -0.5), (64, 0, 1.5)], indices=[0, 1, 2])])
    d = f.depth()
    cov = d < 1.0
    assert cov.any() and (d[cov] >= 0).all()
    assert not cov[1, 1] and not cov[0, 60]


def test_parts_disp_indexed_by_parts_no():
    a = dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2], parts_no=1)
    b = dict(verts=[(20, 0, .5), (20, 8, .5), (28, 0, .5)], indices=[0, 1, 2], parts_no=0)
    md = pixel_model([a, b], parts_disp=[1, 0])  # src/model.rs:318: parts_disp[primitive.parts_no()]
    f = orc.OracleFrame(64, 64)
    f.draw(orc.OracleModel(md), pixel_to_ndc_matrix(64, 64))
    cov = f.depth() < 1.0
    assert not cov[1, 1] and cov[1, 21]  # a: parts_disp[1] = 0 (hidden), b: parts_disp[0] = 1
    md2 = pixel_model([dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2], parts_no=5)])
    with pytest.raises(orc.OracleError):  # the reference would panic on the out-of-bounds index
        orc.OracleFrame(64, 64).draw(orc.OracleModel(md2), pixel_to_ndc_matrix(64, 64))


def test_out_of_range_vertex_drops_triangle():
    f = _frags([dict(verts=[(0, 0, .5), (0, 8, .5), (8, 0, .5)], indices=[0, 1, 2, 0, 1, 7])])
    assert f.stats()["tris_in"] == 2 and f.stats()["tris_setup"] == 1


def test_unsupported_formats_error():
    md = scene.cube_model()
    md.layouts = [[(scene.SEM_POSITION, scene.IEF_U16, 2, 0)]]  # todo!() arm, src/rshader2.rs:548-551 maps to Uint16x2
    with pytest.raises(orc.OracleError) as e:
        orc.OracleFrame(16, 16).draw(orc.OracleModel(md), np.eye(4, dtype=np.float32))
    assert e.value.code == 2
    with pytest.raises(orc.OracleError) as e:
        orc.decode_texture(99, 4, 4, bytes(64))  # src/rtexture.rs:159
    assert e.value.code == 2


def test_lbs_single_bone_identity_is_bit_identical():
    md = scene.skinned_capsule_model([((0.1, -0.2, 0.05), 0.3, 1.2)], rows=6, cols=9)
    vb = md.vertex_buf.reshape(-1, scene.SKINNED_STRIDE)
    vb[:, 20:24] = (255, 0, 0, 0)
    om = orc.OracleModel(md)
    M = scene.to_f32_colmajor(scene.headline_transform(640, 360))
    c0, uv0 = om.vertex_stage(0, M, None)
    c1, uv1 = om.vertex_stage(0, M, np.tile(scene.to_f32_colmajor(np.eye(4)), (64, 1)))
    assert (c0.view(np.uint32) == c1.view(np.uint32)).all() and (uv0 == uv1).all()
    # a rigid single-bone palette equals the unskinned path through the composed chain order
    P = scene.to_f32_colmajor(scene.mat_translate(0.1, 0.2, -0.3) @ scene.mat_rot_y(0.3))
    c2, _ = om.vertex_stage(0, M, np.tile(P, (64, 1)))
    ref = np.zeros_like(c2)
    pos = vb[:, 0:8].copy().view("<i2").reshape(-1, 4)[:, :3].astype(np.float32)
    pos = np.maximum(pos / np.float32(32767.0), np.float32(-1.0)).astype(np.float32)
    Pm, Mm = P.reshape(4, 4).T.astype(np.float64), M.reshape(4, 4).T.astype(np.float64)
    q = (Pm @ np.concatenate([pos.astype(np.float64), np.ones((len(pos), 1))], axis=1).T).T
    ref = (Mm @ np.concatenate([q[:, :3], np.ones((len(pos), 1))], axis=1).T).T
    assert np.allclose(c2, ref, rtol=2e-5, atol=2e-5)


def test_threads_do_not_change_results():
    md = scene.skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows=30, cols=40)
    om = orc.OracleModel(md)
    M = scene.to_f32_colmajor(scene.headline_transform(320, 180))
    out = []
    for nt in (1, 3):
        f = orc.OracleFrame(320, 180)
        f.draw(om, M, scene.bone_palette(), nthreads=nt)
        out.append((f.color(), f.depth(), f.stats()))
    assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all() and out[0][2] == out[1][2]
