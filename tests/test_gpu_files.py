"""-m gpu: Model::new over resource FILES (mtr_model_create_from_files, src/model.rs:36-293) renders bit for bit what
the oracle renders from the same model given directly: layouts resolved through the shader package by handle,
textures through material name -> rMaterial -> tAlbedoMap, debug ids through boundary joints, textures uploaded from
.tex images (RGBA8 / BC1 / BC7)."""
import numpy as np
import pytest

from mt_renderer_amd import api, files, scene
from tests import mt_files
from tests.helpers import assert_same, render_gpu, render_oracle
from tests.pixel_scenes import pixel_model, pixel_to_ndc_matrix

pytestmark = pytest.mark.gpu


def _from_files(md, drop_texture=None):
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)

    def make(dev):
        sh = files.Shader2File(rshader2)
        mat = files.MaterialFile(rmaterial, sh)
        assert len(mat.textures()) == len(rtextures)
        tex = [None if i == drop_texture else files.TextureFile(b).upload(dev) for i, b in enumerate(rtextures)]
        return files.model_from_files(dev, files.ModelFile(rmodel), sh, mat, tex)
    return make


def test_pixel_model_from_files(gpu_device):
    tex = [scene.checker_rgba8_texture(16, 16, alpha=(255, 120)), scene.random_bc1_texture(8, 8), scene.random_bc7_texture(16, 16, seed=5)]
    prims = [dict(verts=[(1, 1, .5, 0, 0), (1, 60, .5, 0, 1), (60, 60, .5, 1, 1)], indices=[0, 1, 2], texture=1, debug_id=7),
             dict(verts=[(2, 2, .25), (2, 40, .25), (40, 2, .25), (40, 40, .25)], indices=[0, 1, 2, 3], topology=scene.TOPO_STRIP,
                  debug_id=3, parts_no=1),
             dict(verts=[(5, 5, .75, .5, .5), (5, 55, .75, .5, 1), (55, 55, .75, 1, 1)], indices=[0, 1, 2], texture=0, debug_id=11),
             dict(verts=[(30, 3, .1, 0, 0), (30, 33, .1, 0, 1), (62, 33, .1, 1, 1)], indices=[0, 1, 2], texture=2, debug_id=19)]
    md = pixel_model(prims, textures=tex)
    M = pixel_to_ndc_matrix(64, 64)
    ref = render_oracle(64, 64, [dict(md=md, M=M)])
    got = render_gpu(gpu_device, 64, 64, [dict(md=md, M=M, make_model=_from_files(md))])
    assert_same(got, ref, "pixel model from files")


def test_skinned_textured_mesh_from_files(gpu_device):
    W, H = 320, 200
    md = scene.mesh50k(textured=True, textures=[scene.random_bc7_texture(64, 64, seed=9)], rows=12, cols=20)
    M = scene.to_f32_colmajor(scene.headline_transform(W, H))
    pal = scene.bone_palette()
    ref = render_oracle(W, H, [dict(md=md, M=M, palette=pal)])
    got = render_gpu(gpu_device, W, H, [dict(md=md, M=M, palette=pal, make_model=_from_files(md))])
    assert_same(got, ref, "skinned mesh from files")


def test_unloaded_texture_is_an_error_only_when_needed(gpu_device):
    tex = [scene.checker_rgba8_texture(8, 8), scene.checker_rgba8_texture(8, 8)]
    prims = [dict(verts=[(1, 1, .5, 0, 0), (1, 30, .5, 0, 1), (30, 30, .5, 1, 1)], indices=[0, 1, 2], texture=1)]
    md = pixel_model(prims, textures=tex)
    m = _from_files(md, drop_texture=0)(gpu_device)  # texture 0 failed to load, nothing uses it: fine (src/model.rs:46-58)
    m.close()
    with pytest.raises(api.MtrError) as e:           # texture 1 is the primitive's albedo: "no texture found!" (src/model.rs:167)
        _from_files(md, drop_texture=1)(gpu_device)
    assert e.value.code == api.MTR_E_INVALID and "no texture found" in str(e.value)


def test_unknown_input_layout_handle_is_an_error(gpu_device):
    md = pixel_model([dict(verts=[(1, 1, .5), (1, 30, .5), (30, 30, .5)], indices=[0, 1, 2])])
    rmodel, rshader2, rmaterial, _ = mt_files.files_from_model_data(md)
    bad = mt_files.write_rmodel(md, [mt_files.handle_of("IANotThere")], ["mat_0"], [0])
    sh = files.Shader2File(rshader2)
    with pytest.raises(api.MtrError) as e:  # panic!("invalid inputlayout ..."), src/model.rs:182-183
        files.model_from_files(gpu_device, files.ModelFile(bad), sh, files.MaterialFile(rmaterial, sh), [])
    assert "invalid inputlayout" in str(e.value)
    # a handle that names a non-layout object: unreachable!() in the reference
    bad = mt_files.write_rmodel(md, [mt_files.handle_of("RSMesh")], ["mat_0"], [0])
    with pytest.raises(api.MtrError):
        files.model_from_files(gpu_device, files.ModelFile(bad), sh, None, [])


def test_material_states_and_mip_chains_from_files(gpu_device):
    """row f-4 end to end over files: the materials name state objects (BSAddAlpha, DSZTest, RSMeshCN ...), the textures carry
    mip chains behind their offset tables; states applied by name and chains uploaded give the oracle's pixels"""
    rng = np.random.default_rng(4)

    def chain(w, h, levels, seed):
        data, lw, lh = b"", w, h
        for l in range(levels):
            img = np.random.default_rng(seed + l).integers(0, 256, size=(lh, lw, 4), dtype=np.uint8)
            img[..., 3] = np.where(img[..., 3] > 128, 255, 100)
            data += img.tobytes()
            lw, lh = max(1, lw >> 1), max(1, lh >> 1)
        return scene.TextureData(w, h, scene.TEX_RGBA8, data, levels=levels)
    tex = [chain(64, 64, 5, 1), chain(32, 16, 4, 9)]
    prims = []
    for i in range(10):
        x0, y0 = rng.uniform(0, 30, size=2)
        s = rng.uniform(4, 30)
        z = float(rng.integers(1, 15)) / 16
        v = [(x0, y0, z, 0, 0), (x0, y0 + s, z, 0, 1), (x0 + s, y0 + s, z, 1, 1), (x0 + s, y0, z, 1, 0)]
        prims.append(dict(verts=v, indices=[0, 1, 2, 0, 2, 3] if i % 3 else [0, 2, 1, 0, 3, 2], texture=i % 2, debug_id=i))
    md = pixel_model(prims, textures=tex)
    md.prim_states = np.array([(rng.integers(0, 3), rng.integers(0, 2), rng.integers(0, 2), rng.integers(0, 3)) for _ in prims], dtype=np.uint8)
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)

    def make(dev):
        sh = files.Shader2File(rshader2)
        mat = files.MaterialFile(rmaterial, sh)
        mf = files.ModelFile(rmodel)
        t = [files.TextureFile(b).upload(dev, max_levels=8) for b in rtextures]
        m = files.model_from_files(dev, mf, sh, mat, t)
        m.set_prim_states(files.states_from_files(mf, sh, mat))
        return m
    M = pixel_to_ndc_matrix(64, 64)
    ref = render_oracle(64, 64, [dict(md=md, M=M)])
    got = render_gpu(gpu_device, 64, 64, [dict(md=md, M=M, make_model=make)])
    assert_same(got, ref, "states + mips from files")


def _skeleton(jn):
    """jn joints in a chain at exactly representable offsets; imat = inverse bind world matrix"""
    offs = [(0.0, 0.03125 * (j % 7) - 0.0625, 0.015625 * j) for j in range(jn)]
    lm, im, acc = [], [], np.zeros(3)
    for o in offs:
        acc = acc + np.array(o)
        t = np.eye(4); t[:3, 3] = o
        ti = np.eye(4); ti[:3, 3] = -acc
        lm.append(scene.to_f32_colmajor(t)); im.append(scene.to_f32_colmajor(ti))
    joints = [(j, (j - 1) if j else 255, offs[j]) for j in range(jn)]
    return joints, np.stack(lm), np.stack(im)


def test_bind_pose_palette_from_the_file_skins_like_no_palette_at_all(gpu_device):
    """row f-3 KAT: the palette formed from lmats / imats in bind pose is the identity, so the skinned render is bit-identical
    to the unskinned one; and the per-joint cubes of Model::render (src/model.rs:309-315) come out of the file's joint offsets"""
    W, H = 320, 200
    md = scene.mesh50k(rows=12, cols=20)
    # rigid weights (255, 0, 0, 0): w_0 = 1 exactly, so an identity palette reproduces the position bit for bit (blended
    # weights b/255 sum to 1 only up to rounding, and then neither does the skinned position)
    vb = md.vertex_buf.reshape(-1, 24).copy()
    vb[:, 20:24] = (255, 0, 0, 0)
    md.vertex_buf = vb.reshape(-1)
    joints, lm, im = _skeleton(64)
    rmodel, rshader2, rmaterial, _ = mt_files.files_from_model_data(md)
    names = [f"mat_{p}" for p in range(md.nprims)]
    rmodel = mt_files.write_rmodel(md, [mt_files.handle_of("IATest0", low=p) for p in range(md.nprims)], names, list(range(md.nprims)),
                                   joints=joints, lmats=lm, imats=im)
    sh = files.Shader2File(rshader2)
    mat = files.MaterialFile(rmaterial, sh)
    mf = files.ModelFile(rmodel)
    pal = mf.palette()
    assert (pal == np.eye(4, dtype=np.float32).reshape(16)).all()
    M = scene.to_f32_colmajor(scene.headline_transform(W, H))
    model = files.model_from_files(gpu_device, mf, sh, mat, [])
    out = {}
    for key, p in (("skinned", pal), ("unskinned", None)):
        model.set_palette(p)
        fr = api.Frame(gpu_device, W, H); model.render(fr, M); fr.end()
        out[key] = (fr.color(), fr.depth(), fr.stats()); fr.close()
    assert (out["skinned"][0] == out["unskinned"][0]).all() and (out["skinned"][1].view(np.uint32) == out["unskinned"][1].view(np.uint32)).all()
    assert_same(out["unskinned"], render_oracle(W, H, [dict(md=md, M=M, palette=None)]), "unskinned from files")
    # joint cubes: scale 0.005 at offset * 0.01, after the model, through the overlay path
    model.set_palette(scene.bone_palette())
    cam = scene.to_f32_colmajor(scene.reference_view_proj(W, H) @ scene.mat_translate(-5.0, 0.0, 1.0 - 0.06))
    fr = api.Frame(gpu_device, W, H); model.render(fr, cam, joints=True); fr.end()
    got = (fr.color(), fr.depth(), fr.stats()); fr.close()
    cubes = np.zeros((64, 16), dtype=np.float32)
    for j, (_no, _parent, off) in enumerate(joints):
        cubes[j, 0] = cubes[j, 5] = cubes[j, 10] = np.float32(0.005)
        cubes[j, 12:15] = np.array(off, dtype=np.float32) * np.float32(0.01)
        cubes[j, 15] = 1.0
    ref = render_oracle(W, H, [dict(md=md, M=cam, palette=scene.bone_palette()), dict(md=md, vp=cam, overlay=cubes)])
    assert_same(got, ref, "model + joint cubes")
    assert got[2]["tris_in"] == md.input_triangles() + 64 * 12
    model.close(); mat.close(); sh.close()


def test_sdl_tracks_drive_parts_disp_and_instances_of_gpu_frames(gpu_device):
    """row f-2 end to end: a .sdl image (src/rscheduler.rs:119-210: BOOL / U32 / MATRIX / VECTOR / FLOAT tracks) is read,
    its tracks are bound to parts_disp entries and to the instances of a batch, evaluated at three frame numbers
    (mtr_rscheduler_apply), and what comes out drives Model::set_parts_disp and the batch of the GPU frame.  Every frame
    equals the oracle's render of the same evaluated arrays; the three frames differ the way the tracks say."""
    import copy
    W, H = 480, 270
    md = scene.skinned_capsule_model([((-0.45, 0.0, 0.0), 0.2, 1.2), ((0.0, 0.0, 0.0), 0.2, 1.2), ((0.45, 0.0, 0.0), 0.2, 1.2)], rows=16, cols=24)
    w = md.prims.view(np.uint32).reshape(-1, 14)
    w[:, 1] = (w[:, 1] & ~np.uint32(0xFFF)) | np.arange(3, dtype=np.uint32)  # one part per capsule (PrimitiveInfo::parts_no)
    base, pals = scene.instance_lattice(3, 1)
    base = base.astype(np.float32)
    moved = base[1].copy(); moved[12] += np.float32(0.3); moved[13] -= np.float32(0.2); moved[0] *= np.float32(0.8)
    p0 = tuple(float(v) for v in base[0, 12:15]) + (0.0,)
    p1 = (p0[0] - 0.25, p0[1] + 0.3, p0[2], 0.0)
    tracks = [dict(type=1, name="root"),
              dict(type=11, prop=files.PROP_BOOL, name="PartsDisp1", keys=[(0, 0, True), (10, 0, False), (20, 0, True)]),
              dict(type=6, prop=files.PROP_U32, name="PartsDisp2", keys=[(15, 0, 0)]),
              dict(type=16, prop=22, name="Inst1Matrix", keys=[(0, 0, tuple(float(v) for v in base[1])), (12, 0, tuple(float(v) for v in moved))]),
              dict(type=8, prop=21, name="Inst0Pos", keys=[(0, 0, p0), (18, 0, p1)]),
              dict(type=9, prop=files.PROP_F32, name="Inst2Y", keys=[(5, 0, float(base[2, 13]) + 0.35)])]
    sf = files.SchedulerFile(mt_files.write_rscheduler(tracks))
    T = files
    bindings = [(sf.find_track("PartsDisp1"), T.SDL_PARTS_DISP, 1), (sf.find_track("PartsDisp2"), T.SDL_PARTS_DISP, 2),
                (sf.find_track("Inst1Matrix"), T.SDL_INSTANCE_MATRIX, 1), (sf.find_track("Inst0Pos"), T.SDL_INSTANCE_TRANSLATION, 0),
                (sf.find_track("Inst2Y"), T.SDL_INSTANCE_TRANSLATE_Y, 2)]
    vp = scene.to_f32_colmajor(scene.reference_view_proj(W, H))
    model = api.Model.new(gpu_device, md)
    seen = []
    try:
        for frame_no, want_pd in ((0, [1, 1, 1]), (13, [1, 0, 1]), (25, [1, 1, 0])):
            pd = np.ones(3, dtype=np.uint8)
            mm = base.copy()
            sf.apply(frame_no, bindings, pd, mm)
            assert list(pd) == want_pd, frame_no
            model.set_parts_disp(pd)
            batch = api.Batch(gpu_device, model, mm, pals)
            fr = api.Frame(gpu_device, W, H)
            fr.draw_batch(batch, vp)
            fr.end()
            got = (fr.color(), fr.depth(), fr.stats())
            fr.close(); batch.close()
            md_f = copy.copy(md)
            md_f.parts_disp = pd.copy()
            assert_same(got, render_oracle(W, H, [dict(md=md_f, vp=vp, model_mats=mm, palettes=pals)]), f".sdl frame {frame_no}")
            assert got[2]["tris_in"] == 3 * int(pd.sum()) * (md.input_triangles() // 3)
            seen.append((mm.copy(), got[0]))
        assert (seen[0][0][1] == base[1]).all() and (seen[1][0][1] == moved).all()            # MATRIX key at 12
        assert tuple(seen[1][0][0, 12:15]) == p0[:3] and tuple(seen[2][0][0, 12:15]) == tuple(np.float32(v) for v in p1[:3])  # VECTOR key at 18
        assert seen[0][0][2, 13] == base[2, 13] and seen[1][0][2, 13] == np.float32(float(base[2, 13]) + 0.35)  # FLOAT key at 5
        assert (seen[0][1] != seen[1][1]).any() and (seen[1][1] != seen[2][1]).any()
    finally:
        model.close()
        sf.close()
