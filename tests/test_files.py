"""CPU: the resource-file readers of include/mtr_files.h against the reference's own
numbers -- struct sizes (src/rmodel.rs:487-494, src/rtexture.rs:171, src/rshader2.rs:573-582, src/rmaterial.rs:317-322,
src/rscheduler.rs:222), the crc32 / DTI-hash known answers, the bit-field accessors -- and against synthetic files
written by tests/mt_files.py (the reference ships no asset files: full-file parsing is "parity unpinned")."""
import os
import re
import struct

import numpy as np
import pytest

from mt_renderer_amd import api, files, scene
from tests import mt_files
from tests.pixel_scenes import pixel_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mtr_files.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mtr_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(files.EXPORTED_SYMBOLS)
    for n in names:
        assert hasattr(api.lib, n), n


def test_struct_sizes_match_the_reference_size_tests():
    want = dict(ModelHdr=0xA0, PrimitiveInfo=0x38, PartsInfo=0x20, BoundaryInfo=0x90, JointInfo=24, MtMatrix=64,
                TextureHeader=0x10, Shader2Header=0x20, RawShader2Object=0x28, RawShader2InputElement=0x10,
                RawShader2InputLayout=16, RawShader2Struct=16, RawShader2Variable=0x30, RawShader2CBuffer=24,
                MaterialHeader=0x28, RawTextureInfo=0x98, RawMaterialInfo=0x48, RawMaterialState=0x18, SchedulerTrack=0x30,
                SchedulerHeader=0x20, ArchiveHeader=8, RawResourceInfo=0x90)
    for k, v in want.items():
        assert files.struct_size(k) == v, k
    assert api.lib.mtr_file_struct_size(99) == 0


def test_crc_and_dti_known_answers():
    assert mt_files.crc32_mt(b"MtObject") == 0x2EA10CEB == api.crc32(b"MtObject")  # src/util/crc.rs:55
    assert mt_files.RTEXTURE_DTI == 606035435  # src/dti.txt "rTexture", rule src/dti.rs:174
    assert api.crc32(b"rTexture") & 0x7FFFFFFF == 606035435


def _model():
    tex = [scene.checker_rgba8_texture(16, 16), scene.random_bc1_texture(8, 8)]
    prims = [dict(verts=[(1, 1, .5, 0, 0), (1, 30, .5, 0, 1), (30, 30, .5, 1, 1)], indices=[0, 1, 2], texture=1, debug_id=7),
             dict(verts=[(2, 2, .25), (2, 20, .25), (20, 2, .25), (20, 20, .25)], indices=[0, 1, 2, 3], topology=scene.TOPO_STRIP,
                  debug_id=3, parts_no=1),
             dict(verts=[(5, 5, .75, .5, .5), (5, 25, .75, .5, 1), (25, 25, .75, 1, 1)], indices=[0, 1, 2], texture=0, debug_id=11)]
    return pixel_model(prims, textures=tex)


def test_rmodel_roundtrip_and_accessors():
    md = _model()
    rmodel, _, _, _ = mt_files.files_from_model_data(md)
    mf = files.ModelFile(rmodel)
    assert mf.v.magic == 0x444F4D and mf.v.primitive_num == 3 and mf.v.boundary_num == 3 and mf.v.material_num == 3
    assert mf.material_names() == ["mat_0", "mat_1", "mat_2"]
    assert np.array_equal(mf.vertex_buf(), md.vertex_buf) and np.array_equal(mf.index_buf(), md.index_buf)
    for p in range(3):
        f = scene.unpack_primitive(md.prims[p])
        assert mf.primitive_field(p, files.PRIM_VERTEX_NUM) == f["vertex_num"]
        assert mf.primitive_field(p, files.PRIM_VERTEX_STRIDE) == f["vertex_stride"]
        assert mf.primitive_field(p, files.PRIM_TOPOLOGY) == f["topology"]
        assert mf.primitive_field(p, files.PRIM_PARTS_NO) == f["parts_no"]
        assert mf.primitive_field(p, files.PRIM_INDEX_OFS) == f["index_ofs"] and mf.primitive_field(p, files.PRIM_INDEX_NUM) == f["index_num"]
        assert mf.primitive_field(p, files.PRIM_VERTEX_BASE) == f["vertex_base"] and mf.primitive_field(p, files.PRIM_INDEX_BASE) == f["index_base"]
        assert mf.primitive_field(p, files.PRIM_MATERIAL_NO) == p and mf.primitive_field(p, files.PRIM_BOUNDARY_NUM) == p
        assert mf.boundary_joint(p) == int(md.prim_debug_id[p])
    assert (mf.joint_table() == 255).all() and mf.lmats().shape == (0, 16)  # no joints: src/rmodel.rs:415-421


def test_rmodel_joints_block():
    md = _model()
    jn = 5
    lm = np.arange(jn * 16, dtype=np.float32).reshape(jn, 16)
    im = -lm
    joints = [(i, max(i - 1, 0) if i else 255, (0.5 * i, 1.0, -2.0)) for i in range(jn)]
    rmodel = mt_files.write_rmodel(md, [0, 0, 0], ["a", "b", "c"], [0, 1, 2], joints=joints, lmats=lm, imats=im)
    mf = files.ModelFile(rmodel)
    assert mf.v.jnt_num == jn
    assert np.array_equal(mf.lmats(), lm) and np.array_equal(mf.imats(), im)
    assert mf.joint(3) == dict(no=3, parent=2, symmetry=255, offset=(1.5, 1.0, -2.0))
    assert list(mf.joint_table()[:6]) == [0, 1, 2, 3, 4, 255]
    with pytest.raises(api.MtrError):
        mf.joint(jn)


def test_rmodel_rejects_what_the_reference_panics_on():
    md = _model()
    good = mt_files.files_from_model_data(md)[0]
    for cut in (0, 0x50, 0xA2, len(good) // 2, len(good) - 1):  # truncated anywhere
        with pytest.raises(api.MtrError) as e:
            files.ModelFile(good[:cut])
        assert e.value.code == api.MTR_E_INVALID
    bad = bytearray(good)
    struct.pack_into("<Q", bad, 0x48, 0xFFFFFFFFFFFFFFF0)  # vertex_data offset far outside
    with pytest.raises(api.MtrError):
        files.ModelFile(bytes(bad))
    bad = bytearray(good)
    struct.pack_into("<I", bad, 0x10, 0x7FFFFFFF)  # index_num huge
    with pytest.raises(api.MtrError):
        files.ModelFile(bytes(bad))
    # a primitive naming a material / boundary the file does not have (Model::new would index out of bounds)
    with pytest.raises(api.MtrError):
        files.ModelFile(mt_files.write_rmodel(md, [0, 0, 0], ["only"], [0, 1, 0]))


def test_rtexture_header_bitfields_and_errors():
    t = scene.random_bc7_texture(64, 32, seed=3)
    tf = files.TextureFile(mt_files.write_rtexture(64, 32, t.fmt, t.data))
    assert (tf.width(), tf.height(), tf.format()) == (64, 32, scene.TEX_BC7) and tf.data() == t.data
    assert tf.v.level_count == 1 and tf.v.array_count == 1 and tf.v.type == 2 and tf.v.version == 0x9D
    # prebias shifts the stored size back up (src/rtexture.rs:57-62)
    tf = files.TextureFile(mt_files.write_rtexture(64, 32, 7, bytes(64 * 32 * 4), prebias=2))
    assert (tf.width(), tf.height(), tf.v.prebias) == (64, 32, 2)
    # several images: only offsets[0] is used, and the data runs to the end of the file
    tf = files.TextureFile(mt_files.write_rtexture(8, 8, 7, bytes(range(256)) + b"tail", level_count=4))
    assert tf.v.level0_offset == 16 + 8 * 4 and tf.data().endswith(b"tail")
    for bad in (mt_files.write_rtexture(8, 8, 7, bytes(256), magic=b"XET\0"),   # assert_eq!(magic), src/rtexture.rs:105
                mt_files.write_rtexture(8, 8, 7, bytes(256), tex_type=6),        # assert_eq!(TT_2D), :106
                mt_files.write_rtexture(8, 8, 7, bytes(256), tex_type=12),       # from_repr(...).unwrap(), :66
                mt_files.write_rtexture(8, 8, 7, bytes(256), level_count=0),     # unk_offsets[0], :126
                mt_files.write_rtexture(8, 8, 7, bytes(256))[:12]):
        with pytest.raises(api.MtrError) as e:
            files.TextureFile(bad)
        assert e.value.code == api.MTR_E_INVALID


def test_rshader2_objects_handles_and_layouts():
    md = _model()
    _, rshader2, _, _ = mt_files.files_from_model_data(md)
    sh = files.Shader2File(rshader2)
    objs = sh.objects()
    assert [o["name"] for o in objs][:6] == ["BSSolid", "DSZTestWrite", "RSMesh", "tAlbedoMap", "SSLinear", "CBMaterial"]
    for i, o in enumerate(objs):
        assert o["name_hash"] == mt_files.crc32_mt(o["name"].encode()) & 0xFFFFF  # src/rshader2.rs:343
        for low in (0, 0x5A5, 0xFFF):  # the low 12 bits of a handle are not part of the lookup (src/rshader2.rs:487-492)
            assert sh.get_object_by_handle(mt_files.handle_of(o["name"], low)) == i
    assert sh.get_object_by_handle(0xDEAD0000) is None
    lay_objs = [i for i, o in enumerate(objs) if o["obj_type"] == 9]
    assert len(lay_objs) == 2  # position-only and position+texcoord
    seen = set()
    for i in lay_objs:
        L = sh.input_layout(i)
        names = [e["name"] for e in L["elements"]]
        assert names[0] == "Normal" and names[-1] == "Tangent" and "Position" in names
        # bound elements: Position / TexCoord only; Normal is not a bound name, the SCMP3N Tangent is skipped
        assert all(sem in (scene.SEM_POSITION, scene.SEM_TEXCOORD) for (sem, _, _, _) in L["bound"])
        seen.add((tuple(L["bound"]), L["stride"]))
    want = {(tuple(md.layouts[p]), scene.unpack_primitive(md.prims[p])["vertex_stride"]) for p in range(md.nprims)}
    assert seen == want
    with pytest.raises(api.MtrError):
        sh.input_layout(0)  # "primitive inputlayout isn't an inputlayout!", src/model.rs:190
    with pytest.raises(api.MtrError):
        files.Shader2File(mt_files.write_rshader2([dict(name="x", obj_type=4)], magic=0x123))  # bad magic, src/rshader2.rs:307
    with pytest.raises(api.MtrError):
        files.Shader2File(mt_files.write_rshader2([dict(name="x", obj_type=40)]))  # from_repr(...).expect, :363
    with pytest.raises(api.MtrError):
        files.Shader2File(mt_files.write_rshader2([dict(name="same", obj_type=4), dict(name="same", obj_type=5)]))  # collision assert
    for cut in (10, 0x28, len(rshader2) - 40):
        with pytest.raises(api.MtrError):
            files.Shader2File(rshader2[:cut])


def test_rmaterial_albedo_binding_and_errors():
    md = _model()
    _, rshader2, rmaterial, _ = mt_files.files_from_model_data(md)
    sh = files.Shader2File(rshader2)
    mf = files.MaterialFile(rmaterial, sh)
    assert mf.textures() == ["model\\tex\\t0_BM", "model\\tex\\t1_BM"]
    mats = mf.materials()
    assert [m["albedo_texture_idx"] for m in mats] == [1, None, 0]  # prim_to_texture of _model()
    assert all(m["state_num"] == (3 if m["albedo_texture_idx"] is not None else 2) for m in mats)
    assert mf.material_by_name("mat_2") == 2 and mf.material_by_name("nope") is None
    assert mats[0]["name_hash"] == mt_files.crc32_mt(b"mat_0")  # full 32 bits (src/rmaterial.rs:304-311)
    assert mats[0]["blend_factor"] == (1.0, 1.0, 1.0, 1.0)
    base = [dict(name="m", albedo=1)]
    with pytest.raises(api.MtrError):  # texture class must be rTexture (assert, src/rmaterial.rs:194)
        files.MaterialFile(mt_files.write_rmaterial(["a"], base, texture_dti=1234), sh)
    with pytest.raises(api.MtrError):  # texture index beyond the list (textures[...] would panic)
        files.MaterialFile(mt_files.write_rmaterial(["a"], [dict(name="m", albedo=5)]), sh)
    with pytest.raises(api.MtrError):  # a state object the package lacks (get_object_by_handle(...).unwrap())
        files.MaterialFile(mt_files.write_rmaterial(["a"], [dict(name="m", albedo=1, bs="BSMissing")]), sh)
    with pytest.raises(api.MtrError):
        files.MaterialFile(mt_files.write_rmaterial(["a"], [dict(name="m", extra_states=[(7, "SSLinear", 0)])]), sh)  # bad state type
    # STATE_TEXTURE with value 0 is only a warning in the reference (src/rmaterial.rs:266-267)
    ok = files.MaterialFile(mt_files.write_rmaterial(["a"], [dict(name="m", extra_states=[(3, "tAlbedoMap", 0)])]), sh)
    assert ok.materials()[0]["albedo_texture_idx"] is None
    for cut in (8, 0x30, len(rmaterial) - 8):
        with pytest.raises(api.MtrError):
            files.MaterialFile(rmaterial[:cut], sh)


def test_rscheduler_tracks_keys_and_eval():
    tracks = [dict(type=1, name="root"),
              dict(type=2, name="uModel", field_10=mt_files.crc32_mt(b"uModel") & 0x7FFFFFFF),
              dict(type=11, prop=files.PROP_BOOL, name="mPartsDisp[3]", parent=1, keys=[(0, 1, True), (30, 1, False), (45, 2, True)]),
              dict(type=9, prop=files.PROP_F32, name="mScale", parent=1, keys=[(10, 0, 1.5), (20, 3, -2.25)]),
              dict(type=6, prop=files.PROP_U32, name="mMotionNo", parent=1, keys=[(0, 0, 7), (5, 0, 0xFFFFFFFE)]),
              dict(type=13, prop=2, name="mpModel", parent=1, keys=[(0, 0, None), (12, 0, (606035435, "chr\\pl\\pl0000_BM"))]),
              dict(type=8, prop=21, name="mPos", parent=1, keys=[(0, 0, (1.0, 2.0, 3.0, 0.0))])]
    sf = files.SchedulerFile(mt_files.write_rscheduler(tracks))
    ts = sf.tracks()
    assert [t["name"] for t in ts] == [t["name"] for t in tracks]
    assert [t["key_num"] for t in ts] == [0, 0, 3, 2, 2, 2, 1]
    assert ts[1]["dti_or_prop"] == mt_files.crc32_mt(b"uModel") & 0x7FFFFFFF and ts[2]["parent"] == 1
    assert sf.key(2, 1) == dict(frame=30, mode=1, value_bits=0, resource=None)
    assert sf.key(2, 2)["mode"] == 2 and sf.key(4, 1)["value_bits"] == 0xFFFFFFFE
    assert sf.key(5, 0)["resource"] is None and sf.key(5, 1) == dict(frame=12, mode=0, value_bits=606035435, resource="chr\\pl\\pl0000_BM")
    # step hold (this build's rule): the last key at or before the frame
    assert [sf.eval(2, f) for f in (0, 29, 30, 44, 45, 1000)] == [1, 1, 0, 0, 1, 1]
    assert sf.eval_float(3, 10) == 1.5 and sf.eval_float(3, 19) == 1.5 and sf.eval_float(3, 20) == -2.25
    with pytest.raises(api.MtrError):
        sf.eval(3, 9)  # before the first key
    with pytest.raises(api.MtrError) as e:
        sf.key(6, 0)  # VECTOR keys: todo!() in the reference
    assert e.value.code == api.MTR_E_UNSUPPORTED
    for bad in (mt_files.write_rscheduler(tracks, version=0x15), mt_files.write_rscheduler(tracks, magic=b"LDS\0"),
                mt_files.write_rscheduler([dict(type=4, name="nested")]), mt_files.write_rscheduler([dict(type=77, name="x")]),
                mt_files.write_rscheduler([dict(type=11, prop=files.PROP_F32, name="b", keys=[(0, 0, True)])]),  # assert_eq!(prop_type, bool)
                mt_files.write_rscheduler(tracks)[:0x60]):
        with pytest.raises(api.MtrError):
            files.SchedulerFile(bad)


def test_cpp_mirror_reads_the_same_files(tmp_path):
    """include/mtr_files.hpp (C++ host mirror): compile a small reader against libmtr.so and run it on synthetic files."""
    import subprocess
    md = _model()
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)
    sdl = mt_files.write_rscheduler([dict(type=1, name="root"), dict(type=2, name="uModel"),
                                     dict(type=11, prop=files.PROP_BOOL, name="mDisp", keys=[(0, 0, True), (30, 0, False)])])
    paths = []
    for name, data in (("m.mod", rmodel), ("s.mfx", rshader2), ("m.mrl", rmaterial), ("t.tex", rtextures[1]), ("s.sdl", sdl)):
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    exe = str(tmp_path / "files_demo")
    lib_dir = os.path.join(ROOT, "mt_renderer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "files_demo.cpp"), "-o", exe, "-L", lib_dir, "-lmtr", "-Wl,-rpath," + lib_dir])
    r = subprocess.run([exe] + paths, capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    out = r.stdout.splitlines()
    assert out[0] == "model prims=3 materials=3 boundaries=3"
    assert out[1].startswith("prim 0 layout=IATest") and "material=mat_0 albedo=1 joint=7" in out[1] and "bound=2" in out[1]
    assert "material=mat_1 albedo=-1 joint=3" in out[2] and "bound=1" in out[2]
    assert "albedo=0 joint=11" in out[3]
    assert out[4] == "texture 8x8 format=19"
    assert out[-1] == "eval bool@31=0"
    bad = tmp_path / "bad.mod"
    bad.write_bytes(rmodel[:100])
    r = subprocess.run([exe, str(bad)] + paths[1:], capture_output=True, text=True)
    assert r.returncode == 2 and "rModel" in r.stderr


def test_rarchive_table_lookup_and_inflate():
    md = _model()
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)
    h = lambda name: mt_files.crc32_mt(name.encode()) & 0x7FFFFFFF
    res = [("chr\\pl\\pl0000", h("rModel"), rmodel), ("chr\\pl\\pl0000", h("rMaterial"), rmaterial),
           ("model\\tex\\t0_BM", h("rTexture"), rtextures[0]), ("empty", h("rTexture"), b"")]
    arc = mt_files.write_rarchive(res)
    af = files.ArchiveFile(arc)
    infos = af.resource_infos()
    assert [i["path"] for i in infos] == [r[0] for r in res] and [i["dti_hash"] for i in infos] == [r[1] for r in res]
    assert [i["size_uncompressed"] for i in infos] == [len(r[2]) for r in res] and all(i["quality"] == 2 for i in infos)
    for i, r in enumerate(res):
        assert af.extract(i) == r[2]
    # same path, two classes: the class hash picks (src/rarchive.rs:150-153); '/' is accepted for the separator
    assert af.get_resource("chr/pl/pl0000", h("rMaterial")) == rmaterial and af.get_resource("chr\\pl\\pl0000", h("rModel")) == rmodel
    assert af.get_resource("chr/pl/pl0000", h("rTexture")) is None and af.get_resource("nope", h("rModel")) is None
    # and the extracted bytes parse: archive -> rModel
    assert files.ModelFile(af.get_resource("chr/pl/pl0000", h("rModel"))).v.primitive_num == md.nprims
    for bad in (mt_files.write_rarchive(res, magic=b"CRA\0"), mt_files.write_rarchive(res, version=8), arc[:8 + 0x90 * 2]):
        with pytest.raises(api.MtrError):
            files.ArchiveFile(bad)
    with pytest.raises(api.MtrError):  # assert_eq!(num_decompressed_bytes, size_uncompressed), src/rarchive.rs:173
        files.ArchiveFile(mt_files.write_rarchive(res, lie_about_size=5)).extract(0)
    with pytest.raises(api.MtrError):  # a stream cut short
        files.ArchiveFile(arc[:-3]).extract(3)
    corrupt = bytearray(arc)
    corrupt[8 + 0x90 * 4 + 20] ^= 0xFF  # inside the first zlib stream
    with pytest.raises(api.MtrError):
        files.ArchiveFile(bytes(corrupt)).extract(0)


def test_parsers_survive_mutated_files():
    """Robustness: random byte flips, field overwrites with huge values and truncations of valid files must end in a
    normal return or an MtrError -- never a crash (every offset and count read from a file is bounds-checked).  A
    parse that succeeds is also walked through its accessors."""
    md = _model()
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)
    sdl = mt_files.write_rscheduler([dict(type=1, name="root"), dict(type=11, prop=files.PROP_BOOL, name="b", keys=[(0, 0, True), (9, 1, False)]),
                                     dict(type=9, prop=files.PROP_F32, name="f", keys=[(0, 0, 1.0)]),
                                     dict(type=13, prop=2, name="r", keys=[(0, 0, (5, "a\\b")), (1, 0, None)])])
    sh_ok = files.Shader2File(rshader2)
    rng = np.random.default_rng(1234)

    def mutate(b: bytes) -> bytes:
        a = bytearray(b)
        kind = rng.integers(0, 4)
        if kind == 0:
            for _ in range(int(rng.integers(1, 6))):
                a[int(rng.integers(0, len(a)))] = int(rng.integers(0, 256))
        elif kind == 1:  # a 32/64-bit field becomes huge
            off = int(rng.integers(0, max(1, min(len(a), 0x100) - 8))) & ~3
            big = [0xFFFFFFFFFFFFFFFF, 0x7FFFFFFF, len(a), len(a) - 1, 0x80000000][int(rng.integers(0, 5))]
            a[off:off + 8] = struct.pack("<Q", big)
        elif kind == 2:
            a = a[:int(rng.integers(0, len(a)))]
        else:
            off = int(rng.integers(0, len(a)))
            a[off:off + 4] = bytes(4)
        return bytes(a)

    def walk_model(b):
        mf = files.ModelFile(b)
        mf.material_names(); mf.vertex_buf(); mf.index_buf(); mf.joint_table()
        for p in range(mf.v.primitive_num):
            mf.boundary_joint(mf.primitive_field(p, files.PRIM_BOUNDARY_NUM))

    def walk_shader(b):
        sh = files.Shader2File(b)
        for i, o in enumerate(sh.objects()):
            if o["obj_type"] == 9:
                sh.input_layout(i)

    def walk_material(b):
        m = files.MaterialFile(b, sh_ok)
        m.textures(); m.materials()

    def walk_sdl(b):
        s = files.SchedulerFile(b)
        for i, t in enumerate(s.tracks()):
            for k in range(t["key_num"]):
                try:
                    s.key(i, k)
                except api.MtrError:
                    pass

    outcomes = {"ok": 0, "err": 0}
    for good, walk in ((rmodel, walk_model), (rshader2, walk_shader), (rmaterial, walk_material),
                       (rtextures[0], lambda b: files.TextureFile(b).data()), (sdl, walk_sdl)):
        walk(good)
        for _ in range(400):
            try:
                walk(mutate(good))
                outcomes["ok"] += 1
            except api.MtrError:
                outcomes["err"] += 1
    assert outcomes["err"] > 300 and outcomes["ok"] > 100, outcomes


def test_parsers_under_address_sanitizer(tmp_path):
    """The same kind of mutations, 30 000 of them, against an AddressSanitizer + UBSan build of csrc/mtr_files.cpp
    (host-only build with stand-ins for the device entry points; sanitizers cannot run on the GPU box)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    md = _model()
    rmodel, rshader2, rmaterial, rtextures = mt_files.files_from_model_data(md)
    joints = [(i, max(i - 1, 0), (0.1 * i, 0.0, 1.0)) for i in range(4)]
    rmodel_j = mt_files.write_rmodel(md, [mt_files.handle_of("IATest0")] * 3, ["mat_0", "mat_1", "mat_2"], [0, 1, 2], joints=joints,
                                     lmats=np.zeros((4, 16), np.float32), imats=np.ones((4, 16), np.float32))
    sdl = mt_files.write_rscheduler([dict(type=1, name="root"), dict(type=11, prop=files.PROP_BOOL, name="b", keys=[(0, 0, True), (9, 1, False)]),
                                     dict(type=6, prop=files.PROP_U32, name="i", keys=[(0, 0, 3)]),
                                     dict(type=13, prop=2, name="r", keys=[(0, 0, (5, "a\\b")), (1, 0, None)])])
    paths = []
    arc = mt_files.write_rarchive([("a\\b", 1, rmodel), ("a\\c", 2, bytes(range(256)) * 9), ("e", 3, b"")])
    for name, data in (("m.mod", rmodel_j), ("s.mfx", rshader2), ("m.mrl", rmaterial), ("t.tex", rtextures[0]), ("s.sdl", sdl), ("a.arc", arc)):
        p = tmp_path / name
        p.write_bytes(data)
        paths.append(str(p))
    exe = str(tmp_path / "files_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "files_fuzz.cpp"), "-o", exe, "-lz"])
    r = subprocess.run([exe, "30000"] + paths, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    ok, err = [int(t.split("=")[1]) for t in r.stdout.split()]
    assert ok > 1000 and err > 5000, r.stdout


def test_state_object_names_map_to_material_states():
    """row f-4: MT Framework state-object names -> (blend, depth write, depth test, cull); a convention of this build"""
    A, OFF, ADD = 0, 1, 2
    cases = {("BSSolid", "DSZTestWrite", "RSMesh"): ((OFF, 1, 1, 0), 3),
             ("BSBlendAlpha", "DSZTest", "RSMeshCN"): ((A, 0, 1, 1), 3),
             ("BSAddAlpha", "DSZWrite", "RSMeshCF"): ((ADD, 1, 0, 2), 3),
             ("BSRevSubBlendAlpha", "DSZTestWriteStencilWrite", "RSMeshBias1"): ((A, 1, 1, 0), 3),
             ("BSBlendBlendAlpha", "DSZTestStencilWrite", "RSTwoSide"): ((A, 0, 1, 1), 3),
             (None, None, None): ((A, 1, 1, 0), 0),            # nothing named: the reference's pipeline state
             ("", "Whatever", "xx"): ((A, 1, 1, 0), 0),        # unknown names fall back to it too
             ("BSSolid", None, None): ((OFF, 1, 1, 0), 1)}
    for names, want in cases.items():
        assert files.state_from_names(*names) == want, names


def test_states_from_files_follow_material_names():
    from tests.pixel_scenes import pixel_model
    prims = [dict(verts=[(1, 1, .5), (1, 9, .5), (9, 9, .5)], indices=[0, 1, 2]) for _ in range(4)]
    md = pixel_model(prims)
    md.prim_states = np.array([(0, 1, 1, 0), (1, 0, 1, 1), (2, 1, 0, 2), (1, 0, 0, 0)], dtype=np.uint8)
    rmodel, rshader2, rmaterial, _ = mt_files.files_from_model_data(md)
    sh = files.Shader2File(rshader2)
    mat = files.MaterialFile(rmaterial, sh)
    got = files.states_from_files(files.ModelFile(rmodel), sh, mat)
    assert (got == md.prim_states).all(), got
    mat.close(); sh.close()


def _translate(x, y, z):
    m = np.eye(4, dtype=np.float64)
    m[:3, 3] = (x, y, z)
    return m


def _chain_skeleton(jn=6):
    """joint j hangs off joint j - 1 at an exactly representable offset; imat = inverse bind world matrix"""
    offs = [(0.5 * j, 0.25, -0.125 * j) for j in range(jn)]
    lm = np.stack([scene.to_f32_colmajor(_translate(*o)) for o in offs])
    world, acc = [], np.zeros(3)
    for o in offs:
        acc = acc + np.array(o)
        world.append(acc.copy())
    im = np.stack([scene.to_f32_colmajor(_translate(*(-w))) for w in world])
    joints = [(j + 10, (j - 1) if j else 255, offs[j]) for j in range(jn)]  # joint numbers differ from indices
    return joints, lm, im, world


def test_skeleton_palette_bind_pose_is_identity_and_animates():
    """row f-3: palette[j] = world_j * imat_j, world_j = world_parent * local_j (this build's rule; the reference parses the
    matrices and never combines them, src/rmodel.rs:392-413)"""
    md = _model()
    joints, lm, im, world = _chain_skeleton()
    mf = files.ModelFile(mt_files.write_rmodel(md, [0, 0, 0], ["a", "b", "c"], [0, 1, 2], joints=joints, lmats=lm, imats=im))
    pal = mf.palette()
    assert pal.shape == (6, 16) and (pal == np.eye(4, dtype=np.float32).reshape(16)).all(), "bind pose must skin with the identity, exactly"
    assert mf.joint_index(12) == 2 and mf.joint_index(10) == 0 and mf.joint_index(3) is None
    # rotate joint 2 about z by 90 degrees: joints 0, 1 stay put, joints 2.. swing around joint 2's position
    local = lm.copy().reshape(6, 4, 4)
    R = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    local[2] = (_translate(*joints[2][2]) @ R).T.astype(np.float32)  # column-major storage = transpose of the row-major array
    pal = mf.palette(local.reshape(6, 16)).reshape(6, 4, 4).transpose(0, 2, 1).astype(np.float64)
    assert np.allclose(pal[0], np.eye(4)) and np.allclose(pal[1], np.eye(4))
    for j in (2, 3, 5):
        # a point at joint j's bind position: stays for j = 2 (the pivot), moves on a quarter circle around it otherwise
        p = np.append(world[j], 1.0)
        rel = world[j] - world[2]
        want = world[2] + np.array([-rel[1], rel[0], rel[2]])
        assert np.allclose((pal[j] @ p)[:3], want, atol=1e-6), j
    # malformed skeletons
    bad = list(joints); bad[1] = (11, 4, joints[1][2]); bad[4] = (14, 1, joints[4][2])  # 1 -> 4 -> 1
    with pytest.raises(api.MtrError, match="ancestor"):
        files.ModelFile(mt_files.write_rmodel(md, [0, 0, 0], ["a", "b", "c"], [0, 1, 2], joints=bad, lmats=lm, imats=im)).palette()
    bad = list(joints); bad[3] = (13, 77, joints[3][2])
    with pytest.raises(api.MtrError, match="parent"):
        files.ModelFile(mt_files.write_rmodel(md, [0, 0, 0], ["a", "b", "c"], [0, 1, 2], joints=bad, lmats=lm, imats=im)).palette()
    with pytest.raises(api.MtrError, match="no joints"):
        files.ModelFile(mt_files.write_rmodel(md, [0, 0, 0], ["a", "b", "c"], [0, 1, 2])).palette()


def test_rscheduler_vector_matrix_keys_and_bindings():
    """row f-2: VECTOR / MATRIX keys (todo!() in the reference) and tracks bound to parts_disp / instance matrices"""
    m0 = tuple(float(i) for i in range(16))
    m1 = tuple(float(100 + i) for i in range(16))
    tracks = [dict(type=1, name="root"),
              dict(type=11, prop=files.PROP_BOOL, name="PartsDisp2", keys=[(0, 0, True), (10, 0, False)]),
              dict(type=6, prop=files.PROP_U32, name="PartsDisp0", keys=[(5, 0, 0)]),
              dict(type=16, prop=22, name="Inst1Matrix", keys=[(0, 0, m0), (20, 0, m1)]),
              dict(type=8, prop=21, name="Inst0Pos", keys=[(0, 0, (1.0, 2.0, 3.0, 9.0)), (15, 0, (-1.0, -2.0, -3.0, 9.0))]),
              dict(type=9, prop=files.PROP_F32, name="Inst2Y", keys=[(8, 0, 0.5)])]
    sf = files.SchedulerFile(mt_files.write_rscheduler(tracks))
    assert sf.find_track("Inst1Matrix") == 3 and sf.find_track("nope") is None
    assert tuple(sf.key_floats(3, 1)) == m1 and tuple(sf.key_floats(4, 0)) == (1.0, 2.0, 3.0, 9.0) and tuple(sf.key_floats(5, 0)) == (0.5,)
    assert tuple(sf.eval_floats(3, 19)) == m0 and tuple(sf.eval_floats(3, 20)) == m1
    with pytest.raises(api.MtrError):
        sf.key_floats(1, 0)  # a BOOL track has no float keys
    T = files
    bindings = [(1, T.SDL_PARTS_DISP, 2), (2, T.SDL_PARTS_DISP, 0), (3, T.SDL_INSTANCE_MATRIX, 1), (4, T.SDL_INSTANCE_TRANSLATION, 0),
                (5, T.SDL_INSTANCE_TRANSLATE_Y, 2)]
    for frame, want_pd, want_t0, want_y2, want_m1 in ((0, [1, 1, 1], (1, 2, 3), 0.0, m0), (7, [0, 1, 1], (1, 2, 3), 0.0, m0),
                                                       (12, [0, 1, 0], (1, 2, 3), 0.5, m0), (25, [0, 1, 0], (-1, -2, -3), 0.5, m1)):
        pd = np.ones(3, dtype=np.uint8)
        mm = np.tile(np.eye(4, dtype=np.float32).reshape(16), (3, 1))
        sf.apply(frame, bindings, pd, mm)
        assert list(pd) == want_pd, frame
        assert tuple(mm[0, 12:15]) == want_t0 and mm[0, 0] == 1.0 and mm[2, 13] == np.float32(want_y2) and tuple(mm[1]) == want_m1, frame
    pd = np.ones(3, dtype=np.uint8)
    mm = np.tile(np.eye(4, dtype=np.float32).reshape(16), (3, 1))
    for bad in ([(3, T.SDL_PARTS_DISP, 0)], [(1, T.SDL_INSTANCE_MATRIX, 0)], [(4, T.SDL_INSTANCE_TRANSLATE_X, 0)], [(1, T.SDL_PARTS_DISP, 3)],
                [(3, T.SDL_INSTANCE_MATRIX, 3)], [(99, T.SDL_PARTS_DISP, 0)], [(5, 17, 0)]):
        with pytest.raises(api.MtrError):
            sf.apply(0, bad, pd, mm)
    sf.close()
