"""ctypes binding of include/mtr_files.h: the MT Framework resource-file readers (rModel, rTexture, rShader2,
rMaterial, rScheduler) and Model::new over parsed files (src/model.rs:36-293).

Mirrors the reference's reader types by name -- ModelFile (src/rmodel.rs:295-484), TextureFile
(src/rtexture.rs:80-166), Shader2File (src/rshader2.rs:246-494), MaterialFile (src/rmaterial.rs:172-312),
SchedulerFile (src/rscheduler.rs:84-217) -- with the same accessors; a malformed file raises MtrError where the
reference panics.  Parsing is host-only (no GPU).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import api
from .api import MtrError, lib

EXPORTED_SYMBOLS = [
    "mtr_files_last_error", "mtr_file_struct_size", "mtr_rmodel_parse", "mtr_primitive_field", "mtr_rmodel_boundary_joint",
    "mtr_rmodel_joint", "mtr_rtexture_parse", "mtr_texture_create_from_file", "mtr_texture_create_from_file_mips",
    "mtr_state_from_names", "mtr_model_states_from_files", "mtr_rmodel_palette", "mtr_rmodel_joint_index", "mtr_rscheduler_key_floats",
    "mtr_rscheduler_eval_floats", "mtr_rscheduler_find_track", "mtr_rscheduler_apply", "mtr_rshader2_parse", "mtr_rshader2_destroy",
    "mtr_rshader2_num_objects", "mtr_rshader2_object", "mtr_rshader2_find", "mtr_rshader2_input_layout",
    "mtr_rmaterial_parse", "mtr_rmaterial_destroy", "mtr_rmaterial_num_textures", "mtr_rmaterial_texture_path",
    "mtr_rmaterial_num_materials", "mtr_rmaterial_info", "mtr_rmaterial_find", "mtr_rscheduler_parse",
    "mtr_rscheduler_destroy", "mtr_rscheduler_num_tracks", "mtr_rscheduler_track", "mtr_rscheduler_key",
    "mtr_rscheduler_eval", "mtr_model_create_from_files", "mtr_rarchive_parse", "mtr_rarchive_info", "mtr_rarchive_find",
    "mtr_rarchive_extract",
]

STRUCT_KINDS = ["ModelHdr", "PrimitiveInfo", "PartsInfo", "BoundaryInfo", "JointInfo", "MtMatrix", "TextureHeader",
                "Shader2Header", "RawShader2Object", "RawShader2InputElement", "RawShader2InputLayout", "RawShader2Struct",
                "RawShader2Variable", "RawShader2CBuffer", "MaterialHeader", "RawTextureInfo", "RawMaterialInfo",
                "RawMaterialState", "SchedulerTrack", "SchedulerHeader", "ArchiveHeader", "RawResourceInfo"]

(PRIM_VERTEX_NUM, PRIM_PARTS_NO, PRIM_MATERIAL_NO, PRIM_WEIGHT_NUM, PRIM_VERTEX_STRIDE, PRIM_TOPOLOGY, PRIM_VERTEX_OFS,
 PRIM_VERTEX_BASE, PRIM_INPUTLAYOUT, PRIM_INDEX_OFS, PRIM_INDEX_NUM, PRIM_INDEX_BASE, PRIM_BOUNDARY_NUM) = range(13)

# SchedulerTrackType (src/rscheduler.rs:15-33)
TRACK_ROOT, TRACK_UNIT, TRACK_SYSTEM, TRACK_OBJECT, TRACK_INT, TRACK_FLOAT, TRACK_BOOL, TRACK_RESOURCE = 1, 2, 3, 5, 6, 9, 11, 13
PROP_BOOL, PROP_U32, PROP_F32 = 3, 6, 12  # dti::PropType (src/dti.rs:6-70)


class _RModelView(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint16), ("jnt_num", C.c_uint16), ("primitive_num", C.c_uint16),
                ("material_num", C.c_uint16), ("vertex_num", C.c_uint32), ("index_num", C.c_uint32),
                ("polygon_num", C.c_uint32), ("vertexbuf_size", C.c_uint32), ("texture_num", C.c_uint32),
                ("parts_num", C.c_uint32), ("boundary_num", C.c_uint32), ("bounding_sphere", C.c_float * 4),
                ("bounding_box_min", C.c_float * 4), ("bounding_box_max", C.c_float * 4),
                ("material_names", C.c_void_p), ("primitives", C.c_void_p), ("boundary_infos", C.c_void_p),
                ("joint_infos", C.c_void_p), ("lmats", C.c_void_p), ("imats", C.c_void_p), ("joint_table", C.c_void_p),
                ("parts", C.c_void_p), ("vertex_buf", C.c_void_p), ("index_buf", C.c_void_p)]


class _RTextureView(C.Structure):
    _fields_ = [("version", C.c_uint32), ("prebias", C.c_uint32), ("type", C.c_uint32), ("level_count", C.c_uint32),
                ("array_count", C.c_uint32), ("format", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("level0_offset", C.c_uint64), ("data", C.c_void_p), ("data_len", C.c_size_t)]


class _RawElement(C.Structure):
    _fields_ = [("name", C.c_char_p), ("sindex", C.c_uint32), ("format", C.c_uint32), ("count", C.c_uint32),
                ("start", C.c_uint32), ("offset", C.c_uint32), ("instance", C.c_uint32)]


class _MaterialInfo(C.Structure):
    _fields_ = [("name_hash", C.c_uint32), ("dti_hash", C.c_uint32), ("albedo_texture", C.c_int32),
                ("bsstate", C.c_uint32), ("dsstate", C.c_uint32), ("rsstate", C.c_uint32), ("state_num", C.c_uint32),
                ("blend_factor", C.c_float * 4)]


class _TrackInfo(C.Structure):
    _fields_ = [("track_type", C.c_uint32), ("prop_type", C.c_uint32), ("key_num", C.c_uint32), ("parent", C.c_uint32),
                ("dti_or_prop", C.c_uint32), ("name", C.c_char_p)]


class _RArchiveView(C.Structure):
    _fields_ = [("num_resources", C.c_uint32), ("table", C.c_void_p), ("file", C.c_void_p), ("file_len", C.c_size_t)]


class _ResourceInfo(C.Structure):
    _fields_ = [("path", C.c_char_p), ("dti_hash", C.c_uint32), ("size_compressed", C.c_uint32),
                ("size_uncompressed", C.c_uint32), ("quality", C.c_uint32), ("offset", C.c_uint32)]


lib.mtr_rarchive_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_RArchiveView)]
lib.mtr_rarchive_info.argtypes = [C.POINTER(_RArchiveView), C.c_uint32, C.POINTER(_ResourceInfo)]
lib.mtr_rarchive_find.argtypes = [C.POINTER(_RArchiveView), C.c_char_p, C.c_uint32]
lib.mtr_rarchive_extract.argtypes = [C.POINTER(_RArchiveView), C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
lib.mtr_files_last_error.restype = C.c_char_p
lib.mtr_file_struct_size.restype = C.c_size_t
lib.mtr_file_struct_size.argtypes = [C.c_uint32]
lib.mtr_rmodel_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_RModelView)]
lib.mtr_primitive_field.restype = C.c_uint32
lib.mtr_primitive_field.argtypes = [C.c_void_p, C.c_uint32]
lib.mtr_rmodel_boundary_joint.argtypes = [C.POINTER(_RModelView), C.c_uint32, C.POINTER(C.c_uint32)]
lib.mtr_rmodel_joint.argtypes = [C.POINTER(_RModelView), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
lib.mtr_rtexture_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_RTextureView)]
lib.mtr_texture_create_from_file.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.mtr_texture_create_from_file_mips.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]
lib.mtr_rmodel_palette.argtypes = [C.POINTER(_RModelView), C.c_void_p, C.c_void_p, C.c_size_t]
lib.mtr_rmodel_joint_index.argtypes = [C.POINTER(_RModelView), C.c_uint32]
lib.mtr_rscheduler_key_floats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
lib.mtr_rscheduler_eval_floats.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
lib.mtr_rscheduler_find_track.argtypes = [C.c_void_p, C.c_char_p]
lib.mtr_rscheduler_apply.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
lib.mtr_state_from_names.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p]
lib.mtr_model_states_from_files.argtypes = [C.POINTER(_RModelView), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.mtr_rshader2_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.mtr_rshader2_destroy.restype = None
lib.mtr_rshader2_destroy.argtypes = [C.c_void_p]
lib.mtr_rshader2_num_objects.restype = C.c_uint32
lib.mtr_rshader2_num_objects.argtypes = [C.c_void_p]
lib.mtr_rshader2_object.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
lib.mtr_rshader2_find.argtypes = [C.c_void_p, C.c_uint32]
lib.mtr_rshader2_input_layout.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(api._Layout),
                                          C.POINTER(_RawElement), C.c_uint32, C.POINTER(C.c_uint32)]
lib.mtr_rmaterial_parse.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p)]
lib.mtr_rmaterial_destroy.restype = None
lib.mtr_rmaterial_destroy.argtypes = [C.c_void_p]
lib.mtr_rmaterial_num_textures.restype = C.c_uint32
lib.mtr_rmaterial_num_textures.argtypes = [C.c_void_p]
lib.mtr_rmaterial_texture_path.restype = C.c_char_p
lib.mtr_rmaterial_texture_path.argtypes = [C.c_void_p, C.c_uint32]
lib.mtr_rmaterial_num_materials.restype = C.c_uint32
lib.mtr_rmaterial_num_materials.argtypes = [C.c_void_p]
lib.mtr_rmaterial_info.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_MaterialInfo)]
lib.mtr_rmaterial_find.argtypes = [C.c_void_p, C.c_char_p]
lib.mtr_rscheduler_parse.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
lib.mtr_rscheduler_destroy.restype = None
lib.mtr_rscheduler_destroy.argtypes = [C.c_void_p]
lib.mtr_rscheduler_num_tracks.restype = C.c_uint32
lib.mtr_rscheduler_num_tracks.argtypes = [C.c_void_p]
lib.mtr_rscheduler_track.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_TrackInfo)]
lib.mtr_rscheduler_key.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                   C.POINTER(C.c_uint64), C.POINTER(C.c_char_p)]
lib.mtr_rscheduler_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
lib.mtr_model_create_from_files.argtypes = [C.c_void_p, C.POINTER(_RModelView), C.c_void_p, C.c_void_p,
                                            C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(C.c_void_p)]


SDL_PARTS_DISP, SDL_INSTANCE_MATRIX, SDL_INSTANCE_TRANSLATION, SDL_INSTANCE_TRANSLATE_X, SDL_INSTANCE_TRANSLATE_Y, SDL_INSTANCE_TRANSLATE_Z = range(6)


def _check(rc: int):
    if rc:
        raise MtrError(rc, lib.mtr_files_last_error().decode(errors="replace"))


def struct_size(kind: str) -> int:
    return int(lib.mtr_file_struct_size(STRUCT_KINDS.index(kind)))


def _buf(data: bytes):
    """The parsers return views into the caller's bytes: keep them alive in a ctypes buffer next to the view."""
    b = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    return b


class ModelFile:
    """src/rmodel.rs:295-484."""

    def __init__(self, data: bytes):
        self._b = _buf(data)
        self.v = _RModelView()
        _check(lib.mtr_rmodel_parse(self._b, len(data), C.byref(self.v)))

    def _arr(self, ptr, nbytes, dtype=np.uint8):
        if not ptr or nbytes == 0:
            return np.zeros(0, dtype=dtype)
        return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=dtype).copy()

    def material_names(self) -> List[str]:
        raw = self._arr(self.v.material_names, self.v.material_num * 128).reshape(-1, 128)
        return [bytes(r).split(b"\0", 1)[0].decode(errors="replace") for r in raw]

    def primitives(self) -> np.ndarray:
        return self._arr(self.v.primitives, self.v.primitive_num * 0x38).reshape(-1, 0x38)

    def primitive_field(self, p: int, field: int) -> int:
        prim = np.ascontiguousarray(self.primitives()[p])
        return int(lib.mtr_primitive_field(prim.ctypes.data, field))

    def vertex_buf(self) -> np.ndarray:
        return self._arr(self.v.vertex_buf, self.v.vertexbuf_size)

    def index_buf(self) -> np.ndarray:
        return self._arr(self.v.index_buf, self.v.index_num * 2, np.uint16)

    def boundary_joint(self, i: int) -> int:
        out = C.c_uint32()
        _check(lib.mtr_rmodel_boundary_joint(C.byref(self.v), i, C.byref(out)))
        return out.value

    def joint(self, i: int) -> dict:
        no, parent, sym = C.c_uint32(), C.c_uint32(), C.c_uint32()
        off = (C.c_float * 3)()
        _check(lib.mtr_rmodel_joint(C.byref(self.v), i, C.byref(no), C.byref(parent), C.byref(sym), off))
        return dict(no=no.value, parent=parent.value, symmetry=sym.value, offset=tuple(off))

    def lmats(self) -> np.ndarray:
        return self._arr(self.v.lmats, self.v.jnt_num * 64, np.float32).reshape(-1, 16)

    def imats(self) -> np.ndarray:
        return self._arr(self.v.imats, self.v.jnt_num * 64, np.float32).reshape(-1, 16)

    def palette(self, local_mats: Optional[np.ndarray] = None) -> np.ndarray:
        """skin palette of the skeleton (include/mtr_files.h: mtr_rmodel_palette): [jnt_num, 16] f32 for Model.set_palette;
        local_mats None = the file's bind pose"""
        n = self.v.jnt_num
        out = np.zeros((max(1, n), 16), dtype=np.float32)
        lm = None if local_mats is None else np.ascontiguousarray(local_mats, dtype=np.float32).reshape(n, 16)
        _check(lib.mtr_rmodel_palette(C.byref(self.v), None if lm is None else lm.ctypes.data_as(C.c_void_p),
                                      out.ctypes.data_as(C.c_void_p), n))
        return out[:n]

    def joint_index(self, no: int) -> Optional[int]:
        i = lib.mtr_rmodel_joint_index(C.byref(self.v), no)
        return None if i < 0 else i

    def joint_table(self) -> np.ndarray:
        return self._arr(self.v.joint_table, 256) if self.v.joint_table else np.full(256, 255, dtype=np.uint8)


class TextureFile:
    """src/rtexture.rs:80-166."""

    def __init__(self, data: bytes):
        self._b = _buf(data)
        self._len = len(data)
        self.v = _RTextureView()
        _check(lib.mtr_rtexture_parse(self._b, len(data), C.byref(self.v)))

    def width(self) -> int:
        return self.v.width

    def height(self) -> int:
        return self.v.height

    def format(self) -> int:
        return self.v.format

    def data(self) -> bytes:
        return bytes((C.c_uint8 * self.v.data_len).from_address(self.v.data)) if self.v.data_len else b""

    def upload(self, dev: api.Device, max_levels: int = 1) -> api.Texture:
        """Texture::new (src/texture.rs:11-30): level 0 only, like the reference; max_levels > 1 also uploads the file's
        mip chain (row f-4)."""
        h = C.c_void_p()
        _check(lib.mtr_texture_create_from_file_mips(dev._h, self._b, self._len, max_levels, C.byref(h)))
        t = api.Texture(dev, h)
        t.width, t.height = self.v.width, self.v.height
        return t


class Shader2File:
    """src/rshader2.rs:246-494."""

    def __init__(self, data: bytes):
        self.h = C.c_void_p()
        _check(lib.mtr_rshader2_parse(_buf(data), len(data), C.byref(self.h)))

    def close(self):
        if self.h:
            lib.mtr_rshader2_destroy(self.h)
            self.h = None

    __del__ = close

    def objects(self) -> List[dict]:
        out = []
        for i in range(lib.mtr_rshader2_num_objects(self.h)):
            name, ot, nh = C.c_char_p(), C.c_uint32(), C.c_uint32()
            _check(lib.mtr_rshader2_object(self.h, i, C.byref(name), C.byref(ot), C.byref(nh)))
            out.append(dict(name=name.value.decode(errors="replace"), obj_type=ot.value, name_hash=nh.value))
        return out

    def get_object_by_handle(self, handle: int) -> Optional[int]:
        i = lib.mtr_rshader2_find(self.h, handle & 0xFFFFFFFF)
        return None if i < 0 else i

    def input_layout(self, i: int) -> dict:
        stride, n = C.c_uint32(), C.c_uint32()
        lay = api._Layout()
        raw = (_RawElement * 64)()
        _check(lib.mtr_rshader2_input_layout(self.h, i, C.byref(stride), C.byref(lay), raw, 64, C.byref(n)))
        els = [dict(name=raw[e].name.decode(errors="replace"), sindex=raw[e].sindex, format=raw[e].format, count=raw[e].count,
                    start=raw[e].start, offset=raw[e].offset, instance=raw[e].instance) for e in range(min(n.value, 64))]
        bound = [(lay.elements[e].semantic, lay.elements[e].format, lay.elements[e].count, lay.elements[e].offset)
                 for e in range(lay.num_elements)]
        return dict(stride=stride.value, elements=els, bound=bound)


class MaterialFile:
    """src/rmaterial.rs:172-312."""

    def __init__(self, data: bytes, shader2: Shader2File):
        self.h = C.c_void_p()
        _check(lib.mtr_rmaterial_parse(_buf(data), len(data), shader2.h, C.byref(self.h)))

    def close(self):
        if self.h:
            lib.mtr_rmaterial_destroy(self.h)
            self.h = None

    __del__ = close

    def textures(self) -> List[str]:
        return [lib.mtr_rmaterial_texture_path(self.h, i).decode(errors="replace") for i in range(lib.mtr_rmaterial_num_textures(self.h))]

    def materials(self) -> List[dict]:
        out = []
        for i in range(lib.mtr_rmaterial_num_materials(self.h)):
            mi = _MaterialInfo()
            _check(lib.mtr_rmaterial_info(self.h, i, C.byref(mi)))
            out.append(dict(name_hash=mi.name_hash, dti_hash=mi.dti_hash,
                            albedo_texture_idx=None if mi.albedo_texture < 0 else mi.albedo_texture, bsstate=mi.bsstate,
                            dsstate=mi.dsstate, rsstate=mi.rsstate, state_num=mi.state_num, blend_factor=tuple(mi.blend_factor)))
        return out

    def material_by_name(self, name: str) -> Optional[int]:
        i = lib.mtr_rmaterial_find(self.h, name.encode())
        return None if i < 0 else i


class SchedulerFile:
    """src/rscheduler.rs:84-217, plus key access and a step-hold evaluation (this build's; the reference only logs)."""

    def __init__(self, data: bytes):
        self.h = C.c_void_p()
        _check(lib.mtr_rscheduler_parse(_buf(data), len(data), C.byref(self.h)))

    def close(self):
        if self.h:
            lib.mtr_rscheduler_destroy(self.h)
            self.h = None

    __del__ = close

    def tracks(self) -> List[dict]:
        out = []
        for i in range(lib.mtr_rscheduler_num_tracks(self.h)):
            t = _TrackInfo()
            _check(lib.mtr_rscheduler_track(self.h, i, C.byref(t)))
            out.append(dict(track_type=t.track_type, prop_type=t.prop_type, key_num=t.key_num, parent=t.parent,
                            dti_or_prop=t.dti_or_prop, name=t.name.decode(errors="replace")))
        return out

    def key(self, track: int, k: int) -> dict:
        fr, mode, val, res = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_char_p()
        _check(lib.mtr_rscheduler_key(self.h, track, k, C.byref(fr), C.byref(mode), C.byref(val), C.byref(res)))
        return dict(frame=fr.value, mode=mode.value, value_bits=val.value, resource=None if res.value is None else res.value.decode(errors="replace"))

    def eval(self, track: int, frame: int) -> int:
        val = C.c_uint64()
        _check(lib.mtr_rscheduler_eval(self.h, track, frame, C.byref(val)))
        return val.value

    def key_floats(self, track: int, k: int) -> np.ndarray:
        """FLOAT (1) / VECTOR (4) / MATRIX (16) key as f32"""
        out = np.zeros(16, dtype=np.float32)
        n = C.c_uint32()
        _check(lib.mtr_rscheduler_key_floats(self.h, track, k, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[:n.value].copy()

    def eval_floats(self, track: int, frame: int) -> np.ndarray:
        out = np.zeros(16, dtype=np.float32)
        n = C.c_uint32()
        _check(lib.mtr_rscheduler_eval_floats(self.h, track, frame, out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out[:n.value].copy()

    def find_track(self, name: str) -> Optional[int]:
        i = lib.mtr_rscheduler_find_track(self.h, name.encode())
        return None if i < 0 else i

    def apply(self, frame: int, bindings, parts_disp: Optional[np.ndarray] = None, model_mats: Optional[np.ndarray] = None):
        """bindings: (track, target, index) triples (include/mtr_files.h: MTR_SDL_*); the arrays are updated in place"""
        b = np.ascontiguousarray(bindings, dtype=np.uint32).reshape(-1, 3)
        pd = None if parts_disp is None else parts_disp
        mm = None if model_mats is None else model_mats
        assert pd is None or (pd.dtype == np.uint8 and pd.flags.c_contiguous)
        assert mm is None or (mm.dtype == np.float32 and mm.flags.c_contiguous)
        _check(lib.mtr_rscheduler_apply(self.h, frame, b.ctypes.data_as(C.c_void_p), b.shape[0],
                                        None if pd is None else pd.ctypes.data_as(C.c_void_p), 0 if pd is None else pd.size,
                                        None if mm is None else mm.ctypes.data_as(C.c_void_p), 0 if mm is None else mm.size // 16))

    def eval_float(self, track: int, frame: int) -> float:
        return float(np.array([self.eval(track, frame) & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0])


class ArchiveFile:
    """src/rarchive.rs:66-176: a table of zlib-compressed resources."""

    def __init__(self, data: bytes):
        self._b = _buf(data)
        self.v = _RArchiveView()
        _check(lib.mtr_rarchive_parse(self._b, len(data), C.byref(self.v)))

    def resource_infos(self) -> List[dict]:
        out = []
        for i in range(self.v.num_resources):
            ri = _ResourceInfo()
            _check(lib.mtr_rarchive_info(C.byref(self.v), i, C.byref(ri)))
            out.append(dict(path=ri.path.decode(errors="replace"), dti_hash=ri.dti_hash, size_compressed=ri.size_compressed,
                            size_uncompressed=ri.size_uncompressed, quality=ri.quality, offset=ri.offset))
        return out

    def get_resource(self, path: str, dti_hash: int) -> Optional[bytes]:
        """ArchiveFile::get_resource_with_path: None when the archive has no such (path, class)."""
        i = lib.mtr_rarchive_find(C.byref(self.v), path.encode(), dti_hash & 0xFFFFFFFF)
        return None if i < 0 else self.extract(i)

    def extract(self, i: int) -> bytes:
        ri = _ResourceInfo()
        _check(lib.mtr_rarchive_info(C.byref(self.v), i, C.byref(ri)))
        out = (C.c_uint8 * max(1, ri.size_uncompressed))()
        n = C.c_size_t()
        _check(lib.mtr_rarchive_extract(C.byref(self.v), i, out, ri.size_uncompressed, C.byref(n)))
        return bytes(out[:n.value])


def state_from_names(bs: Optional[str], ds: Optional[str], rs: Optional[str]):
    """(blend, depth_write, depth_test, cull), and how many of the names an explicit rule recognised (include/mtr_files.h)"""
    st = (C.c_uint8 * 4)()
    enc = lambda n: None if n is None else n.encode()
    known = lib.mtr_state_from_names(enc(bs), enc(ds), enc(rs), st)
    return tuple(st), known


def states_from_files(model: "ModelFile", shader2: "Shader2File", material: "MaterialFile") -> np.ndarray:
    """uint8 [primitives, 4] for Model.set_prim_states: each primitive's material -> its state objects' names -> states"""
    n = model.v.primitive_num
    out = np.zeros((n, 4), dtype=np.uint8)
    _check(lib.mtr_model_states_from_files(C.byref(model.v), shader2.h, material.h, out.ctypes.data_as(C.c_void_p), n))
    return out


def model_from_files(dev: api.Device, model: ModelFile, shader2: Shader2File, material: Optional[MaterialFile],
                     textures: Sequence[Optional[api.Texture]] = ()) -> api.Model:
    """Model::new (src/model.rs:36-293): textures[i] is the loaded rMaterial texture i, or None if it failed to load."""
    arr = (C.c_void_p * max(1, len(textures)))(*[t._h if t is not None else None for t in textures])
    h = C.c_void_p()
    _check(lib.mtr_model_create_from_files(dev._h, C.byref(model.v), shader2.h, material.h if material else None, arr,
                                           len(textures), C.byref(h)))
    from .scene import ModelData
    n = model.v.primitive_num
    # host-side description for the mirror's helpers (vertex_stage reads prims); the device copy is complete
    md = ModelData(vertex_buf=model.vertex_buf(), index_buf=model.index_buf(), prims=model.primitives(), layouts=[],
                   prim_to_texture=np.full(n, -1, dtype=np.int32), prim_debug_id=np.zeros(n, dtype=np.uint32),
                   parts_disp=np.ones(n, dtype=np.uint8), textures=[])
    return api.Model(dev, h, [t for t in textures if t is not None], md)
