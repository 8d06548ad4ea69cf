"""ctypes front-end of the CPU oracle (oracle/mtr_oracle.c).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by mt_renderer_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmtr_oracle.so")


class _Element(C.Structure):
    _fields_ = [("semantic", C.c_uint8), ("format", C.c_uint8), ("count", C.c_uint8), ("pad0", C.c_uint8),
                ("offset", C.c_uint16), ("pad1", C.c_uint16)]


class _Layout(C.Structure):
    _fields_ = [("num", C.c_uint32), ("el", _Element * 8)]


class _Texture(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("rgba", C.c_void_p), ("levels", C.c_uint32)]


class _Model(C.Structure):
    _fields_ = [("vertex_buf", C.c_void_p), ("vertex_len", C.c_size_t), ("index_buf", C.c_void_p),
                ("index_num", C.c_size_t), ("prims", C.c_void_p), ("nprims", C.c_size_t),
                ("layouts", C.POINTER(_Layout)), ("prim_to_texture", C.c_void_p), ("prim_debug_id", C.c_void_p),
                ("parts_disp", C.c_void_p), ("nparts", C.c_size_t), ("textures", C.POINTER(_Texture)),
                ("ntextures", C.c_size_t), ("prim_state", C.c_void_p)]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("mtr_oracle.c", "mtr_oracle.h", "bc7_tables.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
        for n in ("vertex_stride", "parts_no", "material_no", "weight_num", "inputlayout", "vertex_base", "index_ofs",
                  "index_base", "index_num", "topology", "vertex_num", "boundary_num"):
            f = getattr(L, "orc_prim_" + n)
            f.restype = C.c_uint32
            f.argtypes = [C.c_void_p]
        L.orc_texture_decode.restype = C.c_int
        L.orc_texture_decode.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_bc1_decode_block.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_bc7_decode_block.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_vertex_stage.restype = C.c_int
        L.orc_vertex_stage.argtypes = [C.POINTER(_Model), C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_mat4_mul.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_frame_create.restype = C.c_void_p
        L.orc_frame_create.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_float]
        L.orc_frame_destroy.argtypes = [C.c_void_p]
        L.orc_frame_color.restype = C.c_void_p
        L.orc_frame_color.argtypes = [C.c_void_p]
        L.orc_frame_depth.restype = C.c_void_p
        L.orc_frame_depth.argtypes = [C.c_void_p]
        for n in ("tris_in", "tris_setup", "frags"):
            f = getattr(L, "orc_frame_" + n)
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p]
        L.orc_draw.restype = C.c_int
        L.orc_draw.argtypes = [C.c_void_p, C.POINTER(_Model), C.c_void_p, C.c_void_p, C.c_size_t, C.c_int32, C.c_int]
        L.orc_draw_overlay_cubes.restype = C.c_int
        L.orc_draw_overlay_cubes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    def __init__(self, code: int):
        super().__init__({1: "invalid argument", 2: "unsupported format"}.get(code, f"error {code}"))
        self.code = code


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def crc32(data: bytes, init: int = 0xFFFFFFFF) -> int:
    buf = np.frombuffer(data, dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    return int(lib().orc_crc32(_ptr(buf), len(data), init))


def prim_field(name: str, prim_bytes: np.ndarray) -> int:
    b = np.ascontiguousarray(prim_bytes, dtype=np.uint8)
    assert b.size == 0x38
    return int(getattr(lib(), "orc_prim_" + name)(_ptr(b)))


def decode_texture(fmt: int, w: int, h: int, data: bytes) -> np.ndarray:
    out = np.zeros((h, w, 4), dtype=np.uint8)
    buf = np.frombuffer(data, dtype=np.uint8)
    rc = lib().orc_texture_decode(fmt, w, h, _ptr(buf), len(data), _ptr(out))
    if rc:
        raise OracleError(rc)
    return out


def level_sizes(w: int, h: int, levels: int):
    out = []
    for _ in range(max(1, levels)):
        out.append((w, h))
        w, h = max(1, w >> 1), max(1, h >> 1)
    return out


def level_bytes(fmt: int, w: int, h: int) -> int:
    return w * h * 4 if fmt == 7 else ((w + 3) // 4) * ((h + 3) // 4) * (8 if fmt == 19 else 16)


def decode_texture_levels(fmt: int, w: int, h: int, data: bytes, levels: int = 1) -> np.ndarray:
    """every mip level decoded to RGBA8, concatenated (level l is max(1, w >> l) x max(1, h >> l))"""
    parts, off = [], 0
    for lw, lh in level_sizes(w, h, levels):
        n = level_bytes(fmt, lw, lh)
        parts.append(decode_texture(fmt, lw, lh, data[off:off + n]).reshape(-1))
        off += n
    return np.concatenate(parts)


def decode_bc7_blocks(blocks: np.ndarray) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 16)
    out = np.zeros((blocks.shape[0], 16, 4), dtype=np.uint8)
    for i in range(blocks.shape[0]):
        lib().orc_bc7_decode_block(_ptr(blocks[i]), _ptr(out[i]))
    return out


def decode_bc1_blocks(blocks: np.ndarray) -> np.ndarray:
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, 8)
    out = np.zeros((blocks.shape[0], 16, 4), dtype=np.uint8)
    for i in range(blocks.shape[0]):
        lib().orc_bc1_decode_block(_ptr(blocks[i]), _ptr(out[i]))
    return out


def mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(16)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.float32)
    lib().orc_mat4_mul(_ptr(a), _ptr(b), _ptr(out))
    return out


class OracleModel:
    """Holds the C view of a scene.ModelData (keeps every numpy buffer alive)."""

    def __init__(self, md):
        self.md = md
        self._keep = []
        k = self._keep
        vb = np.ascontiguousarray(md.vertex_buf, dtype=np.uint8)
        ib = np.ascontiguousarray(md.index_buf, dtype=np.uint16)
        pr = np.ascontiguousarray(md.prims, dtype=np.uint8).reshape(-1, 0x38)
        p2t = np.ascontiguousarray(md.prim_to_texture, dtype=np.int32)
        did = np.ascontiguousarray(md.prim_debug_id, dtype=np.uint32)
        pd = np.ascontiguousarray(md.parts_disp, dtype=np.uint8)
        lays = (_Layout * len(md.layouts))()
        for i, els in enumerate(md.layouts):
            lays[i].num = len(els)
            for j, el in enumerate(els[:8]):  # (semantic, format, count, offset[, flags])
                lays[i].el[j].semantic, lays[i].el[j].format, lays[i].el[j].count, lays[i].el[j].offset = el[:4]
                lays[i].el[j].pad0 = el[4] if len(el) > 4 else 0
        texs = (_Texture * max(1, len(md.textures)))()
        for i, t in enumerate(md.textures):
            dec = decode_texture_levels(t.fmt, t.width, t.height, t.data, getattr(t, "levels", 1))
            k.append(dec)
            texs[i].w, texs[i].h, texs[i].rgba, texs[i].levels = t.width, t.height, dec.ctypes.data, getattr(t, "levels", 1)
        k += [vb, ib, pr, p2t, did, pd, lays, texs]
        m = _Model()
        m.vertex_buf, m.vertex_len = vb.ctypes.data, vb.size
        m.index_buf, m.index_num = ib.ctypes.data, ib.size
        m.prims, m.nprims = pr.ctypes.data, pr.shape[0]
        m.layouts = lays
        m.prim_to_texture, m.prim_debug_id = p2t.ctypes.data, did.ctypes.data
        m.parts_disp, m.nparts = pd.ctypes.data, pd.size
        m.textures, m.ntextures = texs, len(md.textures)
        st = getattr(md, "prim_states", None)
        if st is not None:
            st = np.ascontiguousarray(st, dtype=np.uint8).reshape(-1, 4)
            assert st.shape[0] == pr.shape[0]
            k.append(st)
            m.prim_state = st.ctypes.data
        self.c = m

    def vertex_stage(self, prim: int, M: np.ndarray, palette: Optional[np.ndarray] = None):
        nv = prim_field("vertex_num", self.md.prims[prim])
        clip = np.zeros((nv, 4), dtype=np.float32)
        uv = np.zeros((nv, 2), dtype=np.float32)
        M = np.ascontiguousarray(M, dtype=np.float32).reshape(16)
        pal = None if palette is None else np.ascontiguousarray(palette, dtype=np.float32).reshape(-1, 16)
        rc = lib().orc_vertex_stage(C.byref(self.c), prim, _ptr(M), _ptr(pal), 0 if pal is None else pal.shape[0],
                                    _ptr(clip), _ptr(uv))
        if rc:
            raise OracleError(rc)
        return clip, uv


class OracleFrame:
    def __init__(self, w: int, h: int, clear_rgba=(1.0, 1.0, 1.0, 1.0), clear_depth: float = 1.0):
        # defaults: clear white / depth 1.0, src/bin/modelviewer.rs:196,203
        self.w, self.h = w, h
        c = np.asarray(clear_rgba, dtype=np.float32)
        self._f = lib().orc_frame_create(w, h, _ptr(c), clear_depth)
        if not self._f:
            raise OracleError(1)

    def close(self):
        if self._f:
            lib().orc_frame_destroy(self._f)
            self._f = None

    __del__ = close

    def draw(self, model: OracleModel, M: np.ndarray, palette: Optional[np.ndarray] = None, tex_override: int = -1,
             nthreads: int = 1):
        M = np.ascontiguousarray(M, dtype=np.float32).reshape(16)
        pal = None if palette is None else np.ascontiguousarray(palette, dtype=np.float32).reshape(-1, 16)
        rc = lib().orc_draw(self._f, C.byref(model.c), _ptr(M), _ptr(pal), 0 if pal is None else pal.shape[0],
                            tex_override, nthreads)
        if rc:
            raise OracleError(rc)

    def draw_instances(self, model: OracleModel, view_proj: np.ndarray, model_mats: np.ndarray,
                       palettes: Optional[np.ndarray] = None, tex_override: Optional[Sequence[int]] = None,
                       nthreads: int = 1):
        mm = np.ascontiguousarray(model_mats, dtype=np.float32).reshape(-1, 16)
        for i in range(mm.shape[0]):
            M = mat4_mul(view_proj, mm[i])
            pal = None if palettes is None else palettes[i]
            self.draw(model, M, pal, -1 if tex_override is None else int(tex_override[i]), nthreads)

    def draw_overlay_cubes(self, cam: np.ndarray, inst_mats: np.ndarray):
        cam = np.ascontiguousarray(cam, dtype=np.float32).reshape(16)
        im = np.ascontiguousarray(inst_mats, dtype=np.float32).reshape(-1, 16)
        rc = lib().orc_draw_overlay_cubes(self._f, _ptr(cam), _ptr(im), im.shape[0])
        if rc:
            raise OracleError(rc)

    def color(self) -> np.ndarray:
        p = lib().orc_frame_color(self._f)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(self.h, self.w, 4)).copy()

    def depth(self) -> np.ndarray:
        p = lib().orc_frame_depth(self._f)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(self.h, self.w)).copy()

    def stats(self) -> dict:
        L = lib()
        return dict(tris_in=int(L.orc_frame_tris_in(self._f)), tris_setup=int(L.orc_frame_tris_setup(self._f)),
                    frags=int(L.orc_frame_frags(self._f)))
