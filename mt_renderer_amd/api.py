"""ctypes binding of libmtr.so (include/mtr.h) + a host-side mirror of the reference's interface.

The classes keep the reference's names and argument roles -- ``Texture.new`` (src/texture.rs:11),
``Model.new`` / ``Model.set_parts_disp`` / ``Model.render`` (src/model.rs:36-45, :295, :299-305) --
with ``wgpu::Device`` / ``wgpu::RenderPass`` replaced by :class:`Device` / :class:`Frame`.

There is NO CPU fallback here: if libmtr.so is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from .scene import ModelData, TextureData

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTR_LIB_PATH") or os.path.join(_PKG, "libmtr.so")  # MTR_LIB_PATH: A/B runs of two builds on one box

MTR_OK, MTR_E_INVALID, MTR_E_UNSUPPORTED, MTR_E_NOMEM, MTR_E_HIP, MTR_E_OVERFLOW = range(6)
TILE_AUTO, TILE_ORDERED, TILE_VISIBILITY, TILE_MIXED = 0, 1, 2, 3
OWN_INTERLEAVED, OWN_BANDS, OWN_SUPERTILES = 0, 1, 2
TEXRES_DECODED, TEXRES_BLOCKS = 0, 1
GEOM_CULL_OFF, GEOM_CULL_SHARDED, GEOM_CULL_ALL_FRAMES = 0, 1, 2
STAGE_NAMES = ("geom", "scan", "fill", "tile")

# every symbol include/mtr.h declares (tests check that the library exports each one)
EXPORTED_SYMBOLS = [
    "mtr_abi_version", "mtr_device_create", "mtr_device_create_on_stream", "mtr_device_destroy", "mtr_last_error",
    "mtr_device_set_profiling", "mtr_texture_create", "mtr_texture_destroy", "mtr_texture_read_rgba8",
    "mtr_model_create", "mtr_model_destroy", "mtr_model_set_parts_disp", "mtr_model_set_palette",
    "mtr_batch_create", "mtr_batch_destroy", "mtr_frame_begin", "mtr_frame_set_shard", "mtr_frame_draw_model",
    "mtr_frame_draw_batch", "mtr_frame_draw_instances", "mtr_frame_draw_overlay_cubes", "mtr_frame_submit",
    "mtr_frame_wait", "mtr_frame_end", "mtr_frame_read_color", "mtr_frame_read_depth", "mtr_frame_color_devptr",
    "mtr_frame_depth_devptr", "mtr_frame_get_stats", "mtr_frame_get_timings", "mtr_frame_destroy",
    "mtr_model_vertex_stage", "mtr_crc32", "mtr_shard_bytes", "mtr_frame_pack_color_shard",
    "mtr_device_unpack_color_shards", "mtr_frame_read_bin_counts", "mtr_device_set_tile_mode", "mtr_device_set_binning",
    "mtr_frame_pack_color_shard_on_stream", "mtr_device_unpack_color_shards_on_stream",
    "mtr_device_synchronize", "mtr_model_set_joint_positions", "mtr_frame_draw_model_joints", "mtr_model_set_prim_states", "mtr_texture_create_mips", "mtr_frame_set_shard_map", "mtr_device_set_culling", "mtr_device_set_texture_residency", "mtr_shard_bytes_map", "mtr_frame_shard_bytes",
    "mtr_frame_unpack_color_shards_on_stream", "mtr_device_exchange_start", "mtr_device_exchange_add_lane", "mtr_frame_submit_exchange", "mtr_device_exchange_drain", "mtr_device_exchange_stop",
    "mtr_group_create", "mtr_group_destroy", "mtr_group_size", "mtr_group_device", "mtr_group_last_error", "mtr_group_frame_begin",
    "mtr_group_frame_part", "mtr_group_frame_end", "mtr_group_frame_read_color", "mtr_group_frame_color_devptr", "mtr_group_frame_destroy",
]


class MtrError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mtr error {code}: {msg}")
        self.code = code


class _Primitive(C.Structure):
    _fields_ = [("w", C.c_uint32 * 14)]


class _Element(C.Structure):
    _fields_ = [("semantic", C.c_uint8), ("format", C.c_uint8), ("count", C.c_uint8), ("flags", C.c_uint8),
                ("offset", C.c_uint16), ("pad1", C.c_uint16)]


class _Layout(C.Structure):
    _fields_ = [("num_elements", C.c_uint32), ("elements", _Element * 8)]


class FrameStats(C.Structure):
    _fields_ = [("tris_in", C.c_uint64), ("tris_setup", C.c_uint64), ("bin_entries", C.c_uint64),
                ("segments", C.c_uint64), ("width", C.c_uint32), ("height", C.c_uint32), ("nbins", C.c_uint32),
                ("ndraws", C.c_uint32), ("tile_kernel", C.c_uint32), ("binning", C.c_uint32),
                ("chunks", C.c_uint64), ("chunks_culled", C.c_uint64), ("shard_map", C.c_uint32), ("shard_bins", C.c_uint32)]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the HIP draw path)")
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If libmtr.so maps the system runtime
    # first and torch is imported later, the process ends up with two HSA runtimes and torch reports "No HIP GPUs
    # are available"; with torch mapped first, libmtr.so binds to the runtime that is already there.  So: when
    # torch is installed, load it before the library (MTR_NO_TORCH_PRELOAD=1 skips this).
    if os.environ.get("MTR_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, sz = C.c_void_p, C.c_int32, C.c_uint32, C.c_size_t
    sig = {
        "mtr_abi_version": (i32, []),
        "mtr_device_create": (i32, [i32, C.POINTER(vp)]),
        "mtr_device_create_on_stream": (i32, [i32, vp, C.POINTER(vp)]),
        "mtr_device_destroy": (None, [vp]),
        "mtr_last_error": (C.c_char_p, [vp]),
        "mtr_device_set_profiling": (i32, [vp, i32]),
        "mtr_device_synchronize": (i32, [vp]),
        "mtr_texture_create": (i32, [vp, u32, u32, u32, vp, sz, C.POINTER(vp)]),
        "mtr_texture_create_mips": (i32, [vp, u32, u32, u32, u32, vp, sz, C.POINTER(vp)]),
        "mtr_model_set_prim_states": (i32, [vp, vp, sz]),
        "mtr_model_set_joint_positions": (i32, [vp, vp, sz]),
        "mtr_frame_draw_model_joints": (i32, [vp, vp, vp]),
        "mtr_texture_destroy": (None, [vp]),
        "mtr_texture_read_rgba8": (i32, [vp, vp, sz]),
        "mtr_model_create": (i32, [vp, vp, sz, vp, sz, vp, sz, vp, vp, vp, sz, vp, C.POINTER(vp)]),
        "mtr_model_destroy": (None, [vp]),
        "mtr_model_set_parts_disp": (i32, [vp, vp, sz]),
        "mtr_model_set_palette": (i32, [vp, vp, sz]),
        "mtr_batch_create": (i32, [vp, vp, sz, vp, vp, sz, vp, C.POINTER(vp)]),
        "mtr_batch_destroy": (None, [vp]),
        "mtr_frame_begin": (i32, [vp, u32, u32, vp, C.c_float, C.POINTER(vp)]),
        "mtr_frame_set_shard": (i32, [vp, u32, u32]),
        "mtr_frame_set_shard_map": (i32, [vp, u32, u32, u32, u32, vp]),
        "mtr_device_set_culling": (i32, [vp, i32]),
        "mtr_device_set_texture_residency": (i32, [vp, u32]),
        "mtr_shard_bytes_map": (sz, [u32, u32, u32, u32, u32, vp]),
        "mtr_frame_shard_bytes": (sz, [vp]),
        "mtr_frame_unpack_color_shards_on_stream": (i32, [vp, vp, vp, vp]),
        "mtr_frame_draw_model": (i32, [vp, vp, vp]),
        "mtr_frame_draw_batch": (i32, [vp, vp, vp]),
        "mtr_frame_draw_instances": (i32, [vp, vp, vp, vp, sz, sz, vp]),
        "mtr_frame_draw_overlay_cubes": (i32, [vp, vp, vp, sz]),
        "mtr_frame_submit": (i32, [vp]),
        "mtr_frame_wait": (i32, [vp]),
        "mtr_frame_end": (i32, [vp]),
        "mtr_frame_read_color": (i32, [vp, vp, sz]),
        "mtr_frame_read_depth": (i32, [vp, vp, sz]),
        "mtr_frame_color_devptr": (vp, [vp]),
        "mtr_frame_depth_devptr": (vp, [vp]),
        "mtr_frame_get_stats": (i32, [vp, C.POINTER(FrameStats)]),
        "mtr_frame_get_timings": (i32, [vp, C.POINTER(C.c_float * 4)]),
        "mtr_frame_destroy": (None, [vp]),
        "mtr_model_vertex_stage": (i32, [vp, sz, vp, vp, vp]),
        "mtr_crc32": (u32, [vp, sz, u32]),
        "mtr_shard_bytes": (sz, [u32, u32, u32]),
        "mtr_frame_pack_color_shard": (i32, [vp, vp, sz]),
        "mtr_device_unpack_color_shards": (i32, [vp, vp, u32, u32, u32, vp]),
        "mtr_frame_pack_color_shard_on_stream": (i32, [vp, vp, sz, vp]),
        "mtr_device_unpack_color_shards_on_stream": (i32, [vp, vp, u32, u32, u32, vp, vp]),
        "mtr_device_exchange_start": (i32, [vp, vp, vp, i32, vp, sz, vp, vp, u32, vp]),
        "mtr_device_exchange_add_lane": (i32, [vp, vp, vp, vp, vp, vp]),
        "mtr_frame_submit_exchange": (i32, [vp]),
        "mtr_device_exchange_drain": (i32, [vp]),
        "mtr_device_exchange_stop": (i32, [vp]),
        "mtr_frame_read_bin_counts": (i32, [vp, vp, vp, sz]),
        "mtr_device_set_tile_mode": (i32, [vp, i32]),
        "mtr_device_set_binning": (i32, [vp, i32, u32]),
        "mtr_group_create": (i32, [vp, i32, C.POINTER(vp)]),
        "mtr_group_destroy": (None, [vp]),
        "mtr_group_size": (i32, [vp]),
        "mtr_group_device": (vp, [vp, i32]),
        "mtr_group_last_error": (C.c_char_p, [vp]),
        "mtr_group_frame_begin": (i32, [vp, u32, u32, vp, C.c_float, u32, u32, vp, C.POINTER(vp)]),
        "mtr_group_frame_part": (vp, [vp, i32]),
        "mtr_group_frame_end": (i32, [vp]),
        "mtr_group_frame_read_color": (i32, [vp, vp, sz]),
        "mtr_group_frame_color_devptr": (vp, [vp]),
        "mtr_group_frame_destroy": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


lib = _load()


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None) -> np.ndarray:
    r = np.ascontiguousarray(a, dtype=np.float32)
    return r if shape is None else r.reshape(shape)


def shard_bytes_map(width: int, height: int, world: int, own_map: int = OWN_INTERLEAVED, param: int = 0, band_rows=None) -> int:
    br = None if band_rows is None else np.ascontiguousarray(band_rows, dtype=np.uint32)
    return int(lib.mtr_shard_bytes_map(width, height, world, own_map, param, _p(br)))


def crc32(data: bytes, init: int = 0xFFFFFFFF) -> int:
    buf = np.frombuffer(data + b"\0", dtype=np.uint8)
    return int(lib.mtr_crc32(_p(buf), len(data), init))


class Device:
    """Stands where ``wgpu::Device`` + ``wgpu::Queue`` do (src/renderer_app_manager.rs:103-115)."""

    def __init__(self, hip_device: int = 0, stream: Optional[int] = None):
        h = C.c_void_p()
        rc = lib.mtr_device_create_on_stream(hip_device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc:
            raise MtrError(rc, (lib.mtr_last_error(None) or b"").decode())
        self._h = h

    def check(self, rc: int):
        if rc:
            raise MtrError(rc, (lib.mtr_last_error(self._h) or b"").decode())

    def unpack_color_shards(self, gathered_devptr: int, world: int, width: int, height: int, dst_devptr: int,
                            stream: Optional[int] = None):
        """gathered [rank][k][16][16] RGBA8 blocks (device) -> linear RGBA8 framebuffer (device); on the device's public
        stream, or on `stream` (a hipStream_t handle)."""
        if stream is None:
            self.check(lib.mtr_device_unpack_color_shards(self._h, C.c_void_p(gathered_devptr), world, width, height,
                                                          C.c_void_p(dst_devptr)))
        else:
            self.check(lib.mtr_device_unpack_color_shards_on_stream(self._h, C.c_void_p(gathered_devptr), world, width, height,
                                                                    C.c_void_p(dst_devptr), C.c_void_p(stream)))

    def exchange_start(self, allgather_fn_addr: int, comm: int, dtype_u8: int, send_devptr: int, send_bytes: int,
                       gathered_devptr: int, dst_devptr: int, world: int, stream: int):
        """start the device's exchange thread (include/mtr.h): per handed-over frame, on `stream`, pack -> all-gather
        (a C function with ncclAllGather's signature, e.g. rccl.Rccl().allgather_addr) -> unpack into dst -> destroy."""
        self.check(lib.mtr_device_exchange_start(self._h, C.c_void_p(allgather_fn_addr), C.c_void_p(comm), dtype_u8,
                                                 C.c_void_p(send_devptr), send_bytes, C.c_void_p(gathered_devptr),
                                                 C.c_void_p(dst_devptr), world, C.c_void_p(stream)))

    def exchange_add_lane(self, comm: int, send_devptr: int, gathered_devptr: int, dst_devptr: int, stream: int):
        """one more (communicator, buffers, stream) set for the exchange thread; frames go to the lanes in turn"""
        self.check(lib.mtr_device_exchange_add_lane(self._h, C.c_void_p(comm), C.c_void_p(send_devptr), C.c_void_p(gathered_devptr),
                                                    C.c_void_p(dst_devptr), C.c_void_p(stream)))

    def exchange_drain(self):
        """returns once every frame handed to the exchange thread has been issued; raises its first error"""
        self.check(lib.mtr_device_exchange_drain(self._h))

    def exchange_stop(self):
        self.check(lib.mtr_device_exchange_stop(self._h))

    def set_culling(self, mode):
        """skip geometry whose bounds cannot reach the rank's bins: False / True (sharded frames, default) or
        GEOM_CULL_ALL_FRAMES (unsharded frames too: what is off the target is skipped; include/mtr.h)"""
        self.check(lib.mtr_device_set_culling(self._h, int(mode)))

    def set_texture_residency(self, mode: int):
        """what BC textures created from now on keep in HBM: TEXRES_DECODED (RGBA8, default) or TEXRES_BLOCKS (include/mtr.h)"""
        self.check(lib.mtr_device_set_texture_residency(self._h, mode))

    def set_tile_mode(self, mode: int):
        """0 auto, 1 force the ordered tile kernel, 2 visibility-key kernel when eligible (include/mtr.h)."""
        self.check(lib.mtr_device_set_tile_mode(self._h, mode))

    def set_binning(self, single_pass: bool, queue_capacity: int = 0):
        """single-pass bounded bin queues (default) or the exact two-pass queues; see include/mtr.h."""
        self.check(lib.mtr_device_set_binning(self._h, 1 if single_pass else 0, queue_capacity))

    def synchronize(self):
        """every submitted frame has left the GPU; raises if one that was never waited for dropped triangles"""
        self.check(lib.mtr_device_synchronize(self._h))

    def set_profiling(self, on: bool):
        self.check(lib.mtr_device_set_profiling(self._h, 1 if on else 0))

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):  # a group's device is destroyed with the group
                lib.mtr_device_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Texture:
    def __init__(self, dev: Device, h):
        self.dev, self._h = dev, h

    @staticmethod
    def new(dev: Device, resource: TextureData) -> "Texture":
        """Texture::new(device, queue, resource) -- src/texture.rs:11."""
        h = C.c_void_p()
        buf = np.frombuffer(resource.data, dtype=np.uint8)
        dev.check(lib.mtr_texture_create_mips(dev._h, resource.width, resource.height, resource.fmt, getattr(resource, "levels", 1),
                                              _p(buf), buf.size, C.byref(h)))
        t = Texture(dev, h)
        t.width, t.height = resource.width, resource.height
        return t

    def read_rgba8(self) -> np.ndarray:
        out = np.zeros((self.height, self.width, 4), dtype=np.uint8)
        self.dev.check(lib.mtr_texture_read_rgba8(self._h, _p(out), out.size))
        return out

    def close(self):
        if self._h:
            lib.mtr_texture_destroy(self._h)
            self._h = None


class Model:
    def __init__(self, dev: Device, h, textures: List[Texture], md: ModelData):
        self.dev, self._h, self.textures, self.md = dev, h, textures, md

    @staticmethod
    def new(dev: Device, md: ModelData) -> "Model":
        """Model::new(model_file, material_file, shader2, ...) -- src/model.rs:36-45.  ``md`` carries what
        those three parsed files contribute to the draw path."""
        textures = [Texture.new(dev, t) for t in md.textures]
        vb = np.ascontiguousarray(md.vertex_buf, dtype=np.uint8)
        ib = np.ascontiguousarray(md.index_buf, dtype=np.uint16)
        pr = np.ascontiguousarray(md.prims, dtype=np.uint8).reshape(-1, 0x38)
        lays = (_Layout * len(md.layouts))()
        for i, els in enumerate(md.layouts):
            if len(els) > 8:
                raise MtrError(MTR_E_INVALID, "more than 8 layout elements")
            lays[i].num_elements = len(els)
            for j, el in enumerate(els):  # (semantic, format, count, offset[, flags])
                e = lays[i].elements[j]
                e.semantic, e.format, e.count, e.offset = el[:4]
                e.flags = el[4] if len(el) > 4 else 0
        p2t = np.ascontiguousarray(md.prim_to_texture, dtype=np.int32)
        did = np.ascontiguousarray(md.prim_debug_id, dtype=np.uint32)
        th = (C.c_void_p * max(1, len(textures)))(*[t._h for t in textures])
        h = C.c_void_p()
        dev.check(lib.mtr_model_create(dev._h, _p(vb), vb.size, _p(ib), ib.size, _p(pr), pr.shape[0], C.cast(lays, C.c_void_p),
                                       _p(p2t), C.cast(th, C.c_void_p), len(textures), _p(did), C.byref(h)))
        m = Model(dev, h, textures, md)
        pd = np.ascontiguousarray(md.parts_disp, dtype=np.uint8)
        if pd.size != pr.shape[0] or not pd.all():
            m.set_parts_disp(pd)
        if getattr(md, "prim_states", None) is not None:
            m.set_prim_states(md.prim_states)
        return m

    def set_prim_states(self, states):
        """material state per primitive: uint8 [nprims, 4] = blend, depth write, depth test, cull (include/mtr.h), or None"""
        if states is None:
            self.dev.check(lib.mtr_model_set_prim_states(self._h, None, 0))
        else:
            st = np.ascontiguousarray(states, dtype=np.uint8).reshape(-1, 4)
            self.dev.check(lib.mtr_model_set_prim_states(self._h, _p(st), st.shape[0]))

    def set_parts_disp(self, parts_disp: Sequence[bool]):
        """Model::set_parts_disp -- src/model.rs:295."""
        pd = np.ascontiguousarray(parts_disp, dtype=np.uint8)
        self.dev.check(lib.mtr_model_set_parts_disp(self._h, _p(pd), pd.size))

    def set_palette(self, mats: Optional[np.ndarray]):
        if mats is None:
            self.dev.check(lib.mtr_model_set_palette(self._h, None, 0))
        else:
            m = _f32(mats, (-1, 16))
            self.dev.check(lib.mtr_model_set_palette(self._h, _p(m), m.shape[0]))

    def render(self, frame: "Frame", view_proj: np.ndarray, joints: bool = False):
        """Model::render(rpass, queue, transform_bind_group, debug_overlay) -- src/model.rs:299-305; the
        transform uniform (src/bin/modelviewer.rs:217-221) is passed directly.  joints: also the per-joint debug cubes the
        reference adds to its overlay every frame (src/model.rs:309-315)."""
        frame.draw_model(self, view_proj)
        if joints:
            vp = _f32(view_proj, 16)
            self.dev.check(lib.mtr_frame_draw_model_joints(frame._h, self._h, _p(vp)))

    def set_joint_positions(self, xyz: np.ndarray):
        """JointInfo::offset of every joint (src/model.rs:283-291)"""
        a = _f32(xyz).reshape(-1, 3)
        self.dev.check(lib.mtr_model_set_joint_positions(self._h, _p(a), a.shape[0]))

    def vertex_stage(self, prim: int, M: np.ndarray):
        from .scene import unpack_primitive
        nv = unpack_primitive(self.md.prims[prim])["vertex_num"]
        clip = np.zeros((nv, 4), dtype=np.float32)
        uv = np.zeros((nv, 2), dtype=np.float32)
        Mf = _f32(M, 16)
        self.dev.check(lib.mtr_model_vertex_stage(self._h, prim, _p(Mf), _p(clip), _p(uv)))
        return clip, uv

    def close(self):
        if self._h:
            lib.mtr_model_destroy(self._h)
            self._h = None
        for t in self.textures:
            t.close()
        self.textures = []


class Batch:
    """n instances of one Model resident in HBM (the build's scheduler submission)."""

    def __init__(self, dev: Device, model: Model, model_mats: np.ndarray, palettes: Optional[np.ndarray] = None,
                 texture_override: Optional[Sequence[int]] = None):
        mm = _f32(model_mats, (-1, 16))
        n = mm.shape[0]
        pal, npal = None, 0
        if palettes is not None:
            pal = _f32(palettes).reshape(n, -1, 16)
            npal = pal.shape[1]
        to = None if texture_override is None else np.ascontiguousarray(texture_override, dtype=np.int32)
        h = C.c_void_p()
        dev.check(lib.mtr_batch_create(dev._h, model._h, n, _p(mm), _p(pal), npal, _p(to), C.byref(h)))
        self.dev, self._h, self.n, self.model = dev, h, n, model

    def close(self):
        if self._h:
            lib.mtr_batch_destroy(self._h)
            self._h = None


class FrameLoop:
    """A render loop body with every argument converted once: begin -> [set_shard] -> draw model / batch -> submit
    [-> exchange] -> destroy, as five or six bare C calls per frame.  The classes above convert numpy arrays and build
    ctypes objects on every call (about 20 us of interpreter time per frame, more than a sharded rank's GPU share of a
    small frame); a native host of the C ABI pays about 10 us per frame (tools/probe/host_cost.cpp).  Same calls, same
    order, same error checks -- only the argument marshalling is hoisted out of the loop."""

    def __init__(self, dev: Device, width: int, height: int, *, model: Optional[Model] = None, batch: Optional[Batch] = None,
                 view_proj: np.ndarray, clear_rgba=(1.0, 1.0, 1.0, 1.0), clear_depth: float = 1.0, shard=None, exchange: bool = False):
        assert (model is None) != (batch is None)
        self.dev = dev
        self._clear = _f32(clear_rgba, 4)
        self._vp = _f32(view_proj, 16)
        self._keep = (model, batch)
        self._begin_args = (dev._h, C.c_uint32(width), C.c_uint32(height), _p(self._clear), C.c_float(clear_depth))
        self._draw = lib.mtr_frame_draw_model if model is not None else lib.mtr_frame_draw_batch
        self._obj = (model or batch)._h
        self._vp_p = _p(self._vp)
        self._shard = None
        if shard is not None:
            sh = tuple(shard)
            rank, world, own_map, param, bands = sh + (OWN_INTERLEAVED, 0, None)[len(sh) - 2:]
            self._bands = None if bands is None else np.ascontiguousarray(bands, dtype=np.uint32)
            self._shard = (C.c_uint32(rank), C.c_uint32(world), C.c_uint32(own_map), C.c_uint32(param), _p(self._bands))
        self._exchange = exchange
        self._h = C.c_void_p()
        self._href = C.byref(self._h)

    def run(self, n: int = 1):
        """n frames: submitted, not waited for (dev.synchronize() / exchange_drain() afterwards reports any error)"""
        begin, draw, submit, destroy, shard_fn = lib.mtr_frame_begin, self._draw, lib.mtr_frame_submit, lib.mtr_frame_destroy, lib.mtr_frame_set_shard_map
        xchg = lib.mtr_frame_submit_exchange
        h, href, ba, obj, vp, sh, check = self._h, self._href, self._begin_args, self._obj, self._vp_p, self._shard, self.dev.check
        for _ in range(n):
            rc = begin(*ba, href)
            if rc:
                check(rc)
            if sh is not None:
                rc = shard_fn(h, *sh)
            rc = rc or draw(h, obj, vp)
            if not rc:
                if self._exchange:
                    rc = xchg(h)
                    if not rc:
                        continue  # the exchange thread owns the frame now
                else:
                    rc = submit(h)
            destroy(h)
            if rc:
                check(rc)


class Frame:
    """One render pass: clear colour / depth as in src/bin/modelviewer.rs:190-210 (white, 1.0)."""

    def __init__(self, dev: Device, width: int, height: int, clear_rgba=(1.0, 1.0, 1.0, 1.0), clear_depth: float = 1.0):
        c = _f32(clear_rgba, 4)
        h = C.c_void_p()
        dev.check(lib.mtr_frame_begin(dev._h, width, height, _p(c), clear_depth, C.byref(h)))
        self.dev, self._h, self.w, self.h = dev, h, width, height

    def set_shard(self, rank: int, world: int, own_map: int = OWN_INTERLEAVED, param: int = 0, band_rows=None):
        """render only the bins `rank` of `world` owns under the ownership map (include/mtr.h: MTR_OWN_*)"""
        br = None if band_rows is None else np.ascontiguousarray(band_rows, dtype=np.uint32)
        if br is not None and br.size != world + 1:
            raise MtrError(MTR_E_INVALID, "band_rows needs world + 1 entries")
        self.dev.check(lib.mtr_frame_set_shard_map(self._h, rank, world, own_map, param, _p(br)))

    def shard_bytes(self) -> int:
        return int(lib.mtr_frame_shard_bytes(self._h))

    def unpack_color_shards(self, gathered_devptr: int, dst_devptr: int, stream: Optional[int] = None):
        """gathered blocks (this frame's ownership map) -> linear RGBA8 at dst_devptr"""
        self.dev.check(lib.mtr_frame_unpack_color_shards_on_stream(self._h, C.c_void_p(gathered_devptr), C.c_void_p(dst_devptr),
                                                                   C.c_void_p(stream) if stream else None))

    def draw_model(self, model: Model, view_proj: np.ndarray):
        vp = _f32(view_proj, 16)
        self.dev.check(lib.mtr_frame_draw_model(self._h, model._h, _p(vp)))

    def draw_batch(self, batch: Batch, view_proj: np.ndarray):
        vp = _f32(view_proj, 16)
        self.dev.check(lib.mtr_frame_draw_batch(self._h, batch._h, _p(vp)))

    def draw_instances(self, model: Model, view_proj: np.ndarray, model_mats: np.ndarray,
                       palettes: Optional[np.ndarray] = None):
        vp = _f32(view_proj, 16)
        mm = _f32(model_mats, (-1, 16))
        pal, npal = None, 0
        if palettes is not None:
            pal = _f32(palettes).reshape(mm.shape[0], -1, 16)
            npal = pal.shape[1]
        self.dev.check(lib.mtr_frame_draw_instances(self._h, model._h, _p(mm), _p(pal), npal, mm.shape[0], _p(vp)))

    def draw_overlay_cubes(self, camera: np.ndarray, inst_mats: np.ndarray):
        cam = _f32(camera, 16)
        im = _f32(inst_mats, (-1, 16))
        self.dev.check(lib.mtr_frame_draw_overlay_cubes(self._h, _p(cam), _p(im), im.shape[0]))

    def submit(self):
        self.dev.check(lib.mtr_frame_submit(self._h))

    def submit_exchange(self):
        """submit (if not yet) and hand the frame to the device's exchange thread, which owns it from here on"""
        self.dev.check(lib.mtr_frame_submit_exchange(self._h))
        self._h = None

    def wait(self):
        self.dev.check(lib.mtr_frame_wait(self._h))

    def end(self):
        self.dev.check(lib.mtr_frame_end(self._h))

    def color(self) -> np.ndarray:
        out = np.zeros((self.h, self.w, 4), dtype=np.uint8)
        self.dev.check(lib.mtr_frame_read_color(self._h, _p(out), out.size))
        return out

    def depth(self) -> np.ndarray:
        out = np.zeros((self.h, self.w), dtype=np.float32)
        self.dev.check(lib.mtr_frame_read_depth(self._h, _p(out), out.size))
        return out

    def pack_color_shard(self, dst_devptr: int, dst_bytes: int, stream: Optional[int] = None):
        """this rank's bins, bin-major, into the all-gather send buffer (device pointer); on the device's public stream,
        or on `stream` (a hipStream_t handle; the pack then waits there for this frame's completion)."""
        if stream is None:
            self.dev.check(lib.mtr_frame_pack_color_shard(self._h, C.c_void_p(dst_devptr), dst_bytes))
        else:
            self.dev.check(lib.mtr_frame_pack_color_shard_on_stream(self._h, C.c_void_p(dst_devptr), dst_bytes, C.c_void_p(stream)))

    def color_devptr(self) -> int:
        return int(lib.mtr_frame_color_devptr(self._h) or 0)

    def stats(self) -> dict:
        s = FrameStats()
        self.dev.check(lib.mtr_frame_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def bin_counts(self):
        """(entries, segments) per 16x16 bin of the frame just rendered (tuning / test hook)."""
        n = self.stats()["nbins"]
        e = np.zeros(n, dtype=np.uint32)
        s = np.zeros(n, dtype=np.uint32)
        self.dev.check(lib.mtr_frame_read_bin_counts(self._h, _p(e), _p(s), n))
        return e, s

    def timings_ms(self) -> dict:
        ms = (C.c_float * 4)()
        self.dev.check(lib.mtr_frame_get_timings(self._h, C.byref(ms)))
        return {STAGE_NAMES[i]: float(ms[i]) for i in range(4)}

    def close(self):
        if self._h:
            lib.mtr_frame_destroy(self._h)
            self._h = None


class Group:
    """One host thread, N devices (include/mtr.h: mtr_group_*): rank r is a Device on hip_devices[r]; a group frame is one
    sharded Frame per rank, and ending it gathers every part's colour into one image on rank 0's device."""

    def __init__(self, hip_devices: Sequence[int]):
        ids = np.ascontiguousarray(hip_devices, dtype=np.int32)
        h = C.c_void_p()
        rc = lib.mtr_group_create(_p(ids), ids.size, C.byref(h))
        if rc:
            raise MtrError(rc, (lib.mtr_group_last_error(None) or b"").decode())
        self._h = h
        self.devices = []
        for r in range(lib.mtr_group_size(h)):
            d = Device.__new__(Device)
            d._h, d._borrowed = C.c_void_p(lib.mtr_group_device(h, r)), True
            self.devices.append(d)

    def __len__(self):
        return len(self.devices)

    def device(self, rank: int) -> Device:
        return self.devices[rank]

    def check(self, rc: int):
        if rc:
            raise MtrError(rc, (lib.mtr_group_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            lib.mtr_group_destroy(self._h)
            self._h = None
            for d in self.devices:
                d._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class GroupFrame:
    """begin -> draw into part(r) with rank r's models / batches -> end() -> color()"""

    def __init__(self, group: Group, width: int, height: int, clear_rgba=(1.0, 1.0, 1.0, 1.0), clear_depth: float = 1.0,
                 own_map: int = OWN_INTERLEAVED, param: int = 0, band_rows=None):
        c = _f32(clear_rgba, 4)
        br = None if band_rows is None else np.ascontiguousarray(band_rows, dtype=np.uint32)
        if br is not None and br.size != len(group) + 1:
            raise MtrError(MTR_E_INVALID, "band_rows needs world + 1 entries")
        h = C.c_void_p()
        group.check(lib.mtr_group_frame_begin(group._h, width, height, _p(c), clear_depth, own_map, param, _p(br), C.byref(h)))
        self.group, self._h, self.w, self.h = group, h, width, height
        self.parts = []
        for r in range(len(group)):
            f = Frame.__new__(Frame)
            f.dev, f._h, f.w, f.h = group.device(r), None, width, height  # _h None: Frame.close() never destroys a part
            f._part = C.c_void_p(lib.mtr_group_frame_part(h, r))
            self.parts.append(f)

    def part(self, rank: int) -> "Frame":
        """rank's frame with its handle live for the draw calls (owned by the group frame)"""
        f = self.parts[rank]
        f._h = f._part
        return f

    def end(self):
        self.group.check(lib.mtr_group_frame_end(self._h))

    def color(self) -> np.ndarray:
        out = np.zeros((self.h, self.w, 4), dtype=np.uint8)
        self.group.check(lib.mtr_group_frame_read_color(self._h, _p(out), out.size))
        return out

    def color_devptr(self) -> int:
        return int(lib.mtr_group_frame_color_devptr(self._h) or 0)

    def close(self):
        if self._h:
            for f in self.parts:
                f._h = None
            lib.mtr_group_frame_destroy(self._h)
            self._h = None
