"""Synthetic rModel / rTexture data in the reference's own byte formats (host side, numpy only).

Everything here produces *inputs*: packed ``PrimitiveInfo`` records (0x38 bytes, bit-fields of
/root/reference src/rmodel.rs:135-171), raw vertex bytes, u16 strip indices with 0xFFFF restarts
(src/model.rs:251), decoded input-layout elements (src/rshader2.rs:425-442) and rTexture payloads
(src/rtexture.rs:152-161).  The same objects are fed to the HIP library and to the CPU oracle.

Scene definitions follow SURVEY.md section 8(d): ``mesh50k`` (125x200-quad strip grid on a capsule,
24-byte stride, 64 bones), the 20-primitive headline model (1 000 000 triangles) and the
instanced lattices of configs C3-C5.
"""
from __future__ import annotations

import dataclasses
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

# semantics / formats -- numeric values of InputElementFormat (src/rshader2.rs:73-90)
SEM_POSITION, SEM_TEXCOORD, SEM_JOINT, SEM_WEIGHT = 0, 1, 2, 3
IEF_F32, IEF_F16, IEF_S16, IEF_U16, IEF_S16N, IEF_U16N = 1, 2, 3, 4, 5, 6
IEF_S8, IEF_U8, IEF_S8N, IEF_U8N, IEF_SCMP3N, IEF_UCMP3N, IEF_U8NL, IEF_COLOR4N = 7, 8, 9, 10, 11, 12, 13, 14

TEX_RGBA8, TEX_BC1, TEX_BC7, TEX_BC7_ALT = 7, 19, 42, 54  # src/rtexture.rs:152-161

TOPO_LIST, TOPO_STRIP = 3, 4  # 4 = TriangleStrip (src/rmodel.rs:121-123); 3 = list (build extension)

Element = Tuple[int, int, int, int]  # (semantic, format, count, byte offset)


@dataclasses.dataclass
class TextureData:
    width: int
    height: int
    fmt: int
    data: bytes
    levels: int = 1  # mip levels in `data`, level 0 first (the reference uploads one: src/texture.rs:21)


@dataclasses.dataclass
class ModelData:
    """The byte-level content of a parsed rModel + the bindings Model::new derives
    (src/model.rs:36-293): per-primitive layout, texture index and debug id."""

    vertex_buf: np.ndarray  # uint8
    index_buf: np.ndarray  # uint16
    prims: np.ndarray  # uint8 [nprims, 0x38]
    layouts: List[List[Element]]
    prim_to_texture: np.ndarray  # int32 [nprims], -1 = none
    prim_debug_id: np.ndarray  # uint32 [nprims]
    parts_disp: np.ndarray  # uint8, default all-true with len = nprims (src/model.rs:270)
    textures: List[TextureData] = dataclasses.field(default_factory=list)
    # material state per primitive, uint8 [nprims, 4] = blend (0 alpha, 1 off, 2 additive), depth write, depth test,
    # cull (0 back, 1 none, 2 front); None = the reference's pipeline state (src/model.rs:240-262).  SPEC 10.
    prim_states: Optional[np.ndarray] = None

    @property
    def nprims(self) -> int:
        return int(self.prims.shape[0])

    def input_triangles(self) -> int:
        """Triangles per SURVEY 8(d): strip positions that form a triangle, before cull/clip,
        over the primitives that parts_disp leaves visible."""
        total = 0
        for p in range(self.nprims):
            f = unpack_primitive(self.prims[p])
            if not self.parts_disp[f["parts_no"]]:
                continue
            idx = self.index_buf[f["index_ofs"] : f["index_ofs"] + f["index_num"]]
            if f["topology"] == TOPO_LIST:
                total += len(idx) // 3
            else:
                cuts = np.concatenate([[-1], np.nonzero(idx == 0xFFFF)[0], [len(idx)]])
                runs = np.diff(cuts) - 1
                total += int(np.clip(runs - 2, 0, None).sum())
        return total


def pack_primitive(*, vertex_num: int, parts_no: int = 0, material_no: int = 0, weight_num: int = 0,
                   vertex_stride: int, topology: int = TOPO_STRIP, vertex_ofs: int = 0, vertex_base: int = 0,
                   inputlayout: int = 0, index_ofs: int = 0, index_num: int = 0, index_base: int = 0,
                   boundary_num: int = 0, draw_mode: int = 0, lod: int = 0xFF) -> np.ndarray:
    """Packs one PrimitiveInfo exactly as src/rmodel.rs:135-171 lays it out."""
    w = np.zeros(14, dtype=np.uint32)
    w[0] = (draw_mode & 0xFFFF) | ((vertex_num & 0xFFFF) << 16)
    w[1] = (parts_no & 0xFFF) | ((material_no & 0xFFF) << 12) | ((lod & 0xFF) << 24)
    w[2] = 1 | ((weight_num & 0x1F) << 3) | ((vertex_stride & 0xFF) << 16) | ((topology & 0x3F) << 24)
    w[3], w[4], w[5], w[6], w[7], w[8] = vertex_ofs, vertex_base, inputlayout, index_ofs, index_num, index_base
    w[9] = (boundary_num & 0xFF) << 8
    return w.view(np.uint8).copy()


def unpack_primitive(b: np.ndarray) -> dict:
    w = np.frombuffer(np.ascontiguousarray(b, dtype=np.uint8).tobytes(), dtype="<u4")
    return dict(
        vertex_num=int((w[0] >> 16) & 0xFFFF), parts_no=int(w[1] & 0xFFF), material_no=int((w[1] >> 12) & 0xFFF),
        weight_num=int((w[2] >> 3) & 0x1F), vertex_stride=int((w[2] >> 16) & 0xFF), topology=int((w[2] >> 24) & 0x3F),
        vertex_ofs=int(w[3]), vertex_base=int(w[4]), inputlayout=int(w[5]), index_ofs=int(w[6]), index_num=int(w[7]),
        index_base=int(w[8]), boundary_num=int((w[9] >> 8) & 0xFF),
    )


# ---------------------------------------------------------------------------------------------
# deterministic random numbers: splitmix64
# ---------------------------------------------------------------------------------------------
def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# ---------------------------------------------------------------------------------------------
# matrices (column-major, M*v; glam conventions of src/camera.rs:30-47)
# ---------------------------------------------------------------------------------------------
def mat_translate(x, y, z) -> np.ndarray:
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def mat_scale(x, y, z) -> np.ndarray:
    return np.diag([x, y, z, 1.0])


def mat_rot_y(a) -> np.ndarray:
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def mat_rot_x(a) -> np.ndarray:
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def perspective_rh(fov_y_rad: float, aspect: float, near: float, far: float) -> np.ndarray:
    """glam 0.25 Mat4::perspective_rh (depth 0..1), as used by src/camera.rs:40-43."""
    h = 1.0 / math.tan(0.5 * fov_y_rad)
    w = h / aspect
    r = far / (near - far)
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1], m[2, 2], m[2, 3], m[3, 2] = w, h, r, r * near, -1.0
    return m


def reference_view_proj(width: int, height: int, position=(-5.0, 0.0, 1.0), yaw_deg=0.0, pitch_deg=0.0,
                        fov_deg=50.0) -> np.ndarray:
    """Camera::view_proj (src/camera.rs:30-47) with the modelviewer defaults
    (src/bin/modelviewer.rs:168: pos (-5,0,1), yaw 0, pitch 0, fov 50; near 0.01, far 50)."""
    view = np.linalg.inv(mat_translate(*position) @ mat_rot_y(math.radians(yaw_deg)) @ mat_rot_x(math.radians(pitch_deg)))
    return perspective_rh(math.radians(fov_deg), width / height, 0.01, 50.0) @ view


def to_f32_colmajor(m: np.ndarray) -> np.ndarray:
    """4x4 (row, col) -> 16 float32 in column-major order (glam / WGSL mat4x4 bytes)."""
    return np.ascontiguousarray(np.asarray(m, dtype=np.float64).T.reshape(16).astype(np.float32))


def bone_palette(nbones: int = 64, t: float = 0.25) -> np.ndarray:
    """bone b = T(0,y_b,0) R_y(theta_b) T(0,-y_b,0), theta_b = 0.35 sin(2 pi b/64 + t).
    Returns float32 [nbones,16], column-major."""
    out = np.zeros((nbones, 16), dtype=np.float32)
    for b in range(nbones):
        yb = -0.8 + 1.6 * (b + 0.5) / nbones
        th = 0.35 * math.sin(2.0 * math.pi * b / 64.0 + t)
        out[b] = to_f32_colmajor(mat_translate(0, yb, 0) @ mat_rot_y(th) @ mat_translate(0, -yb, 0))
    return out


# ---------------------------------------------------------------------------------------------
# meshes
# ---------------------------------------------------------------------------------------------
SKINNED_LAYOUT: List[Element] = [
    (SEM_POSITION, IEF_S16N, 3, 0),   # -> Snorm16x4 (src/rshader2.rs:536-541)
    (SEM_TEXCOORD, IEF_F16, 2, 12),   # -> Float16x2 (src/rshader2.rs:552-555)
    (SEM_JOINT, IEF_U8, 4, 16),       # build extension (reference skips it: src/rshader2.rs:506)
    (SEM_WEIGHT, IEF_U8N, 4, 20),
]
SKINNED_STRIDE = 24


def _capsule_grid(rows: int, cols: int, center, radius: float, height: float):
    """(rows+1)x(cols+1) vertices on a capsule about the local y axis; returns positions, uv, normals."""
    r_idx = np.arange(rows + 1)[:, None]
    c_idx = np.arange(cols + 1)[None, :]
    y = -0.5 * height + height * r_idx / rows
    half_cyl = 0.5 * height - radius
    dy = np.clip(np.abs(y) - half_cyl, 0.0, None)
    rad = np.sqrt(np.clip(radius * radius - dy * dy, 0.0, None))
    phi = 2.0 * math.pi * c_idx / cols
    x = rad * np.cos(phi)
    z = rad * np.sin(phi)
    pos = np.stack([x + center[0], np.broadcast_to(y, x.shape) + center[1], z + center[2]], axis=-1)
    nrm = np.stack([np.cos(phi) * rad / radius, np.broadcast_to(np.sign(y) * dy / radius, x.shape),
                    np.sin(phi) * rad / radius], axis=-1)
    uv = np.stack([np.broadcast_to(c_idx / cols, x.shape), np.broadcast_to(r_idx / rows, x.shape)], axis=-1)
    return pos, uv, nrm


def _strip_indices(rows: int, cols: int) -> np.ndarray:
    """rows strips of 2*(cols+1) indices joined by 0xFFFF (125 strips of 402)."""
    out = []
    for r in range(rows):
        a = r * (cols + 1) + np.arange(cols + 1)
        b = (r + 1) * (cols + 1) + np.arange(cols + 1)
        # (row r+1, row r) interleave makes the outward side counter-clockwise (front, src/model.rs:252)
        s = np.stack([b, a], axis=1).reshape(-1)
        out.append(s)
        if r != rows - 1:
            out.append(np.array([0xFFFF]))
    return np.concatenate(out).astype(np.uint16)


def _skinned_vertex_bytes(pos, uv, nrm, rows: int, cols: int, nbones: int, seed: int) -> np.ndarray:
    nv = (rows + 1) * (cols + 1)
    vb = np.zeros((nv, SKINNED_STRIDE), dtype=np.uint8)
    p = np.clip(np.rint(pos.reshape(nv, 3) * 32767.0), -32767, 32767).astype("<i2")
    vb[:, 0:6] = p.view(np.uint8).reshape(nv, 6)
    vb[:, 6:8] = np.array([0xFF, 0x7F], dtype=np.uint8)  # w = 32767 (ignored: position.xyz)
    n8 = np.clip(np.rint(nrm.reshape(nv, 3) * 127.0), -127, 127).astype(np.int8)
    vb[:, 8:11] = n8.view(np.uint8)
    vb[:, 12:16] = uv.reshape(nv, 2).astype("<f2").view(np.uint8).reshape(nv, 4)
    r_of_v = np.repeat(np.arange(rows + 1), cols + 1)
    b0 = (r_of_v * nbones) // (rows + 1)
    joints = np.minimum(b0[:, None] + np.arange(4)[None, :], nbones - 1).astype(np.uint8)
    vb[:, 16:20] = joints
    rnd = splitmix64(np.uint64(seed) ^ np.arange(nv, dtype=np.uint64))
    raw = np.stack([(rnd >> np.uint64(8 * k)) & np.uint64(0xFF) for k in range(4)], axis=1).astype(np.int64) + 1
    w = (raw * 255) // raw.sum(axis=1, keepdims=True)
    w[:, 0] += 255 - w.sum(axis=1)
    assert (w.sum(axis=1) == 255).all() and (w >= 0).all() and (w <= 255).all()
    vb[:, 20:24] = w.astype(np.uint8)
    return vb.reshape(-1)


def skinned_capsule_model(parts: Sequence[Tuple[Tuple[float, float, float], float, float]], rows: int = 125,
                          cols: int = 200, nbones: int = 64, seed: int = 0x6D74726D6F64,
                          textured: bool = False, textures: Optional[List[TextureData]] = None) -> ModelData:
    """One primitive per ``(center, radius, height)`` capsule, each with its own vertex_base /
    index_ofs as SURVEY 8(d) prescribes for the headline scene."""
    vbs, ibs, prims = [], [], []
    vbase = 0
    iofs = 0
    nv = (rows + 1) * (cols + 1)
    assert nv < 0xFFFF
    for k, (center, radius, height) in enumerate(parts):
        pos, uv, nrm = _capsule_grid(rows, cols, center, radius, height)
        vb = _skinned_vertex_bytes(pos, uv, nrm, rows, cols, nbones, seed + k)
        ib = _strip_indices(rows, cols)
        prims.append(pack_primitive(vertex_num=nv, parts_no=0, material_no=0, weight_num=4,
                                    vertex_stride=SKINNED_STRIDE, topology=TOPO_STRIP, vertex_base=vbase,
                                    index_ofs=iofs, index_num=len(ib), index_base=0, boundary_num=k & 0xFF))
        vbs.append(vb)
        ibs.append(ib)
        vbase += len(vb)
        iofs += len(ib)
    n = len(parts)
    return ModelData(
        vertex_buf=np.concatenate(vbs), index_buf=np.concatenate(ibs), prims=np.stack(prims),
        layouts=[list(SKINNED_LAYOUT) for _ in range(n)],
        prim_to_texture=np.full(n, 0 if textured else -1, dtype=np.int32),
        prim_debug_id=np.arange(n, dtype=np.uint32), parts_disp=np.ones(n, dtype=np.uint8),
        textures=list(textures or []),
    )


def mesh50k(textured: bool = False, textures: Optional[List[TextureData]] = None, rows: int = 125,
            cols: int = 200) -> ModelData:
    """SURVEY 8(d) ``mesh50k``: 50 000 triangles, 25 326 vertices, capsule r=0.35 h=1.6."""
    return skinned_capsule_model([((0.0, 0.0, 0.0), 0.35, 1.6)], rows, cols, textured=textured, textures=textures)


def headline_model(rows: int = 125, cols: int = 200) -> ModelData:
    """20 primitives x mesh50k topology = 1 000 000 triangles, 506 520 vertices, one palette."""
    parts = []
    for j in range(4):
        for i in range(5):
            parts.append(((-0.8 + 0.4 * i, -0.735 + 0.49 * j, 0.0), 0.19, 0.47))
    return skinned_capsule_model(parts, rows, cols)


def headline_transform(width: int, height: int) -> np.ndarray:
    """view_proj * model for the headline / C2 scenes: reference camera, model pushed 2.3 units in
    front of it and stretched to the 16:9 target so that it covers roughly 60 % of the pixels."""
    vp = reference_view_proj(width, height)
    model = mat_translate(-5.0, 0.0, 1.0 - 2.3) @ mat_scale(1.6, 0.9, 1.0)
    return vp @ model


def instance_lattice(nx: int, ny: int, seed: int = 7) -> Tuple[np.ndarray, np.ndarray]:
    """Configs C3-C5: nx*ny instances of mesh50k on a lattice with seeded jitter, each with its own
    palette phase.  Returns (model matrices [n,16], palettes [n,64,16]) as float32 column-major."""
    n = nx * ny
    rnd = splitmix64(np.uint64(seed) + np.arange(3 * n, dtype=np.uint64)).astype(np.float64) / 2.0**64
    mats = np.zeros((n, 16), dtype=np.float32)
    pals = np.zeros((n, 64, 16), dtype=np.float32)
    sx, sy = 3.3 / nx, 1.86 / ny
    s = 0.5 * min(sx / 0.7, sy / 1.6) * 1.9
    for k in range(n):
        i, j = k % nx, k // nx
        x = -1.65 + sx * (i + 0.5) + 0.1 * sx * (rnd[3 * k] - 0.5)
        y = -0.93 + sy * (j + 0.5) + 0.1 * sy * (rnd[3 * k + 1] - 0.5)
        z = -0.3 * rnd[3 * k + 2]
        m = mat_translate(-5.0 + x, y, 1.0 - 2.3 + z) @ mat_rot_y(2.0 * math.pi * rnd[3 * k + 2]) @ mat_scale(s, s, s)
        mats[k] = to_f32_colmajor(m)
        pals[k] = bone_palette(64, t=0.25 + 0.37 * k)
    return mats, pals


# ---------------------------------------------------------------------------------------------
# the reference's debug cube (src/debug_overlay.rs:10-35) as a one-primitive list-topology model
# ---------------------------------------------------------------------------------------------
CUBE_VERTS = np.array([1, 1, -1, 1, -1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1], dtype=np.float32)
CUBE_INDICES = np.array([4, 2, 0, 2, 7, 3, 6, 5, 7, 1, 7, 5, 0, 3, 1, 4, 1, 5, 4, 6, 2, 2, 6, 7, 6, 4, 5, 1, 3, 7, 0, 2, 3, 4, 0, 1],
                        dtype=np.uint16)


def cube_model(debug_id: int = 0) -> ModelData:
    prim = pack_primitive(vertex_num=8, vertex_stride=12, topology=TOPO_LIST, index_num=36)
    return ModelData(
        vertex_buf=CUBE_VERTS.view(np.uint8).copy(), index_buf=CUBE_INDICES.copy(), prims=prim[None, :],
        layouts=[[(SEM_POSITION, IEF_F32, 3, 0)]], prim_to_texture=np.array([-1], dtype=np.int32),
        prim_debug_id=np.array([debug_id], dtype=np.uint32), parts_disp=np.ones(1, dtype=np.uint8),
    )


def cube_transform(width: int, height: int) -> np.ndarray:
    """Config C1: the cube seen from outside with the reference camera."""
    vp = reference_view_proj(width, height)
    return vp @ mat_translate(-5.0, 0.0, 1.0 - 5.0) @ mat_rot_y(0.6) @ mat_rot_x(0.4)


# ---------------------------------------------------------------------------------------------
# textures
# ---------------------------------------------------------------------------------------------
def random_bc7_texture(width: int, height: int, seed: int = 1, opaque_modes_only: bool = False) -> TextureData:
    """Random BC7 blocks with the mode bit forced valid."""
    nb = ((width + 3) // 4) * ((height + 3) // 4)
    r = splitmix64(np.uint64(seed) * np.uint64(0x10001) + np.arange(2 * nb, dtype=np.uint64))
    blocks = r.view(np.uint8).reshape(nb, 16).copy()
    modes = (splitmix64(np.uint64(seed + 99) + np.arange(nb, dtype=np.uint64)) % np.uint64(8)).astype(np.uint8)
    if opaque_modes_only:
        modes = np.array([0, 1, 2, 3], dtype=np.uint8)[modes % 4]
    low_mask = ((1 << (modes.astype(np.uint16) + 1)) - 1).astype(np.uint16)
    b0 = blocks[:, 0].astype(np.uint16)
    blocks[:, 0] = ((b0 & ~low_mask) | (1 << modes.astype(np.uint16))).astype(np.uint8)
    return TextureData(width, height, TEX_BC7, blocks.tobytes())


def random_bc1_texture(width: int, height: int, seed: int = 2) -> TextureData:
    nb = ((width + 3) // 4) * ((height + 3) // 4)
    r = splitmix64(np.uint64(seed) * np.uint64(0x20003) + np.arange(nb, dtype=np.uint64))
    return TextureData(width, height, TEX_BC1, r.view(np.uint8).tobytes())


def checker_rgba8_texture(width: int, height: int, cell: int = 8, alpha: Tuple[int, int] = (255, 255)) -> TextureData:
    y, x = np.mgrid[0:height, 0:width]
    k = ((x // cell) + (y // cell)) & 1
    img = np.zeros((height, width, 4), dtype=np.uint8)
    img[..., 0] = np.where(k, 230, 30) + (x % 7)
    img[..., 1] = np.where(k, 60, 200) + (y % 5)
    img[..., 2] = (x * 255 // max(1, width - 1)).astype(np.uint8)
    img[..., 3] = np.where(k, alpha[0], alpha[1])
    return TextureData(width, height, TEX_RGBA8, img.tobytes())
