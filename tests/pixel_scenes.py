"""Hand-computable scenes: F32x3 vertices given directly in framebuffer pixels (power-of-two target so
the pixel -> NDC -> pixel round trip is exact in binary32), list or strip topology."""
from __future__ import annotations

import numpy as np

from mt_renderer_amd import scene


def pixel_to_ndc_matrix(w: int, h: int) -> np.ndarray:
    """x_ndc = 2x/W - 1, y_ndc = 1 - 2y/H, z passthrough, w = 1 (column-major f32[16])."""
    m = np.eye(4)
    m[0, 0], m[0, 3] = 2.0 / w, -1.0
    m[1, 1], m[1, 3] = -2.0 / h, 1.0
    return scene.to_f32_colmajor(m)


def pixel_model(prims, textures=None, parts_disp=None) -> scene.ModelData:
    """prims: list of dicts {verts: (n,3) or (n,5 with uv), indices, topology, debug_id, texture, parts_no}"""
    vbs, ibs, recs, lays, p2t, dids = [], [], [], [], [], []
    vbase = iofs = 0
    for p in prims:
        v = np.asarray(p["verts"], dtype=np.float32)
        has_uv = v.shape[1] == 5
        stride = 20 if has_uv else 12
        vb = np.zeros((v.shape[0], stride), dtype=np.uint8)
        vb[:, 0:12] = v[:, 0:3].copy().view(np.uint8).reshape(-1, 12)
        lay = [(scene.SEM_POSITION, scene.IEF_F32, 3, 0)]
        if has_uv:
            # texcoord as F32x3 reading (u, v, <next bytes>) is not possible inside the stride; use F16x2
            stride = 16
            vb = np.zeros((v.shape[0], stride), dtype=np.uint8)
            vb[:, 0:12] = v[:, 0:3].copy().view(np.uint8).reshape(-1, 12)
            vb[:, 12:16] = v[:, 3:5].astype("<f2").view(np.uint8).reshape(-1, 4)
            lay.append((scene.SEM_TEXCOORD, scene.IEF_F16, 2, 12))
        ib = np.asarray(p["indices"], dtype=np.uint16)
        recs.append(scene.pack_primitive(vertex_num=v.shape[0], parts_no=p.get("parts_no", 0), vertex_stride=stride,
                                         topology=p.get("topology", scene.TOPO_LIST), vertex_base=vbase, index_ofs=iofs,
                                         index_num=len(ib), index_base=p.get("index_base", 0)))
        vbs.append(vb.reshape(-1))
        ibs.append(ib)
        lays.append(lay)
        p2t.append(p.get("texture", -1))
        dids.append(p.get("debug_id", 0))
        vbase += vb.size
        iofs += len(ib)
    n = len(prims)
    return scene.ModelData(
        vertex_buf=np.concatenate(vbs), index_buf=np.concatenate(ibs), prims=np.stack(recs), layouts=lays,
        prim_to_texture=np.array(p2t, dtype=np.int32), prim_debug_id=np.array(dids, dtype=np.uint32),
        parts_disp=np.ones(n, dtype=np.uint8) if parts_disp is None else np.asarray(parts_disp, dtype=np.uint8),
        textures=list(textures or []))


PALETTE = np.array([  # src/shaders/debug_ids.wgsl:23-44
    (215, 62, 103), (95, 190, 80), (133, 95, 213), (180, 184, 53), (213, 87, 180), (72, 138, 55), (145, 79, 158),
    (91, 196, 153), (206, 78, 55), (74, 174, 209), (225, 133, 58), (92, 122, 198), (207, 162, 81), (188, 144, 216),
    (152, 173, 92), (161, 71, 103), (53, 133, 98), (225, 131, 152), (111, 111, 40), (162, 99, 55)], dtype=np.uint8)
