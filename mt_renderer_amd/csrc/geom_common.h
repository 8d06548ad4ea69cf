// geom_common.h -- device helpers shared by the geometry and bin-fill kernels.
#pragma once
#include "mtr_internal.h"

namespace mtr {

// ---------------------------------------------------------------------------------------------
// vertex fetch: byte address = vertex_base + (index + index_base) * stride + element.offset
// (src/model.rs:337-342,357-361); format table of src/rshader2.rs:516-564.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld32(const uint8_t* p, bool al4) {
    if (al4) return *reinterpret_cast<const uint32_t*>(p);
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// x / d for a small integer x: bit-identical to the IEEE division (tests/test_div_exact.py) in three instructions instead
// of the ten of the IEEE expansion.  Seven of these per vertex (SNORM16 position, UNORM8 weights).  Rounds 1-2 measured it
// slower in k_geom (the 80-VGPR allocation then spilled in the prologue); with the vertex stage feeding LDS instead of
// living across the whole kernel it is the default.  -DMTR_DIV_IEEE restores the plain division for A/B runs.
__device__ __forceinline__ float div_small(float x, float d, float r) {
#ifndef MTR_DIV_IEEE
    const float q0 = x * r;
    return fmaf(fmaf(-q0, d, x), r, q0);
#else
    (void)r;
    return x / d;
#endif
}
__device__ __forceinline__ float snorm16f(uint32_t lo16) {
    float f = div_small((float)(int16_t)lo16, 32767.0f, __uint_as_float(0x38000100u));
    return f < -1.0f ? -1.0f : f;
}
__device__ __forceinline__ float snorm8f(uint32_t lo8) {
    float f = div_small((float)(int8_t)lo8, 127.0f, __uint_as_float(0x3c010204u));
    return f < -1.0f ? -1.0f : f;
}
__device__ __forceinline__ float unorm8f(uint32_t lo8) { return div_small((float)(lo8 & 0xffu), 255.0f, __uint_as_float(0x3b808081u)); }
__device__ __forceinline__ float half_bits_to_float(uint32_t lo16) {
    return (float)__builtin_bit_cast(_Float16, (unsigned short)lo16);  // v_cvt_f32_f16: exact, denormals kept
}

// Decodes one float-class element to (x,y,z) (w is never consumed: position.xyz / texcoord.xy).
__device__ __forceinline__ void decode_elem(uint32_t fmt, uint32_t cnt, const uint8_t* p, bool al4, float& x,
                                            float& y, float& z) {
    x = 0.0f; y = 0.0f; z = 0.0f;
    switch (fmt) {
    case 10: /* U8N */
    case 13: /* U8NL */
        if (fmt == 10 && cnt == 1) { x = unorm8f(p[0]); y = unorm8f(p[1]); }
        else { uint32_t w = ld32(p, al4); x = unorm8f(w); y = unorm8f(w >> 8); z = unorm8f(w >> 16); }
        break;
    case 9: /* S8N */
        if (cnt == 1) { x = snorm8f(p[0]); y = snorm8f(p[1]); }
        else { uint32_t w = ld32(p, al4); x = snorm8f(w & 0xff); y = snorm8f((w >> 8) & 0xff); z = snorm8f((w >> 16) & 0xff); }
        break;
    case 5: /* S16N */ {
        uint32_t w0 = ld32(p, al4);
        x = snorm16f(w0 & 0xffff); y = snorm16f(w0 >> 16);
        if (cnt == 3) { uint32_t w1 = ld32(p + 4, al4); z = snorm16f(w1 & 0xffff); }
        break;
    }
    case 2: /* F16 x2 */ {
        uint32_t w0 = ld32(p, al4);
        x = half_bits_to_float(w0 & 0xffff); y = half_bits_to_float(w0 >> 16);
        break;
    }
    case 1: /* F32 x3 */
        x = __uint_as_float(ld32(p, al4)); y = __uint_as_float(ld32(p + 4, al4)); z = __uint_as_float(ld32(p + 8, al4));
        break;
    case 11: /* SCMP3N, opted into by MTR_ELEM_DECODE_SCMP3N: three signed 10-bit fields, max(v / 511, -1) */ {
        const uint32_t w = ld32(p, al4);
        const float fx = (float)((int32_t)(w << 22) >> 22) / 511.0f, fy = (float)((int32_t)(w << 12) >> 22) / 511.0f,
                    fz = (float)((int32_t)(w << 2) >> 22) / 511.0f;
        x = fx < -1.0f ? -1.0f : fx; y = fy < -1.0f ? -1.0f : fy; z = fz < -1.0f ? -1.0f : fz;
        break;
    }
    default: break;
    }
}

// The same table decoding an element that is already in registers (w0 = its first four bytes, w1, w2 the next eight;
// little-endian): the vertex stage issues every load of a vertex first and decodes afterwards, so that a wave pays one
// memory round trip per vertex instead of one per element (decode_elem waits for each element's load in turn).
__device__ __forceinline__ void decode_regs(uint32_t fmt, uint32_t cnt, uint32_t w0, uint32_t w1, uint32_t w2, float& x, float& y, float& z) {
    x = 0.0f; y = 0.0f; z = 0.0f;
    switch (fmt) {
    case 10: /* U8N */
    case 13: /* U8NL */
        x = unorm8f(w0); y = unorm8f(w0 >> 8);
        if (!(fmt == 10 && cnt == 1)) z = unorm8f(w0 >> 16);
        break;
    case 9: /* S8N */
        x = snorm8f(w0 & 0xff); y = snorm8f((w0 >> 8) & 0xff);
        if (cnt != 1) z = snorm8f((w0 >> 16) & 0xff);
        break;
    case 5: /* S16N */
        x = snorm16f(w0 & 0xffff); y = snorm16f(w0 >> 16);
        if (cnt == 3) z = snorm16f(w1 & 0xffff);
        break;
    case 2: /* F16 x2 */
        x = half_bits_to_float(w0 & 0xffff); y = half_bits_to_float(w0 >> 16);
        break;
    case 1: /* F32 x3 */
        x = __uint_as_float(w0); y = __uint_as_float(w1); z = __uint_as_float(w2);
        break;
    case 11: /* SCMP3N, opted into by MTR_ELEM_DECODE_SCMP3N: three signed 10-bit fields, max(v / 511, -1) */ {
        const float fx = (float)((int32_t)(w0 << 22) >> 22) / 511.0f, fy = (float)((int32_t)(w0 << 12) >> 22) / 511.0f,
                    fz = (float)((int32_t)(w0 << 2) >> 22) / 511.0f;
        x = fx < -1.0f ? -1.0f : fx; y = fy < -1.0f ? -1.0f : fy; z = fz < -1.0f ? -1.0f : fz;
        break;
    }
    default: break;
    }
}
// bytes of an element the decode reads (host: elem_bytes in mtr_api.cpp); 0 for formats the table does not hold
__device__ __forceinline__ uint32_t elem_nbytes(uint32_t fmt, uint32_t cnt) {
    switch (fmt) {
    case 10: return cnt == 1 ? 2u : 4u;
    case 13: return 4u;
    case 9: return cnt == 1 ? 2u : 4u;
    case 5: return cnt == 1 ? 4u : 8u;
    case 2: return 4u;
    case 1: return 12u;
    case 11: return 4u;
    default: return 0u;
    }
}
// the raw bytes of an element: up to three dwords, loads only (nothing waits here).  The vertex buffer is padded by 16
// bytes on the device, so a whole-dword read of a 2-byte element at the end of the last vertex stays inside it.
__device__ __forceinline__ void load_elem(const uint8_t* p, bool al4, uint32_t nbytes, uint32_t& w0, uint32_t& w1, uint32_t& w2) {
    w0 = 0; w1 = 0; w2 = 0;
    if (nbytes == 0) return;
    if (al4) {
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        w0 = q[0];
        if (nbytes > 4) w1 = q[1];
        if (nbytes > 8) w2 = q[2];
    } else {
        w0 = nbytes >= 4 ? ld32(p, false) : ld16(p);
        if (nbytes > 4) w1 = ld32(p + 4, false);
        if (nbytes > 8) w2 = ld32(p + 8, false);
    }
}

struct VOut {
    float x, y, z, w, u, v;
};

// The vertex shader: linear-blend skinning against the LDS-staged palette (build extension,
// SPEC.md "LBS"), then clip = M * (q,1) (src/shaders/textured.wgsl:15, debug_ids.wgsl:13).
// Both contractions are k-ordered fmaf chains starting from 0 -- the exact arithmetic of the
// f32 MFMA (v_mfma_f32_4x4x1_16b_f32) used by the batched variant in k_geom.
__device__ __forceinline__ VOut shade_vertex(const uint8_t* vbuf, const DPrim& pr, uint32_t vid, const float (&M)[16],
                                             const float* s_pal, uint32_t npal, bool skinned) {
    const uint8_t* vp = vbuf + pr.vertex_base + (size_t)vid * pr.stride;
    const bool al4 = pr.aligned4 != 0;
    float px, py, pz, tu = 0.0f, tv = 0.0f, tz;
    decode_elem(pr.pos_fmt, pr.pos_cnt, vp + pr.pos_off, al4, px, py, pz);
    if (pr.has_uv) decode_elem(pr.uv_fmt, pr.uv_cnt, vp + pr.uv_off, al4, tu, tv, tz);
    float q0 = px, q1 = py, q2 = pz;
    if (skinned) {
        uint32_t jw = ld32(vp + pr.joint_off, al4), ww = ld32(vp + pr.weight_off, al4);
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
        const float pin[4] = {px, py, pz, 1.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t j = (jw >> (8 * k)) & 0xff;
            if (j >= npal) j = npal - 1;
            const float4* P = reinterpret_cast<const float4*>(s_pal + j * 16);
            float wk = unorm8f(ww >> (8 * k));
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float4 col = P[c];
                float s = wk * pin[c];
                a0 = fmaf(col.x, s, a0);
                a1 = fmaf(col.y, s, a1);
                a2 = fmaf(col.z, s, a2);
            }
        }
        q0 = a0; q1 = a1; q2 = a2;
    }
    const float q[4] = {q0, q1, q2, 1.0f};
    float cl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) a = fmaf(M[c * 4 + i], q[c], a);
        cl[i] = a;
    }
    VOut r = {cl[0], cl[1], cl[2], cl[3], tu, tv};
    return r;
}

// ---------------------------------------------------------------------------------------------
// The same vertex shader on the matrix cores: v_mfma_f32_4x4x1_16b_f32 = 16 independent
// (4x1)*(1x4) outer products per wave; block = 4 consecutive lanes, D[lane][v] = A[block*4+v] * B[lane]
// + C, and a k-step chain is bitwise an fmaf chain (tools/mfma_probe.hip, run on gfx950).  One lane =
// one vertex supplies B (its own w_k*p_c, or q_c) and row (lane & 3) of the 4x4 matrix as A:
//   * clip = M * (q,1): A is the wave-uniform M -> 4 MFMAs for 64 vertices;
//   * skinning: A is the bone matrix P[j_k], so a block must share its four joint indices (rows of a
//     skinned mesh do) -> 16 MFMAs; blocks that do not are redone by their lanes with the VALU chain,
//     which is the identical arithmetic.
// Must be called by all 64 lanes of the wave (the MFMA is wave-wide); `active` masks the loads.
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ VOut shade_vertex_mfma(const uint8_t* vbuf, const DPrim& pr, uint32_t vid, bool active,
                                                  const float* s_M, const float* s_pal, uint32_t npal, bool skinned, bool want_uv) {
    const uint32_t lane = threadIdx.x & 63, row = lane & 3;
    const bool al4 = pr.aligned4 != 0;
    float px = 0.0f, py = 0.0f, pz = 0.0f, tu = 0.0f, tv = 0.0f, tz;
    uint32_t jw = 0, ww = 0;
    want_uv = want_uv && pr.has_uv;
    {
        // every load of the vertex first (element sizes are wave-uniform), decode once they are all on their way
        uint32_t p0 = 0, p1 = 0, p2 = 0, u0 = 0, u1 = 0, u2 = 0;
        if (active) {
            const uint8_t* vp = vbuf + pr.vertex_base + (size_t)vid * pr.stride;
            load_elem(vp + pr.pos_off, al4, elem_nbytes(pr.pos_fmt, pr.pos_cnt), p0, p1, p2);
            if (want_uv) load_elem(vp + pr.uv_off, al4, elem_nbytes(pr.uv_fmt, pr.uv_cnt), u0, u1, u2);
            if (skinned) { jw = ld32(vp + pr.joint_off, al4); ww = ld32(vp + pr.weight_off, al4); }
        }
        decode_regs(pr.pos_fmt, pr.pos_cnt, p0, p1, p2, px, py, pz);
        if (want_uv) decode_regs(pr.uv_fmt, pr.uv_cnt, u0, u1, u2, tu, tv, tz);
        // inactive lanes hold zero bits, which every format decodes to 0.0
    }
    float q0 = px, q1 = py, q2 = pz;
    if (skinned) {  // wave-uniform
        const float pin[4] = {px, py, pz, 1.0f};
        const uint32_t jw0 = (uint32_t)__shfl((int)jw, (int)(lane & ~3u));
        const uint64_t okm = __ballot(active && jw == jw0);
        const bool coherent = ((okm >> (lane & ~3u)) & 0xFull) == 0xFull;
        v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t j = (jw0 >> (8 * k)) & 0xff;
            if (j >= npal) j = npal - 1;
            const float wk = unorm8f(ww >> (8 * k));
#pragma unroll
            for (int c = 0; c < 4; c++)
                acc = __builtin_amdgcn_mfma_f32_4x4x1f32(s_pal[j * 16 + c * 4 + row], wk * pin[c], acc, 0, 0, 0);
        }
        q0 = acc[0]; q1 = acc[1]; q2 = acc[2];
        if (__ballot(active && !coherent)) {
            if (active && !coherent) {
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    uint32_t j = (jw >> (8 * k)) & 0xff;
                    if (j >= npal) j = npal - 1;
                    const float4* P = reinterpret_cast<const float4*>(s_pal + j * 16);
                    const float wk = unorm8f(ww >> (8 * k));
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const float4 col = P[c];
                        const float s = wk * pin[c];
                        a0 = fmaf(col.x, s, a0);
                        a1 = fmaf(col.y, s, a1);
                        a2 = fmaf(col.z, s, a2);
                    }
                }
                q0 = a0; q1 = a1; q2 = a2;
            }
        }
    }
    // clip = M * (q, 1): A = row (lane & 3) of the workgroup's matrix, straight from LDS (s_M: 16 floats, column-major)
    const float q[4] = {q0, q1, q2, 1.0f};
    v4f cl = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < 4; c++) cl = __builtin_amdgcn_mfma_f32_4x4x1f32(s_M[c * 4 + row], q[c], cl, 0, 0, 0);
    VOut r = {cl[0], cl[1], cl[2], cl[3], tu, tv};
    return r;
}

// ---------------------------------------------------------------------------------------------
// Geometry culling of sharded frames (multi-GPU v2).  A rank must not spend vertex work on
// geometry that cannot reach one of its bins, and it must never skip geometry that does: the test is conservative.
//
// A BoneBox holds the object-space box (centre c, half extents e) of the vertices that one joint influences.  When the
// weight bytes of every vertex sum to 255 the skinned position is a convex combination of the points P_j * (p,1) over
// the joints j that carry weight, so every clip coordinate of every vertex lies in the union over those joints of
//     [ v_i - r_i - m_i ,  v_i + r_i + m_i ],     v_i = (C * (c,1))_i,   r_i = sum_c |C[c][i]| * e_c,   C = M * P_j,
// where m_i = 2^-16 * (|M| |P_j| (|c|+e, 1))_i covers the rounding of both this evaluation and of the vertex shader's own
// fma chains (about 40 operations at 2^-24 each, relative to the same sum of magnitudes), the error of the weight sum
// (4 * 2^-25) included.  Near-plane clipping only adds convex combinations of clip-space vertices, so the interval also
// holds for the vertices it creates.  With w_lo > 0 the screen rectangle follows by interval division; a pixel of
// slack and 2^-20 of the coordinate cover the divide, the viewport fma and the 1/256 snap.  Anything that cannot be
// bounded (w_lo <= 0, NaN, weights that are not normalised) is kept.
// ---------------------------------------------------------------------------------------------
struct ClipBox {
    float lo[3], hi[3];  // x, y, w
};

// interval of one box under M * [P; 0 0 0 1] (Pm: 16 floats column-major, rows 0..2 used; nullptr: identity)
__device__ __forceinline__ ClipBox box_clip_interval(const BoneBox& b, const float* Pm, const float (&M)[16]) {
    const float cen[3] = {b.cx, b.cy, b.cz}, ext[3] = {b.ex, b.ey, b.ez};
    ClipBox r;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int i = t == 2 ? 3 : t;  // clip x, y, w
        float v = 0.0f, rad = 0.0f, mag = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float C, A;
            if (Pm) {
                C = M[0 + i] * Pm[c * 4 + 0] + M[4 + i] * Pm[c * 4 + 1] + M[8 + i] * Pm[c * 4 + 2];
                A = fabsf(M[0 + i]) * fabsf(Pm[c * 4 + 0]) + fabsf(M[4 + i]) * fabsf(Pm[c * 4 + 1]) + fabsf(M[8 + i]) * fabsf(Pm[c * 4 + 2]);
                if (c == 3) { C += M[12 + i]; A += fabsf(M[12 + i]); }
            } else {
                C = M[c * 4 + i];
                A = fabsf(C);
            }
            if (c < 3) {
                v += C * cen[c];
                rad += fabsf(C) * ext[c];
                mag += A * (fabsf(cen[c]) + ext[c]);
            } else {
                v += C;
                mag += A;
            }
        }
        const float m = mag * 1.52587890625e-05f;  // 2^-16
        r.lo[t] = v - rad - m;
        r.hi[t] = v + rad + m;
    }
    return r;
}

// the composite of one joint for the chunk tests: rows x, y, w of M * [P; 0 0 0 1] and of |M| * |P|
__device__ __forceinline__ CompMat make_comp(const float* Pm, const float (&M)[16]) {
    CompMat cm;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        const int i = t == 2 ? 3 : t;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float C, A;
            if (Pm) {
                C = M[0 + i] * Pm[c * 4 + 0] + M[4 + i] * Pm[c * 4 + 1] + M[8 + i] * Pm[c * 4 + 2];
                A = fabsf(M[0 + i]) * fabsf(Pm[c * 4 + 0]) + fabsf(M[4 + i]) * fabsf(Pm[c * 4 + 1]) + fabsf(M[8 + i]) * fabsf(Pm[c * 4 + 2]);
                if (c == 3) { C += M[12 + i]; A += fabsf(M[12 + i]); }
            } else {
                C = M[c * 4 + i];
                A = fabsf(C);
            }
            cm.C[c * 3 + t] = C;
            cm.A[c * 3 + t] = A;
        }
    }
    return cm;
}

// the same interval as box_clip_interval from a prepared composite: 11 multiply-adds per clip coordinate
__device__ __forceinline__ ClipBox box_comp_interval(const BoneBox& b, const CompMat& cm) {
    const float cen[3] = {b.cx, b.cy, b.cz}, ext[3] = {b.ex, b.ey, b.ez};
    ClipBox r;
#pragma unroll
    for (int t = 0; t < 3; t++) {
        float v = cm.C[9 + t], rad = 0.0f, mag = cm.A[9 + t];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v += cm.C[c * 3 + t] * cen[c];
            rad += fabsf(cm.C[c * 3 + t]) * ext[c];
            mag += cm.A[c * 3 + t] * (fabsf(cen[c]) + ext[c]);
        }
        const float m = mag * 1.52587890625e-05f;  // 2^-16
        r.lo[t] = v - rad - m;
        r.hi[t] = v + rad + m;
    }
    return r;
}

__device__ __forceinline__ bool clipbox_finite(const ClipBox& r) {
    bool ok = true;
#pragma unroll
    for (int t = 0; t < 3; t++) ok = ok && fabsf(r.lo[t]) < 3.0e38f && fabsf(r.hi[t]) < 3.0e38f;  // false for NaN and inf
    return ok;
}

// May geometry whose clip coordinates lie in `u` produce a fragment in a bin of this rank?  (wave-uniform arithmetic)
__device__ __forceinline__ bool clipbox_may_touch_rank(const ClipBox& u, const FrameBuffers& fb) {
    const float xlo = u.lo[0], xhi = u.hi[0], ylo = u.lo[1], yhi = u.hi[1], wlo = u.lo[2], whi = u.hi[2];
    if (!(wlo > 0.0f)) return true;  // reaches w <= 0 (or NaN): no screen bound
    const float sx_lo = xlo >= 0.0f ? xlo / whi : xlo / wlo, sx_hi = xhi >= 0.0f ? xhi / wlo : xhi / whi;
    const float sy_lo = ylo >= 0.0f ? ylo / whi : ylo / wlo, sy_hi = yhi >= 0.0f ? yhi / wlo : yhi / whi;
    const float fW = (float)fb.W, fH = (float)fb.H, hw = 0.5f * fW, hh = 0.5f * fH;
    float fx_lo = sx_lo * hw + hw, fx_hi = sx_hi * hw + hw;
    float fy_lo = hh - sy_hi * hh, fy_hi = hh - sy_lo * hh;
    const float mx = 1.0f + 9.5367431640625e-07f * fmaxf(fabsf(fx_lo), fabsf(fx_hi));  // 1 px + 2^-20 relative
    const float my = 1.0f + 9.5367431640625e-07f * fmaxf(fabsf(fy_lo), fabsf(fy_hi));
    fx_lo -= mx; fx_hi += mx; fy_lo -= my; fy_hi += my;
    if (fx_hi < 0.0f || fx_lo > fW || fy_hi < 0.0f || fy_lo > fH) return false;  // provably off the target (NaN: kept)
    if (!(fx_lo == fx_lo && fx_hi == fx_hi && fy_lo == fy_lo && fy_hi == fy_hi)) return true;
    const uint32_t bx0 = (uint32_t)fminf(fmaxf(fx_lo, 0.0f), fW) >> MTR_BIN_SHIFT, by0 = (uint32_t)fminf(fmaxf(fy_lo, 0.0f), fH) >> MTR_BIN_SHIFT;
    const uint32_t bx1 = min((uint32_t)fminf(fmaxf(fx_hi, 0.0f), fW) >> MTR_BIN_SHIFT, fb.nbx - 1u);
    const uint32_t by1 = min((uint32_t)fminf(fmaxf(fy_hi, 0.0f), fH) >> MTR_BIN_SHIFT, fb.nby - 1u);
    return rect_owned_any(fb.own, min(bx0, fb.nbx - 1u), min(by0, fb.nby - 1u), bx1, by1, fb.nbx);
}

// Is every bin that geometry with clip coordinates in `u` can produce a fragment in a bin of this rank?  (Then no part of
// it needs testing against the rank's border.)  Same rectangle as clipbox_may_touch_rank; false whenever in doubt.
__device__ __forceinline__ bool clipbox_all_in_rank(const ClipBox& u, const FrameBuffers& fb) {
    const float ylo = u.lo[1], yhi = u.hi[1], wlo = u.lo[2], whi = u.hi[2];
    if (!(wlo > 0.0f)) return false;
    const float sy_lo = ylo >= 0.0f ? ylo / whi : ylo / wlo, sy_hi = yhi >= 0.0f ? yhi / wlo : yhi / whi;
    const float fH = (float)fb.H, hh = 0.5f * fH;
    float fy_lo = hh - sy_hi * hh, fy_hi = hh - sy_lo * hh;
    const float my = 1.0f + 9.5367431640625e-07f * fmaxf(fabsf(fy_lo), fabsf(fy_hi));
    fy_lo -= my; fy_hi += my;
    if (!(fy_lo == fy_lo && fy_hi == fy_hi)) return false;
    if (fb.own.world <= 1) {  // an unsharded frame (MTR_GEOM_CULL_ALL_FRAMES): "all in" = wholly on the target, so that the
                              // chunks of an instance that hangs over its edge are still tested one by one
        const float xlo = u.lo[0], xhi = u.hi[0];
        const float sx_lo = xlo >= 0.0f ? xlo / whi : xlo / wlo, sx_hi = xhi >= 0.0f ? xhi / wlo : xhi / whi;
        const float fW = (float)fb.W, hw = 0.5f * fW;
        const float fx_lo = sx_lo * hw + hw, fx_hi = sx_hi * hw + hw;
        const float mx = 1.0f + 9.5367431640625e-07f * fmaxf(fabsf(fx_lo), fabsf(fx_hi));
        return fx_lo - mx >= 0.0f && fx_hi + mx <= fW && fy_lo >= 0.0f && fy_hi <= fH;  // false for NaN
    }
    const uint32_t by0 = (uint32_t)fminf(fmaxf(fy_lo, 0.0f), fH) >> MTR_BIN_SHIFT;
    const uint32_t by1 = min((uint32_t)fminf(fmaxf(fy_hi, 0.0f), fH) >> MTR_BIN_SHIFT, fb.nby - 1u);
    return rect_owned_all(fb.own, min(by0, fb.nby - 1u), by1);
}

// union of intervals inside each row of 16 lanes (DPP rotations; min / max are idempotent, so every lane of a row ends up
// with the row's result): one wave bounds four chunks at once, a chunk has <= 16 boxes
__device__ __forceinline__ float row_min_f32(float v) {
#define MTR_ROR(x, c) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), c, 0xf, 0xf, false))
    v = fminf(v, MTR_ROR(v, 0x128)); v = fminf(v, MTR_ROR(v, 0x124)); v = fminf(v, MTR_ROR(v, 0x122)); v = fminf(v, MTR_ROR(v, 0x121));
    return v;
}
__device__ __forceinline__ float row_max_f32(float v) {
    v = fmaxf(v, MTR_ROR(v, 0x128)); v = fmaxf(v, MTR_ROR(v, 0x124)); v = fmaxf(v, MTR_ROR(v, 0x122)); v = fmaxf(v, MTR_ROR(v, 0x121));
#undef MTR_ROR
    return v;
}
// wave-wide union of the lanes' intervals (lanes that hold none pass lo = +inf, hi = -inf); every lane gets the result:
// the row reduction, then the four rows' results through v_readlane (six ds_bpermute shuffles per value did this before)
__device__ __forceinline__ float wave_min_f32(float v) {
    v = row_min_f32(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fminf(fminf(r0, r1), fminf(r2, r3));
}
__device__ __forceinline__ float wave_max_f32(float v) {
    v = row_max_f32(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---------------------------------------------------------------------------------------------
// bin iteration shared by k_geom (count) and k_fill (fill): one round = up to 64 records, one per
// lane, in record order.  Lanes whose current bin equals the wave-minimum current bin form a group;
// f(bin, group_mask, is_member) runs once per group, groups in increasing bin order, so both
// kernels see identical (bin, count) sequences.
// ---------------------------------------------------------------------------------------------
// wave-wide minimum as a scalar: rotate-and-min inside each row of 16 lanes with DPP (row_ror 8/4/2/1; min is
// idempotent, so rotations give every lane its row's minimum), then four v_readlane + scalar min across the rows.
// Six ds_bpermute shuffles did this before and were the long pole of the binning loop (~700 cycles per group).
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(r0, r1), min(r2, r3));
}

template <class F>
__device__ __forceinline__ void for_each_bin_group(RecHdr h, bool act, const FrameBuffers& fb, F f) {
    const uint32_t nbx = fb.nbx;
    uint32_t bx = h.bx0, by = h.by0;
    // position on the first owned bin
    while (act && !bin_owned(fb.own, bx, by, nbx)) {
        if (++bx > h.bx1) { bx = h.bx0; if (++by > h.by1) act = false; }
    }
    for (;;) {
        uint64_t m_act = __ballot(act);
        if (!m_act) break;
        // every lane walks its bins in increasing order and the wave always serves the SMALLEST current
        // bin, so each bin is served exactly once per round, by all of its lanes together, in lane
        // (= submission) order: one ordered segment per (chunk, round, bin)
        uint32_t mybin = by * nbx + bx;
        uint32_t b = act ? mybin : 0xFFFFFFFFu;
        b = wave_min_u32(b);
        bool hit = act && mybin == b;
        uint64_t m = __ballot(hit);
        f(b, m, hit);
        if (hit) {
            do {
                if (++bx > h.bx1) { bx = h.bx0; if (++by > h.by1) act = false; }
            } while (act && !bin_owned(fb.own, bx, by, nbx));
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One round = up to 64 records of a chunk's run, lane = record.  A record whose bin rectangle holds more
// than MTR_WIDE_BINS bins (a big triangle) is "wide": walking its bins with one lane would serialise the
// wave (a 12-triangle cube cost 1 ms that way), so wide records are emitted cooperatively, lane = bin.
// To keep submission order exact a wide record splits the round: the records before it are grouped and
// emitted first, then the wide record, then the rest.  Every segment is keyed by the submission order of
// its first entry, which makes segment keys unique and totally ordered per bin.
//   on_groups(act_sub): bin grouping of the lanes with act_sub set;  on_wide(wl): emit the record of lane wl.
// ---------------------------------------------------------------------------------------------
#define MTR_WIDE_BINS 16

template <class FG, class FW>
__device__ __forceinline__ void walk_round(RecHdr h, bool act, uint32_t lane, FG on_groups, FW on_wide) {
    const uint32_t nb = act ? (uint32_t)(h.bx1 - h.bx0 + 1) * (uint32_t)(h.by1 - h.by0 + 1) : 0u;
    uint64_t mw = __ballot(nb > MTR_WIDE_BINS);
    if (!mw) {
        on_groups(act);
        return;
    }
    uint32_t lo = 0;
    for (;;) {
        const uint32_t wl = mw ? __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)mw) - 1) : 64u;
        on_groups(act && lane >= lo && lane < wl && nb <= MTR_WIDE_BINS);
        if (wl == 64) break;
        on_wide(wl);
        mw &= mw - 1;
        lo = wl + 1;
    }
}

// bins of a wide record: lane i of a 64-lane step serves bin number `i` of the rectangle (row-major)
template <class F>
__device__ __forceinline__ void for_each_wide_bin(RecHdr hw, uint32_t lane, const FrameBuffers& fb, F f) {
    const uint32_t w = (uint32_t)(hw.bx1 - hw.bx0 + 1), n = w * (uint32_t)(hw.by1 - hw.by0 + 1);
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t bx = hw.bx0 + i % w, by = hw.by0 + i / w;
        if (bin_owned(fb.own, bx, by, fb.nbx)) f(by * fb.nbx + bx);
    }
}

__device__ __forceinline__ RecHdr hdr_of_lane(RecHdr h, uint32_t wl) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)h.bx0 | ((uint32_t)h.by0 << 16)), wl);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)((uint32_t)h.bx1 | ((uint32_t)h.by1 << 16)), wl);
    RecHdr r = {(uint16_t)(lo & 0xffff), (uint16_t)(lo >> 16), (uint16_t)(hi & 0xffff), (uint16_t)(hi >> 16)};
    return r;
}

// count pass of the two-pass path (k_geom<false>): one non-returning atomic per (wave, bin) group / wide bin
__device__ __forceinline__ void count_bins(const FrameBuffers& fb, RecHdr h, bool act, uint32_t lane) {
    walk_round(
        h, act, lane,
        [&](bool act_sub) {
            for_each_bin_group(h, act_sub, fb, [&](uint32_t bin, uint64_t m, bool hit) {
                if (hit && lane == (uint32_t)__ffsll((long long)m) - 1)
                    atomicAdd(&fb.bin_count[bin], (unsigned long long)__popcll(m) | (1ull << 32));
            });
        },
        [&](uint32_t wl) {
            for_each_wide_bin(hdr_of_lane(h, wl), lane, fb,
                              [&](uint32_t bin) { atomicAdd(&fb.bin_count[bin], 1ull | (1ull << 32)); });
        });
}

// ---------------------------------------------------------------------------------------------
// Hands one round of records (lane = record `round*64 + lane` of chunk `gid`'s run) to the bin queues.
// The (bin, lanes) groups are enumerated first; queue space for up to 64 groups is then reserved by ONE
// wave-wide returning atomic (lane g reserves for group g), so the atomic round trip is paid once per
// round instead of once per group; finally every member lane writes its entry and every group leader its
// segment descriptor.  DIRECT: bounded per-bin queues (single-pass binning, k_geom); otherwise the exact
// two-pass layout positioned by k_scan (k_fill).
// ---------------------------------------------------------------------------------------------
template <bool DIRECT>
__device__ __forceinline__ void emit_bins(const FrameBuffers& fb, RecHdr h, bool act, uint32_t gid, uint32_t round, uint32_t lane) {
    const uint32_t ord0 = gid * 128u + round * 64u;  // submission order of lane 0's record
    auto put = [&](uint32_t bin, unsigned long long t, uint32_t cnt, uint32_t rank, uint32_t order, bool leader, uint32_t key) {
        const uint32_t off = (uint32_t)t, si = (uint32_t)(t >> 32);
        const uint32_t qb = DIRECT ? bin * fb.qcap : fb.bin_start[bin];
        const uint32_t sb = DIRECT ? bin * fb.scap : fb.seg_start[bin];
        if (!DIRECT || (off + cnt <= fb.qcap && si < fb.scap)) {
            fb.entries[qb + off + rank] = order;
            if (leader) {
                Seg sg = {key, off, cnt, 0u};
                fb.segs[sb + si] = sg;
            }
        } else if (leader) {
            atomicOr(&fb.counters[CTR_OVERFLOW], 4u);
        }
    };
    walk_round(
        h, act, lane,
        [&](bool act_sub) {
            uint32_t gbin = 0, ng = 0;
            uint64_t gmask = 0;
            auto flush = [&]() {
                if (ng == 0) return;
                unsigned long long t = 0;
                if (lane < ng) t = atomicAdd(&fb.bin_fill[gbin], (unsigned long long)__popcll(gmask) | (1ull << 32));
                for (uint32_t gi = 0; gi < ng; gi++) {
                    const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gmask >> 32), gi) << 32) |
                                       (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gmask, gi);
                    const unsigned long long tg = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(t >> 32), gi) << 32) |
                                                  (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, gi);
                    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)gbin, gi);
                    if ((m >> lane) & 1ull) {
                        const uint32_t first = (uint32_t)__ffsll((long long)m) - 1;
                        put(b, tg, (uint32_t)__popcll(m), (uint32_t)__popcll(m & ((1ull << lane) - 1ull)), ord0 + lane, lane == first,
                            ord0 + first);
                    }
                }
                ng = 0;
            };
            for_each_bin_group(h, act_sub, fb, [&](uint32_t bin, uint64_t m, bool) {
                if (lane == ng) { gbin = bin; gmask = m; }
                if (++ng == 64) flush();
            });
            flush();
        },
        [&](uint32_t wl) {
            for_each_wide_bin(hdr_of_lane(h, wl), lane, fb, [&](uint32_t bin) {
                const unsigned long long t = atomicAdd(&fb.bin_fill[bin], 1ull | (1ull << 32));
                put(bin, t, 1u, 0u, ord0 + wl, true, ord0 + wl);
            });
        });
}

// ---------------------------------------------------------------------------------------------
// Single-pass binning for frames the visibility-key tile kernel renders (every material opaque): the winner of a
// pixel does not depend on the order of the queue, so there is no order to keep and no segment to describe.
// Groups form around the FIRST ACTIVE lane's current bin (one v_readlane instead of a wave-wide minimum); a bin may
// then be served more than once per round, which only costs one more reservation.  Same queues, same fill words
// (entries in the low half, reservations in the high half) as the ordered builder.
// ---------------------------------------------------------------------------------------------
typedef unsigned short mtr_us2 __attribute__((ext_vector_type(2)));
// both 16-bit halves at once (v_pk_min_u16 / v_pk_max_u16): wave-wide minimum / maximum of a packed (x, y) pair
template <bool MAX>
__device__ __forceinline__ uint32_t pk_minmax(uint32_t a, uint32_t b) {
    const mtr_us2 x = __builtin_bit_cast(mtr_us2, a), y = __builtin_bit_cast(mtr_us2, b);
    return __builtin_bit_cast(uint32_t, MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
}
template <bool MAX>
__device__ __forceinline__ uint32_t wave_pk_minmax(uint32_t v) {
    v = pk_minmax<MAX>(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    v = pk_minmax<MAX>(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = pk_minmax<MAX>(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
    v = pk_minmax<MAX>(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return __builtin_amdgcn_readfirstlane(pk_minmax<MAX>(pk_minmax<MAX>(r0, r1), pk_minmax<MAX>(r2, r3)));
}

// `slot`: 128 dwords of LDS private to the wave (entry counts, then queue offsets, of an 8x8 window of bins).
__device__ __forceinline__ void emit_bins_unordered(const FrameBuffers& fb, RecHdr h, bool act, uint32_t gid, uint32_t round, uint32_t lane,
                                                    uint32_t* slot) {
    const uint32_t ord0 = gid * 128u + round * 64u;
    const uint32_t nb = act ? (uint32_t)(h.bx1 - h.bx0 + 1) * (uint32_t)(h.by1 - h.by0 + 1) : 0u;
    const uint64_t lt = (1ull << lane) - 1ull;
    // ---- fast path, lane-parallel: every record covers <= 4 bins and the round's bins fit an 8x8 window (a strip
    //      chunk of small triangles always does).  Each lane counts itself into the LDS slot of each of its bins
    //      (the returned count is its place in the group), lane s then reserves queue space for slot s with one
    //      global atomic, and every lane stores its entries: no loop over groups at all. ----
    {
        // The window: a chunk's triangles are neighbours on screen, so try the 8 x 8 bins that start three bins up and left of the
        // first active lane's rectangle (two v_readlane and a ballot); only when some lane does not fit is the round's true bounding
        // window worked out (two wave-wide packed min / max reductions, ~36 instructions: what every round used to pay).
        const uint64_t am = __ballot(act);
        if (!am) return;
        const uint32_t fl = (uint32_t)__ffsll((long long)am) - 1u;
        const uint32_t fx = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h.bx0, fl), fy = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)h.by0, fl);
        uint32_t wx0 = fx > 3u ? fx - 3u : 0u, wy0 = fy > 3u ? fy - 3u : 0u;
        bool fits = !__ballot(act && !((uint32_t)h.bx0 >= wx0 && (uint32_t)h.bx1 < wx0 + 8u && (uint32_t)h.by0 >= wy0 && (uint32_t)h.by1 < wy0 + 8u));
        if (!fits) {
            const uint32_t lo = wave_pk_minmax<false>(act ? ((uint32_t)h.bx0 | ((uint32_t)h.by0 << 16)) : 0xFFFFFFFFu);
            const uint32_t hi = wave_pk_minmax<true>(act ? ((uint32_t)h.bx1 | ((uint32_t)h.by1 << 16)) : 0u);
            wx0 = lo & 0xffffu; wy0 = lo >> 16;
            fits = (hi & 0xffffu) - wx0 < 8u && (hi >> 16) - wy0 < 8u;
        }
        if (fits && !__ballot(nb > 4u)) {
            const uint32_t w = (uint32_t)(h.bx1 - h.bx0) + 1u;  // 1..4; w >= 3 means a single row
            slot[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            uint32_t ranks = 0, own = 0;
            for (uint32_t j = 0; j < 4u; j++) {
                const bool on = j < nb;
                if (!__ballot(on)) break;
                const uint32_t dx = w == 1u ? 0u : (w == 2u ? (j & 1u) : j), dy = w == 1u ? j : (w == 2u ? (j >> 1) : 0u);
                const uint32_t bx = h.bx0 + dx, by = h.by0 + dy;
                if (on && bin_owned(fb.own, bx, by, fb.nbx)) {
                    ranks |= atomicAdd(&slot[(by - wy0) * 8u + (bx - wx0)], 1u) << (8u * j);
                    own |= 1u << j;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t c = slot[lane];
            if (c) {
                const uint32_t bin = (wy0 + (lane >> 3)) * fb.nbx + wx0 + (lane & 7u);
                uint32_t o = (uint32_t)atomicAdd(&fb.bin_fill[bin], (unsigned long long)c | (1ull << 32));
                if (o + c > fb.qcap) { atomicOr(&fb.counters[CTR_OVERFLOW], 4u); o = 0x80000000u; }
                slot[64 + lane] = o;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t j = 0; j < 4u; j++) {
                if (!__ballot((own >> j) & 1u)) { if (!__ballot(j < nb)) break; continue; }
                const uint32_t dx = w == 1u ? 0u : (w == 2u ? (j & 1u) : j), dy = w == 1u ? j : (w == 2u ? (j >> 1) : 0u);
                const uint32_t bx = h.bx0 + dx, by = h.by0 + dy;
                if ((own >> j) & 1u) {
                    const uint32_t o = slot[64 + (by - wy0) * 8u + (bx - wx0)];
                    if (!(o & 0x80000000u)) fb.entries[(by * fb.nbx + bx) * fb.qcap + o + ((ranks >> (8u * j)) & 0xffu)] = ord0 + lane;
                }
            }
            // the next round (or the caller) may reuse the slots at once: LDS operations of one wave are ordered
            return;
        }
    }
    uint32_t gbin = 0, ng = 0;
    uint64_t gmask = 0;
    auto flush = [&]() {
        if (ng == 0) return;
        uint32_t off = 0;
        if (lane < ng) {
            const uint32_t cnt = (uint32_t)__popcll(gmask);
            off = (uint32_t)atomicAdd(&fb.bin_fill[gbin], (unsigned long long)cnt | (1ull << 32));
            if (off + cnt > fb.qcap) atomicOr(&fb.counters[CTR_OVERFLOW], 4u);
        }
        for (uint32_t gi = 0; gi < ng; gi++) {
            const uint64_t m = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gmask >> 32), gi) << 32) |
                               (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gmask, gi);
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)off, gi);
            const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)gbin, gi);
            if (((m >> lane) & 1ull) && o + (uint32_t)__popcll(m) <= fb.qcap)
                fb.entries[b * fb.qcap + o + (uint32_t)__popcll(m & lt)] = ord0 + lane;
        }
        ng = 0;
    };
    bool a = act && nb <= MTR_WIDE_BINS;
    uint32_t bx = h.bx0, by = h.by0;
    while (a && !bin_owned(fb.own, bx, by, fb.nbx)) {
        if (++bx > h.bx1) { bx = h.bx0; if (++by > h.by1) a = false; }
    }
    for (;;) {
        const uint64_t m_act = __ballot(a);
        if (!m_act) break;
        const uint32_t mybin = by * fb.nbx + bx;
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)mybin, (uint32_t)__ffsll((long long)m_act) - 1);
        const bool hit = a && mybin == b;
        const uint64_t m = __ballot(hit);
        if (lane == ng) { gbin = b; gmask = m; }
        if (++ng == 64) flush();
        if (hit) {
            do {
                if (++bx > h.bx1) { bx = h.bx0; if (++by > h.by1) a = false; }
            } while (a && !bin_owned(fb.own, bx, by, fb.nbx));
        }
    }
    flush();
    // big triangles: lane = bin of the rectangle, one reservation each
    for (uint64_t mw = __ballot(nb > MTR_WIDE_BINS); mw; mw &= mw - 1) {
        const uint32_t wl = (uint32_t)__ffsll((long long)mw) - 1;
        for_each_wide_bin(hdr_of_lane(h, wl), lane, fb, [&](uint32_t bin) {
            const uint32_t o = (uint32_t)atomicAdd(&fb.bin_fill[bin], 1ull | (1ull << 32));
            if (o < fb.qcap) fb.entries[bin * fb.qcap + o] = ord0 + wl;
            else atomicOr(&fb.counters[CTR_OVERFLOW], 4u);
        });
    }
}

}  // namespace mtr
