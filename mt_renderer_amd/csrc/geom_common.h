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

__device__ __forceinline__ float snorm16f(uint32_t lo16) {
    float f = (float)(int16_t)lo16 / 32767.0f;
    return f < -1.0f ? -1.0f : f;
}
__device__ __forceinline__ float snorm8f(uint32_t lo8) {
    float f = (float)(int8_t)lo8 / 127.0f;
    return f < -1.0f ? -1.0f : f;
}
__device__ __forceinline__ float unorm8f(uint32_t lo8) { return (float)(lo8 & 0xffu) / 255.0f; }
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
    default: break;
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
// bin iteration shared by k_geom (count) and k_fill (fill): one round = up to 64 records, one per
// lane, in record order.  Lanes whose current bin equals the wave-minimum current bin form a group;
// f(bin, group_mask, is_member) runs once per group, groups in increasing bin order, so both
// kernels see identical (bin, count) sequences.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bin_owned(uint32_t bin, uint32_t rank, uint32_t world) {
    return world <= 1 || (bin % world) == rank;
}

template <class F>
__device__ __forceinline__ void for_each_bin_group(RecHdr h, bool act, uint32_t nbx, uint32_t rank, uint32_t world, F f) {
    uint32_t bx = h.bx0, by = h.by0;
    // position on the first owned bin
    while (act && !bin_owned(by * nbx + bx, rank, world)) {
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
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) b = min(b, (uint32_t)__shfl_xor((int)b, d));
        b = __builtin_amdgcn_readfirstlane(b);
        bool hit = act && mybin == b;
        uint64_t m = __ballot(hit);
        f(b, m, hit);
        if (hit) {
            do {
                if (++bx > h.bx1) { bx = h.bx0; if (++by > h.by1) act = false; }
            } while (act && !bin_owned(by * nbx + bx, rank, world));
        }
    }
}

}  // namespace mtr
