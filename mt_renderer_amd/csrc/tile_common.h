// tile_common.h -- fragment-stage helpers shared by the ordered (k_tile.hip) and visibility
// (k_tile_vis.hip) tile kernels.  Every formula is SPEC.md section 7, bit for bit.
#pragma once
#include "bc_sample.h"
#include "mtr_internal.h"

namespace mtr {

// every workgroup of a tile kernel clears its slice of the counter block the next frame on these framebuffers will use
__device__ __forceinline__ void zero_next_counters(const TileParams& P) {
    if (P.zero_words)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < P.zero_nwords; i += gridDim.x * blockDim.x) {
            for (uint32_t k = 0; k < P.nhint; k++)  // the thread that clears a word the host wants to see publishes it first
                if (i - P.hint_word[k] < 2u)
                    __hip_atomic_store(&P.hint_out[2u * P.hint_slot[k] + (i - P.hint_word[k])], 0x80000000u | P.zero_words[i], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
            P.zero_words[i] = 0u;
        }
    if (!P.zero_next) return;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)CTR_NUM; i += gridDim.x * blockDim.x) P.zero_next[i] = 0u;
}

// Start of every tile workgroup.  Returns the frame's overflow flags (final: the kernels that raise them precede this
// one on the stream) and publishes them to the host's status word.  A frame with a flag set has incomplete queues --
// slots that were reserved and never written -- so the tile kernels must not read them: they clear their bin, park the
// fill word and leave; the host re-runs the frame (mtr_frame_wait, the exchange thread) or latches an error.
__device__ __forceinline__ uint32_t tile_prologue(const TileParams& P) {
    zero_next_counters(P);
    const uint32_t ovf = P.fb.counters[CTR_OVERFLOW];
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.host_status)
        __hip_atomic_store(P.host_status, 0x80000000u | ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return ovf;
}

// one record, decoded (the rare long-edged triangle costs a second, dependent load)
__device__ __forceinline__ RecA load_rec(const FrameBuffers& fb, uint32_t r) {
    const RecP p = fb.rec_a[r];
    int4 l = make_int4(0, 0, 0, 0);
    if ((p.q0.z & 0xFFFFu) == MTR_REC_LARGE_SENTINEL) l = fb.rec_l[r];
    return rec_unpack(p, l);
}

// byte / 255, bit-identical to the IEEE division SPEC.md spells, in three instructions instead of the ten of the
// correctly rounded divide expansion: q0 = x * r, e = fma(-q0, 255, x), q = fma(e, r, q0), r = RN(1 / 255).  Verified
// for all 256 bytes with exact rational arithmetic (tests/test_div_exact.py; the bare product x * r is wrong for 126
// of them).  Texel decode and blending do 4-19 of these per fragment: C5 translucent 2.75 -> 2.40 ms.  (The same
// sequence for the SNORM16 / UNORM8 vertex decode made k_geom SLOWER, 37.8 -> 41.9 us, and is not used there.)
__device__ __forceinline__ float unorm8f(uint32_t v) {
    const float x = (float)(v & 0xffu), r = __uint_as_float(0x3b808081u);
    const float q0 = x * r;
    return fmaf(fmaf(-q0, 255.0f, x), r, q0);
}
__device__ __forceinline__ uint32_t quant8(float x) {
    if (!(x > 0.0f)) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    return (uint32_t)rintf(x * 255.0f);
}
__device__ __forceinline__ int32_t clamp_texel(float f, uint32_t n) {
    if (!(f >= 0.0f)) f = 0.0f;
    if (f > (float)(n - 1)) f = (float)(n - 1);
    return (int32_t)f;
}
struct TexRef {
    const uint8_t* tex;
    uint32_t tw, th;
    uint32_t levels;  // DMat::tlevels: mip levels present (the reference uploads one, src/texture.rs:21; more: row f-4)
                      // in the low byte, MTR_TR_* (what `tex` holds) above it
};
// one texel of mip level `level` (lw x lh) as RGBA8: from the image decoded at upload, or from the BC block it lives in
__device__ __forceinline__ uint32_t texel_u32(const TexRef& m, int level, uint32_t lw, int32_t x, int32_t y) {
    const uint32_t res = m.levels >> 8;
    uint32_t w = m.tw, h = m.th;
    size_t off = 0;
    if (res == MTR_TR_RGBA8) {
        for (int l = 0; l < level; l++) { off += (size_t)w * h; w = w > 1u ? w >> 1 : 1u; h = h > 1u ? h >> 1 : 1u; }
        return reinterpret_cast<const uint32_t*>(m.tex)[off + (size_t)y * lw + (size_t)x];
    }
    for (int l = 0; l < level; l++) { off += (size_t)((w + 3u) >> 2) * ((h + 3u) >> 2); w = w > 1u ? w >> 1 : 1u; h = h > 1u ? h >> 1 : 1u; }
    const size_t b = off + (size_t)((uint32_t)y >> 2) * ((lw + 3u) >> 2) + ((uint32_t)x >> 2);
    const uint32_t i = ((uint32_t)y & 3u) * 4u + ((uint32_t)x & 3u);
    if (res == MTR_TR_BC1) return bc1_texel(reinterpret_cast<const uint2*>(m.tex)[b], i);
    return bc7_texel(reinterpret_cast<const ulonglong2*>(m.tex)[b], i);
}
__device__ __forceinline__ void texel_f(const TexRef& m, int32_t x, int32_t y, float (&o)[4]) {
    const uint32_t t = texel_u32(m, 0, m.tw, x, y);
    o[0] = unorm8f(t); o[1] = unorm8f(t >> 8); o[2] = unorm8f(t >> 16); o[3] = unorm8f(t >> 24);
}

// textureSample: clamp-to-edge, mag linear / min nearest (src/texture.rs:33-42).  flt < 0: linear on level 0
// (magnification); flt >= 0: the nearest texel of mip level flt (level 0 always when the texture has one level).
__device__ __forceinline__ void sample_texture(const TexRef& m, float u, float v, int flt, float (&o)[4]) {
    const float fw = (float)m.tw, fh = (float)m.th;
    if (flt >= 0) {
        uint32_t lw = m.tw, lh = m.th;
        for (int l = 0; l < flt; l++) { lw = lw > 1u ? lw >> 1 : 1u; lh = lh > 1u ? lh >> 1 : 1u; }
        const int32_t x = clamp_texel(floorf(u * (float)lw), lw), y = clamp_texel(floorf(v * (float)lh), lh);
        const uint32_t t = texel_u32(m, flt, lw, x, y);
        o[0] = unorm8f(t); o[1] = unorm8f(t >> 8); o[2] = unorm8f(t >> 16); o[3] = unorm8f(t >> 24);
        return;
    }
    float x = u * fw - 0.5f, y = v * fh - 0.5f;
    float x0 = floorf(x), y0 = floorf(y);
    float fx = x - x0, fy = y - y0;
    int32_t ix0 = clamp_texel(x0, m.tw), ix1 = clamp_texel(x0 + 1.0f, m.tw);
    int32_t iy0 = clamp_texel(y0, m.th), iy1 = clamp_texel(y0 + 1.0f, m.th);
    float c00[4], c10[4], c01[4], c11[4];
    texel_f(m, ix0, iy0, c00); texel_f(m, ix1, iy0, c10); texel_f(m, ix0, iy1, c01); texel_f(m, ix1, iy1, c11);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        float top = fmaf(fx, c10[c] - c00[c], c00[c]);
        float bot = fmaf(fx, c11[c] - c01[c], c01[c]);
        o[c] = fmaf(fy, bot - top, top);
    }
}

// SPEC.md section 7: -1 = linear (every derivative product <= 1: magnification), else the mip level of a nearest
// sample: level l while m > 2^(l - 1/2), i.e. m * m > 2^(2l - 1), m = the largest product (NaN: level 0)
__device__ __forceinline__ int filter_select(float dudx, float dvdx, float dudy, float dvdy, uint32_t tw, uint32_t th, uint32_t levels) {
    levels &= 0xffu;  // DMat::tlevels carries MTR_TR_* above the count
    const float fw = (float)tw, fh = (float)th;
    const float a = fabsf(dudx) * fw, b = fabsf(dvdx) * fh, c = fabsf(dudy) * fw, d = fabsf(dvdy) * fh;
    if ((a <= 1.0f) && (b <= 1.0f) && (c <= 1.0f) && (d <= 1.0f)) return -1;
    int level = 0;
    if (levels > 1u) {
        const float m = fmaxf(fmaxf(a, b), fmaxf(c, d));
        const float m2 = m * m;
        float thr = 2.0f;
        while ((uint32_t)level + 1u < levels && m2 > thr) { level++; thr *= 4.0f; }
    }
    return level;
}

// blend 1: SrcAlpha / OneMinusSrcAlpha colour, One / Zero alpha (src/model.rs:243-246); 2: additive, SrcAlpha / One
// (material state, SPEC.md section 10); 0: replace.  UNORM8 store.
__device__ __forceinline__ uint32_t blend_store(uint32_t dst, const float (&src)[4], uint32_t blend) {
    uint32_t out = 0;
    if (blend == 2u) {
        const float a = src[3];
#pragma unroll
        for (int c = 0; c < 3; c++) out |= quant8(fmaf(src[c], a, unorm8f(dst >> (8 * c)))) << (8 * c);
        out |= quant8(src[3]) << 24;
    } else if (blend) {
        const float a = src[3], ia = 1.0f - a;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float d = unorm8f(dst >> (8 * c));
            float t = d * ia;
            out |= quant8(fmaf(src[c], a, t)) << (8 * c);
        }
        out |= quant8(src[3]) << 24;
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) out |= quant8(src[c]) << (8 * c);
    }
    return out;
}

// LDS traffic between lanes of ONE wave: ds operations of a wave complete in order; this only stops
// the compiler from moving accesses across the hand-off point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// XCD-aware bin order: blocks b, b+8, ... share an XCD's L2: give each XCD a contiguous run of this rank's bins
// (own_list is row-major for interleaved / band ownership and super-tile-major for super-tiles).
// Returns false when this block has no bin.
__device__ __forceinline__ bool block_to_bin(const FrameBuffers& fb, uint32_t& bin, uint32_t run = 0) {
    const uint32_t per = (gridDim.x + 7) / 8;
    const uint32_t x = blockIdx.x & 7, i = blockIdx.x >> 3;
    // run == 0: XCD x takes one contiguous eighth of the bins; run > 0: runs of `run` consecutive bins are dealt to the
    // XCDs in turn (the launch covers whole runs), so every XCD sees every region of the frame
    const uint32_t slot = run ? ((i / run) * 8 + x) * run + (i % run) : x * per + i;
    if (slot >= fb.own.own_count) return false;
    bin = fb.own.own_list ? fb.own.own_list[slot] : slot;
    return bin < fb.nbx * fb.nby;
}

}  // namespace mtr
