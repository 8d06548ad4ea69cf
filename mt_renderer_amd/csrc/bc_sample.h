// bc_sample.h -- ONE texel of a BC1 / BC7 block, for textures kept block-compressed in HBM
// (mtr_device_set_texture_residency(MTR_TEXRES_BLOCKS)): what the reference's texture unit does per fetch
// (device feature TEXTURE_COMPRESSION_BC, src/renderer_app_manager.rs:107; formats src/rtexture.rs:152-161).
// Same arithmetic as the whole-block decoders of k_texture.hip, restricted to one index: the two must agree
// bit for bit (tests/test_gpu_texture_blocks.py compares both against the oracle's decode).
#pragma once
#include "bc7_tables_dev.h"
#include "mtr_internal.h"

namespace mtr {

__device__ __forceinline__ uint32_t bc1_texel(uint2 raw, uint32_t i) {
    const uint32_t c0 = raw.x & 0xffff, c1 = raw.x >> 16;
    const uint32_t s = (raw.y >> (2 * i)) & 3u;
    uint32_t e[2][3];
    const uint32_t c[2] = {c0, c1};
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const uint32_t r5 = (c[k] >> 11) & 31, g6 = (c[k] >> 5) & 63, b5 = c[k] & 31;
        e[k][0] = (r5 << 3) | (r5 >> 2);
        e[k][1] = (g6 << 2) | (g6 >> 4);
        e[k][2] = (b5 << 3) | (b5 >> 2);
    }
    uint32_t out = 0xff000000u;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        uint32_t v;
        if (s < 2u) v = e[s][k];
        else if (c0 > c1) v = s == 2u ? (2 * e[0][k] + e[1][k] + 1) / 3 : (e[0][k] + 2 * e[1][k] + 1) / 3;
        else v = s == 2u ? (e[0][k] + e[1][k] + 1) / 2 : 0u;
        out |= v << (8 * k);
    }
    if (s == 3u && c0 <= c1) out &= 0x00ffffffu;
    return out;
}

// n <= 8 bits at bit position pos >= 1 of the 128-bit block
__device__ __forceinline__ uint32_t bc7_bits(unsigned long long lo, unsigned long long hi, uint32_t pos, uint32_t n) {
    const unsigned long long v = pos >= 64u ? hi >> (pos - 64u) : (lo >> pos) | ((hi << 1) << (63u - pos));
    return (uint32_t)v & ((1u << n) - 1u);
}
// field f of the mode: per-mode parameters packed 4 bits each into a literal (mode 0 in the low nibble)
__device__ __forceinline__ uint32_t bc7_mode_field(unsigned long long packed, uint32_t mode) { return (uint32_t)(packed >> (4u * mode)) & 15u; }

__device__ __forceinline__ uint32_t bc7_weight(uint32_t idx, uint32_t bits) {
    // {0,21,43,64}, {0,9,18,27,37,46,55,64}, {0,4,9,13,17,21,26,30,34,38,43,47,51,55,60,64}, 7 bits each
    const unsigned long long w2 = 0ull | (21ull << 7) | (43ull << 14) | (64ull << 21);
    const unsigned long long w3 = 0ull | (9ull << 7) | (18ull << 14) | (27ull << 21) | (37ull << 28) | (46ull << 35) | (55ull << 42) | (64ull << 49);
    const unsigned long long w4a = 0ull | (4ull << 7) | (9ull << 14) | (13ull << 21) | (17ull << 28) | (21ull << 35) | (26ull << 42) | (30ull << 49);
    const unsigned long long w4b = 34ull | (38ull << 7) | (43ull << 14) | (47ull << 21) | (51ull << 28) | (55ull << 35) | (60ull << 42) | (64ull << 49);
    const unsigned long long t = bits == 2u ? w2 : bits == 3u ? w3 : (idx < 8u ? w4a : w4b);
    return (uint32_t)(t >> (7u * (idx & 7u))) & 127u;
}

__device__ __forceinline__ uint32_t bc7_texel(ulonglong2 raw, uint32_t i) {
    const unsigned long long lo = raw.x, hi = raw.y;
    const uint32_t byte0 = (uint32_t)lo & 0xffu;
    if (byte0 == 0u) return 0u;  // reserved mode: transparent black
    const uint32_t mode = (uint32_t)__ffs((int)byte0) - 1u;
    const uint32_t ns = bc7_mode_field(0x21112323ull, mode), pb = bc7_mode_field(0x60006664ull, mode);
    const uint32_t rb = bc7_mode_field(0x00220000ull, mode), isb = bc7_mode_field(0x00010000ull, mode);
    const uint32_t cb = bc7_mode_field(0x57757564ull, mode), ab = bc7_mode_field(0x57860000ull, mode);
    const uint32_t epb = bc7_mode_field(0x11001001ull, mode), spb = bc7_mode_field(0x00000010ull, mode);
    const uint32_t ib = bc7_mode_field(0x24222233ull, mode), ib2 = bc7_mode_field(0x00230000ull, mode);
    uint32_t pos = mode + 1u;
    const uint32_t part = pb ? bc7_bits(lo, hi, pos, pb) : 0u; pos += pb;
    const uint32_t rot = rb ? bc7_bits(lo, hi, pos, rb) : 0u; pos += rb;
    const uint32_t isel = isb ? bc7_bits(lo, hi, pos, isb) : 0u; pos += isb;
    const uint32_t ne = 2u * ns;
    uint32_t s = 0u, anchor1 = 255u, anchor2 = 255u;
    if (ns == 2u) { s = BC7_PART2[part][i]; anchor1 = BC7_ANCHOR2_1[part]; }
    if (ns == 3u) { s = BC7_PART3[part][i]; anchor1 = BC7_ANCHOR3_1[part]; anchor2 = BC7_ANCHOR3_2[part]; }
    const uint32_t cbase = pos, abase = cbase + 3u * ne * cb, pbase = abase + ne * ab;
    const uint32_t ibase1 = pbase + (epb ? ne : spb ? ns : 0u), ibase2 = ibase1 + 16u * ib - ns;
    const bool anch = i == 0u || i == anchor1 || i == anchor2;
    const uint32_t before = (i > 0u ? 1u : 0u) + (i > anchor1 ? 1u : 0u) + (i > anchor2 ? 1u : 0u);  // anchors below i
    const uint32_t idx1 = bc7_bits(lo, hi, ibase1 + i * ib - before, ib - (anch ? 1u : 0u));
    const uint32_t idx2 = ib2 ? bc7_bits(lo, hi, ibase2 + i * ib2 - (i > 0u ? 1u : 0u), ib2 - (i == 0u ? 1u : 0u)) : 0u;
    uint32_t ci = idx1, cbits = ib, ai = idx1, abits = ib;
    if (ib2) {
        if (isel) { ci = idx2; cbits = ib2; }
        else { ai = idx2; abits = ib2; }
    }
    const uint32_t e0 = 2u * s, e1 = e0 + 1u;
    uint32_t p0 = 0u, p1 = 0u;
    if (epb) { p0 = bc7_bits(lo, hi, pbase + e0, 1); p1 = bc7_bits(lo, hi, pbase + e1, 1); }
    else if (spb) { p0 = p1 = bc7_bits(lo, hi, pbase + s, 1); }
    const bool has_p = (epb | spb) != 0u;
    const uint32_t cprec = cb + (has_p ? 1u : 0u), aprec = ab + ((epb && ab) ? 1u : 0u);
    const uint32_t wc = bc7_weight(ci, cbits), wa = bc7_weight(ai, abits);
    uint32_t ch[4];
#pragma unroll
    for (uint32_t c = 0; c < 3; c++) {
        uint32_t a = bc7_bits(lo, hi, cbase + (c * ne + e0) * cb, cb), b = bc7_bits(lo, hi, cbase + (c * ne + e1) * cb, cb);
        if (has_p) { a = (a << 1) | p0; b = (b << 1) | p1; }
        a <<= 8u - cprec; a |= a >> cprec;
        b <<= 8u - cprec; b |= b >> cprec;
        ch[c] = ((64u - wc) * a + wc * b + 32u) >> 6;
    }
    ch[3] = 255u;
    if (ab) {
        uint32_t a = bc7_bits(lo, hi, abase + e0 * ab, ab), b = bc7_bits(lo, hi, abase + e1 * ab, ab);
        if (epb) { a = (a << 1) | p0; b = (b << 1) | p1; }
        a <<= 8u - aprec; a |= a >> aprec;
        b <<= 8u - aprec; b |= b >> aprec;
        ch[3] = ((64u - wa) * a + wa * b + 32u) >> 6;
    }
    if (rot) {
        const uint32_t rc = rot == 1u ? ch[0] : rot == 2u ? ch[1] : ch[2];
        const uint32_t a = ch[3];
        ch[3] = rc;
        if (rot == 1u) ch[0] = a; else if (rot == 2u) ch[1] = a; else ch[2] = a;
    }
    return ch[0] | (ch[1] << 8) | (ch[2] << 16) | (ch[3] << 24);
}

}  // namespace mtr
