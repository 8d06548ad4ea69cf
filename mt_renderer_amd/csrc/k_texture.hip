// k_texture.hip -- rTexture block decode on upload (BC1 / BC7 -> RGBA8 resident in HBM).
// The reference hands BC blocks to the GPU's texture unit (device feature TEXTURE_COMPRESSION_BC,
// src/renderer_app_manager.rs:107; formats src/rtexture.rs:152-161); here they are decoded once at
// Texture::new time so the fragment stage samples plain RGBA8.  One thread per 4x4 block.
#include "mtr_internal.h"
#include "bc7_tables_dev.h"

namespace mtr {

__device__ __forceinline__ void store_block(uint8_t* rgba, uint32_t w, uint32_t h, uint32_t bx, uint32_t by,
                                            const uint32_t (&px)[16]) {
#pragma unroll
    for (uint32_t y = 0; y < 4; y++)
#pragma unroll
        for (uint32_t x = 0; x < 4; x++) {
            const uint32_t X = bx * 4 + x, Y = by * 4 + y;
            if (X < w && Y < h) reinterpret_cast<uint32_t*>(rgba)[(size_t)Y * w + X] = px[y * 4 + x];
        }
}

// BC1 (Bc1RgbaUnorm), SPEC.md "BC1"
__global__ __launch_bounds__(256) void k_bc1_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h) {
    const uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bw * bh) return;
    const uint2 raw = reinterpret_cast<const uint2*>(blocks)[b];
    const uint32_t c0 = raw.x & 0xffff, c1 = raw.x >> 16;
    uint32_t col[4][3], alpha3 = 255;
    const uint32_t c[2] = {c0, c1};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const uint32_t r5 = (c[i] >> 11) & 31, g6 = (c[i] >> 5) & 63, b5 = c[i] & 31;
        col[i][0] = (r5 << 3) | (r5 >> 2);
        col[i][1] = (g6 << 2) | (g6 >> 4);
        col[i][2] = (b5 << 3) | (b5 >> 2);
    }
    if (c0 > c1) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            col[2][k] = (2 * col[0][k] + col[1][k] + 1) / 3;
            col[3][k] = (col[0][k] + 2 * col[1][k] + 1) / 3;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            col[2][k] = (col[0][k] + col[1][k] + 1) / 2;
            col[3][k] = 0;
        }
        alpha3 = 0;
    }
    uint32_t pal[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pal[i] = col[i][0] | (col[i][1] << 8) | (col[i][2] << 16) | ((i == 3 ? alpha3 : 255u) << 24);
    uint32_t px[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const uint32_t s = (raw.y >> (2 * i)) & 3;
        px[i] = s == 0 ? pal[0] : s == 1 ? pal[1] : s == 2 ? pal[2] : pal[3];
    }
    store_block(rgba, w, h, b % bw, b / bw, px);
}

// ---- BC7 (Bc7RgbaUnorm), Khronos Data Format Specification "BPTC": bit-exact ----
struct Bits128 {
    unsigned long long lo, hi;
    uint32_t pos;
    __device__ __forceinline__ uint32_t get(uint32_t n) {  // n <= 8
        if (n == 0) return 0;
        unsigned long long v;
        if (pos >= 64) v = hi >> (pos - 64);
        else if (pos + n <= 64) v = lo >> pos;
        else v = (lo >> pos) | (hi << (64 - pos));
        pos += n;
        return (uint32_t)v & ((1u << n) - 1u);
    }
};

__device__ __constant__ static const uint8_t BC7_NS[8] = {3, 2, 3, 2, 1, 1, 1, 2};
__device__ __constant__ static const uint8_t BC7_PB[8] = {4, 6, 6, 6, 0, 0, 0, 6};
__device__ __constant__ static const uint8_t BC7_RB[8] = {0, 0, 0, 0, 2, 2, 0, 0};
__device__ __constant__ static const uint8_t BC7_ISB[8] = {0, 0, 0, 0, 1, 0, 0, 0};
__device__ __constant__ static const uint8_t BC7_CB[8] = {4, 6, 5, 7, 5, 7, 7, 5};
__device__ __constant__ static const uint8_t BC7_AB[8] = {0, 0, 0, 0, 6, 8, 7, 5};
__device__ __constant__ static const uint8_t BC7_EPB[8] = {1, 0, 0, 1, 0, 0, 1, 1};
__device__ __constant__ static const uint8_t BC7_SPB[8] = {0, 1, 0, 0, 0, 0, 0, 0};
__device__ __constant__ static const uint8_t BC7_IB[8] = {3, 3, 2, 2, 2, 2, 4, 2};
__device__ __constant__ static const uint8_t BC7_IB2[8] = {0, 0, 0, 0, 3, 2, 0, 0};
__device__ __constant__ static const uint8_t BC7_WT2[4] = {0, 21, 43, 64};
__device__ __constant__ static const uint8_t BC7_WT3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
__device__ __constant__ static const uint8_t BC7_WT4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

__device__ __forceinline__ uint32_t bc7_lerp(uint32_t e0, uint32_t e1, uint32_t idx, uint32_t bits) {
    const uint32_t wgt = bits == 2 ? BC7_WT2[idx] : bits == 3 ? BC7_WT3[idx] : BC7_WT4[idx];
    return ((64 - wgt) * e0 + wgt * e1 + 32) >> 6;
}

__global__ __launch_bounds__(64) void k_bc7_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h) {
    const uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bw * bh) return;
    const ulonglong2 raw = reinterpret_cast<const ulonglong2*>(blocks)[b];
    uint32_t px[16];
    const uint32_t byte0 = (uint32_t)raw.x & 0xff;
    if (byte0 == 0) {  // reserved mode: transparent black
#pragma unroll
        for (int i = 0; i < 16; i++) px[i] = 0;
        store_block(rgba, w, h, b % bw, b / bw, px);
        return;
    }
    const uint32_t mode = (uint32_t)__ffs((int)byte0) - 1;
    Bits128 br = {raw.x, raw.y, mode + 1};
    const uint32_t ns = BC7_NS[mode], cb = BC7_CB[mode], ab = BC7_AB[mode], ib = BC7_IB[mode], ib2 = BC7_IB2[mode];
    const uint32_t part = br.get(BC7_PB[mode]);
    const uint32_t rot = br.get(BC7_RB[mode]);
    const uint32_t isel = br.get(BC7_ISB[mode]);
    const uint32_t ne = ns * 2;
    uint32_t ep[6][4];
    for (int c = 0; c < 3; c++)
        for (uint32_t e = 0; e < ne; e++) ep[e][c] = br.get(cb);
    for (uint32_t e = 0; e < ne; e++) ep[e][3] = ab ? br.get(ab) : 255u;
    uint32_t cprec = cb, aprec = ab;
    if (BC7_EPB[mode]) {
        for (uint32_t e = 0; e < ne; e++) {
            const uint32_t p = br.get(1);
            for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p;
            if (ab) ep[e][3] = (ep[e][3] << 1) | p;
        }
        cprec++;
        if (ab) aprec++;
    } else if (BC7_SPB[mode]) {
        for (uint32_t s = 0; s < ns; s++) {
            const uint32_t p = br.get(1);
            for (uint32_t e = 2 * s; e < 2 * s + 2; e++)
                for (int c = 0; c < 3; c++) ep[e][c] = (ep[e][c] << 1) | p;
        }
        cprec++;
    }
    for (uint32_t e = 0; e < ne; e++) {
        for (int c = 0; c < 3; c++) {
            const uint32_t v = ep[e][c] << (8 - cprec);
            ep[e][c] = v | (v >> cprec);
        }
        if (ab) {
            const uint32_t v = ep[e][3] << (8 - aprec);
            ep[e][3] = v | (v >> aprec);
        }
    }
    uint32_t anchor1 = 255, anchor2 = 255;
    if (ns == 2) anchor1 = BC7_ANCHOR2_1[part];
    if (ns == 3) { anchor1 = BC7_ANCHOR3_1[part]; anchor2 = BC7_ANCHOR3_2[part]; }
    uint32_t sub[16], i1[16], i2[16];
    for (uint32_t i = 0; i < 16; i++) {
        sub[i] = ns == 1 ? 0u : ns == 2 ? (uint32_t)BC7_PART2[part][i] : (uint32_t)BC7_PART3[part][i];
        const bool anch = i == 0 || i == anchor1 || i == anchor2;
        i1[i] = br.get(ib - (anch ? 1u : 0u));
    }
    for (uint32_t i = 0; i < 16; i++) i2[i] = ib2 ? br.get(ib2 - (i == 0 ? 1u : 0u)) : 0u;
    for (uint32_t i = 0; i < 16; i++) {
        const uint32_t s = sub[i];
        uint32_t ci = i1[i], cbits = ib, ai = i1[i], abits = ib;
        if (ib2) {
            if (isel) { ci = i2[i]; cbits = ib2; }
            else { ai = i2[i]; abits = ib2; }
        }
        uint32_t ch[4];
        for (int c = 0; c < 3; c++) ch[c] = bc7_lerp(ep[2 * s][c], ep[2 * s + 1][c], ci, cbits);
        ch[3] = ab ? bc7_lerp(ep[2 * s][3], ep[2 * s + 1][3], ai, abits) : 255u;
        if (rot) {
            const uint32_t t = ch[3];
            ch[3] = ch[rot - 1];
            ch[rot - 1] = t;
        }
        px[i] = ch[0] | (ch[1] << 8) | (ch[2] << 16) | (ch[3] << 24);
    }
    store_block(rgba, w, h, b % bw, b / bw, px);
}

// minimum alpha over a decoded RGBA8 texture (Texture::new time): 255 <=> the texture is opaque
__global__ __launch_bounds__(256) void k_alpha_min(const uint32_t* rgba, size_t n, uint32_t* out) {
    uint32_t mn = 255;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) mn = min(mn, rgba[i] >> 24);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) mn = min(mn, (uint32_t)__shfl_xor((int)mn, d));
    if ((threadIdx.x & 63) == 0 && mn != 255) atomicMin(out, mn);
}

}  // namespace mtr

void mtr_launch_alpha_min(const uint8_t* rgba, size_t npixels, uint32_t* out_min, hipStream_t s) {
    const uint32_t grid = (uint32_t)((npixels + 255) / 256 < 1024 ? (npixels + 255) / 256 : 1024);
    hipLaunchKernelGGL(mtr::k_alpha_min, dim3(grid ? grid : 1), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(rgba), npixels, out_min);
}

void mtr_launch_bc1_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t s) {
    const uint32_t nb = ((w + 3) / 4) * ((h + 3) / 4);
    hipLaunchKernelGGL(mtr::k_bc1_decode, dim3((nb + 255) / 256), dim3(256), 0, s, blocks, rgba, w, h);
}
void mtr_launch_bc7_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t s) {
    const uint32_t nb = ((w + 3) / 4) * ((h + 3) / 4);
    hipLaunchKernelGGL(mtr::k_bc7_decode, dim3((nb + 63) / 64), dim3(64), 0, s, blocks, rgba, w, h);
}
