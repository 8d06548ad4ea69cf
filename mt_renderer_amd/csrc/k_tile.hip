// k_tile.hip -- tile raster + depth + fragment shading + blend + write-out (gfx950, wave64).
//
// One 256-thread workgroup per 32x32-pixel bin; one wave per row of four 8x8 sub-tiles;
// lane = pixel, with that pixel's depth and colour held in registers for the whole bin.
//   1. the bin's segment descriptors are sorted by submission key in LDS (a few dozen items),
//   2. the concatenated, now fully ordered triangle list is set up 128 triangles at a time into
//      LDS (integer edge equations relative to the bin origin, top-left bias folded in),
//   3. each wave walks, per sub-tile and in order, only the triangles whose bbox touches it
//      (one wave ballot per sub-tile per pass), evaluating the three edge functions per lane,
//      depth LessEqual (src/model.rs:255-261), the fragment shader (src/shaders/debug_ids.wgsl,
//      src/shaders/textured.wgsl + sampler src/texture.rs:33-42) and the blend
//      (src/model.rs:240-247) in submission order,
//   4. colour + depth leave the CU once (clear is fused: no separate clear pass).
#include "mtr_internal.h"

namespace mtr {

#define TRI_PASS 128

struct TriS {
    // edge i: E_i(lx,ly) = C_i + A_i*lx + B_i*ly   (lx,ly = pixel offset inside the bin)
    // small class: everything fits i32 and C_i already carries the top-left bias (tl_i - 1);
    // large class: A/B are unscaled (dy, -dx), C is i64 split in Clo/Chi, bias likewise folded in.
    int32_t A[3], B[3], Clo[3], Chi[3];
    int32_t unb[3];  // 1 - tl_i: add back to recover the unbiased edge value
    uint32_t flags;  // bit0 large
    float z0, dz1, dz2, rcpA;
    uint32_t mat, submask;
    float iw0, diw1, diw2, up0, dup1, dup2, vp0, dvp1, dvp2;
    uint32_t pad;
};
static_assert(sizeof(TriS) == 32 * 4, "TriS is 32 dwords");

__device__ __forceinline__ float unorm8f(uint32_t v) { return (float)(v & 0xffu) / 255.0f; }
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
__device__ __forceinline__ void texel_f(const DMat& m, int32_t x, int32_t y, float (&o)[4]) {
    uint32_t t = reinterpret_cast<const uint32_t*>(m.tex)[(size_t)y * m.tw + (size_t)x];
    o[0] = unorm8f(t); o[1] = unorm8f(t >> 8); o[2] = unorm8f(t >> 16); o[3] = unorm8f(t >> 24);
}

// textureSample: clamp-to-edge, mag linear / min nearest, one level (src/texture.rs:21,33-42).
// du/dx etc. are fine quad differences of the per-lane (u,v): SPEC.md "sampling".
__device__ __forceinline__ void sample_texture(const DMat& m, float u, float v, bool linear, float (&o)[4]) {
    const float fw = (float)m.tw, fh = (float)m.th;
    if (!linear) {
        texel_f(m, clamp_texel(floorf(u * fw), m.tw), clamp_texel(floorf(v * fh), m.th), o);
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

__device__ __forceinline__ uint32_t blend_store(uint32_t dst, const float (&src)[4], bool blend) {
    uint32_t out = 0;
    if (blend) {
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

// per-entry set-up by one thread: record -> bin-relative edge equations in LDS
__device__ __forceinline__ void setup_entry(const TileParams& P, uint32_t r, int32_t binx0, int32_t biny0, TriS& t) {
    const RecA a = P.fb.rec_a[r];
    const int32_t X[3] = {a.X0, a.X1, a.X2}, Y[3] = {a.Y0, a.Y1, a.Y2};
    const long long A2 = (long long)(X[2] - X[0]) * (long long)(Y[1] - Y[0]) - (long long)(X[1] - X[0]) * (long long)(Y[2] - Y[0]);
    const int32_t xmin = min(X[0], min(X[1], X[2])), xmax = max(X[0], max(X[1], X[2]));
    const int32_t ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    const bool large = (xmax - xmin) > 16384 || (ymax - ymin) > 16384;
    const long long Px = (long long)binx0 * 256 + 128, Py = (long long)biny0 * 256 + 128;
    // edge 0: v1->v2, edge 1: v2->v0, edge 2: v0->v1;  E = dy*(Px-Xa) - dx*(Py-Ya)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int ia = (i + 1) % 3, ib = (i + 2) % 3;
        const int32_t dx = X[ib] - X[ia], dy = Y[ib] - Y[ia];
        const int32_t tl = (dy > 0 || (dy == 0 && dx < 0)) ? 1 : 0;
        long long C = (long long)dy * (Px - X[ia]) - (long long)dx * (Py - Y[ia]) + (tl - 1);
        t.unb[i] = 1 - tl;
        if (large) {
            t.A[i] = dy; t.B[i] = -dx;
            t.Clo[i] = (int32_t)(uint32_t)(unsigned long long)C;
            t.Chi[i] = (int32_t)(C >> 32);
        } else {
            t.A[i] = dy * 256; t.B[i] = -dx * 256;
            t.Clo[i] = (int32_t)C; t.Chi[i] = 0;
        }
    }
    t.flags = large ? 1u : 0u;
    t.z0 = a.z0; t.dz1 = a.z1 - a.z0; t.dz2 = a.z2 - a.z0;
    t.rcpA = 1.0f / (float)A2;
    t.mat = a.mat;
    // sub-tiles (8x8 px) of this bin touched by the pixel-centre bbox
    int32_t px0 = ((xmin + 127) >> 8) - binx0, px1 = ((xmax - 128) >> 8) - binx0;
    int32_t py0 = ((ymin + 127) >> 8) - biny0, py1 = ((ymax - 128) >> 8) - biny0;
    px0 = max(px0, 0); py0 = max(py0, 0); px1 = min(px1, MTR_BIN - 1); py1 = min(py1, MTR_BIN - 1);
    uint32_t sm = 0;
    if (px0 <= px1 && py0 <= py1) {
        const uint32_t rowbits = ((1u << ((px1 >> 3) + 1)) - 1u) & ~((1u << (px0 >> 3)) - 1u);
        for (int32_t sy = py0 >> 3; sy <= (py1 >> 3); sy++) sm |= rowbits << (4 * sy);
    }
    t.submask = sm;
    if (P.mats[a.mat].shader == MTR_SH_TEXTURED) {
        const RecB b = P.fb.rec_b[r];
        t.iw0 = b.iw0; t.diw1 = b.iw1 - b.iw0; t.diw2 = b.iw2 - b.iw0;
        t.up0 = b.up0; t.dup1 = b.up1 - b.up0; t.dup2 = b.up2 - b.up0;
        t.vp0 = b.vp0; t.dvp1 = b.vp1 - b.vp0; t.dvp2 = b.vp2 - b.vp0;
    }
}

__global__ __launch_bounds__(256) void k_tile(TileParams P) {
    __shared__ unsigned long long s_sort[MTR_SEG_CAP];
    __shared__ uint32_t s_off[MTR_SEG_CAP];
    __shared__ uint32_t s_pre[MTR_SEG_CAP + 1];
    __shared__ __align__(16) TriS s_tri[TRI_PASS];
    __shared__ uint32_t s_wsum[4];
    __shared__ uint32_t s_n;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nbx = P.fb.nbx, nbins = nbx * P.fb.nby;
    const uint32_t world = P.fb.shard_world ? P.fb.shard_world : 1u;
    // XCD-aware bin order: blocks b, b+8, ... share an XCD's L2, give each XCD a contiguous bin range
    const uint32_t per = (gridDim.x + 7) / 8;
    const uint32_t slot = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    const uint32_t bin = slot * world + P.fb.shard_rank;
    if (slot >= (nbins + world - 1 - P.fb.shard_rank) / world || bin >= nbins) return;
    const int32_t binx0 = (int32_t)(bin % nbx) * MTR_BIN, biny0 = (int32_t)(bin / nbx) * MTR_BIN;

    float dep[4];
    uint32_t col[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { dep[i] = P.clear_depth; col[i] = P.clear_rgba8; }

    const uint32_t seg_lo = P.fb.seg_start[bin], S = P.fb.seg_start[bin + 1] - seg_lo;
    const uint32_t ent_lo = P.fb.bin_start[bin];
    const Seg* segs = P.fb.segs + seg_lo;

    // ---- passes over key ranges [lo, hi): a single pass unless the bin holds > MTR_SEG_CAP segments ----
    unsigned long long lo = 0, hi = 1ull << 32;
    if (S > MTR_SEG_CAP) hi = ((1ull << 32) * (MTR_SEG_CAP / 2)) / S + 1;
    while (S != 0 && lo < (1ull << 32)) {
        // gather the segments of this key range
        if (tid == 0) s_n = 0;
        __syncthreads();
        if (S <= MTR_SEG_CAP) {
            for (uint32_t i = tid; i < S; i += 256) {
                const Seg sg = segs[i];
                s_sort[i] = ((unsigned long long)sg.key << 32) | i;
            }
            if (tid == 0) s_n = S;
        } else {
            uint32_t cnt = 0;
            for (uint32_t i = tid; i < S; i += 256) {
                const uint32_t k = segs[i].key;
                if (k >= lo && k < hi) cnt++;
            }
            atomicAdd(&s_n, cnt);
            __syncthreads();
            if (s_n > MTR_SEG_CAP) {  // too many: halve the range (keys are unique, so this ends)
                hi = lo + (hi - lo) / 2;
                __syncthreads();
                continue;
            }
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
            for (uint32_t i = tid; i < S; i += 256) {
                const uint32_t k = segs[i].key;
                if (k >= lo && k < hi) {
                    const uint32_t at = atomicAdd(&s_n, 1u);
                    s_sort[at] = ((unsigned long long)k << 32) | i;
                }
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        if (n) {
            // bitonic sort of n (padded to a power of two) keys
            uint32_t np2 = 1;
            while (np2 < n) np2 <<= 1;
            for (uint32_t i = n + tid; i < np2; i += 256) s_sort[i] = ~0ull;
            __syncthreads();
            for (uint32_t k = 2; k <= np2; k <<= 1)
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t i = tid; i < np2; i += 256) {
                        const uint32_t ixj = i ^ j;
                        if (ixj > i) {
                            const unsigned long long a = s_sort[i], b = s_sort[ixj];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { s_sort[i] = b; s_sort[ixj] = a; }
                        }
                    }
                    __syncthreads();
                }
            // sorted (offset, count) + exclusive prefix of counts
            uint32_t offs[MTR_SEG_CAP / 256], cnts[MTR_SEG_CAP / 256], tsum = 0;
#pragma unroll
            for (int e = 0; e < MTR_SEG_CAP / 256; e++) {
                const uint32_t k = tid * (MTR_SEG_CAP / 256) + e;
                offs[e] = 0; cnts[e] = 0;
                if (k < n) {
                    const Seg sg = segs[(uint32_t)s_sort[k]];
                    offs[e] = sg.off; cnts[e] = sg.cnt;
                }
                tsum += cnts[e];
            }
            uint32_t inc = tsum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d);
                if ((int)lane >= d) inc += t;
            }
            if (lane == 63) s_wsum[wave] = inc;
            __syncthreads();
            uint32_t run = inc - tsum;
            for (uint32_t w = 0; w < wave; w++) run += s_wsum[w];
#pragma unroll
            for (int e = 0; e < MTR_SEG_CAP / 256; e++) {
                const uint32_t k = tid * (MTR_SEG_CAP / 256) + e;
                if (k < n) { s_off[k] = offs[e]; s_pre[k] = run; }
                run += cnts[e];
            }
            if (tid == 255) s_pre[n] = run;  // thread 255 holds the grand total (k >= n adds 0)
            __syncthreads();
            const uint32_t N = s_pre[n];

            // ---- ordered triangle list of this pass, TRI_PASS at a time ----
            for (uint32_t e0 = 0; e0 < N; e0 += TRI_PASS) {
                const uint32_t cntp = min((uint32_t)TRI_PASS, N - e0);
                if (tid < cntp) {
                    const uint32_t e = e0 + tid;
                    uint32_t a = 0, b = n;  // largest k with s_pre[k] <= e
                    while (b - a > 1) {
                        const uint32_t mid = (a + b) >> 1;
                        if (s_pre[mid] <= e) a = mid; else b = mid;
                    }
                    const uint32_t r = P.fb.entries[ent_lo + s_off[a] + (e - s_pre[a])];
                    setup_entry(P, r, binx0, biny0, s_tri[tid]);
                }
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint32_t s = wave * 4 + i;
                    const int32_t lx = (int32_t)((s & 3) * 8 + (lane & 7)), ly = (int32_t)((s >> 2) * 8 + (lane >> 3));
                    const bool in_vp = (uint32_t)(binx0 + lx) < P.fb.W && (uint32_t)(biny0 + ly) < P.fb.H;
                    for (uint32_t half = 0; half * 64 < cntp; half++) {
                        const uint32_t ti = half * 64 + lane;
                        uint64_t m = __ballot(ti < cntp && ((s_tri[ti].submask >> s) & 1u));
                        while (m) {
                            const uint32_t t = __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)m) - 1 + half * 64);
                            m &= m - 1;
                            const TriS& T = s_tri[t];
                            bool inside;
                            float e1f, e2f;
                            if (!(T.flags & 1u)) {
                                const int32_t eb0 = T.Clo[0] + __mul24(T.A[0], lx) + __mul24(T.B[0], ly);
                                const int32_t eb1 = T.Clo[1] + __mul24(T.A[1], lx) + __mul24(T.B[1], ly);
                                const int32_t eb2 = T.Clo[2] + __mul24(T.A[2], lx) + __mul24(T.B[2], ly);
                                inside = (eb0 | eb1 | eb2) >= 0;
                                e1f = (float)(eb1 + T.unb[1]);
                                e2f = (float)(eb2 + T.unb[2]);
                            } else {
                                const long long X = (long long)lx * 256, Y = (long long)ly * 256;
                                long long eb[3];
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    const long long C = ((long long)T.Chi[k] << 32) | (unsigned long long)(uint32_t)T.Clo[k];
                                    eb[k] = C + (long long)T.A[k] * X + (long long)T.B[k] * Y;
                                }
                                inside = (eb[0] | eb[1] | eb[2]) >= 0;
                                e1f = (float)(eb[1] + T.unb[1]);
                                e2f = (float)(eb[2] + T.unb[2]);
                            }
                            const float b1 = e1f * T.rcpA, b2 = e2f * T.rcpA;
                            const float z = fmaf(b2, T.dz2, fmaf(b1, T.dz1, T.z0));
                            const DMat& M = P.mats[T.mat];
                            if (M.shader != MTR_SH_TEXTURED) {
                                const bool pass = inside && in_vp && z >= 0.0f && z <= 1.0f && z <= dep[i];
                                if (pass) { dep[i] = z; col[i] = M.rgba8; }
                            } else {
                                // every lane evaluates (u,v) so quad differences exist for helper pixels too
                                const float iw = fmaf(b2, T.diw2, fmaf(b1, T.diw1, T.iw0));
                                const float up = fmaf(b2, T.dup2, fmaf(b1, T.dup1, T.up0));
                                const float vp = fmaf(b2, T.dvp2, fmaf(b1, T.dvp1, T.vp0));
                                const float u = up / iw, v = vp / iw;
                                const float dudx = __shfl(u, (int)(lane | 1)) - __shfl(u, (int)(lane & ~1u));
                                const float dvdx = __shfl(v, (int)(lane | 1)) - __shfl(v, (int)(lane & ~1u));
                                const float dudy = __shfl(u, (int)(lane | 8)) - __shfl(u, (int)(lane & ~8u));
                                const float dvdy = __shfl(v, (int)(lane | 8)) - __shfl(v, (int)(lane & ~8u));
                                const float fw = (float)M.tw, fh = (float)M.th;
                                const bool linear = (fabsf(dudx) * fw <= 1.0f) && (fabsf(dvdx) * fh <= 1.0f) &&
                                                    (fabsf(dudy) * fw <= 1.0f) && (fabsf(dvdy) * fh <= 1.0f);
                                const bool pass = inside && in_vp && z >= 0.0f && z <= 1.0f && z <= dep[i];
                                if (pass) {
                                    dep[i] = z;
                                    float src[4];
                                    sample_texture(M, u, v, linear, src);
                                    col[i] = blend_store(col[i], src, M.blend != 0);
                                }
                            }
                        }
                    }
                }
                __syncthreads();
            }
        }
        lo = hi;
        if (S > MTR_SEG_CAP) {
            unsigned long long width = ((1ull << 32) * (MTR_SEG_CAP / 2)) / S + 1;
            hi = lo + width;
            if (hi > (1ull << 32)) hi = 1ull << 32;
        }
        __syncthreads();
    }

    // ---- write-out: the only framebuffer traffic of the frame ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t s = wave * 4 + i;
        const uint32_t x = (uint32_t)binx0 + (s & 3) * 8 + (lane & 7), y = (uint32_t)biny0 + (s >> 2) * 8 + (lane >> 3);
        if (x < P.fb.W && y < P.fb.H) {
            const size_t pi = (size_t)y * P.fb.W + x;
            reinterpret_cast<uint32_t*>(P.color)[pi] = col[i];
            P.depth[pi] = dep[i];
        }
    }
}

}  // namespace mtr

void mtr_launch_tile(const TileParams& p, hipStream_t s) {
    const uint32_t nbins = p.fb.nbx * p.fb.nby;
    const uint32_t world = p.fb.shard_world ? p.fb.shard_world : 1u;
    uint32_t mine = (nbins + world - 1 - p.fb.shard_rank) / world;
    if (mine == 0) return;
    uint32_t grid = (mine + 7) / 8 * 8;
    hipLaunchKernelGGL(mtr::k_tile, dim3(grid), dim3(256), 0, s, p);
}
