// k_tile.hip -- tile raster + depth + fragment shading + blend + write-out (gfx950, wave64).
//
// One WAVE per 16x16-pixel bin (64-thread workgroups, no workgroup barriers anywhere): bins are
// the scheduling unit, so the hardware dispatcher balances heavy and light screen regions.
// lane = pixel of an 8x8 sub-tile; the wave owns the bin's four sub-tiles and keeps every pixel's
// depth and colour in registers for the whole bin.
//   1. the bin's segment descriptors (runs of records contributed by one geometry wave, k_bin.hip)
//      are sorted by submission key with an in-register bitonic network (wave shuffles),
//   2. the concatenated, now fully ordered triangle list is set up 64 triangles at a time, one per
//      lane, into LDS (integer edge equations relative to the bin origin, top-left bias folded in),
//   3. a pass of small triangles is resolved through per-pixel FRAGMENT LISTS (the pairs flattened over the wave,
//      each covered pair appends its triangle number to its pixel's list, lane = pixel then walks its list in
//      submission order); any other pass: per sub-tile the wave walks, in order, only the triangles whose bbox
//      touches it (one ballot per sub-tile per pass), evaluating the three edge functions per lane.  Either way: depth LessEqual
//      (src/model.rs:255-261), the fragment shader (src/shaders/debug_ids.wgsl,
//      src/shaders/textured.wgsl + sampler src/texture.rs:33-42) and the blend
//      (src/model.rs:240-247) in submission order,
//   4. colour + depth leave the CU once (clear is fused: no separate clear pass).
#include "tile_common.h"

namespace mtr {

#define TRI_PASS 64
#ifndef FRAG_K
#define FRAG_K 8u  // fragment-list path: triangle numbers kept per pixel and pass (4, 8 or 16)
#endif

// per-triangle set-up in LDS.  TriC (64 B) is read by every (triangle, sub-tile) visit as four
// broadcast ds_read_b128; TriX (64 B) only by textured triangles.
struct TriC {
    // edge i: E_i(lx,ly) = C_i + A_i*lx + B_i*ly   (lx,ly = pixel offset inside the bin); C_i already
    // carries the top-left bias (tl_i - 1).  small class: everything is i32 with A,B pre-scaled by 256;
    // large class: A/B are the unscaled (dy, -dx), C is i64 with its high word in s_chi.
    int32_t A0, B0, C0, A1;
    int32_t B1, C1, A2, B2;
    int32_t C2;
    uint32_t flags;  // bit0 large, bit1 textured, bit2 alpha blend, bit3 additive blend, bits 4..6: (1 - tl_i), added back
                     // for barycentrics, bit7 no depth write, bit8 no depth test, bit9 colour through blend_store
    float z0, dz1;
    float dz2, rcpA;
    uint32_t rgba8;  // quantised source colour of the debug / constant shaders
    uint32_t pad;    // bbox inside the bin (fragment-list path), see setup_entry
};
struct TriX {
    float iw0, diw1, diw2, up0;
    float dup1, dup2, vp0, dvp1;
    float dvp2;
    uint32_t tw, th, tlevels;
    const uint8_t* tex;
    uint64_t pad2;
};
static_assert(sizeof(TriC) == 64 && sizeof(TriX) == 64, "LDS triangle records are 64 B");
enum { TF_LARGE = 1, TF_TEX = 2, TF_BLEND = 4, TF_ADD = 8, TF_NODW = 128, TF_NODT = 256, TF_BLENDSOLID = 512 };

// per-entry set-up by one lane: record -> bin-relative edge equations in LDS
template <bool TEX>
__device__ __forceinline__ void setup_entry(const TileParams& P, uint32_t r, int32_t binx0, int32_t biny0, TriC& t, TriX* x,
                                            int4& chi, uint32_t& submask) {
    const RecA a = load_rec(P.fb, r);
    // a solid record carries its colour and replaces the pixel; anything else names its material
    const bool solid = (a.pad1 & 1u) != 0;
    DMat mat = {};
    if (!solid) mat = P.mats[a.mat];
    const uint32_t mshader = solid ? (uint32_t)MTR_SH_DEBUG : mat.shader;
    const int32_t X[3] = {a.X0, a.X1, a.X2}, Y[3] = {a.Y0, a.Y1, a.Y2};
    const long long A2 = (long long)(X[2] - X[0]) * (long long)(Y[1] - Y[0]) - (long long)(X[1] - X[0]) * (long long)(Y[2] - Y[0]);
    const int32_t xmin = min(X[0], min(X[1], X[2])), xmax = max(X[0], max(X[1], X[2]));
    const int32_t ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    const bool large = (xmax - xmin) > 16384 || (ymax - ymin) > 16384;
    const long long Px = (long long)binx0 * 256 + 128, Py = (long long)biny0 * 256 + 128;
    // edge 0: v1->v2, edge 1: v2->v0, edge 2: v0->v1;  E = dy*(Px-Xa) - dx*(Py-Ya)
    int32_t A[3], B[3], Clo[3], Chi[3];
    uint32_t flags = large ? TF_LARGE : 0u;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int ia = (i + 1) % 3, ib = (i + 2) % 3;
        const int32_t dx = X[ib] - X[ia], dy = Y[ib] - Y[ia];
        const int32_t tl = (dy > 0 || (dy == 0 && dx < 0)) ? 1 : 0;
        const long long C = (long long)dy * (Px - X[ia]) - (long long)dx * (Py - Y[ia]) + (tl - 1);
        flags |= (uint32_t)(1 - tl) << (4 + i);
        A[i] = large ? dy : dy * 256;
        B[i] = large ? -dx : -dx * 256;
        Clo[i] = (int32_t)(uint32_t)(unsigned long long)C;
        Chi[i] = (int32_t)(C >> 32);
    }
    if (TEX && mshader == MTR_SH_TEXTURED) flags |= TF_TEX;
    if (!solid) {
        if (mat.blend == MTR_DB_ALPHA) flags |= TF_BLEND;
        if (mat.blend == MTR_DB_ADD) flags |= TF_ADD;
        if (!(mat.dstate & 1u)) flags |= TF_NODW;
        if (!(mat.dstate & 2u)) flags |= TF_NODT;
        if (mshader != MTR_SH_TEXTURED && mat.blend == MTR_DB_ADD) flags |= TF_BLENDSOLID;
    }
    t.A0 = A[0]; t.B0 = B[0]; t.C0 = Clo[0];
    t.A1 = A[1]; t.B1 = B[1]; t.C1 = Clo[1];
    t.A2 = A[2]; t.B2 = B[2]; t.C2 = Clo[2];
    t.flags = flags;
    t.z0 = a.z0; t.dz1 = a.z1 - a.z0; t.dz2 = a.z2 - a.z0;
    t.rcpA = 1.0f / (float)A2;
    t.rgba8 = solid ? a.pad0 : mat.rgba8;
    chi = make_int4(Chi[0], Chi[1], Chi[2], 0);
    // sub-tiles (8x8 px) of this bin touched by the pixel-centre bbox: bit = sy*2 + sx
    int32_t px0 = ((xmin + 127) >> 8) - binx0, px1 = ((xmax - 128) >> 8) - binx0;
    int32_t py0 = ((ymin + 127) >> 8) - biny0, py1 = ((ymax - 128) >> 8) - biny0;
    uint32_t sm = 0;
    if (px0 <= 15 && px1 >= 0 && py0 <= 15 && py1 >= 0) {
        const uint32_t colbits = (px0 <= 7 ? 1u : 0u) | (px1 >= 8 ? 2u : 0u);
        if (py0 <= 7) sm |= colbits;
        if (py1 >= 8) sm |= colbits << 2;
    }
    submask = sm;
    // bbox clipped to the bin, for the fragment-list path: px0 | py0 << 4 | (w-1) << 8 | (h-1) << 12 | non-empty << 16
    {
        const int32_t cx0 = max(px0, 0), cx1 = min(px1, MTR_BIN - 1), cy0 = max(py0, 0), cy1 = min(py1, MTR_BIN - 1);
        t.pad = (cx0 <= cx1 && cy0 <= cy1) ? ((uint32_t)cx0 | ((uint32_t)cy0 << 4) | ((uint32_t)(cx1 - cx0) << 8) | ((uint32_t)(cy1 - cy0) << 12) | (1u << 16)) : 0u;
    }
    if (TEX && mshader == MTR_SH_TEXTURED) {
        const RecB b = P.fb.rec_b[r];
        x->iw0 = b.iw0; x->diw1 = b.iw1 - b.iw0; x->diw2 = b.iw2 - b.iw0;
        x->up0 = b.up0; x->dup1 = b.up1 - b.up0; x->dup2 = b.up2 - b.up0;
        x->vp0 = b.vp0; x->dvp1 = b.vp1 - b.vp0; x->dvp2 = b.vp2 - b.vp0;
        x->tex = mat.tex; x->tw = mat.tw; x->th = mat.th; x->tlevels = mat.tlevels;
    }
}

template <bool TEX>
__global__ __launch_bounds__(64) void k_tile(TileParams P) {
    __shared__ __align__(16) TriC s_tc[TRI_PASS];
    __shared__ __align__(16) TriX s_tx[TEX ? TRI_PASS : 1];
    __shared__ __align__(16) int4 s_chi[TRI_PASS];
    __shared__ uint32_t s_mask[TRI_PASS];
    __shared__ uint32_t s_off[64], s_pre[65];
    __shared__ Seg s_seg[64];
    // fragment-list path: per pixel a count and up to FRAG_K triangle numbers of the current pass
    __shared__ uint32_t s_cnt[MTR_BIN * MTR_BIN];
    __shared__ __align__(16) uint8_t s_frag[MTR_BIN * MTR_BIN * FRAG_K];
    __shared__ unsigned long long s_start[32];  // <= 2048 pairs per pass: 32 batches of 64
    __shared__ uint32_t s_pm[TRI_PASS];         // pair prefix | magic(bbox width) << 12
    __shared__ uint8_t s_tmap[TRI_PASS];        // compacted index -> triangle of the pass

    const uint32_t ovf = tile_prologue(P);
    const uint32_t lane = threadIdx.x;
    const uint32_t nbx = P.fb.nbx;
    uint32_t bin;
    if (!block_to_bin(P.fb, bin, P.xcd_run)) return;
    if (P.mixed && !P.bin_flag[bin]) return;  // mixed frame: k_tile_vis has rendered this bin (only opaque triangles in it)
    const int32_t binx0 = (int32_t)(bin % nbx) * MTR_BIN, biny0 = (int32_t)(bin / nbx) * MTR_BIN;

    float dep[4];
    uint32_t col[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { dep[i] = P.clear_depth; col[i] = P.clear_rgba8; }
    for (uint32_t i = lane; i < MTR_BIN * MTR_BIN; i += 64) s_cnt[i] = 0u;

    uint32_t seg_lo, S, ent_lo, n_ent;
    bin_queue(P.fb, bin, ent_lo, n_ent, seg_lo, S);
    if (ovf) { n_ent = 0; S = 0; }  // incomplete queues: read nothing, the bin stays cleared (tile_prologue)
    if (P.fb.direct && lane == 0 && n_ent) {  // queue statistics (direct mode has no scan to count them)
        atomicAdd(&P.fb.counters[MTR_CTR(CTR_ENT, bin)], n_ent);
        atomicAdd(&P.fb.counters[MTR_CTR(CTR_SEG, bin)], S);
    }
    const Seg* segs = P.fb.segs + seg_lo;

    // ---- key ranges [lo, hi): one pass when the bin holds <= 64 segments ----
    unsigned long long lo = 0, hi = 1ull << 32, width = 1ull << 32, end = 1ull << 32;
    if (S > 64) {
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
        for (uint32_t i = lane; i < S; i += 64) { const uint32_t k = segs[i].key; kmin = min(kmin, k); kmax = max(kmax, k); }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d)); kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d)); }
        width = ((unsigned long long)(kmax - kmin) + 1) * 40 / S + 1;
        lo = kmin; hi = lo + width; end = (unsigned long long)kmax + 1;
    }
    while (S != 0 && lo < end) {
        uint32_t n;
        uint32_t mkey = 0xFFFFFFFFu, moff = 0, mcnt = 0;
        if (S <= 64) {
            n = S;
            if (lane < S) { const Seg sg = segs[lane]; mkey = sg.key; moff = sg.off; mcnt = sg.cnt; }
        } else {
            uint32_t cnt = 0;
            for (uint32_t i = lane; i < S; i += 64) { const uint32_t k = segs[i].key; cnt += (k >= lo && k < hi) ? 1u : 0u; }
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d);
            if (cnt > 64) {  // keys are unique per bin, so halving always terminates
                hi = lo + (hi - lo) / 2;
                continue;
            }
            n = cnt;
            uint32_t filled = 0;
            for (uint32_t i0 = 0; i0 < S && filled < n; i0 += 64) {
                const uint32_t i = i0 + lane;
                Seg sg = {0, 0, 0, 0};
                bool match = false;
                if (i < S) { sg = segs[i]; match = sg.key >= lo && sg.key < hi; }
                const uint64_t m = __ballot(match);
                if (match) s_seg[filled + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = sg;
                filled += (uint32_t)__popcll(m);
            }
            wave_lds_sync();
            if (lane < n) { const Seg sg = s_seg[lane]; mkey = sg.key; moff = sg.off; mcnt = sg.cnt; }
            wave_lds_sync();
        }
        if (n) {
            // in-register bitonic sort by key (unused lanes hold key 0xFFFFFFFF and sink to the end)
#pragma unroll
            for (uint32_t k = 2; k <= 64; k <<= 1)
#pragma unroll
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    const uint32_t okey = (uint32_t)__shfl_xor((int)mkey, (int)j);
                    const uint32_t ooff = (uint32_t)__shfl_xor((int)moff, (int)j);
                    const uint32_t ocnt = (uint32_t)__shfl_xor((int)mcnt, (int)j);
                    const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                    // one (chunk, round) can contribute several runs to a bin (k_bin.hip groups lanes by
                    // their CURRENT bin); their offsets were handed out in program order by one wave,
                    // so (key, offset) is a total submission order
                    const unsigned long long mk = ((unsigned long long)mkey << 32) | moff, ok = ((unsigned long long)okey << 32) | ooff;
                    const bool take = (lower == up) ? (ok < mk) : (ok > mk);
                    if (take) { mkey = okey; moff = ooff; mcnt = ocnt; }
                }
            // exclusive prefix of the counts over the sorted segments
            uint32_t inc = mcnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)inc, d);
                if ((int)lane >= d) inc += t;
            }
            s_off[lane] = moff;
            s_pre[lane] = inc - mcnt;
            if (lane == 63) s_pre[64] = inc;
            wave_lds_sync();
            const uint32_t N = s_pre[64];

            // ---- ordered triangle list of this range, 64 at a time ----
            for (uint32_t e0 = 0; e0 < N; e0 += TRI_PASS) {
                const uint32_t cntp = min((uint32_t)TRI_PASS, N - e0);
                if (lane < cntp) {
                    const uint32_t e = e0 + lane;
                    uint32_t a = 0, b = 64;  // largest k with s_pre[k] <= e (zero-count tails never win)
#pragma unroll
                    for (int it = 0; it < 6; it++) {
                        const uint32_t mid = (a + b) >> 1;
                        if (s_pre[mid] <= e) a = mid; else b = mid;
                    }
                    const uint32_t ord = P.fb.entries[ent_lo + s_off[a] + (e - s_pre[a])];
                    const uint32_t r = (ord >> 7) * MTR_CHUNK_SLOTS + (ord & 127u);
                    setup_entry<TEX>(P, r, binx0, biny0, s_tc[lane], TEX ? &s_tx[lane] : nullptr, s_chi[lane], s_mask[lane]);
                }
                wave_lds_sync();
                // ---- fragment-list path: a pass of small triangles.  The (triangle, pixel) pairs of the pass are spread
                //      over the wave 64 at a time (as in k_tile_vis.hip); every covered pair appends its triangle number to
                //      its pixel's list; then lane = pixel walks its own short list in submission order: depth test,
                //      shade, blend.  A whole-wave visit per (triangle, 8x8 sub-tile) -- the loop below -- costs the same
                //      for a 2-pixel sliver as for a full sub-tile; this path costs per covered pixel.  Passes with a
                //      64-bit-edge triangle, more than 2048 pairs, or a pixel hit more than FRAG_K times take the loop. ----
                bool listed = false;
                {
                    const uint32_t box = lane < cntp ? s_tc[lane].pad : 0u;
                    const uint32_t bw = ((box >> 8) & 15u) + 1u, bh = ((box >> 12) & 15u) + 1u;
                    const uint32_t mine = (box >> 16) ? bw * bh : 0u;
                    const bool large = lane < cntp && (s_tc[lane].flags & TF_LARGE);
                    const uint32_t inc = wave_incl_scan_u32(mine);
                    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
                    if (!__ballot(large) && total != 0 && total <= 2048u) {
                        const uint64_t nzm = __ballot(mine != 0);
                        const uint32_t cidx = __builtin_amdgcn_mbcnt_hi((uint32_t)(nzm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nzm, 0u));
                        const uint32_t pre = inc - mine;
                        if (lane < 32) s_start[lane] = 0ull;
                        wave_lds_sync();
                        if (mine) {
                            s_tmap[cidx] = (uint8_t)lane;
                            s_pm[lane] = pre | (((65536u + bw - 1u) / bw) << 12);  // exact k / bw for k < 256, bw <= 16
                            atomicOr(&s_start[pre >> 6], 1ull << (pre & 63u));
                        }
                        wave_lds_sync();
                        const unsigned long long my_start = lane < 32 ? s_start[lane] : 0ull;
                        const uint32_t nbat = (total + 63u) >> 6;
                        uint32_t base = 0;
                        for (uint32_t b = 0; b < nbat; b++) {
                            const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_start, b);
                            const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_start >> 32), b);
                            const uint64_t mm = ((uint64_t)mhi << 32) | mlo;
                            const uint32_t ci = base + __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + (uint32_t)((mm >> lane) & 1ull) - 1u;
                            base += (uint32_t)__popcll(mm);
                            const uint32_t pidx = b * 64u + lane;
                            if (pidx < total) {
                                const uint32_t t = s_tmap[ci];
                                const int4* tc = reinterpret_cast<const int4*>(&s_tc[t]);
                                const int4 q0 = tc[0], q1 = tc[1], q2 = tc[2], q3 = tc[3];
                                const uint32_t tb = (uint32_t)q3.w, pm = s_pm[t];
                                const uint32_t k = pidx - (pm & 0xfffu);
                                const uint32_t row = (k * (pm >> 12)) >> 16;
                                const int32_t lx = (int32_t)((tb & 15u) + (k - row * (((tb >> 8) & 15u) + 1u)));
                                const int32_t ly = (int32_t)(((tb >> 4) & 15u) + row);
                                const int32_t eb0 = q0.z + __mul24(q0.x, lx) + __mul24(q0.y, ly);
                                const int32_t eb1 = q1.y + __mul24(q0.w, lx) + __mul24(q1.x, ly);
                                const int32_t eb2 = q2.x + __mul24(q1.z, lx) + __mul24(q1.w, ly);
                                if ((eb0 | eb1 | eb2) >= 0) {
                                    const uint32_t pix = (uint32_t)(ly * MTR_BIN + lx);
                                    const uint32_t slot = atomicAdd(&s_cnt[pix], 1u);
                                    if (slot < FRAG_K) s_frag[pix * FRAG_K + slot] = (uint8_t)t;
                                }
                            }
                        }
                        wave_lds_sync();
                        uint32_t cn[4];
                        bool over = false;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t pix = (uint32_t)(((i >> 1) * 8 + (lane >> 3)) * MTR_BIN + (i & 1) * 8 + (lane & 7));
                            cn[i] = s_cnt[pix];
                            s_cnt[pix] = 0u;
                            over = over || cn[i] > FRAG_K;
                        }
                        if (!__ballot(over)) {
                            listed = true;
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const int32_t lx = (int32_t)((i & 1) * 8 + (lane & 7)), ly = (int32_t)((i >> 1) * 8 + (lane >> 3));
                                const uint32_t pix = (uint32_t)(ly * MTR_BIN + lx);
                                const bool in_vp = (uint32_t)(binx0 + lx) < P.fb.W && (uint32_t)(biny0 + ly) < P.fb.H;
                                uint32_t fw[FRAG_K / 4];
#pragma unroll
                                for (uint32_t j = 0; j < FRAG_K / 4; j++) fw[j] = reinterpret_cast<const uint32_t*>(&s_frag[pix * FRAG_K])[j];
                                const uint32_t n = in_vp ? cn[i] : 0u;
                                int32_t last = -1;
                                for (;;) {
                                    // the smallest triangle number above `last` in this pixel's list: submission order
                                    uint32_t best = 255u;
#pragma unroll
                                    for (uint32_t j = 0; j < FRAG_K; j++) {
                                        const uint32_t f = (fw[j >> 2] >> (8u * (j & 3u))) & 0xffu;
                                        if (j < n && (int32_t)f > last && f < best) best = f;
                                    }
                                    if (!__ballot(best != 255u)) break;
                                    if (best == 255u) continue;
                                    last = (int32_t)best;
                                    const int4* tc = reinterpret_cast<const int4*>(&s_tc[best]);
                                    const int4 q0 = tc[0], q1 = tc[1], q2 = tc[2], q3 = tc[3];
                                    const uint32_t flags = (uint32_t)q2.y;
                                    const float z0 = __int_as_float(q2.z), dz1 = __int_as_float(q2.w), dz2 = __int_as_float(q3.x),
                                                rcpA = __int_as_float(q3.y);
                                    // barycentrics at pixel (x, y) of the bin: the arithmetic of the loop below, lane by lane
                                    auto bary = [&](int32_t x, int32_t y, float& b1, float& b2) {
                                        const int32_t e1 = q1.y + __mul24(q0.w, x) + __mul24(q1.x, y);
                                        const int32_t e2 = q2.x + __mul24(q1.z, x) + __mul24(q1.w, y);
                                        b1 = (float)(e1 + (int32_t)((flags >> 5) & 1u)) * rcpA;
                                        b2 = (float)(e2 + (int32_t)((flags >> 6) & 1u)) * rcpA;
                                    };
                                    float b1, b2;
                                    bary(lx, ly, b1, b2);
                                    const float z = fmaf(b2, dz2, fmaf(b1, dz1, z0));
                                    if (!(z >= 0.0f && z <= 1.0f && ((flags & TF_NODT) || z <= dep[i]))) continue;
                                    if (!(flags & TF_NODW)) dep[i] = z;
                                    const uint32_t bmode = (flags & TF_ADD) ? 2u : ((flags & TF_BLEND) ? 1u : 0u);
                                    if (!TEX || !(flags & TF_TEX)) {
                                        if (flags & TF_BLENDSOLID) {
                                            const uint32_t c8 = (uint32_t)q3.z;
                                            const float src[4] = {unorm8f(c8), unorm8f(c8 >> 8), unorm8f(c8 >> 16), unorm8f(c8 >> 24)};
                                            col[i] = blend_store(col[i], src, bmode);
                                        } else {
                                            col[i] = (uint32_t)q3.z;
                                        }
                                    } else {
                                        const TriX& Xt = s_tx[TEX ? best : 0];
                                        auto uv_at = [&](int32_t x, int32_t y, float& u, float& v) {
                                            float c1, c2;
                                            bary(x, y, c1, c2);
                                            const float iw = fmaf(c2, Xt.diw2, fmaf(c1, Xt.diw1, Xt.iw0));
                                            const float up = fmaf(c2, Xt.dup2, fmaf(c1, Xt.dup1, Xt.up0));
                                            const float vp = fmaf(c2, Xt.dvp2, fmaf(c1, Xt.dvp1, Xt.vp0));
                                            u = up / iw;
                                            v = vp / iw;
                                        };
                                        // fine quad differences: (odd position) - (even position) along each axis; this
                                        // pixel is one end of both, so two more evaluations give all four derivatives
                                        float u, v, uh, vh, uw, vw;
                                        uv_at(lx, ly, u, v);
                                        uv_at(lx ^ 1, ly, uh, vh);
                                        uv_at(lx, ly ^ 1, uw, vw);
                                        const float dudx = (lx & 1) ? u - uh : uh - u, dvdx = (lx & 1) ? v - vh : vh - v;
                                        const float dudy = (ly & 1) ? u - uw : uw - u, dvdy = (ly & 1) ? v - vw : vw - v;
                                        const TexRef tr = {Xt.tex, Xt.tw, Xt.th, Xt.tlevels};
                                        float src[4];
                                        sample_texture(tr, u, v, filter_select(dudx, dvdx, dudy, dvdy, tr.tw, tr.th, tr.levels), src);
                                        col[i] = blend_store(col[i], src, bmode);
                                    }
                                }
                            }
                        }
                    }
                }
                if (listed) { wave_lds_sync(); continue; }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int32_t lx = (int32_t)((i & 1) * 8 + (lane & 7)), ly = (int32_t)((i >> 1) * 8 + (lane >> 3));
                    const bool in_vp = (uint32_t)(binx0 + lx) < P.fb.W && (uint32_t)(biny0 + ly) < P.fb.H;
                    uint64_t m = __ballot(lane < cntp && ((s_mask[lane] >> i) & 1u));
                    while (m) {
                        const uint32_t t = __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)m) - 1);
                        m &= m - 1;
                        const int4* tc = reinterpret_cast<const int4*>(&s_tc[t]);
                        const int4 q0 = tc[0], q1 = tc[1], q2 = tc[2], q3 = tc[3];
                        const int32_t A0 = q0.x, B0 = q0.y, C0 = q0.z, A1 = q0.w, B1 = q1.x, C1 = q1.y, A2 = q1.z, B2 = q1.w, C2 = q2.x;
                        const uint32_t flags = (uint32_t)q2.y;
                        const float z0 = __int_as_float(q2.z), dz1 = __int_as_float(q2.w), dz2 = __int_as_float(q3.x),
                                    rcpA = __int_as_float(q3.y);
                        bool inside;
                        float e1f, e2f;
                        if (!(flags & TF_LARGE)) {
                            const int32_t eb0 = C0 + __mul24(A0, lx) + __mul24(B0, ly);
                            const int32_t eb1 = C1 + __mul24(A1, lx) + __mul24(B1, ly);
                            const int32_t eb2 = C2 + __mul24(A2, lx) + __mul24(B2, ly);
                            inside = (eb0 | eb1 | eb2) >= 0;
                            e1f = (float)(eb1 + (int32_t)((flags >> 5) & 1u));
                            e2f = (float)(eb2 + (int32_t)((flags >> 6) & 1u));
                        } else {
                            const int4 ch = s_chi[t];
                            const long long Xp = (long long)lx * 256, Yp = (long long)ly * 256;
                            const long long c0 = ((long long)ch.x << 32) | (unsigned long long)(uint32_t)C0;
                            const long long c1 = ((long long)ch.y << 32) | (unsigned long long)(uint32_t)C1;
                            const long long c2 = ((long long)ch.z << 32) | (unsigned long long)(uint32_t)C2;
                            const long long eb0 = c0 + (long long)A0 * Xp + (long long)B0 * Yp;
                            const long long eb1 = c1 + (long long)A1 * Xp + (long long)B1 * Yp;
                            const long long eb2 = c2 + (long long)A2 * Xp + (long long)B2 * Yp;
                            inside = (eb0 | eb1 | eb2) >= 0;
                            e1f = (float)(eb1 + (long long)((flags >> 5) & 1u));
                            e2f = (float)(eb2 + (long long)((flags >> 6) & 1u));
                        }
                        const float b1 = e1f * rcpA, b2 = e2f * rcpA;
                        const float z = fmaf(b2, dz2, fmaf(b1, dz1, z0));
                        const bool pass = inside && in_vp && z >= 0.0f && z <= 1.0f && ((flags & TF_NODT) || z <= dep[i]);
                        const uint32_t bmode = (flags & TF_ADD) ? 2u : ((flags & TF_BLEND) ? 1u : 0u);
                        if (!TEX || !(flags & TF_TEX)) {
                            if (pass) {
                                if (!(flags & TF_NODW)) dep[i] = z;
                                if (flags & TF_BLENDSOLID) {
                                    const uint32_t c8 = (uint32_t)q3.z;
                                    const float src[4] = {unorm8f(c8), unorm8f(c8 >> 8), unorm8f(c8 >> 16), unorm8f(c8 >> 24)};
                                    col[i] = blend_store(col[i], src, bmode);
                                } else {
                                    col[i] = (uint32_t)q3.z;
                                }
                            }
                        } else {
                            // every lane evaluates (u,v) so quad differences exist for helper pixels too
                            const TriX& Xt = s_tx[TEX ? t : 0];
                            const float iw = fmaf(b2, Xt.diw2, fmaf(b1, Xt.diw1, Xt.iw0));
                            const float up = fmaf(b2, Xt.dup2, fmaf(b1, Xt.dup1, Xt.up0));
                            const float vp = fmaf(b2, Xt.dvp2, fmaf(b1, Xt.dvp1, Xt.vp0));
                            const float u = up / iw, v = vp / iw;
                            const float dudx = __shfl(u, (int)(lane | 1)) - __shfl(u, (int)(lane & ~1u));
                            const float dvdx = __shfl(v, (int)(lane | 1)) - __shfl(v, (int)(lane & ~1u));
                            const float dudy = __shfl(u, (int)(lane | 8)) - __shfl(u, (int)(lane & ~8u));
                            const float dvdy = __shfl(v, (int)(lane | 8)) - __shfl(v, (int)(lane & ~8u));
                            const TexRef tr = {Xt.tex, Xt.tw, Xt.th, Xt.tlevels};
                            const int flt = filter_select(dudx, dvdx, dudy, dvdy, tr.tw, tr.th, tr.levels);
                            if (pass) {
                                if (!(flags & TF_NODW)) dep[i] = z;
                                float src[4];
                                sample_texture(tr, u, v, flt, src);
                                col[i] = blend_store(col[i], src, bmode);
                            }
                        }
                    }
                }
                wave_lds_sync();
            }
        }
        lo = hi;
        hi = lo + width;
        if (hi > (1ull << 32)) hi = 1ull << 32;
    }

    if (lane == 0) bin_queue_done(P.fb, bin);
    // ---- write-out: the only framebuffer traffic of the frame ----
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t x = (uint32_t)binx0 + (i & 1) * 8 + (lane & 7), y = (uint32_t)biny0 + (i >> 1) * 8 + (lane >> 3);
        if (x < P.fb.W && y < P.fb.H) {
            const size_t pi = (size_t)y * P.fb.W + x;
            reinterpret_cast<uint32_t*>(P.color)[pi] = col[i];
            P.depth[pi] = dep[i];
        }
    }
}

}  // namespace mtr

void mtr_launch_tile(const TileParams& p, bool textured, hipStream_t s) {
    const uint32_t mine = p.fb.own.own_count;
    if (mine == 0) return;
    uint32_t grid = (mine + 7) / 8 * 8;
    if (p.xcd_run) grid = (grid / 8 + p.xcd_run - 1) / p.xcd_run * p.xcd_run * 8;  // whole runs
    if (textured) hipLaunchKernelGGL(mtr::k_tile<true>, dim3(grid), dim3(64), 0, s, p);
    else hipLaunchKernelGGL(mtr::k_tile<false>, dim3(grid), dim3(64), 0, s, p);
}
