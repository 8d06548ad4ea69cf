// k_tile_vis.hip -- visibility-key tile kernel for frames whose every material is opaque
// (debug-id / overlay colours, or textures with alpha == 255 everywhere: blend = replace).
//
// For such frames the per-pixel result of the ordered pipeline (depth LessEqual + write, blend with
// a = 1) is a pure reduction: the winner of a pixel is the fragment with the smallest z, the LATEST
// in submission order among equals.  So each pixel keeps one 64-bit key  (~bits(z) : order+1)  in
// LDS and every fragment is an order-independent, fire-and-forget `ds_max_u64`:
//   * 64 triangles are set up per pass, one per lane;
//   * the (triangle, pixel-of-bbox) pairs of every i32-edge-class triangle are FLATTENED over the wave, in rounds of
//     <= 4096 pairs: a pass costs sum(bbox pixels)/64 iterations whatever the mix of 1-pixel slivers and bin-filling
//     triangles (a lane = triangle walk ran max(bbox pixels) iterations at 29 % lane efficiency on the headline scene).
//     The pair -> triangle map needs no search: triangle t sets bit (prefix_t mod 64) of a 64-bit start mask per
//     batch of 64 pairs (one ds_or_b64), and pair p's triangle is  #starts before its batch + popcount(mask bits
//     <= p) - 1  (v_mbcnt).  Triangles over 64 px across (64-bit edge functions) are rasterised by the whole wave, one
//     at a time (v_readlane);
//   * shading is deferred: the winner's record is addressable from its order (chunk runs live at
//     chunk * MTR_CHUNK_SLOTS), so the resolve does one colour lookup -- or one texture sample with the
//     quad derivatives evaluated from the winner's plane equations exactly as SPEC.md section 7
//     defines them -- per pixel, then the only framebuffer write of the frame.
// Same arithmetic per fragment as k_tile.hip, bit for bit; the host picks this kernel only when the
// frame is eligible (mtr_api.cpp); tests run both kernels on the same scenes.
// VIS_WAVES waves per 16x16 bin (passes dealt round-robin, two workgroup barriers in total), no segment
// sort (the submission order rides in the entry).
//
// STAIR (frames with translucent materials, round 3): alpha blending in the default depth state (test LessEqual + write)
// depends on submission order only through the fragments that PASS the depth test, and those are exactly the prefix
// minima of z in submission order: fragment f passes iff z_f <= z_e for every earlier fragment e of the pixel (a
// fragment that fails leaves the depth buffer alone, so the depth f meets is the minimum over all its predecessors).
// The set is order-independent to build: next to the max key each pixel keeps a short list of submission orders; a
// fragment is appended unless the key it meets proves it dominated (an EARLIER fragment that is STRICTLY nearer -- with
// entries arriving roughly in order that leaves little more than the prefix minima themselves).  The resolve walks a
// pixel's list in increasing order, recomputes z from the record, keeps the running minimum, and shades + blends the
// fragments that pass -- the ordered kernel's arithmetic, without its segment sort, ordered passes and per-pass set-up
// (C5 with translucent textures: tile stage 675 -> see DESIGN.md).  A bin with a HARD order-dependent triangle (additive
// blend, depth write or test off) or a pixel whose list overflows is flagged and left to the ordered kernel.
#include "tile_common.h"

namespace mtr {

// LDS triangle record of one pass (64 B): everything a (triangle, pixel) work item needs
struct VisTri {
    int32_t A0, B0, C0, A1;  // edge i: E_i(lx,ly) = C_i + A_i*lx + B_i*ly, top-left bias folded into C_i
    int32_t B1, C1, A2, B2;  // small class: i32, A/B pre-scaled by 256; large class: unscaled, C high words in s_chi
    int32_t C2;
    uint32_t flags;          // bit0 large, bits 4..6: 1 - tl_i
    float z0, dz1;
    float dz2, rcpA;
    uint32_t ordk;           // submission order + 1
    uint32_t box;            // px0 | py0 << 4 | (bw-1) << 8 | magic(iw) << 12   (k / iw = k * magic >> 16; iw: items per bbox row)
};
static_assert(sizeof(VisTri) == 64, "VisTri is 64 B");

struct Setup {
    VisTri t;
    int4 chi;
    int32_t npx;   // pixels of the triangle's bbox in this bin
    uint32_t bhm1; // bbox height - 1
};

__device__ __forceinline__ void setup_tri(const RecA& a, uint32_t ord, int32_t binx0, int32_t biny0, int32_t vw, int32_t vh, Setup& s) {
    const int32_t xmin = min(a.X0, min(a.X1, a.X2)), xmax = max(a.X0, max(a.X1, a.X2));
    const int32_t ymin = min(a.Y0, min(a.Y1, a.Y2)), ymax = max(a.Y0, max(a.Y1, a.Y2));
    const bool large = (xmax - xmin) > 16384 || (ymax - ymin) > 16384;
    const int32_t X[3] = {a.X0, a.X1, a.X2}, Y[3] = {a.Y0, a.Y1, a.Y2};
    int32_t A[3], B[3], Clo[3], Chi[3] = {0, 0, 0};
    uint32_t flags = large ? 1u : 0u;
    float fA2;
    if (!large) {
        // the bin overlaps the bbox, so every operand is < 2^16 and every product < 2^31: 24-bit multiplies
        const int32_t Px = binx0 * 256 + 128, Py = biny0 * 256 + 128;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int ia = (i + 1) % 3, ib = (i + 2) % 3;
            const int32_t dx = X[ib] - X[ia], dy = Y[ib] - Y[ia];
            const int32_t tl = (dy > 0 || (dy == 0 && dx < 0)) ? 1 : 0;
            flags |= (uint32_t)(1 - tl) << (4 + i);
            A[i] = dy * 256; B[i] = -dx * 256;
            Clo[i] = __mul24(dy, Px - X[ia]) - __mul24(dx, Py - Y[ia]) + (tl - 1);
        }
        fA2 = (float)(__mul24(X[2] - X[0], Y[1] - Y[0]) - __mul24(X[1] - X[0], Y[2] - Y[0]));
    } else {
        const long long Px = (long long)binx0 * 256 + 128, Py = (long long)biny0 * 256 + 128;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int ia = (i + 1) % 3, ib = (i + 2) % 3;
            const int32_t dx = X[ib] - X[ia], dy = Y[ib] - Y[ia];
            const int32_t tl = (dy > 0 || (dy == 0 && dx < 0)) ? 1 : 0;
            const long long C = (long long)dy * (Px - X[ia]) - (long long)dx * (Py - Y[ia]) + (tl - 1);
            flags |= (uint32_t)(1 - tl) << (4 + i);
            A[i] = dy; B[i] = -dx;
            Clo[i] = (int32_t)(uint32_t)(unsigned long long)C;
            Chi[i] = (int32_t)(C >> 32);
        }
        fA2 = (float)((long long)(X[2] - X[0]) * (long long)(Y[1] - Y[0]) - (long long)(X[1] - X[0]) * (long long)(Y[2] - Y[0]));
    }
    const int32_t px0 = max(((xmin + 127) >> 8) - binx0, 0), px1 = min(((xmax - 128) >> 8) - binx0, min(MTR_BIN, vw) - 1);
    const int32_t py0 = max(((ymin + 127) >> 8) - biny0, 0), py1 = min(((ymax - 128) >> 8) - biny0, min(MTR_BIN, vh) - 1);
    const int32_t bw = px1 - px0 + 1, bh = py1 - py0 + 1;
    s.npx = (bw > 0 && bh > 0) ? bw * bh : 0;  // pixels of the bbox in this bin; the caller turns it into work items (walk_items)
    s.bhm1 = (uint32_t)max(bh - 1, 0);
    s.t.A0 = A[0]; s.t.B0 = B[0]; s.t.C0 = Clo[0];
    s.t.A1 = A[1]; s.t.B1 = B[1]; s.t.C1 = Clo[1];
    s.t.A2 = A[2]; s.t.B2 = B[2]; s.t.C2 = Clo[2];
    s.t.flags = flags;
    s.t.z0 = a.z0; s.t.dz1 = a.z1 - a.z0; s.t.dz2 = a.z2 - a.z0;
    s.t.rcpA = 1.0f / fA2;
    s.t.ordk = ord + 1u;
    s.t.box = (uint32_t)(px0 & 15) | ((uint32_t)(py0 & 15) << 4) | ((uint32_t)((bw - 1) & 15) << 8);  // + magic << 12 (walk_items)
    s.chi = make_int4(Chi[0], Chi[1], Chi[2], 0);
}

#ifndef MTR_QUAD_WALK
#define MTR_QUAD_WALK 1
#endif
#define STAIR_K 8u  // submission orders kept per pixel (STAIR)

// one fragment that passed coverage and the z range.  STAIR: also remember its order unless the key it meets proves it
// dominated: `old` is SOME fragment of this pixel; if it is earlier and strictly nearer, this one fails the depth test
// whatever else arrives (the depth it meets is <= z_old < z).
template <bool STAIR>
__device__ __forceinline__ void put_fragment(unsigned long long* s_key, uint32_t* s_cnt, uint32_t* s_list, uint32_t pix, unsigned long long key) {
    if (!STAIR) {
        atomicMax(&s_key[pix], key);
        return;
    }
    const unsigned long long old = atomicMax(&s_key[pix], key);
    if (old != 0ull && (uint32_t)old < (uint32_t)key && (uint32_t)(old >> 32) > (uint32_t)(key >> 32)) return;
    const uint32_t slot = atomicAdd(&s_cnt[pix], 1u);
    if (slot < STAIR_K) s_list[pix * STAIR_K + slot] = (uint32_t)key;
}

__device__ __forceinline__ unsigned long long make_key(float z, uint32_t ordk) {
    // 0 <= z <= 1 and z is never -0 (SPEC.md: vertex z comes from an fma chain started at +0), so the bit
    // pattern is monotonic; larger key = nearer, then later
    return ((unsigned long long)(~__float_as_uint(z)) << 32) | ordk;
}

// deferred textured shading of the winner at pixel (px,py): SPEC.md section 7, same operations as the
// per-fragment path (the quad neighbours are evaluated from the same triangle's plane equations)
__device__ __forceinline__ void sample_textured(const RecA& a, const RecB& b, const DMat& mat, int32_t px, int32_t py, float (&src)[4]) {
    const long long A2 = (long long)(a.X2 - a.X0) * (long long)(a.Y1 - a.Y0) - (long long)(a.X1 - a.X0) * (long long)(a.Y2 - a.Y0);
    const float rcpA = 1.0f / (float)A2;
    const float diw1 = b.iw1 - b.iw0, diw2 = b.iw2 - b.iw0, dup1 = b.up1 - b.up0, dup2 = b.up2 - b.up0,
                dvp1 = b.vp1 - b.vp0, dvp2 = b.vp2 - b.vp0;
    auto uv_at = [&](int32_t qx, int32_t qy, float& u, float& v) {
        const long long Px = (long long)qx * 256 + 128, Py = (long long)qy * 256 + 128;
        const long long E1 = (long long)(a.Y0 - a.Y2) * (Px - a.X2) - (long long)(a.X0 - a.X2) * (Py - a.Y2);
        const long long E2 = (long long)(a.Y1 - a.Y0) * (Px - a.X0) - (long long)(a.X1 - a.X0) * (Py - a.Y0);
        const float b1 = (float)E1 * rcpA, b2 = (float)E2 * rcpA;
        const float iw = fmaf(b2, diw2, fmaf(b1, diw1, b.iw0));
        const float up = fmaf(b2, dup2, fmaf(b1, dup1, b.up0));
        const float vp = fmaf(b2, dvp2, fmaf(b1, dvp1, b.vp0));
        u = up / iw;
        v = vp / iw;
    };
    // fine quad differences: (odd position) - (even position) along each axis; this pixel is one end of both, so two
    // more evaluations give all four derivatives
    float u, v, uh, vh, uw, vw;
    uv_at(px, py, u, v);
    uv_at(px ^ 1, py, uh, vh);
    uv_at(px, py ^ 1, uw, vw);
    const float dudx = (px & 1) ? u - uh : uh - u, dvdx = (px & 1) ? v - vh : vh - v;
    const float dudy = (py & 1) ? u - uw : uw - u, dvdy = (py & 1) ? v - vw : vw - v;
    const TexRef tr = {mat.tex, mat.tw, mat.th, mat.tlevels};
    sample_texture(tr, u, v, filter_select(dudx, dvdx, dudy, dvdy, mat.tw, mat.th, mat.tlevels), src);
}
__device__ __forceinline__ uint32_t shade_textured(const RecA& a, const RecB& b, const DMat& mat, int32_t px, int32_t py) {
    float src[4];
    sample_textured(a, b, mat, px, py, src);
    return blend_store(0u, src, 0u);  // order-free: blending is off, or the texture is opaque (a == 1 exactly) and the blend a replace
}
// z of the record's triangle at the centre of pixel (px, py): the edge values are the exact integers the rasteriser
// compares, so this is the float the flattened walk computed from its bin-relative form (SPEC.md section 6)
__device__ __forceinline__ float z_at(const RecA& a, int32_t px, int32_t py) {
    const long long A2 = (long long)(a.X2 - a.X0) * (long long)(a.Y1 - a.Y0) - (long long)(a.X1 - a.X0) * (long long)(a.Y2 - a.Y0);
    const float rcpA = 1.0f / (float)A2;
    const long long Px = (long long)px * 256 + 128, Py = (long long)py * 256 + 128;
    const long long E1 = (long long)(a.Y0 - a.Y2) * (Px - a.X2) - (long long)(a.X0 - a.X2) * (Py - a.Y2);
    const long long E2 = (long long)(a.Y1 - a.Y0) * (Px - a.X0) - (long long)(a.X1 - a.X0) * (Py - a.Y0);
    const float b1 = (float)E1 * rcpA, b2 = (float)E2 * rcpA;
    return fmaf(b2, a.z2 - a.z0, fmaf(b1, a.z1 - a.z0, a.z0));
}

// waves per bin: the passes (64 triangles each) of a bin are dealt round-robin to the waves of its workgroup;
// the keys are order-independent, so the waves only meet at the two barriers around the raster loop.
// measured on the unsharded headline scene (tools/sweep_vis_waves.sh): 1 wave per bin: 104 us, 2: 73 us, 4: 83 us, 8: 129 us
#ifndef VIS_OCC
#define VIS_OCC 6    // waves per SIMD the register allocator must leave room for (no spills at 6)
#endif

// VIS_WAVES: 2 for unsharded frames (see above); a sharded rank has few bins and the frame then takes as long as its
// heaviest bin (629 triangles = 183 batches of 64 pairs on the headline scene: 23 us with two waves), so the host
// gives such frames 4 or 8 waves per bin (mtr_launch_tile_vis).
template <bool TEX, int VIS_WAVES, bool STAIR, bool QW>
__global__ __launch_bounds__(64 * VIS_WAVES, VIS_OCC) void k_tile_vis(TileParams P) {
    __shared__ unsigned long long s_key[MTR_BIN * MTR_BIN];
    __shared__ uint4 s_flat[VIS_WAVES][64 * 4];              // flat-class triangles of the current pass, 64 B each
    __shared__ unsigned long long s_start[VIS_WAVES][64];     // per batch of 64 pairs: which pairs start a triangle
    __shared__ uint32_t s_trans;                               // mixed frames: the queue holds an order-dependent triangle
    __shared__ uint32_t s_hard;                                // ... one that prefix minima of z do not resolve: the ordered kernel's bin
    __shared__ uint32_t s_cnt[STAIR ? MTR_BIN * MTR_BIN : 1];                      // STAIR: orders listed per pixel
    __shared__ __align__(16) uint32_t s_list[STAIR ? MTR_BIN * MTR_BIN * STAIR_K : 4];  // STAIR: the orders (+ 1)

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t magic_lut = (65536u + (lane & 15u)) / ((lane & 15u) + 1u);  // lane i: ceil(65536 / (i + 1)), fetched by __shfl per pass
    const uint32_t ovf = tile_prologue(P);
    uint32_t bin;
    if (!block_to_bin(P.fb, bin, P.xcd_run)) return;  // uniform over the workgroup, before any barrier
    const uint32_t nbx = P.fb.nbx;
    const int32_t binx0 = (int32_t)(bin % nbx) * MTR_BIN, biny0 = (int32_t)(bin / nbx) * MTR_BIN;
    const float cd = P.clear_depth;
    // fragments pass 0 <= z <= 1 and z <= clear depth (nothing else is in the depth buffer before the resolve)
    const bool zlim_ok = cd >= 0.0f;
    const uint32_t zlim = __float_as_uint(fminf(cd, 1.0f));
    const int32_t vw = (int32_t)P.fb.W - binx0, vh = (int32_t)P.fb.H - biny0;  // viewport edge in bin coordinates
    // direct mode: bin b's queue starts at b * qcap whatever its fill, so this wave's first two entry loads are
    // issued together with the fill-count load instead of after it (one dependent round trip less per bin)
    const uint32_t stride = 64 * VIS_WAVES, first = wv * 64;
    uint32_t spec0 = 0, spec1 = 0;
    if (P.fb.direct) {
        const uint32_t qb = bin * P.fb.qcap;
        if (first + lane < P.fb.qcap) spec0 = P.fb.entries[qb + first + lane];
        if (first + stride + lane < P.fb.qcap) spec1 = P.fb.entries[qb + first + stride + lane];
    }
    uint32_t ent_lo, N, seg_lo_unused, n_seg;
    bin_queue(P.fb, bin, ent_lo, N, seg_lo_unused, n_seg);  // the same word for every thread of the workgroup
    if (ovf) {  // incomplete queues: read nothing, leave the bin cleared and its fill word parked for the next frame
        N = 0;
        __syncthreads();  // every thread has read the fill word
        if (threadIdx.x == 0) bin_queue_done(P.fb, bin);
    }
    if (N == 0) {
        // an empty bin (a third of the headline frame): clear colour / depth and leave, no LDS, no barrier
        for (uint32_t pidx = threadIdx.x; pidx < MTR_BIN * MTR_BIN; pidx += 64 * VIS_WAVES) {
            const int32_t lx = (int32_t)(pidx & (MTR_BIN - 1)), ly = (int32_t)(pidx >> MTR_BIN_SHIFT);
            if (lx >= vw || ly >= vh) continue;
            const size_t pi = (size_t)(biny0 + ly) * P.fb.W + (size_t)(binx0 + lx);
            reinterpret_cast<uint32_t*>(P.color)[pi] = P.clear_rgba8;
            P.depth[pi] = cd;
        }
        if (P.mixed && threadIdx.x == 0) P.bin_flag[bin] = 0;
        // the parked count of an empty bin is 0 too (mtr_frame_read_bin_counts: what bench.py balances bands by); without
        // this store the word keeps whatever the allocation, or an earlier scene, left there
        if (P.fb.direct && threadIdx.x == 0 && !ovf) P.fb.bin_count[bin] = 0ull;
        return;
    }
    for (uint32_t i = threadIdx.x; i < MTR_BIN * MTR_BIN; i += 64 * VIS_WAVES) {
        s_key[i] = 0ull;
        if (STAIR) s_cnt[i] = 0u;
    }
    if (threadIdx.x == 0) { s_trans = 0u; s_hard = 0u; }
    __syncthreads();

    const RecA zero_rec = {0, 0, 0, 0, 0, 0, 0.0f, 0.0f, 0.0f, 0u, 0u, 0u};
    // two-deep software pipeline over the dependent loads entries[] -> rec_a[]: while pass k is rasterised the
    // record loads of this wave's next pass and the entry loads of the one after are in flight.  The record of an
    // entry is addressable from its submission order: chunk run base = chunk * MTR_CHUNK_SLOTS.
    uint32_t ord_cur = 0, ord_nxt = 0;
    if (P.fb.direct) {
        if (first + lane < N) ord_cur = spec0;
        if (first + stride + lane < N) ord_nxt = spec1;
    } else {
        if (first + lane < N) ord_cur = P.fb.entries[ent_lo + first + lane];
        if (first + stride + lane < N) ord_nxt = P.fb.entries[ent_lo + first + stride + lane];
    }
    RecA a_cur = zero_rec;
    if (first + lane < N) a_cur = load_rec(P.fb, (ord_cur >> 7) * MTR_CHUNK_SLOTS + (ord_cur & 127u));
    for (uint32_t e0 = first; e0 < N; e0 += stride) {
        const bool valid = e0 + lane < N;
        RecA a_nxt = zero_rec;
        if (e0 + stride + lane < N) a_nxt = load_rec(P.fb, (ord_nxt >> 7) * MTR_CHUNK_SLOTS + (ord_nxt & 127u));
        uint32_t ord_nn = 0;
        if (e0 + 2 * stride + lane < N) ord_nn = P.fb.entries[ent_lo + e0 + 2 * stride + lane];

        if (P.mixed && valid && (a_cur.pad1 >> 16)) {  // benign races: every writer stores 1
            s_trans = 1u;
            if (!STAIR || (a_cur.pad1 >> 17)) s_hard = 1u;
        }
        Setup s = {};
        if (valid) setup_tri(a_cur, ord_cur, binx0, biny0, vw, vh, s);
        const bool large = (s.t.flags & 1u) != 0;
        // The wave's work item is a (triangle, pixel of its bbox) pair, or -- when a good part of the pass's triangles have a bbox
        // of more than four pixels -- a (triangle, 2 x 2 quad) pair: one staged record read and one index decode per four pixels,
        // the three edge functions stepped by an add.  A pass of one-pixel boxes (the instanced configs) is cheaper pixel by pixel,
        // a pass of the headline model's 10-pixel boxes in quads: 9 % fewer VALU and 29 % fewer LDS instructions, the kernel alone
        // 50.4 -> 45.3 us -- and frames in flight 1.3 % SLOWER (paired builds, three runs each), so the host asks for quads only
        // for a frame that has the GPU to itself (TileParams::quad_walk -> the QW instantiation: latency, not throughput).  Uniform
        // over the wave.
        const bool quads = QW && (uint32_t)__popcll(__ballot(!large && s.npx > 4)) * 4u >= (uint32_t)__popcll(__ballot(!large && s.npx > 0));
        const uint32_t bwm1_l = (s.t.box >> 8) & 15u;
        const uint32_t iw = (quads && !large) ? (bwm1_l >> 1) + 1u : bwm1_l + 1u, ih = (quads && !large) ? (s.bhm1 >> 1) + 1u : s.bhm1 + 1u;
        const uint32_t npx = s.npx ? iw * ih : 0u;
        // k / iw = k * magic >> 16, exact for k < 256 and iw <= 16: magic = ceil(65536 / iw), from the lanes' table (QW) or divided out
        s.t.box |= (QW ? (uint32_t)__shfl((int)magic_lut, (int)(iw - 1u)) : (65536u + iw - 1u) / iw) << 12;
        // ---- lane = pixel of the bbox: triangles that need 64-bit edge functions (more than 64 px across: rare),
        //      broadcast one at a time with v_readlane ----
        for (uint64_t mb = __ballot(npx != 0 && large); mb; mb &= mb - 1) {
            const uint32_t t = __builtin_amdgcn_readfirstlane((uint32_t)__ffsll((long long)mb) - 1);
#define RL(x) __builtin_amdgcn_readlane((int)(x), t)
            const int32_t A0 = RL(s.t.A0), B0 = RL(s.t.B0), C0 = RL(s.t.C0), A1 = RL(s.t.A1), B1 = RL(s.t.B1), C1 = RL(s.t.C1);
            const int32_t A2 = RL(s.t.A2), B2 = RL(s.t.B2), C2 = RL(s.t.C2);
            const uint32_t flags = (uint32_t)RL(s.t.flags), box = (uint32_t)RL(s.t.box), tord = (uint32_t)RL(s.t.ordk), tn = (uint32_t)RL(npx);
            const float z0 = __int_as_float(RL(__float_as_int(s.t.z0))), dz1 = __int_as_float(RL(__float_as_int(s.t.dz1)));
            const float dz2 = __int_as_float(RL(__float_as_int(s.t.dz2))), rcpA = __int_as_float(RL(__float_as_int(s.t.rcpA)));
            const int32_t H0 = RL(s.chi.x), H1 = RL(s.chi.y), H2 = RL(s.chi.z);
#undef RL
            const uint32_t bw = ((box >> 8) & 15u) + 1u, magic = box >> 12;
            for (uint32_t k = lane; k < tn; k += 64) {
                const uint32_t row = (k * magic) >> 16;
                const int32_t lx = (int32_t)((box & 15u) + (k - row * bw)), ly = (int32_t)(((box >> 4) & 15u) + row);
                const long long Xp = (long long)lx * 256, Yp = (long long)ly * 256;
                const long long c0 = ((long long)H0 << 32) | (unsigned long long)(uint32_t)C0;
                const long long c1 = ((long long)H1 << 32) | (unsigned long long)(uint32_t)C1;
                const long long c2 = ((long long)H2 << 32) | (unsigned long long)(uint32_t)C2;
                const long long eb0 = c0 + (long long)A0 * Xp + (long long)B0 * Yp;
                const long long eb1 = c1 + (long long)A1 * Xp + (long long)B1 * Yp;
                const long long eb2 = c2 + (long long)A2 * Xp + (long long)B2 * Yp;
                const bool inside = (eb0 | eb1 | eb2) >= 0;
                const float e1f = (float)(eb1 + (long long)((flags >> 5) & 1u));
                const float e2f = (float)(eb2 + (long long)((flags >> 6) & 1u));
                const float b1 = e1f * rcpA, b2 = e2f * rcpA;
                const float z = fmaf(b2, dz2, fmaf(b1, dz1, z0));
                if (inside && z >= 0.0f && z <= 1.0f && z <= cd) put_fragment<STAIR>(s_key, s_cnt, s_list, (uint32_t)(ly * MTR_BIN + lx), make_key(z, tord));
            }
        }

        // ---- lane = (triangle, pixel) pair, 64 pairs per step, for every i32-class triangle.  A round stages a
        //      prefix of the remaining triangles holding <= 4096 pairs (64 start masks, one per lane); one round
        //      is the rule, a pass of 64 bin-filling triangles takes four. ----
        for (uint64_t todo = zlim_ok ? __ballot(npx != 0 && !large) : 0ull; todo;) {
            const bool cand = (todo >> lane) & 1ull;
            const uint32_t mine = cand ? npx : 0u;
            const uint32_t inc = wave_incl_scan_u32(mine);
            const bool take = cand && inc <= 4096u;  // npx <= 256: the first candidate always fits
            const uint64_t tm = __ballot(take);
            todo &= ~tm;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63 - __builtin_clzll(tm));
            const uint32_t cidx = __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u));
            const uint32_t pre = inc - mine;
            s_start[wv][lane] = 0ull;
            wave_lds_sync();
            if (take) {
                uint4* dst = &s_flat[wv][cidx * 4];
                // edge functions rebased to the bbox origin: a pair evaluates E(col, row) with no bin coordinates
                const int32_t ox = (int32_t)(s.t.box & 15u), oy = (int32_t)((s.t.box >> 4) & 15u);
                const int32_t c0 = s.t.C0 + __mul24(s.t.A0, ox) + __mul24(s.t.B0, oy);
                const int32_t c1 = s.t.C1 + __mul24(s.t.A1, ox) + __mul24(s.t.B1, oy);
                const int32_t c2 = s.t.C2 + __mul24(s.t.A2, ox) + __mul24(s.t.B2, oy);
                dst[0] = make_uint4((uint32_t)s.t.A0, (uint32_t)s.t.B0, (uint32_t)c0, (uint32_t)s.t.A1);
                dst[1] = make_uint4((uint32_t)s.t.B1, (uint32_t)c1, (uint32_t)s.t.A2, (uint32_t)s.t.B2);
                dst[2] = make_uint4((uint32_t)c2, pre | (s.bhm1 << 16), __float_as_uint(s.t.z0), __float_as_uint(s.t.dz1));
                // box bits 29 / 30: 1 - tl of edges 1 / 2 (flags bits 5 / 6)
                dst[3] = make_uint4(__float_as_uint(s.t.dz2), __float_as_uint(s.t.rcpA), s.t.ordk, s.t.box | ((s.t.flags & 0x60u) << 24));
                atomicOr(&s_start[wv][pre >> 6], 1ull << (pre & 63u));
            }
            wave_lds_sync();
            const unsigned long long my_start = s_start[wv][lane];
            const uint32_t nb = (total + 63u) >> 6;
            uint32_t base = 0;
            for (uint32_t b = 0; b < nb; b++) {
                const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_start, b);
                const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_start >> 32), b);
                const uint64_t m = ((uint64_t)mhi << 32) | mlo;
                // triangles started before this batch + starts at or below this lane - 1
                const uint32_t tri = base + __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) + (uint32_t)((m >> lane) & 1ull) - 1u;
                base += (uint32_t)__popcll(m);
                const uint32_t p = b * 64u + lane;
                if (p < total && (!QW || !quads)) {
                    const uint4* src = &s_flat[wv][tri * 4];
                    const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                    const uint32_t box = q3.w, k = p - (q2.y & 0xFFFFu);
                    const int32_t row = (int32_t)((k * ((box >> 12) & 0x1ffffu)) >> 16);
                    const int32_t col = (int32_t)k - __mul24(row, (int32_t)((box >> 8) & 15u) + 1);
                    const int32_t eb0 = (int32_t)q0.z + __mul24((int32_t)q0.x, col) + __mul24((int32_t)q0.y, row);
                    const int32_t eb1 = (int32_t)q1.y + __mul24((int32_t)q0.w, col) + __mul24((int32_t)q1.x, row);
                    const int32_t eb2 = (int32_t)q2.x + __mul24((int32_t)q1.z, col) + __mul24((int32_t)q1.w, row);
                    const float b1 = (float)(eb1 + (int32_t)((box >> 29) & 1u)) * __uint_as_float(q3.y);
                    const float b2 = (float)(eb2 + (int32_t)((box >> 30) & 1u)) * __uint_as_float(q3.y);
                    const float z = fmaf(b2, __uint_as_float(q3.x), fmaf(b1, __uint_as_float(q2.w), __uint_as_float(q2.z)));
                    // 0 <= z <= min(1, clear depth) as ONE unsigned compare of the bit patterns (z is never -0, SPEC.md;
                    // negative and NaN patterns are above every non-negative bound)
                    if ((eb0 | eb1 | eb2) >= 0 && __float_as_uint(z) <= zlim)
                        put_fragment<STAIR>(s_key, s_cnt, s_list, (box & 0xffu) + (uint32_t)(row * MTR_BIN + col), make_key(z, q3.z));
                }
                if (QW && p < total && quads) {
                    const uint4* src = &s_flat[wv][tri * 4];
                    const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                    const uint32_t box = q3.w, k = p - (q2.y & 0xFFFFu);
                    const uint32_t bwm1 = (box >> 8) & 15u, bhm1 = q2.y >> 16;
                    const int32_t qr = (int32_t)((k * ((box >> 12) & 0x1ffffu)) >> 16);
                    const int32_t qc = (int32_t)k - __mul24(qr, (int32_t)(bwm1 >> 1) + 1);
                    const int32_t col = qc * 2, row = qr * 2;  // the quad's first pixel in the bbox
                    const int32_t A0 = (int32_t)q0.x, B0 = (int32_t)q0.y, A1 = (int32_t)q0.w, B1 = (int32_t)q1.x, A2 = (int32_t)q1.z, B2 = (int32_t)q1.w;
                    const int32_t e0 = (int32_t)q0.z + __mul24(A0, col) + __mul24(B0, row);
                    const int32_t e1 = (int32_t)q1.y + __mul24(A1, col) + __mul24(B1, row);
                    const int32_t e2 = (int32_t)q2.x + __mul24(A2, col) + __mul24(B2, row);
                    const int32_t f1 = (int32_t)((box >> 29) & 1u), f2 = (int32_t)((box >> 30) & 1u);
                    const float rcp = __uint_as_float(q3.y), dz1 = __uint_as_float(q2.w), dz2 = __uint_as_float(q3.x), z0 = __uint_as_float(q2.z);
                    const uint32_t pix0 = (box & 0xffu) + (uint32_t)(row * MTR_BIN + col), ordk = q3.z;
                    // a pixel past the bbox's last column / row belongs to the neighbouring bin (or lies off the target): not ours
                    const bool vx = (uint32_t)col < bwm1, vy = (uint32_t)row < bhm1;
                    auto fragment = [&](int32_t a0, int32_t a1, int32_t a2, bool ok, uint32_t pix) {
                        if (ok && (a0 | a1 | a2) >= 0) {
                            const float b1 = (float)(a1 + f1) * rcp, b2 = (float)(a2 + f2) * rcp;
                            const float z = fmaf(b2, dz2, fmaf(b1, dz1, z0));
                            if (__float_as_uint(z) <= zlim) put_fragment<STAIR>(s_key, s_cnt, s_list, pix, make_key(z, ordk));
                        }
                    };
                    fragment(e0, e1, e2, true, pix0);
                    fragment(e0 + A0, e1 + A1, e2 + A2, vx, pix0 + 1u);
                    fragment(e0 + B0, e1 + B1, e2 + B2, vy, pix0 + MTR_BIN);
                    fragment(e0 + A0 + B0, e1 + A1 + B1, e2 + A2 + B2, vx && vy, pix0 + MTR_BIN + 1u);
                }
            }
        }
        a_cur = a_nxt;
        ord_cur = ord_nxt;
        ord_nxt = ord_nn;
    }
    __syncthreads();
    bool stair = false;  // this bin resolves through its per-pixel order lists
    if (P.mixed) {  // an order-dependent triangle in the queue
        bool leave = s_hard != 0u;  // leave the bin (and its queue) to the ordered kernel
        if (STAIR && !leave && s_trans) {
            bool over = false;
            for (uint32_t i = threadIdx.x; i < MTR_BIN * MTR_BIN; i += 64 * VIS_WAVES) over = over || s_cnt[i] > STAIR_K;
            leave = __syncthreads_or(over ? 1 : 0) != 0;  // a pixel listed more orders than it has room for
            stair = !leave;
        }
        if (threadIdx.x == 0) P.bin_flag[bin] = leave ? 1 : 0;
        if (leave) return;
    }
    if (threadIdx.x == 0) {
        if (P.fb.direct && N) {  // queue statistics (direct mode has no scan to count them)
            atomicAdd(&P.fb.counters[MTR_CTR(CTR_ENT, bin)], N);
            atomicAdd(&P.fb.counters[MTR_CTR(CTR_SEG, bin)], n_seg);
        }
        bin_queue_done(P.fb, bin);  // every wave has read its queue bounds by now
    }

    // ---- resolve: deferred shading of each pixel's winner, the only framebuffer traffic of the frame;
    //      one pixel per thread, rows of 16 pixels = 64 contiguous bytes ----
    // (unrolled: a thread's pixels are independent, so their winner-record loads are in flight together)
#pragma unroll
    for (uint32_t pidx = threadIdx.x; pidx < MTR_BIN * MTR_BIN; pidx += 64 * VIS_WAVES) {
        const int32_t lx = (int32_t)(pidx & (MTR_BIN - 1)), ly = (int32_t)(pidx >> MTR_BIN_SHIFT);
        if (lx >= vw || ly >= vh) continue;
        const uint32_t x = (uint32_t)(binx0 + lx), y = (uint32_t)(biny0 + ly);
        const unsigned long long key = s_key[ly * MTR_BIN + lx];
        uint32_t col = P.clear_rgba8;
        float dep = cd;
        if (STAIR && stair) {
            // the pixel's listed fragments in submission order; those that pass the depth test (z <= every earlier one's:
            // running minimum, starting from the clear depth) are shaded and blended, exactly as the ordered kernel does
            const uint32_t pix = (uint32_t)(ly * MTR_BIN + lx);
            const uint32_t n = s_cnt[pix];
            const uint4 l0 = reinterpret_cast<const uint4*>(&s_list[pix * STAIR_K])[0], l1 = reinterpret_cast<const uint4*>(&s_list[pix * STAIR_K])[1];
            const uint32_t lst[STAIR_K] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
            uint32_t last = 0;  // orders are stored + 1
            for (;;) {
                uint32_t best = 0xFFFFFFFFu;  // the smallest listed order above `last`
#pragma unroll
                for (uint32_t j = 0; j < STAIR_K; j++)
                    if (j < n && lst[j] > last && lst[j] < best) best = lst[j];
                if (best == 0xFFFFFFFFu) break;
                last = best;
                const uint32_t ord = best - 1u;
                const uint32_t r = (ord >> 7) * MTR_CHUNK_SLOTS + (ord & 127u);
                const RecA a = load_rec(P.fb, r);
                const float z = z_at(a, (int32_t)x, (int32_t)y);
                if (!(z <= dep)) continue;  // fails LessEqual against the nearest of its predecessors
                dep = z;
                if (a.pad1 & 1u) {  // solid: its colour replaces the pixel
                    col = a.pad0;
                } else {
                    const DMat mat = P.mats[a.mat];
                    if (TEX && mat.shader == MTR_SH_TEXTURED) {
                        const RecB b = P.fb.rec_b[r];
                        float src[4];
                        sample_textured(a, b, mat, (int32_t)x, (int32_t)y, src);
                        col = blend_store(col, src, mat.blend == MTR_DB_ALPHA ? 1u : 0u);
                    } else {
                        col = mat.rgba8;
                    }
                }
            }
        } else if (key != 0ull) {
            dep = __uint_as_float(~(uint32_t)(key >> 32));
            const uint32_t ord = (uint32_t)key - 1u;
            const uint32_t r = (ord >> 7) * MTR_CHUNK_SLOTS + (ord & 127u);
            // the last word of the record: the source colour of a solid triangle (top byte 0xFF) or a material id
            const uint32_t payload = P.fb.rec_a[r].q1.w;
            if ((payload >> 24) != 0xFFu) {  // needs its material: in this kernel (order-free triangles only) a textured one
                const DMat mat = P.mats[payload & 0xFFFFFFu];
                if (TEX && mat.shader == MTR_SH_TEXTURED) {
                    const RecA a = load_rec(P.fb, r);
                    const RecB b = P.fb.rec_b[r];
                    col = shade_textured(a, b, mat, (int32_t)x, (int32_t)y);
                } else {
                    col = mat.rgba8;
                }
            } else {
                col = payload;
            }
        }
        const size_t pi = (size_t)y * P.fb.W + x;
        reinterpret_cast<uint32_t*>(P.color)[pi] = col;
        P.depth[pi] = dep;
    }
}

}  // namespace mtr

void mtr_launch_tile_vis(const TileParams& p, bool textured, hipStream_t s) {
    const uint32_t mine = p.fb.own.own_count;
    if (mine == 0) return;
    uint32_t grid = (mine + 7) / 8 * 8;
    if (p.xcd_run) grid = (grid / 8 + p.xcd_run - 1) / p.xcd_run * p.xcd_run * 8;  // whole runs
    int waves = 2;
    if (p.vis_waves) waves = (int)p.vis_waves;
    else if (mine <= 1536) waves = 8;   // 256 CUs: every bin is resident at once, the heaviest bin bounds the frame
    else if (mine <= 4096) waves = 4;
// QW (2 x 2 quad walk where a pass's boxes are large enough): all-opaque frames that have the GPU to themselves (latency);
// its own instantiation, so that the kernel of frames in flight keeps its registers and code
#define MTR_LAUNCH_VIS(T, W, S)                                                                                         \
    do {                                                                                                                \
        if (!S && p.quad_walk && MTR_QUAD_WALK) hipLaunchKernelGGL((mtr::k_tile_vis<T, W, false, true>), dim3(grid), dim3(64 * W), 0, s, p);  \
        else hipLaunchKernelGGL((mtr::k_tile_vis<T, W, S, false>), dim3(grid), dim3(64 * W), 0, s, p);                  \
    } while (0)
#define MTR_LAUNCH_VIS_W(T, S) do { if (waves >= 8) MTR_LAUNCH_VIS(T, 8, S); else if (waves >= 4) MTR_LAUNCH_VIS(T, 4, S); else MTR_LAUNCH_VIS(T, 2, S); } while (0)
    // frames with translucent materials keep per-pixel order lists (STAIR); all-opaque frames need only the key
    if (p.mixed) { if (textured) MTR_LAUNCH_VIS_W(true, true); else MTR_LAUNCH_VIS_W(false, true); }
    else { if (textured) MTR_LAUNCH_VIS_W(true, false); else MTR_LAUNCH_VIS_W(false, false); }
#undef MTR_LAUNCH_VIS_W
#undef MTR_LAUNCH_VIS
}
