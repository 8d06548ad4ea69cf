// k_geom.hip -- fused geometry stage of the rModel draw path (gfx950, wave64).
//
// One wave = one "chunk": 64 consecutive index-buffer positions of one primitive (2 halo + 62 new).
// lane = strip position.  Per lane: index fetch, vertex fetch + decode, linear-blend skinning
// against the LDS-staged palette, clip = M*(q,1), perspective divide, viewport, 1/256-px snap.
// Triangle (l-2, l-1, l) is assembled by lane l from its neighbours' registers (wave shuffles);
// primitive restart (0xFFFF, src/model.rs:251) and strip parity come from one wave ballot.
// Survivors of trivial reject / near clip / back-face cull / empty-bbox are compacted with
// ballot + popcount into ONE contiguous, submission-ordered run of records per wave, and the
// (triangle -> 16x16 bin) counts are accumulated with one 64-bit atomic per (wave, bin) group.
//
// Replaces everything between `rpass.draw_indexed` (src/model.rs:357-361) and the rasteriser.
#include "geom_common.h"

#include <algorithm>

namespace mtr {

struct PV {  // projected vertex
    int32_t X, Y;
    float z, iw, up, vp;
    uint32_t flags;  // bit0 valid index, bit1 projectable, bits 2..7 clip outcode
};

enum { OC_XN = 1, OC_XP = 2, OC_YN = 4, OC_YP = 8, OC_ZN = 16, OC_ZP = 32 };

__device__ __forceinline__ uint32_t outcode(const VOut& c) {
    uint32_t oc = 0;
    if (c.x < -c.w) oc |= OC_XN;
    if (c.x > c.w) oc |= OC_XP;
    if (c.y < -c.w) oc |= OC_YN;
    if (c.y > c.w) oc |= OC_YP;
    if (c.z < 0.0f) oc |= OC_ZN;
    if (c.z > c.w) oc |= OC_ZP;
    return oc;
}

// perspective divide + viewport (WebGPU: y down, depth range [0,1]) + snap; SPEC.md "projection"
__device__ __forceinline__ PV project(const VOut& c, uint32_t W, uint32_t H) {
    PV p;
    p.X = 0; p.Y = 0; p.z = 0.0f; p.iw = 0.0f; p.up = 0.0f; p.vp = 0.0f; p.flags = 0;
    if (!(c.w > 0.0f)) return p;
    float iw = 1.0f / c.w;
    float xn = c.x * iw, yn = c.y * iw, zn = c.z * iw;
    float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    float xf = fmaf(xn, hw, hw);
    float yf = fmaf(-yn, hh, hh);
    if (!(fabsf(xf) <= MTR_GUARD_BAND && fabsf(yf) <= MTR_GUARD_BAND)) return p;
    p.flags = 2;
    p.X = (int32_t)rintf(xf * 256.0f);
    p.Y = (int32_t)rintf(yf * 256.0f);
    p.z = zn;
    p.iw = iw;
    p.up = c.u * iw;
    p.vp = c.v * iw;
    return p;
}

struct Rec {
    RecA a;
    RecB b;
    RecHdr h;
};

// back-face cull (front = CCW, cull = Back: src/model.rs:252), pixel-centre bbox, record fill.
// *small (when asked for): the triangle is a candidate for sample_cull below.
__device__ __forceinline__ bool setup_tri(const PV& a, const PV& b_in, const PV& c_in, uint32_t W, uint32_t H, uint32_t mat,
                                          uint32_t cull, Rec& r, bool* small = nullptr) {
    if (!((a.flags & b_in.flags & c_in.flags) & 2)) return false;
    long long A2 = (long long)(c_in.X - a.X) * (long long)(b_in.Y - a.Y) - (long long)(b_in.X - a.X) * (long long)(c_in.Y - a.Y);
    // cull back (the reference) / none / front (material state): a kept back face is set up with b and c exchanged
    if (cull == 0u ? A2 <= 0 : (A2 == 0 || (cull == 2u && A2 > 0))) return false;
    const bool flip = A2 < 0;
    const PV& b = flip ? c_in : b_in;
    const PV& c = flip ? b_in : c_in;
    int32_t xmin = min(a.X, min(b.X, c.X)), xmax = max(a.X, max(b.X, c.X));
    int32_t ymin = min(a.Y, min(b.Y, c.Y)), ymax = max(a.Y, max(b.Y, c.Y));
    int32_t px0 = (xmin + 127) >> 8, px1 = (xmax - 128) >> 8;
    int32_t py0 = (ymin + 127) >> 8, py1 = (ymax - 128) >> 8;
    px0 = max(px0, 0); py0 = max(py0, 0);
    px1 = min(px1, (int32_t)W - 1); py1 = min(py1, (int32_t)H - 1);
    if (px0 > px1 || py0 > py1) return false;
    if (small) *small = ((xmax - xmin) | (ymax - ymin)) < 512;  // under 2 px across: its bbox holds at most 2 x 2 pixel centres
    r.h.bx0 = (uint16_t)(px0 >> MTR_BIN_SHIFT); r.h.bx1 = (uint16_t)(px1 >> MTR_BIN_SHIFT);
    r.h.by0 = (uint16_t)(py0 >> MTR_BIN_SHIFT); r.h.by1 = (uint16_t)(py1 >> MTR_BIN_SHIFT);
    r.a.X0 = a.X; r.a.Y0 = a.Y; r.a.X1 = b.X; r.a.Y1 = b.Y; r.a.X2 = c.X; r.a.Y2 = c.Y;
    r.a.z0 = a.z; r.a.z1 = b.z; r.a.z2 = c.z; r.a.mat = mat; r.a.pad0 = 0; r.a.pad1 = 0;
    r.b.iw0 = a.iw; r.b.iw1 = b.iw; r.b.iw2 = c.iw;
    r.b.up0 = a.up; r.b.up1 = b.up; r.b.up2 = c.up;
    r.b.vp0 = a.vp; r.b.vp1 = b.vp; r.b.vp2 = c.vp;
    r.b.pad0 = r.b.pad1 = r.b.pad2 = 0.0f;
    return true;
}

// Exact coverage of the (<= 2 x 2) pixel centres in the bbox of a triangle under 2 px across, with the tile kernels' inside test bit for bit: E_i(S) =
// dy (Sx - Xa) - dx (Sy - Ya) + (top-left ? 0 : -1) >= 0 for the three edges, which with (u_k, v_k) = S - P_k is
// u_b v_a - u_a v_b (exact in i32 at this size); a pixel to the right adds 256 dy, one down subtracts 256 dx.  Returns false
// when no centre is covered: the triangle is set up -- it counts in tris_setup like any triangle whose bbox holds a pixel
// centre (SPEC.md defines the statistic by the bbox) -- but nothing downstream could produce a fragment from it, so it gets
// no record and no queue entry.  Otherwise the bin rectangle shrinks to the covered centres: a covered pair seldom
// straddles two bins.  On the instanced configs two thirds of the set-up triangles go (C5: 4.04 M -> 1.25 M queue entries).
#ifndef MTR_SAMPLE_CULL
#define MTR_SAMPLE_CULL 1
#endif
#ifndef MTR_SAMPLE_FRAC
#define MTR_SAMPLE_FRAC 2u  // run the test in a wave when small triangles are at least 1 / MTR_SAMPLE_FRAC of what it set up
#endif
__device__ __forceinline__ bool sample_cull(Rec& r, uint32_t W, uint32_t H) {
    // the bbox of pixel centres again (setup_tri), here only for the waves that run the test: <= 2 x 2 centres at this size
    const int32_t xmin = min(r.a.X0, min(r.a.X1, r.a.X2)), xmax = max(r.a.X0, max(r.a.X1, r.a.X2));
    const int32_t ymin = min(r.a.Y0, min(r.a.Y1, r.a.Y2)), ymax = max(r.a.Y0, max(r.a.Y1, r.a.Y2));
    const int32_t px0 = max((xmin + 127) >> 8, 0), py0 = max((ymin + 127) >> 8, 0);
    const bool two_x = min((xmax - 128) >> 8, (int32_t)W - 1) > px0, two_y = min((ymax - 128) >> 8, (int32_t)H - 1) > py0;
    const int32_t Sx = px0 * 256 + 128, Sy = py0 * 256 + 128;
    const int32_t u[3] = {Sx - r.a.X0, Sx - r.a.X1, Sx - r.a.X2}, v[3] = {Sy - r.a.Y0, Sy - r.a.Y1, Sy - r.a.Y2};
    int32_t e[3], ex[3], ey[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int ia = (i + 1) % 3, ib = (i + 2) % 3;
        const int32_t ndx = u[ib] - u[ia], dy = v[ia] - v[ib];  // -dx, dy of the edge; |dx|, |dy| < 2^10
        // top-left (dy > 0, or dy == 0 and dx < 0)  <=>  dy * 4096 - dx > 0; the bias is 0 for such an edge, -1 otherwise
        const int32_t bias = (((dy << 12) + ndx) - 1) >> 31;
        e[i] = __mul24(u[ib], v[ia]) - __mul24(u[ia], v[ib]) + bias;
        ex[i] = dy << 8; ey[i] = ndx << 8;
    }
    const bool m00 = (e[0] | e[1] | e[2]) >= 0;
    bool m10 = false, m01 = false, m11 = false;
    if (__ballot(two_x || two_y)) {  // some lane's bbox holds a second column or row (a sub-pixel mesh: seldom)
        m10 = two_x && ((e[0] + ex[0]) | (e[1] + ex[1]) | (e[2] + ex[2])) >= 0;
        m01 = two_y && ((e[0] + ey[0]) | (e[1] + ey[1]) | (e[2] + ey[2])) >= 0;
        m11 = two_x && two_y && ((e[0] + ex[0] + ey[0]) | (e[1] + ex[1] + ey[1]) | (e[2] + ex[2] + ey[2])) >= 0;
    }
    if (!(m00 || m10 || m01 || m11)) return false;  // r.h keeps the bins of the bbox: a sharded rank counts it if it owns one
    if (two_x || two_y) {
        const int32_t qx0 = (m00 || m01) ? px0 : px0 + 1, qx1 = (m10 || m11) ? px0 + 1 : px0;
        const int32_t qy0 = (m00 || m10) ? py0 : py0 + 1, qy1 = (m01 || m11) ? py0 + 1 : py0;
        r.h.bx0 = (uint16_t)(qx0 >> MTR_BIN_SHIFT); r.h.bx1 = (uint16_t)(qx1 >> MTR_BIN_SHIFT);
        r.h.by0 = (uint16_t)(qy0 >> MTR_BIN_SHIFT); r.h.by1 = (uint16_t)(qy1 >> MTR_BIN_SHIFT);
    }
    return true;
}

__device__ __forceinline__ VOut clip_lerp(const VOut& in, const VOut& out) {  // in.z >= 0 > out.z
    float t = in.z / (in.z - out.z);
    VOut r;
    r.x = fmaf(t, out.x - in.x, in.x);
    r.y = fmaf(t, out.y - in.y, in.y);
    r.z = 0.0f;
    r.w = fmaf(t, out.w - in.w, in.w);
    r.u = fmaf(t, out.u - in.u, in.u);
    r.v = fmaf(t, out.v - in.v, in.v);
    return r;
}

// guard-band planes (SPEC.md 5.3): 0: x <= 64 w, 1: x >= -64 w, 2: y <= 64 w, 3: y >= -64 w; inside: distance >= 0
#define MTR_MAX_POLY 8
__device__ __forceinline__ float guard_dist(const VOut& v, int plane) {
    return plane == 0 ? fmaf(64.0f, v.w, -v.x) : plane == 1 ? fmaf(64.0f, v.w, v.x) : plane == 2 ? fmaf(64.0f, v.w, -v.y) : fmaf(64.0f, v.w, v.y);
}
__device__ __forceinline__ VOut plane_lerp(const VOut& in, const VOut& out, float din, float dout) {  // din >= 0 > dout
    const float t = din / (din - dout);
    VOut r;
    r.x = fmaf(t, out.x - in.x, in.x);
    r.y = fmaf(t, out.y - in.y, in.y);
    r.z = fmaf(t, out.z - in.z, in.z);
    r.w = fmaf(t, out.w - in.w, in.w);
    r.u = fmaf(t, out.u - in.u, in.u);
    r.v = fmaf(t, out.v - in.v, in.v);
    return r;
}

// + the instance's clip matrix M = view_proj * model, ONCE per workgroup: sixteen threads compute one element each (a k-ordered
// fma chain from 0, like every contraction of the numeric contract) into s_M.  Every wave used to compose its own copy -- 32 packed fmas + 30 moves per
// wave, 11 % of k_geom's VALU instructions on the instanced configs (PMC ablation, round 3) -- and then hold it in 16
// VGPRs for the whole kernel; the vertex stage now reads the row it needs straight from LDS.
__device__ __forceinline__ void stage_palette(const GeomParams& P, uint32_t inst, float* s_pal, float* s_M) {
    if (threadIdx.x < 16) {
        const uint32_t c = threadIdx.x >> 2, i = threadIdx.x & 3u;
        float a = P.vp[threadIdx.x];
        if (P.model_mats) {
            const float* B = P.model_mats + (size_t)inst * 16;
            a = 0.0f;
#pragma unroll
            for (int k = 0; k < 4; k++) a = fmaf(P.vp[k * 4 + i], B[c * 4 + k], a);
        }
        s_M[threadIdx.x] = a;
    }
    if (P.palettes && P.npal) {
        const float4* src = reinterpret_cast<const float4*>(P.palettes + (size_t)inst * P.pal_stride);
        for (uint32_t i = threadIdx.x; i < P.npal * 4; i += blockDim.x) reinterpret_cast<float4*>(s_pal)[i] = src[i];
    }
    __syncthreads();
}

// the strip position shading lane `lane` takes (rail order for strips, identity for lists; geom_chunk)
__device__ __forceinline__ uint32_t shading_position(const DPrim& pr, uint32_t lane) {
    return pr.topology == 4 ? (((lane & 31u) << 1) | (lane >> 5)) : lane;
}
// index fetch (Uint16, src/model.rs:307) for that position; 0xFFFF outside the primitive's index range
__device__ __forceinline__ uint32_t fetch_index(const GeomParams& P, const DChunk& ch, const DPrim& pr, uint32_t lane) {
    const int32_t ps = (int32_t)ch.start - 2 + (int32_t)shading_position(pr, lane);
    return (ps >= 0 && (uint32_t)ps < pr.index_num) ? (uint32_t)P.ibuf[pr.index_ofs + (uint32_t)ps] : 0xFFFFu;
}

// MODE 0: count pass of the exact two-pass queues; 1: single-pass binning, ordered segments; 2: single-pass binning
// for frames the visibility-key tile kernel renders (no order kept, no segments)
template <int MODE>
__device__ __forceinline__ void geom_chunk(const GeomParams& P, uint32_t inst, uint32_t c, const DChunk& ch, const DPrim& pr,
                                           bool skinned, const float* s_M, const float* s_pal, RecHdr* s_hdr, uint32_t* s_slot,
                                           float4* s_pv, uint32_t lane, uint32_t idx) {
    const uint32_t gid = P.chunk_base + inst * P.nchunks + c;
    const uint32_t mat = P.mat_base + inst * P.mat_inst_stride + ch.prim;
    const uint32_t W = P.fb.W, H = P.fb.H;
    const bool strip = pr.topology == 4;
    const DMat dmat = P.mats[mat];  // wave-uniform
    // texcoord planes are only read by textured materials: no registers, LDS or HBM traffic for them otherwise
    const bool want_b = pr.has_uv && dmat.shader == MTR_SH_TEXTURED;

    // ---- vertex stage, in RAIL order.  The matrix-core skinning needs the four lanes of a block to share their joint
    //      indices (geom_common.h: shade_vertex_mfma).  In strip order neighbouring lanes alternate between the two rails of
    //      the strip -- two rows of the mesh, which as often as not hang on different joints -- and every wave then also ran
    //      the whole VALU fallback for its incoherent blocks: 128 of ~600 VALU instructions per chunk (PMC, C5).  So the
    //      vertex stage runs permuted: shading lane s takes strip position pi(s) = the even positions first, then the odd
    //      ones; a block is four consecutive vertices of ONE rail.  The projected vertices go through 1-2 KB of LDS private
    //      to the wave (written at pi(s), read back at the lane's own position): no shuffle, no workgroup barrier.
    //      Triangle lists keep the identity order. ----
    const uint32_t sp = shading_position(pr, lane);
    {
        // `idx`: the index at this lane's shading position (fetch_index: the caller issues the load before the palette
        // barrier, so its round trip overlaps with the staging instead of following it)
        const int32_t ps = (int32_t)ch.start - 2 + (int32_t)sp;
        const bool in_s = ps >= 0 && (uint32_t)ps < pr.index_num;
        const bool restart_s = !in_s || (strip && idx == 0xFFFFu);
        const uint32_t vid_s = idx + pr.index_base;  // base_vertex = index_base, src/model.rs:359
        const bool valid_s = !restart_s && vid_s < pr.vertex_num;
        // skinning + MVP on the matrix cores, wave-wide (all 64 lanes issue the MFMAs; invalid lanes carry zeros)
#ifndef MTR_ABL  // instruction-count ablations (tools/abl_geom.sh): what each part of the kernel costs, by PMC difference
#define MTR_ABL 0
#endif
#if MTR_ABL == 2   // no vertex shader at all
        VOut v; v.x = (float)(vid_s & 1023u) * 0.001f - 0.5f; v.y = (float)(vid_s >> 10) * 0.01f - 0.3f; v.z = 0.5f; v.w = 1.0f; v.u = v.v = 0.0f;
#else
        const VOut v = shade_vertex_mfma(P.vbuf, pr, vid_s, valid_s, s_M, s_pal, P.npal, MTR_ABL == 1 ? false : skinned, want_b);
#endif
        PV q;
        q.X = 0; q.Y = 0; q.z = 0.0f; q.iw = 0.0f; q.up = 0.0f; q.vp = 0.0f; q.flags = 0;
        if (valid_s) {
#if MTR_ABL == 6   // no projection
            q.X = (int32_t)(v.x * 1000.0f); q.Y = (int32_t)(v.y * 1000.0f); q.z = v.z; q.flags = 2u;
#else
            q = project(v, W, H);
#endif
            q.flags |= 1u | (outcode(v) << 2);
        }
        // flags word: bits 0..7 PV flags, bit 8 restart, bits 16..31 the vertex id (valid ids are < 65536: vertex_num is 16 bits)
        const uint32_t fw = q.flags | (restart_s ? 0x100u : 0u) | (vid_s << 16);
        s_pv[sp] = make_float4(__int_as_float(q.X), __int_as_float(q.Y), q.z, __uint_as_float(fw));
        if (want_b) s_pv[64 + sp] = make_float4(q.iw, q.up, q.vp, 0.0f);
    }
    // s_pv is private to this wave: no workgroup barrier, LDS ops of one wave are ordered
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int32_t p = (int32_t)ch.start - 2 + (int32_t)lane;
    const bool in_range = p >= 0 && (uint32_t)p < pr.index_num;
    PV me;
    me.iw = 0.0f; me.up = 0.0f; me.vp = 0.0f;
    uint32_t vid;
    bool is_restart;
    {
        const float4 a = s_pv[lane];
        const uint32_t fw = __float_as_uint(a.w);
        me.X = __float_as_int(a.x); me.Y = __float_as_int(a.y); me.z = a.z; me.flags = fw & 0xFFu;
        is_restart = (fw & 0x100u) != 0;
        vid = fw >> 16;
        if (want_b) {
            const float4 b = s_pv[64 + lane];
            me.iw = b.x; me.up = b.y; me.vp = b.z;
        }
    }

#if MTR_ABL == 3   // vertex stage only
    if (me.X == 0x7fffffff) P.fb.counters[CTR_OVERFLOW] = vid;  // keep the loads alive
    return;
#endif
    // ---- strip assembly: which lanes complete a triangle, and its winding parity ----
    const uint64_t R = __ballot(is_restart);
    const uint64_t below = R & ((1ull << lane) - 1ull);
    uint32_t q = 0;  // indices since the last restart, this one included
    if (!is_restart) q = below ? lane - (63u - (uint32_t)__clzll((long long)below)) : ch.q_before + lane + 1u;
    bool tri, odd;
    if (strip) {
        tri = q >= 3 && lane >= 2;
        odd = ((q - 3) & 1) != 0;
    } else {
        tri = lane >= 2 && in_range && ((uint32_t)p % 3u == 2u);
        odd = false;
    }

    // lanes l-1 and l-2 by DPP wave_shr:1 (VALU moves instead of the 14 ds_bpermute behind __shfl_up; same speed in an
    // A/B run, tools/sweep_ab.sh, but the LDS pipe stays free for the palette reads)
    PV v1, v2;
    v1.iw = v1.up = v1.vp = v2.iw = v2.up = v2.vp = 0.0f;
#define MTR_SHR1(x) __builtin_amdgcn_update_dpp((int)(x), (int)(x), 0x138, 0xf, 0xf, false)
    v1.X = MTR_SHR1(me.X); v2.X = MTR_SHR1(v1.X);
    v1.Y = MTR_SHR1(me.Y); v2.Y = MTR_SHR1(v1.Y);
    v1.z = __int_as_float(MTR_SHR1(__float_as_int(me.z))); v2.z = __int_as_float(MTR_SHR1(__float_as_int(v1.z)));
    if (want_b) {
        v1.iw = __int_as_float(MTR_SHR1(__float_as_int(me.iw))); v2.iw = __int_as_float(MTR_SHR1(__float_as_int(v1.iw)));
        v1.up = __int_as_float(MTR_SHR1(__float_as_int(me.up))); v2.up = __int_as_float(MTR_SHR1(__float_as_int(v1.up)));
        v1.vp = __int_as_float(MTR_SHR1(__float_as_int(me.vp))); v2.vp = __int_as_float(MTR_SHR1(__float_as_int(v1.vp)));
    }
    v1.flags = (uint32_t)MTR_SHR1(me.flags); v2.flags = (uint32_t)MTR_SHR1(v1.flags);
    const uint32_t vid1 = (uint32_t)MTR_SHR1(vid), vid2 = (uint32_t)MTR_SHR1(vid1);
#undef MTR_SHR1

    // triangle i of a strip = (v_i, v_{i+1}, v_{i+2}) for even i, (v_i, v_{i+2}, v_{i+1}) for odd i
    const PV& ta = v2;
    const PV& tb = odd ? me : v1;
    const PV& tc = odd ? v1 : me;

    Rec r0, r1;
    uint32_t n_out = 0;   // records held in r0 / r1
    bool empty = false;   // set up, covers no pixel centre: counted, not recorded (sample_cull)
    bool small = false;   // a candidate for sample_cull (setup_tri)
    // guard-band clipping (SPEC.md 5.3): a polygon with a vertex the rasteriser cannot take (w <= 0, or beyond +-2^20 px)
    // is clipped against |x| <= 64 w, |y| <= 64 w and may fan into up to six triangles; its lane keeps the polygon (in
    // scratch) and writes the records in a second pass, once every lane's place in the chunk's run is known
    VOut poly[MTR_MAX_POLY];
    uint32_t n_poly = 0;
    const uint32_t f_and = ta.flags & tb.flags & tc.flags, f_or = ta.flags | tb.flags | tc.flags;
    tri = tri && (f_and & 1u);            // every vertex inside the bound slice (SPEC.md)
    tri = tri && ((f_and >> 2) == 0);     // trivial frustum reject
#if MTR_ABL == 5   // no triangle set-up
    tri = tri && ta.X == 0x7ffffff0;
#endif
    if (tri) {
        if (!((f_or >> 2) & OC_ZN) && (f_and & 2u)) {
            if (setup_tri(ta, tb, tc, W, H, mat, pr.cull, r0, &small)) n_out = 1;
        } else {
            // near-plane clip (z >= 0) and / or guard-band clip: rare, re-shades the three vertices in clip space
            const uint32_t ia = vid2, ib = odd ? vid : vid1, ic = odd ? vid1 : vid;
            float M[16];
#pragma unroll
            for (int i = 0; i < 16; i++) M[i] = s_M[i];
            VOut cv[3];
#pragma nounroll
            for (int i = 0; i < 3; i++) cv[i] = shade_vertex(P.vbuf, pr, i == 0 ? ia : (i == 1 ? ib : ic), M, s_pal, P.npal, skinned);
            int n = 3;
            if ((f_or >> 2) & OC_ZN) {
                n = 0;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const VOut& pp = cv[i];
                    const VOut& qq = cv[(i + 1) % 3];
                    bool pin = !(pp.z < 0.0f), qin = !(qq.z < 0.0f);
                    if (pin) poly[n++] = pp;
                    if (pin != qin) poly[n++] = pin ? clip_lerp(pp, qq) : clip_lerp(qq, pp);
                }
            } else {
                poly[0] = cv[0]; poly[1] = cv[1]; poly[2] = cv[2];
            }
            if (n >= 3) {
                bool all_ok = true;
                for (int i = 0; i < n; i++) all_ok = all_ok && (project(poly[i], W, H).flags & 2u);
                if (all_ok) {
                    PV q0 = project(poly[0], W, H), q1 = project(poly[1], W, H), q2 = project(poly[2], W, H);
                    bool s0 = setup_tri(q0, q1, q2, W, H, mat, pr.cull, r0);
                    bool s1 = false;
                    if (n == 4) {
                        PV q3 = project(poly[3], W, H);
                        s1 = setup_tri(q0, q2, q3, W, H, mat, pr.cull, s0 ? r1 : r0);
                    }
                    n_out = (s0 ? 1u : 0u) + (s1 ? 1u : 0u);
                } else {
                    VOut tmp[MTR_MAX_POLY];
                    for (int plane = 0; plane < 4 && n >= 3; plane++) {
                        int m = 0;
                        for (int i = 0; i < n; i++) {
                            const VOut pp = poly[i], qq = poly[(i + 1) % n];
                            const float dp = guard_dist(pp, plane), dq = guard_dist(qq, plane);
                            const bool pin = !(dp < 0.0f), qin = !(dq < 0.0f);
                            if (pin && m < MTR_MAX_POLY) tmp[m++] = pp;
                            if (pin != qin && m < MTR_MAX_POLY) tmp[m++] = pin ? plane_lerp(pp, qq, dp, dq) : plane_lerp(qq, pp, dq, dp);
                        }
                        for (int i = 0; i < m; i++) poly[i] = tmp[i];
                        n = m;
                    }
                    if (n >= 3) n_poly = (uint32_t)n;
                }
            }
        }
    }

    // ---- sharded frames: a rank keeps only the triangles whose bin rectangle holds one of its bins, so records,
    //      queue traffic and binning work shrink with the shard.  Relative order is preserved.
#if MTR_SAMPLE_CULL
    // the coverage test costs every wave that runs it the same ~40 instructions: worth them where most of what the wave
    // set up is small (the instanced configs), not where a few lanes are (the headline model: 4 % of its entries go)
    {
        const uint64_t sm = __ballot(small);
        if (sm && (uint32_t)__popcll(sm) * MTR_SAMPLE_FRAC >= (uint32_t)__popcll(__ballot(n_out != 0u)) && small && !sample_cull(r0, W, H)) {
            n_out = 0;
            empty = true;
        }
    }
#endif
    if (P.fb.own.world > 1 && empty) empty = rect_owned_any(P.fb.own, r0.h.bx0, r0.h.by0, r0.h.bx1, r0.h.by1, P.fb.nbx);
    if (P.fb.own.world > 1 && n_out) {
        const bool k0 = rect_owned_any(P.fb.own, r0.h.bx0, r0.h.by0, r0.h.bx1, r0.h.by1, P.fb.nbx);
        const bool k1 = n_out == 2 && rect_owned_any(P.fb.own, r1.h.bx0, r1.h.by0, r1.h.bx1, r1.h.by1, P.fb.nbx);
        if (!k0 && k1) r0 = r1;
        n_out = (k0 ? 1u : 0u) + (k1 ? 1u : 0u);
    }

    // ---- wave compaction: one contiguous, ordered run of records per chunk.  A guard-clipped lane reserves a slot for
    //      every triangle of its fan (a culled or empty one leaves a hole: an empty bin rectangle that nothing reads) ----
    const uint32_t n_res = n_poly ? n_poly - 2u : n_out;
    uint32_t rank, total;
    const uint64_t b1 = __ballot(n_out >= 1), b2 = __ballot(n_out == 2);
    const uint32_t n_empty = (uint32_t)__popcll(__ballot(empty));
    if (!__ballot(n_poly != 0)) {
        const uint64_t lt = (1ull << lane) - 1ull;
        rank = (uint32_t)__popcll(b1 & lt) + (uint32_t)__popcll(b2 & lt);
        total = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
    } else {
        const uint32_t inc = wave_incl_scan_u32(n_res);
        rank = inc - n_res;
        total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        if (total > MTR_CHUNK_SLOTS) {  // the run has a fixed place and size: report, emit nothing (the host fails the frame)
            if (lane == 0) atomicOr(&P.fb.counters[CTR_OVERFLOW], 1u);
            total = 0;
        }
    }
    // the run lives at a fixed place (chunk id * MTR_CHUNK_SLOTS): no allocator, no hot counter
    const uint32_t base = gid * MTR_CHUNK_SLOTS;
    if (lane == 0) {
        if (MODE == 0) {  // k_fill walks the runs
            ChunkInfo ci = {base, total};
            P.fb.chunk_info[gid] = ci;
        }
        const uint32_t nrec = (uint32_t)__popcll(b1) + (uint32_t)__popcll(b2);
        const uint32_t nset = (total ? nrec : 0u) + n_empty;
        if (nset) atomicAdd(&P.fb.counters[MTR_CTR(CTR_REC, gid)], nset);  // statistics only (tris_setup)
    }
    total = __builtin_amdgcn_readfirstlane(total);
    if (total == 0) return;
    // a solid colour in the default depth state replaces the pixel whatever the blend (its alpha is 1): the fragment
    // stage finds the colour in the record itself, no dependent material lookup; everything else carries its material id
    const uint32_t solid = (dmat.shader != MTR_SH_TEXTURED && dmat.blend != MTR_DB_ADD && dmat.dstate == 3u) ? 1u : 0u;
    // hard: the pixel depends on the order of ALL its fragments (additive blend, depth write or test off); a translucent
    // material that is not hard depends only on the fragments that pass the depth test, which prefix minima of z identify
    const uint32_t hard = (dmat.blend == MTR_DB_ADD || dmat.dstate != 3u) ? 1u : 0u;
    const uint32_t mflags = solid | ((dmat.blend != MTR_DB_OFF ? 1u : 0u) << 8) | (dmat.translucent << 16) | (hard << 17);
    r0.a.pad0 = dmat.rgba8; r0.a.pad1 = mflags;
    r1.a.pad0 = dmat.rgba8; r1.a.pad1 = mflags;
    auto put_rec = [&](uint32_t slot, const Rec& r) {
        const bool lg = rec_is_large(r.a);
        P.fb.rec_a[base + slot] = rec_pack(r.a, lg);
        if (lg) P.fb.rec_l[base + slot] = make_int4(r.a.X1, r.a.Y1, r.a.X2, r.a.Y2);
        if (MODE == 0) P.fb.rec_hdr[base + slot] = r.h;  // read back by k_fill only
        if (want_b) P.fb.rec_b[base + slot] = r.b;
        s_hdr[slot] = r.h;
    };
    if (n_out >= 1) put_rec(rank, r0);
    if (n_out == 2) put_rec(rank + 1, r1);
    if (n_poly) {  // the fan of a guard-clipped polygon: (p0, p_k+1, p_k+2)
        uint32_t nvalid = 0;
        for (uint32_t k = 0; k + 2u < n_poly; k++) {
            Rec t;
            const PV q0 = project(poly[0], W, H), q1 = project(poly[k + 1], W, H), q2 = project(poly[k + 2], W, H);
            bool ok = setup_tri(q0, q1, q2, W, H, mat, pr.cull, t);
            if (ok && P.fb.own.world > 1) ok = rect_owned_any(P.fb.own, t.h.bx0, t.h.by0, t.h.bx1, t.h.by1, P.fb.nbx);
            if (ok) {
                t.a.pad0 = dmat.rgba8; t.a.pad1 = mflags;
                put_rec(rank + k, t);
                nvalid++;
            } else {
                const RecHdr hole = {1, 1, 0, 0};  // bx0 > bx1: no bin
                s_hdr[rank + k] = hole;
                if (MODE == 0) P.fb.rec_hdr[base + rank + k] = hole;
            }
        }
        if (nvalid) atomicAdd(&P.fb.counters[MTR_CTR(CTR_REC, gid + lane)], nvalid);  // statistics only
    }
    // s_hdr is private to this wave: no workgroup barrier, LDS ops of one wave are ordered
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#if MTR_ABL == 4   // no binning
    return;
#endif
    // ---- triangle -> bin: DIRECT = single-pass binning straight into the bounded per-bin queues;
    //      otherwise count only (one non-returning atomic per (wave, bin) group), k_scan + k_fill follow ----
    for (uint32_t round = 0; round * 64 < total; ++round) {
        const uint32_t j = round * 64 + lane;
        bool act = j < total;
        RecHdr h = {0, 0, 0, 0};
        if (act) h = s_hdr[j];
        act = act && h.bx0 <= h.bx1;  // a hole left by a guard-clipped fan
        if (MODE == 2) {
            emit_bins_unordered(P.fb, h, act, gid, round, lane, s_slot);
        } else if (MODE == 1) {
            emit_bins<true>(P.fb, h, act, gid, round, lane);
        } else {
            count_bins(P.fb, h, act, lane);
        }
    }
}

// Length of the instance list: a SCALAR load of a uniform address.  Written as `count ? *count : ninst` the compiler selects
// between the two ADDRESSES (the kernel argument's and the counter's) and loads through a flat VECTOR instruction -- the first
// thing every wave of a sharded launch waits for, the ones of dead instance slots included (ISA of round 3: flat_load_dword
// + v_cmp + s_and_saveexec).  Hiding the argument's value from that transformation leaves a branch and an s_load_dword.
__device__ __forceinline__ uint32_t live_instances(const uint32_t* count, uint32_t ninst) {
    asm("" : "+s"(ninst));
    if (count) ninst = *count;
    return ninst;
}

#ifndef GEOM_OCC
#define GEOM_OCC 8  // waves per SIMD the register allocator must leave room for: 64 VGPRs.  Rounds 1-2 ran at 6 (80 VGPRs,
                    // the vertex stage's live ranges spilled below that); with the vertex stage feeding LDS the kernel fits 64
                    // without spills and the instanced configs gain 8-10 % (C5 0.903 -> 0.813 ms; tools/sweep_geom_occ.sh)
#define GEOM_OCC_SMALL 7  // a draw that does not fill the GPU (the headline model: 16 k waves) overlaps with the neighbouring
                    // frames' tile kernels for most of its life and does better leaving them a wave slot per SIMD, with the
                    // 72 registers that allows: 0.0485 ms per frame against 0.0509 at 8 and 0.0498 at 6 (three runs each,
                    // tools/sweep_geom_occ.sh); capping the residency of the 64-register build by LDS gets 0.0498
#endif
// 256 threads, wave = one chunk (62 strip positions) of one instance.  Unsharded frames (CULL false): grid = (blocks
// of 4 chunks, instances).  Sharded frames (CULL true): k_cull_chunks has bounded every chunk against the rank's bins
// and written the survivors to a work list; the workgroups stride over it.  (Testing the bounds inside this kernel --
// 80 registers, 6 workgroups per CU -- put the test's chain of dependent loads on every workgroup's critical path:
// +11 us on the headline scene even when nothing was culled.)
template <int MODE, bool CULL, int OCC>
__global__ __launch_bounds__(256, OCC) void k_geom(GeomParams P) {
    extern __shared__ __align__(16) float s_pal[];
    __shared__ RecHdr s_hdr[4][MTR_CHUNK_SLOTS + 4];
    __shared__ uint32_t s_slot[MODE == 2 ? 4 : 1][128];  // unordered binning: per-wave bin-window counters / offsets
    __shared__ float4 s_pv[4][128];                       // per wave: the projected vertices of its 64 strip positions (+ texcoords)
    __shared__ float s_M[16];                             // the instance's clip matrix (stage_palette)
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (!CULL) {  // one instance per blockIdx.y, nothing to test
        // blocks b and b+8 share an XCD (observed round-robin dispatch): give each XCD a contiguous run of
        // chunks so neighbouring strips (which share a vertex row) hit the same L2.  Speed only.
        const uint32_t nblk = gridDim.x;
        const uint32_t per = (nblk + 7) / 8;
        const uint32_t blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        const uint32_t c = blk * 4 + wave;
        if (blk >= nblk) return;  // whole workgroup
        // the chunk, its primitive and the wave's indices are requested BEFORE the palette is staged: three dependent round
        // trips that used to follow the staging barrier now overlap with it
        const bool has = c < P.nchunks;
        DChunk ch = {};
        DPrim pr = {};
        uint32_t idx = 0xFFFFu;
        if (has) { ch = P.chunks[c]; pr = P.prims[ch.prim]; idx = fetch_index(P, ch, pr, lane); }
        stage_palette(P, blockIdx.y, s_pal, s_M);  // every thread of the workgroup copies its share
        if (!has) return;
        geom_chunk<MODE>(P, blockIdx.y, c, ch, pr, pr.skinnable && P.palettes && P.npal, s_M, s_pal, s_hdr[wave], s_slot[wave], s_pv[wave], lane, idx);
        return;
    }
    // sharded: workgroup g = (group x, quarter q) fastest, then instance slot ii: it takes the q-th four survivors of the 16
    // chunks of group x of instance slot ii, from the mask k_cull_chunks stored.  The launch covers every instance slot.
    // Slots past the end of the instance list leave after one scalar load of its length (one address: a scalar-cache hit)
    // and sit at the END of the launch, where they cost next to nothing -- 53 k instead of 262 k workgroups launched
    // changed nothing.  Spread AMONG working workgroups they do cost: instance slot fastest-varying 213 us, a
    // multiplicative walk over the slots 234 us, against 189 us for this order (C5, rank 3 of 8) -- the dispatcher
    // hands out workgroups in order, and four that leave at once per one that works halve the rate at which work starts.
    // (Fetching the group's 16 chunk descriptors together with the mask, lane = chunk, to take one load off the chain in
    // front of the vertex work: 192 vs 189 us, no gain.)
    const uint32_t nxq = P.work_nx * 4u, ii = blockIdx.x / nxq, xq = blockIdx.x - ii * nxq, x = xq >> 2, q = xq & 3u;  // slots [0, grid / nxq)
    const uint32_t nlive = live_instances(P.inst_count, P.ninst);
    if (ii >= nlive) return;  // before touching anything else: a dead slot must cost no memory traffic
    // the mask through a SCALAR load (the aligned dword that holds it: a 16-bit load of a uniform address is still a
    // vector-memory load, ~1 us under load, and the instance-list load waited behind it)
    const size_t mi = (size_t)ii * P.work_nx + x;
    const uint32_t mword = reinterpret_cast<const uint32_t*>(P.work_mask)[mi >> 1];
    const uint32_t inst = P.inst_list ? P.inst_list[ii] : ii;
    const uint32_t m16 = (mi & 1u) ? mword >> 16 : mword & 0xFFFFu;
    const uint32_t k = (uint32_t)__popc(m16);
    if (q * 4u >= k) return;
    const uint32_t nth = q * 4u + wave;  // this wave's survivor
    const bool has = nth < k;
    uint32_t mm = m16;
    for (uint32_t t = 0; t < nth && mm; t++) mm &= mm - 1u;  // drop the nth lowest set bits (wave-uniform)
    const uint32_t c = x * 16u + (mm ? (uint32_t)__ffs((int)mm) - 1u : 0u);
    DChunk ch = {};
    DPrim pr = {};
    uint32_t idx = 0xFFFFu;
    if (has) { ch = P.chunks[c]; pr = P.prims[ch.prim]; idx = fetch_index(P, ch, pr, lane); }  // before the barrier, as above
    stage_palette(P, inst, s_pal, s_M);
    if (!has) return;
    geom_chunk<MODE>(P, inst, c, ch, pr, pr.skinnable && P.palettes && P.npal, s_M, s_pal, s_hdr[wave], s_slot[wave], s_pv[wave], lane, idx);
}

// The instance slots the full-rate launch of k_geom<MODE, true> does not cover (mtr_launch_geom): MTR_GEOM_REST_SPLIT
// workgroups per slot.  Nearly always the slot is past the end of the instance list and they leave; otherwise each walks
// every MTR_GEOM_REST_SPLIT-th mask of the instance itself, four survivors at a time (the palette is staged once per
// workgroup).  Sixteen per slot: a single straggling instance costs ~13 sequential steps (~0.06 ms), hundreds of them fill the
// chip as the full-rate launch would.  Its own kernel: the loop's register allocation must not touch the hot kernel's.
#define MTR_GEOM_REST_SPLIT 16u
template <int MODE>
__global__ __launch_bounds__(256) void k_geom_rest(GeomParams P) {
    extern __shared__ __align__(16) float s_pal[];
    __shared__ RecHdr s_hdr[4][MTR_CHUNK_SLOTS + 4];
    __shared__ uint32_t s_slot[MODE == 2 ? 4 : 1][128];
    __shared__ float4 s_pv[4][128];
    __shared__ float s_M[16];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t split = min(P.work_nx, P.rest_split);
    const uint32_t ii = P.work_slot_base + blockIdx.x / split;
    const uint32_t nlive = live_instances(P.inst_count, P.ninst);
    if (ii >= nlive) return;
    const uint32_t inst = P.inst_list ? P.inst_list[ii] : ii;
    stage_palette(P, inst, s_pal, s_M);
    for (uint32_t x = blockIdx.x % split; x < P.work_nx; x += split) {
        const uint32_t m16 = P.work_mask[(size_t)ii * P.work_nx + x];
        const uint32_t k = (uint32_t)__popc(m16);
        for (uint32_t nth = wave; nth < k; nth += 4) {
            uint32_t mm = m16;
            for (uint32_t t = 0; t < nth; t++) mm &= mm - 1u;
            const uint32_t c = x * 16u + (uint32_t)__ffs((int)mm) - 1u;
            const DChunk ch = P.chunks[c];
            const DPrim pr = P.prims[ch.prim];
            geom_chunk<MODE>(P, inst, c, ch, pr, pr.skinnable && P.palettes && P.npal, s_M, s_pal, s_hdr[wave], s_slot[wave], s_pv[wave], lane,
                             fetch_index(P, ch, pr, lane));
        }
    }
}

__device__ __forceinline__ void compose_vp_model(const float (&vp)[16], const float* model_mats, uint32_t inst, float (&M)[16]) {
    if (model_mats) {
        const float* B = model_mats + (size_t)inst * 16;  // M = VP * Model, the fma chain of stage_palette
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                float a = 0.0f;
#pragma unroll
                for (int k = 0; k < 4; k++) a = fmaf(vp[k * 4 + i], B[c * 4 + k], a);
                M[c * 4 + i] = a;
            }
    } else {
#pragma unroll
        for (int i = 0; i < 16; i++) M[i] = vp[i];
    }
}

// Instance culling of a sharded batch draw, one wave per instance, lane = per-joint box of the whole model: the
// instances that may reach a bin of this rank are appended to `list` (in no particular order: k_geom takes the instance
// number from the work list, so submission-order keys do not change).
__global__ __launch_bounds__(64) void k_cull_instances(CullParams P) {
    const uint32_t inst = blockIdx.x, lane = threadIdx.x;
    if (inst >= P.ninst) return;
    float M[16];
    compose_vp_model(P.vp, P.model_mats, inst, M);
    const bool have_pal = P.palettes && P.npal;
    const float* pal = have_pal ? P.palettes + (size_t)inst * P.pal_stride : nullptr;
    ClipBox cb;
#pragma unroll
    for (int t = 0; t < 3; t++) { cb.lo[t] = __builtin_inff(); cb.hi[t] = -__builtin_inff(); }
    bool bad = false;
    for (uint32_t i = lane; i < P.nboxes; i += 64) {
        const BoneBox bx = P.boxes[i];
        const float* Pm = nullptr;
        if (bx.joint != MTR_BOX_UNSKINNED && have_pal) Pm = pal + (size_t)min(bx.joint, P.npal - 1u) * 16;
        const ClipBox one = box_clip_interval(bx, Pm, M);
        bad = bad || !clipbox_finite(one);
#pragma unroll
        for (int t = 0; t < 3; t++) { cb.lo[t] = fminf(cb.lo[t], one.lo[t]); cb.hi[t] = fmaxf(cb.hi[t], one.hi[t]); }
    }
    bool keep = true, inside = false;
    if (!__ballot(bad) && P.nboxes) {
        ClipBox u;
#pragma unroll
        for (int t = 0; t < 3; t++) { u.lo[t] = wave_min_f32(cb.lo[t]); u.hi[t] = wave_max_f32(cb.hi[t]); }
        FrameBuffers fb = {};
        fb.W = P.W; fb.H = P.H; fb.nbx = P.nbx; fb.nby = P.nby; fb.own = P.own;
        keep = clipbox_may_touch_rank(u, fb) || P.own.cull == 3u || P.own.cull == 4u;  // 3, 4: timing ablations (MTR_CULL_DEBUG), keep everything
        inside = (keep && (clipbox_all_in_rank(u, fb) || P.own.cull == 5u)) || P.own.cull == 4u;  // 5: no chunk tests for kept instances
    }
    if (!keep) {
        if (lane == 0) atomicAdd(&P.counters[MTR_CTR(CTR_CULL, inst)], P.nchunks);  // statistics only
        return;
    }
    uint32_t slot = 0;
    if (lane == 0) {
        slot = atomicAdd(P.count, 1u);
        P.list[slot] = inst;
        if (!inside) P.strad[atomicAdd(P.count + 1, 1u)] = slot;
    }
    if (inside) {  // every chunk of it is this rank's: no chunk tests (k_cull_chunks skips the slot)
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        const uint32_t nx = (P.nchunks + 15u) / 16u, tail = P.nchunks & 15u;
        for (uint32_t x = lane; x < nx; x += 64) P.work_mask[(size_t)slot * nx + x] = (uint16_t)((x == nx - 1u && tail) ? (1u << tail) - 1u : 0xFFFFu);
    }
    if (P.comp && !inside) {  // what the chunk tests of this instance read (k_cull_chunks tests the straddlers only)
        CompMat* out = P.comp + (size_t)inst * P.ncomp;
        for (uint32_t j = lane; j < P.ncomp; j += 64) out[j] = make_comp((have_pal && j + 1 < P.ncomp) ? pal + (size_t)j * 16 : nullptr, M);
    }
}

// Chunk culling of a sharded draw.  256 threads = 16 rows of 16 lanes: row = one chunk, lane = one of its boxes; the
// per-joint composites of the instance are built once per workgroup in LDS.  Which of the workgroup's 16 chunks may
// reach a bin of the rank is stored as one 16-bit mask per (instance slot, group of 16 chunks): no list to append to,
// no atomic; k_geom<.., true> launches four workgroups per mask, each taking four of its set bits (one palette in LDS,
// four waves).  A light kernel (no records, no binning state) at full occupancy: the test's chain of dependent loads
// is not paid inside k_geom's 80-register workgroups.
template <bool LDS_COMP>  // true: the workgroup builds its instance's composites in LDS (a single model); false: they come from k_cull_instances
__global__ __launch_bounds__(256, LDS_COMP ? 4 : 8) void k_cull_chunks(ChunkCullParams P) {
    extern __shared__ __align__(16) unsigned char s_raw[];
    CompMat* s_comp = reinterpret_cast<CompMat*>(s_raw);
    __shared__ uint32_t s_wmask[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, row = lane >> 4, sub = lane & 15u;
    const uint32_t c = blockIdx.x * 16u + wave * 4u + row;
    const bool has = c < P.nchunks;
    DChunk ch = {};
    if (has) ch = P.chunks[c];
    const bool have_pal = P.palettes && P.npal;
    const uint32_t ncomp = have_pal ? P.npal + 1u : 1u;
    const bool skinned = (ch.b_flags & 2u) && have_pal;
    // chunks that cannot be bounded are kept
    const bool unbounded = has && (ch.b_count == 0 || (skinned && (ch.b_flags & 1u)) || (skinned && ch.b_count < 2) || ch.b_count > MTR_CHUNK_MAX_BOXES + 1u);
    const uint32_t first = skinned ? ch.b_first + 1u : ch.b_first, n = has ? (skinned ? ch.b_count - 1u : 1u) : 0u;
    BoneBox bx = {};
    const bool tests = has && !unbounded && sub < n;
    if (tests) bx = P.boxes[first + sub];
    const uint32_t nlive = P.strad ? P.inst_count[1] : live_instances(P.inst_count, P.ninst);
    for (uint32_t si = blockIdx.y; si < nlive; si += gridDim.y) {
        const uint32_t ii = P.strad ? P.strad[si] : si;  // the instance's slot: where its masks go
        const uint32_t inst = P.inst_list ? P.inst_list[ii] : ii;
        __syncthreads();  // the composites and masks of the previous instance are no longer read
        if (LDS_COMP) {
            float M[16];
            compose_vp_model(P.vp, P.model_mats, inst, M);
            const float* pal = have_pal ? P.palettes + (size_t)inst * P.pal_stride : nullptr;
            for (uint32_t j = threadIdx.x; j < ncomp; j += 256) s_comp[j] = make_comp((have_pal && j + 1 < ncomp) ? pal + (size_t)j * 16 : nullptr, M);
        }
        __syncthreads();
        ClipBox cb;
#pragma unroll
        for (int t = 0; t < 3; t++) { cb.lo[t] = __builtin_inff(); cb.hi[t] = -__builtin_inff(); }
        bool bad = false;
        if (tests) {
            const uint32_t j = skinned ? min(bx.joint, P.npal - 1u) : ncomp - 1u;  // the last composite is M itself
            cb = box_comp_interval(bx, LDS_COMP ? s_comp[j] : P.comp[(size_t)inst * ncomp + j]);
            bad = !clipbox_finite(cb);
        }
        const uint64_t badm = __ballot(bad);
        const bool row_bad = ((badm >> (row * 16u)) & 0xFFFFull) != 0;
        ClipBox u;
#pragma unroll
        for (int t = 0; t < 3; t++) { u.lo[t] = row_min_f32(cb.lo[t]); u.hi[t] = row_max_f32(cb.hi[t]); }
        const bool keep = has && (unbounded || row_bad || P.keep_all || clipbox_may_touch_rank(u, P.fb));
        const uint64_t km = __ballot(keep);
        if (lane == 0)
            s_wmask[wave] = (uint32_t)(km & 1ull) | (uint32_t)((km >> 15) & 2ull) | (uint32_t)((km >> 30) & 4ull) | (uint32_t)((km >> 45) & 8ull);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t m16 = s_wmask[0] | (s_wmask[1] << 4) | (s_wmask[2] << 8) | (s_wmask[3] << 12);
            P.work_mask[(size_t)ii * gridDim.x + blockIdx.x] = (uint16_t)m16;
            const uint32_t k = (uint32_t)__popc(m16);
            const uint32_t nhave = blockIdx.x * 16u < P.nchunks ? min(16u, P.nchunks - blockIdx.x * 16u) : 0u;
            if (nhave > k) atomicAdd(&P.fb.counters[MTR_CTR(CTR_CULL, blockIdx.x + inst)], nhave - k);  // statistics only
        }
    }
}

// vertex stage alone (unit-parity hook: mtr_model_vertex_stage)
__global__ __launch_bounds__(256) void k_vertex_stage(GeomParams P, uint32_t prim, float* out_clip, float* out_uv) {
    extern __shared__ __align__(16) float s_pal[];
    __shared__ float s_M[16];
    stage_palette(P, 0, s_pal, s_M);
    const DPrim pr = P.prims[prim];
    const bool skinned = pr.skinnable && P.palettes && P.npal;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    // whole waves call the MFMA path; the tail of the last wave is masked
    if ((v & ~63u) >= pr.vertex_num) return;
    const VOut o = shade_vertex_mfma(P.vbuf, pr, v, v < pr.vertex_num, s_M, s_pal, P.npal, skinned, true);
    if (v >= pr.vertex_num) return;
    out_clip[4 * v + 0] = o.x; out_clip[4 * v + 1] = o.y; out_clip[4 * v + 2] = o.z; out_clip[4 * v + 3] = o.w;
    out_uv[2 * v + 0] = o.u; out_uv[2 * v + 1] = o.v;
}

}  // namespace mtr

// MODE by the frame's queue builder, the register budget by the size of the draw
#define MTR_LAUNCH_GEOM_O(C, O)                                                                                         \
    do {                                                                                                                \
        if (p.fb.direct && p.fb.unordered) hipLaunchKernelGGL((mtr::k_geom<2, C, O>), grid, dim3(256), lds, s, p);      \
        else if (p.fb.direct) hipLaunchKernelGGL((mtr::k_geom<1, C, O>), grid, dim3(256), lds, s, p);                   \
        else hipLaunchKernelGGL((mtr::k_geom<0, C, O>), grid, dim3(256), lds, s, p);                                    \
    } while (0)
#define MTR_LAUNCH_GEOM(C) do { if (p.small_draw) MTR_LAUNCH_GEOM_O(C, GEOM_OCC_SMALL); else MTR_LAUNCH_GEOM_O(C, GEOM_OCC); } while (0)
void mtr_launch_geom(const GeomParams& p, hipStream_t s) {
    if (p.nchunks == 0 || p.ninst == 0) return;
    size_t lds = (size_t)p.npal * 64;
    if (p.work_mask) {
        // sharded: four workgroups per group of 16 chunks and instance slot (what k_cull_chunks kept of them is on the device)
        // The instance list's length is only known on the device, and a slot past its end costs ~0.2 ns per workgroup
        // (204 of them per slot on mesh50k: 160 k idle workgroups were 32 us of the 188 us k_geom took on C5 as rank 3 of
        // 8).  So the full-rate launch covers twice the rank's fair share of slots, and a second, small launch -- sixteen
        // workgroups per remaining slot, which walk the instance's masks themselves -- covers the rest: almost always
        // every one of them leaves at once; when a rank does keep more, they are exact and only a little slower.
        uint32_t slots = p.ninst;
        if (p.inst_count && p.fb.own.world > 1) slots = std::min<uint32_t>(p.ninst, (2u * p.ninst + p.fb.own.world - 1) / p.fb.own.world + 64u);  // 64 more cost 3 us
        // ... unless a recent frame of the same batch under the same ownership reported its list length (through its tile kernel, TileParams::hint_out):
        // a dead slot is 204 workgroups = 816 waves that launch, load the length and leave, and the chip launches ~2.6
        // waves per cycle: the 98 k idle workgroups of C5 as rank 0 of 2 cost 62 us of 461, the 30 k of rank 3 of 8 19 us of
        // 127 (kernel trace, keep-everything ablation against the normal run).  An eighth more than reported + 4; whatever
        // a moving camera adds beyond that goes through the second launch, as before.
        if (p.inst_count && (p.slots_hint & 0x80000000u)) {
            const uint32_t h = p.slots_hint & 0x7FFFFFFFu;
            slots = std::min<uint32_t>(p.ninst, h + h / 8u + 4u);
        }
        // the second launch costs ~4 us of stream time even when every one of its workgroups leaves at once: when the slots it
        // would cover are few (idle workgroups at ~0.2 ns each: 20 000 of them = 4 us), the full-rate launch takes them all
        if ((uint64_t)(p.ninst - slots) * p.work_nx * 4u <= 20000u) slots = p.ninst;
        if (p.slots_override) slots = std::min<uint32_t>(p.ninst, p.slots_override);  // MTR_GEOM_SLOTS at device creation: tests force the second launch
        dim3 grid(p.work_nx * 4u * slots);  // the host checked work_nx * 4 * ninst against the launch limit (2^32 threads)
        MTR_LAUNCH_GEOM(true);
        if (slots < p.ninst) {
            GeomParams r = p;
            r.work_slot_base = slots;
            // sized by a recent frame's count the slots beyond are empty but for what a camera move added: a quarter of the workgroups
            r.rest_split = (p.inst_count && (p.slots_hint & 0x80000000u) && !p.slots_override) ? MTR_GEOM_REST_SPLIT / 4u : MTR_GEOM_REST_SPLIT;
            dim3 rest((p.ninst - slots) * std::min<uint32_t>(p.work_nx, r.rest_split));
            if (p.fb.direct && p.fb.unordered) hipLaunchKernelGGL((mtr::k_geom_rest<2>), rest, dim3(256), lds, s, r);
            else if (p.fb.direct) hipLaunchKernelGGL((mtr::k_geom_rest<1>), rest, dim3(256), lds, s, r);
            else hipLaunchKernelGGL((mtr::k_geom_rest<0>), rest, dim3(256), lds, s, r);
        }
        return;
    }
    uint32_t nblk = (p.nchunks + 3) / 4;
    nblk = (nblk + 7) / 8 * 8;  // whole multiple of 8 for the XCD remap
    dim3 grid(nblk, p.ninst);
    MTR_LAUNCH_GEOM(false);
}

void mtr_launch_cull_instances(const CullParams& p, hipStream_t s) {
    if (p.ninst == 0) return;
    hipLaunchKernelGGL(mtr::k_cull_instances, dim3(p.ninst), dim3(64), 0, s, p);
}

void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t s) {
    if (p.nchunks == 0 || p.ninst == 0) return;
    // instance slots: the kernel strides over the (possibly compacted) instance list; twice the rank's fair share
    uint32_t ny = p.ninst;
    if (p.inst_count && p.fb.own.world > 1) ny = std::max<uint32_t>(1u, std::min<uint32_t>(p.ninst, (2u * p.ninst + p.fb.own.world - 1) / p.fb.own.world));
    // the straddlers a recent frame of the batch reported (the kernel strides over the list: any ny is correct)
    if (p.strad && (p.strad_hint & 0x80000000u)) ny = std::max<uint32_t>(1u, std::min<uint32_t>(p.ninst, (p.strad_hint & 0x7FFFFFFFu) + (p.strad_hint & 0x7FFFFFFFu) / 8u + 2u));
    ny = std::min<uint32_t>(ny, 65535u);
    const uint32_t ncomp = (p.palettes && p.npal) ? p.npal + 1u : 1u;
    if (p.comp) hipLaunchKernelGGL(mtr::k_cull_chunks<false>, dim3((p.nchunks + 15) / 16, ny), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(mtr::k_cull_chunks<true>, dim3((p.nchunks + 15) / 16, ny), dim3(256), (size_t)ncomp * sizeof(CompMat), s, p);
}

void mtr_launch_vertex_stage(const GeomParams& p, uint32_t prim, float* out_clip, float* out_uv, hipStream_t s) {
    // vertex_num is 16 bits (src/rmodel.rs:218-220): at most 256 blocks
    hipLaunchKernelGGL(mtr::k_vertex_stage, dim3(256), dim3(256), (size_t)p.npal * 64, s, p, prim, out_clip, out_uv);
}
