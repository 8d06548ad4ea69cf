// mtr_internal.h -- device-visible data layout of the rModel draw path (gfx950).
//
// HBM layout (DESIGN.md "Data layout"):
//   model   : raw vertex bytes, u16 indices, DPrim[nprims], DChunk[nchunks]   (resident, static)
//   frame   : RecHdr/RecA/RecB[record capacity]  one record per surviving screen triangle, written
//             as one contiguous, submission-ordered run per geometry wave ("chunk run");
//             ChunkInfo[total chunks] = {run base, run length};
//             bin_count/bin_fill[nbins] (u64: lo = entries, hi = segments), bin_start/seg_start;
//             entries[] (record ids, grouped by bin, ordered inside a segment),
//             segs[]    ({order key, offset, count} runs, grouped by bin, sorted by the tile kernel)
//             colour RGBA8 [H][W], depth f32 [H][W].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MTR_BIN 16             // screen bin edge in pixels (one wave of the tile kernel)
#define MTR_BIN_SHIFT 4
#define MTR_SUB 8              // sub-tile edge: one wave64 = one 8x8 pixel block
#define MTR_CHUNK_NEW 62       // strip positions that complete a triangle per geometry wave
#define MTR_CHUNK_SLOTS 124    // worst case records per chunk (near clip: 2 per position)
#define MTR_GUARD_BAND 1048576.0f

enum { MTR_SH_DEBUG = 0, MTR_SH_TEXTURED = 1, MTR_SH_CONST = 2 };

struct DPrim {
    uint32_t vertex_base, vertex_num, stride;
    uint32_t index_ofs, index_num, index_base;
    uint32_t topology;  // 4 strip, 3 list
    uint32_t pos_fmt, pos_cnt, pos_off;
    uint32_t uv_fmt, uv_cnt, uv_off, has_uv;
    uint32_t joint_off, weight_off, skinnable;
    uint32_t aligned4;  // vertex_base, stride and every bound offset are multiples of 4
    uint32_t parts_no;
    uint32_t cull;      // MTR_CULL_*: 0 back (the reference, src/model.rs:252), 1 none, 2 front (material state, row f-4)
};

struct DChunk {
    uint32_t prim;      // primitive index
    uint32_t start;     // first strip position whose triangle this chunk owns
    uint32_t q_before;  // strip: consecutive non-restart indices right before position start-2
    uint32_t ntris;     // input triangles completed inside this chunk (statistics)
    uint32_t b_first;   // object-space bounds of the chunk's vertices: bounds[b_first] is the box of all of them (what an
    uint32_t b_count;   // unskinned draw transforms), bounds[b_first + 1 ..) one box per joint that carries weight; 0: none
    uint32_t b_flags;   // bit0: the skinned position is not a convex combination (weights do not sum to 255, or too many
    uint32_t pad;       //       joints for the table): never culled when skinned; bit1: the primitive can be skinned
};

// object-space box of the vertices one joint influences inside one chunk (or of a whole model: instance culling)
struct BoneBox {
    float cx, cy, cz;   // centre
    float ex, ey, ez;   // half extents, rounded up
    uint32_t joint;     // raw joint index (clamped to the palette like the vertex shader does), 0xFFFFFFFF: unskinned
    uint32_t pad;
};
#define MTR_BOX_UNSKINNED 0xFFFFFFFFu
#define MTR_CULL_CTR_WORDS 32    // per draw (one 128-byte line): word 0 = instance-list length
#define MTR_CHUNK_MAX_BOXES 15  // + the whole-chunk box = one row of 16 lanes

// What the chunk test needs of one joint, per frame and instance: the clip rows x, y, w of C = M * [P_j; 0 0 0 1] and of
// |M| * |P_j| (column-major 4 x 3: element [c * 3 + t], t = x, y, w), written by k_cull_prepare.  Entry npal of an
// instance is M itself (what unskinned geometry is transformed by).
struct CompMat {
    float C[12];
    float A[12];
};

struct DMat {            // one per (draw, [instance,] primitive)
    uint32_t shader;     // MTR_SH_*
    uint32_t rgba8;      // SH_DEBUG / SH_CONST: the quantised source colour
    uint32_t blend;      // MTR_DB_*: alpha blending (src/model.rs:243-246), off (debug overlay), additive (material state)
    uint32_t tw, th;     // texture size
    uint32_t translucent;  // order-dependent: the pixel depends on the order of the fragments (a blend that is not a
                           // replace, depth write or depth test off): the ordered tile kernel's
    const uint8_t* tex;  // decoded RGBA8 texels (or, MTR_TR_BC*: the BC blocks), mip levels one after the other
    uint32_t tlevels;    // mip levels present (>= 1) | MTR_TR_* << 8
    uint32_t dstate;     // bit0 depth write, bit1 depth test
};
enum { MTR_TR_RGBA8 = 0, MTR_TR_BC1 = 1, MTR_TR_BC7 = 2 };    // what DMat::tex holds (mtr_device_set_texture_residency)
enum { MTR_DB_OFF = 0, MTR_DB_ALPHA = 1, MTR_DB_ADD = 2 };  // DMat::blend (0 / 1 as before the material states)

struct RecHdr {          // 8 B: bins covered, inclusive
    uint16_t bx0, by0, bx1, by1;
};
struct RecA {            // what coverage + depth need, in registers (see RecP for the 32 bytes in memory)
    int32_t X0, Y0, X1, Y1;
    int32_t X2, Y2;
    float z0, z1;
    float z2;
    uint32_t mat;
    uint32_t pad0, pad1;  // pad0 = the colour of a solid triangle, pad1 = solid | blending << 8 | order-dependent << 16 |
                          // hard << 17 (order-dependent beyond what prefix minima of z resolve: additive blend, depth write / test off)
};
// RecA as it lives in HBM: 32 B.  A surviving triangle is written once by k_geom and read once per bin it touches by
// the tile kernel (692 k reads per headline frame), so the record is what both kernels' traffic is made of.
//   q0 = { X0, Y0, dX1 | dY1 << 16, dX2 | dY2 << 16 }   vertex 1 / 2 relative to vertex 0, 16 bits each (24.8 fixed
//        point: up to 128 px); a triangle with a longer edge stores dX1 = -32768 and its four coordinates in rec_l[]
//   q1 = { z0, z1, z2, payload }
//   payload: top byte 0xFF = the quantised source colour of a debug-id / overlay triangle in the default depth state
//            (their alpha is 1: whatever the blend, the fragment replaces the pixel); otherwise a triangle whose
//            shading needs its material: material id (24 bits) | blend != off << 24 | order-dependent << 25 | hard << 26
struct RecP {
    uint4 q0, q1;
};
#define MTR_REC_LARGE_SENTINEL 0x8000u
#define MTR_MAX_TEXTURED_MATERIALS (1u << 24)

__device__ __forceinline__ bool rec_is_large(const RecA& a) {
    const int32_t d1x = a.X1 - a.X0, d1y = a.Y1 - a.Y0, d2x = a.X2 - a.X0, d2y = a.Y2 - a.Y0;
    return d1x < -32767 || d1x > 32767 || d1y < -32767 || d1y > 32767 || d2x < -32767 || d2x > 32767 || d2y < -32767 || d2y > 32767;
}
__device__ __forceinline__ RecP rec_pack(const RecA& a, bool large) {
    RecP p;
    const uint32_t solid = a.pad1 & 1u, blend = (a.pad1 >> 8) & 1u, transl = (a.pad1 >> 16) & 1u, hard = (a.pad1 >> 17) & 1u;
    const uint32_t payload = solid ? (a.pad0 | 0xFF000000u) : ((a.mat & 0xFFFFFFu) | (blend << 24) | (transl << 25) | (hard << 26));
    const uint32_t d1 = large ? MTR_REC_LARGE_SENTINEL : (((uint32_t)(a.X1 - a.X0) & 0xFFFFu) | ((uint32_t)(a.Y1 - a.Y0) << 16));
    const uint32_t d2 = large ? 0u : (((uint32_t)(a.X2 - a.X0) & 0xFFFFu) | ((uint32_t)(a.Y2 - a.Y0) << 16));
    p.q0 = make_uint4((uint32_t)a.X0, (uint32_t)a.Y0, d1, d2);
    p.q1 = make_uint4(__float_as_uint(a.z0), __float_as_uint(a.z1), __float_as_uint(a.z2), payload);
    return p;
}
// `l`: the record's entry of rec_l[] (read by the caller only when (q0.z & 0xFFFF) == MTR_REC_LARGE_SENTINEL)
__device__ __forceinline__ RecA rec_unpack(const RecP& p, const int4& l) {
    RecA a;
    a.X0 = (int32_t)p.q0.x; a.Y0 = (int32_t)p.q0.y;
    if ((p.q0.z & 0xFFFFu) == MTR_REC_LARGE_SENTINEL) {
        a.X1 = l.x; a.Y1 = l.y; a.X2 = l.z; a.Y2 = l.w;
    } else {
        a.X1 = a.X0 + (int32_t)(int16_t)(p.q0.z & 0xFFFFu); a.Y1 = a.Y0 + ((int32_t)p.q0.z >> 16);
        a.X2 = a.X0 + (int32_t)(int16_t)(p.q0.w & 0xFFFFu); a.Y2 = a.Y0 + ((int32_t)p.q0.w >> 16);
    }
    a.z0 = __uint_as_float(p.q1.x); a.z1 = __uint_as_float(p.q1.y); a.z2 = __uint_as_float(p.q1.z);
    const uint32_t pl = p.q1.w;
    // pad1: bit0 solid (pad0 is the colour, no material needed), bit8 blending on, bit16 order-dependent, bit17 hard
    if ((pl >> 24) == 0xFFu) { a.mat = 0; a.pad0 = pl; a.pad1 = 1u; }
    else { a.mat = pl & 0xFFFFFFu; a.pad0 = 0; a.pad1 = (((pl >> 24) & 1u) << 8) | (((pl >> 25) & 1u) << 16) | (((pl >> 26) & 1u) << 17); }
    return a;
}

struct RecB {            // 48 B: perspective-correct texcoords (textured primitives only)
    float iw0, iw1, iw2, up0;
    float up1, up2, vp0, vp1;
    float vp2, pad0, pad1, pad2;
};
struct ChunkInfo {
    uint32_t base, n;
};
struct Seg {             // 16 B
    uint32_t key;        // global chunk id * 2 + round: submission order of the run
    uint32_t off;        // offset inside the bin's entry range
    uint32_t cnt;
    uint32_t pad;
};

// Ownership of the 16x16 bins of a sharded frame (multi-GPU, one rank per GPU).  The host picks the map per frame:
//   INTERLEAVED  bin b (row-major) belongs to rank b % world: the finest balance, but every object touches every rank;
//   BANDS        rank r owns the bin rows [band[r], band[r+1]): an object touches the few ranks whose band it crosses;
//   SUPERTILES   squares of (1 << st_shift)^2 bins dealt round-robin in row-major order.
// own_list: this rank's bins in the order the tile kernels take them (nullptr when the frame is not sharded).
#ifndef MTR_H
enum { MTR_OWN_INTERLEAVED = 0, MTR_OWN_BANDS = 1, MTR_OWN_SUPERTILES = 2 };  // include/mtr.h has the same
#endif
struct Ownership {
    uint32_t map, rank, world;
    uint32_t y0, y1;          // BANDS: this rank's bin rows [y0, y1)
    uint32_t st_shift, nsx;   // SUPERTILES: log2 edge in bins, super-tiles per row
    uint32_t own_count;       // bins of this rank (not sharded: every bin)
    const uint32_t* own_list;
    uint32_t cull;            // geometry waves test their chunk's bounds against the map before any vertex work
    uint32_t pad;
};

__device__ __forceinline__ bool bin_owned(const Ownership& o, uint32_t bx, uint32_t by, uint32_t nbx) {
    if (o.world <= 1) return true;
    if (o.map == MTR_OWN_BANDS) return by >= o.y0 && by < o.y1;
    if (o.map == MTR_OWN_SUPERTILES) return ((by >> o.st_shift) * o.nsx + (bx >> o.st_shift)) % o.world == o.rank;
    return (by * nbx + bx) % o.world == o.rank;
}

// true only if EVERY bin of the inclusive rectangle belongs to the rank (bands; the other maps never say so)
__device__ __forceinline__ bool rect_owned_all(const Ownership& o, uint32_t by0, uint32_t by1) {
    if (o.world <= 1) return true;
    return o.map == MTR_OWN_BANDS && by0 >= o.y0 && by1 < o.y1;
}

// true if some bin of the inclusive rectangle belongs to the rank; may say true for a rectangle that holds none (the
// binner filters bin by bin), never false for one that does
__device__ __forceinline__ bool rect_owned_any(const Ownership& o, uint32_t bx0, uint32_t by0, uint32_t bx1, uint32_t by1, uint32_t nbx) {
    if (o.world <= 1) return true;
    if (o.map == MTR_OWN_BANDS) return by1 >= o.y0 && by0 < o.y1 && o.y0 < o.y1;
    uint32_t x0 = bx0, x1 = bx1, y0 = by0, y1 = by1, pitch = nbx;
    if (o.map == MTR_OWN_SUPERTILES) { x0 >>= o.st_shift; x1 >>= o.st_shift; y0 >>= o.st_shift; y1 >>= o.st_shift; pitch = o.nsx; }
    const uint32_t w = x1 - x0;           // width - 1
    if (w + 1 >= o.world) return true;    // `world` consecutive ids hit every residue
    if (y1 - y0 >= 8) return true;        // tall and narrow: rare, keep
    for (uint32_t y = y0; y <= y1; y++) {
        const uint32_t first = (y * pitch + x0) % o.world;
        const uint32_t d = o.rank >= first ? o.rank - first : o.rank + o.world - first;
        if (d <= w) return true;
    }
    return false;
}

#ifdef __HIPCC__  // device only (the host sanitizer builds of tests/cpp include this header with g++)
// wave-wide inclusive prefix sum: Hillis-Steele inside each row of 16 lanes with DPP row_shr (lanes shifted in from
// outside the row read 0), then the totals of the rows below are added (three v_readlane).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t row = (threadIdx.x & 63u) >> 4;
    return v + (row == 0 ? 0u : row == 1 ? t0 : row == 2 ? t0 + t1 : t0 + t1 + t2);
}
#endif

// counters[]: [1] entries, [2] segments (two-pass scan), [3] overflow flags, then CTR_NSHARDS statistics shards of one
// 128-byte line each: {surviving triangles, (triangle, bin) pairs, segments}.  One line per shard: atomics that share
// a line serialise in its L2 channel (64 shards packed into two lines cost k_geom 20 us on the headline scene).
// CTR_OVERFLOW bits: 1 a chunk needed more than MTR_CHUNK_SLOTS records (guard-band clipping fanning many triangles of
// one chunk), 2 two-pass queue capacity, 4 direct-mode per-bin queue full
enum { CTR_ENTRIES = 1, CTR_SEGS = 2, CTR_OVERFLOW = 3, CTR_SHARD_BASE = 32, CTR_SHARD_STRIDE = 32, CTR_NSHARDS = 128,
       CTR_REC = 0, CTR_ENT = 1, CTR_SEG = 2, CTR_CULL = 3 /* geometry chunks culled against the ownership map */,
       CTR_NUM = CTR_SHARD_BASE + CTR_NSHARDS * CTR_SHARD_STRIDE };
#define MTR_CTR(kind, k) (CTR_SHARD_BASE + ((k) & (CTR_NSHARDS - 1)) * CTR_SHARD_STRIDE + (kind))

struct FrameBuffers {
    RecHdr* rec_hdr;
    RecP* rec_a;      // packed records
    int4* rec_l;      // X1, Y1, X2, Y2 of the records whose edges do not fit 16-bit deltas (same index)
    RecB* rec_b;
    ChunkInfo* chunk_info;
    unsigned long long* bin_count;  // lo32 entries, hi32 segments
    unsigned long long* bin_fill;
    uint32_t* bin_start;            // nbins + 1
    uint32_t* seg_start;            // nbins + 1
    uint32_t* entries;              // submission order of each (triangle, bin) pair: chunk*128 + index in the chunk's
                                    // run; its record id is (ord >> 7) * MTR_CHUNK_SLOTS + (ord & 127)
    Seg* segs;
    uint32_t* counters;             // CTR_*
    uint32_t rec_cap, entry_cap, seg_cap;
    uint32_t W, H, nbx, nby;
    Ownership own;
    // direct mode: single-pass binning into bounded per-bin queues (bin b owns entries[b*qcap ..) and
    // segs[b*scap ..), filled through bin_fill); overflow raises CTR_OVERFLOW bit 2 and the host re-runs the frame
    // with the exact two-pass (count, scan, fill) queues
    uint32_t direct, qcap, scap;
    uint32_t unordered;             // direct mode only: the frame goes to the visibility-key kernel, queue order is free
};

// direct mode: the bin's workgroup is the only consumer of bin_fill[bin]; one thread parks the count in bin_count
// (statistics hook) and zeroes the fill word, so the next frame needs no memset.  Call after every wave of the
// workgroup has read its queue bounds.
__device__ __forceinline__ void bin_queue_done(const FrameBuffers& fb, uint32_t bin) {
    if (fb.direct) {
        fb.bin_count[bin] = fb.bin_fill[bin];
        fb.bin_fill[bin] = 0ull;
    }
}

// where bin b's queue lives, for both queue layouts
__device__ __forceinline__ void bin_queue(const FrameBuffers& fb, uint32_t bin, uint32_t& ent_lo, uint32_t& n_ent,
                                          uint32_t& seg_lo, uint32_t& n_seg) {
    if (fb.direct) {
        // every lane of the bin's workgroup reads the same word; the caller cleans it afterwards (bin_queue_done)
        const unsigned long long f = fb.bin_fill[bin];
        ent_lo = bin * fb.qcap; seg_lo = bin * fb.scap;
        n_ent = min((uint32_t)f, fb.qcap); n_seg = min((uint32_t)(f >> 32), fb.scap);
    } else {
        ent_lo = fb.bin_start[bin]; n_ent = fb.bin_start[bin + 1] - ent_lo;
        seg_lo = fb.seg_start[bin]; n_seg = fb.seg_start[bin + 1] - seg_lo;
    }
}

struct GeomParams {
    const uint8_t* vbuf;
    const uint16_t* ibuf;
    const DPrim* prims;
    const DChunk* chunks;
    const BoneBox* boxes;     // chunk bounds (DChunk::b_first indexes it), nullptr: no culling data
    uint32_t nchunks;
    uint32_t ninst;           // instances of the draw
    // sharded draws (k_geom<MODE, true>): what k_cull_chunks decided -- for instance slot ii (the ii-th entry of the
    // compacted instance list, or instance ii itself when there is none) and group x of 16 consecutive chunks, bit b of
    // work_mask[ii * work_nx + x] says chunk 16 x + b may reach a bin of this rank.  No list, no counter: a cull
    // workgroup stores its mask, geometry workgroup (x, quarter q, ii) takes the q-th four set bits.  (Round 2 first
    // appended the survivors to 64 sub-lists: one returning atomic per cull workgroup, one more dependent load in front
    // of every geometry workgroup.)
    const uint16_t* work_mask;   // 4-byte aligned, an even number of masks allocated
    uint32_t work_nx;          // groups of 16 chunks per instance = ceil(nchunks / 16)
    uint32_t work_slot_base;   // k_geom_rest: the first instance slot of its launch
    uint32_t rest_split;       // k_geom_rest: workgroups per instance slot (each walks every rest_split-th mask of the instance)
    uint32_t slots_override;   // 0, or the number of instance slots the full-rate launch covers (MTR_GEOM_SLOTS, tests)
    uint32_t small_draw;       // the draw does not fill the GPU: the build of k_geom that leaves a wave slot per SIMD free
    const uint32_t* inst_list;   // from k_cull_instances, or nullptr: instance slot ii is instance ii
    const uint32_t* inst_count;  // length of inst_list (device), or nullptr: ninst
    uint32_t slots_hint;         // 0, or 0x80000000 | the length a recent frame of this batch reported (launch sizing only)
    const float* model_mats;  // ninst*16 or nullptr
    const float* palettes;    // per instance npal*16 floats (stride pal_stride floats) or nullptr
    uint32_t npal, pal_stride;
    float vp[16];
    uint32_t chunk_base;      // global chunk id of (instance 0, chunk 0)
    uint32_t mat_base, mat_inst_stride;
    const DMat* mats;         // frame material table (read: is this primitive textured?)
    FrameBuffers fb;
};

struct TileParams {
    FrameBuffers fb;
    const DMat* mats;
    uint8_t* color;  // RGBA8
    float* depth;
    uint32_t clear_rgba8;
    float clear_depth;
    // mixed frames (some material translucent): k_tile_vis renders the bins it can -- every triangle opaque, or translucent
    // only through alpha blending in the default depth state (prefix minima of z, k_tile_vis.hip) -- and flags the others
    // in bin_flag[]; k_tile then renders the flagged bins in submission order
    uint8_t* bin_flag;
    uint32_t mixed;
    // the counter block the NEXT frame on these framebuffers will use (CTR_NUM words), zeroed by this frame's tile
    // kernel so that no fill has to be launched per frame; nullptr: nothing to zero
    uint32_t* zero_next;
    // one word of pinned host memory per frame in flight: the tile kernel publishes 0x80000000 | CTR_OVERFLOW there as
    // soon as it starts (the geometry / scan kernels that raise the flags have completed by then), so the host -- the
    // render thread, or the exchange thread before it packs -- learns about a dropped triangle without a read-back
    uint32_t* host_status;
    uint32_t vis_waves;  // waves per bin of the visibility kernel (2, 4, 8), 0: the launcher's choice (tuning hook: MTR_VIS_WAVES)
    uint32_t xcd_run;    // bins per run dealt to the XCDs in turn, 0: one contiguous eighth of the bins per XCD (tile_common.h)
    uint32_t quad_walk;  // visibility kernel: walk bboxes in 2 x 2 quads where a pass's boxes are large enough (a frame alone on the GPU)
    // the slot's culling counters (instance-list and work-list lengths), zeroed here for the slot's next frame
    uint32_t* zero_words;
    uint32_t zero_nwords;
    // feedback for the NEXT frames' launch sizes (k_geom.hip: mtr_launch_geom): before culling counter word hint_word[k] + j
    // (j = 0: instance-list length of a batch draw, 1: its straddlers) is zeroed, 0x80000000 | its value is stored to
    // hint_out[2 * hint_slot[k] + j] -- pinned host memory, like host_status.  (Stored from k_geom itself, ahead of its
    // scalar loads, the store made the compiler turn every one of them into a vector load: 461 -> 686 us on C5 as rank 0 of 2.)
    uint32_t* hint_out;
    uint32_t nhint;
    uint16_t hint_word[4], hint_slot[4];
};

// launchers (defined in the .hip files, called from mtr_api.cpp)
void mtr_launch_geom(const GeomParams& p, hipStream_t s);
void mtr_launch_scan(const FrameBuffers& fb, hipStream_t s);
void mtr_launch_fill(const FrameBuffers& fb, uint32_t total_chunks, hipStream_t s);
void mtr_launch_tile(const TileParams& p, bool textured, hipStream_t s);
void mtr_launch_tile_vis(const TileParams& p, bool textured, hipStream_t s);
void mtr_launch_alpha_min(const uint8_t* rgba, size_t npixels, uint32_t* out_min, hipStream_t s);
void mtr_launch_vertex_stage(const GeomParams& p, uint32_t prim, float* out_clip, float* out_uv, hipStream_t s);
void mtr_launch_bc1_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t s);
void mtr_launch_bc7_decode(const uint8_t* blocks, uint8_t* rgba, uint32_t w, uint32_t h, hipStream_t s);
// own_list / own_count: the packing rank's bins; stride_bins: bins per rank in the gathered buffer (the largest share);
// src_of_bin[b] = rank * stride_bins + k for the k-th bin of its owner
void mtr_launch_pack_shard(const uint8_t* color, uint8_t* dst, uint32_t W, uint32_t H, const uint32_t* own_list, uint32_t own_count,
                           uint32_t stride_bins, hipStream_t s);
void mtr_launch_unpack_shards(const uint8_t* gathered, uint8_t* color, uint32_t W, uint32_t H, const uint32_t* src_of_bin, hipStream_t s);
// instance culling of a sharded batch draw: the instances whose bounds may touch a bin of the rank -> list / count
struct CullParams {
    const BoneBox* boxes;      // per-joint boxes of the whole model, boxes[0] = the box of every vertex
    uint32_t nboxes;
    uint32_t ninst;
    const float* model_mats;
    const float* palettes;
    uint32_t npal, pal_stride;
    float vp[16];
    uint32_t W, H, nbx, nby;
    Ownership own;
    uint32_t* list;
    uint32_t* count;
    CompMat* comp;             // out, for the surviving instances: their per-joint composites (ncomp each, indexed by instance)
    uint32_t ncomp;            // npal + 1 (skinned) or 1
    // An instance whose whole screen rectangle lies in bins of this rank needs no chunk tests: its masks (GeomParams::
    // work_mask, indexed by its slot in `list`) are written here, all ones.  The others -- the instances that straddle
    // the rank's border -- are appended (by slot) to `strad`, length count[1], for k_cull_chunks.
    uint16_t* work_mask;
    uint32_t* strad;
    uint32_t nchunks;
    uint32_t* counters;        // the frame's counter block: the chunks of a culled instance are added to the CTR_CULL statistic
};
void mtr_launch_cull_instances(const CullParams& p, hipStream_t s);
// chunk culling of a sharded draw: every chunk of every (surviving) instance is bounded against the rank's bins; the
// chunks that may reach one are appended to the work list k_geom consumes
struct ChunkCullParams {
    const DChunk* chunks;
    const BoneBox* boxes;
    uint32_t nchunks, ninst;
    const uint32_t* inst_list;   // from k_cull_instances, or nullptr: instances 0 .. ninst-1
    const uint32_t* inst_count;
    const uint32_t* strad;       // slots (indices into inst_list) of the instances to test, length inst_count[1]; nullptr: every slot
    uint32_t strad_hint;         // 0, or 0x80000000 | the straddler count a recent frame of this batch reported (launch sizing only)
    const float* model_mats;
    const float* palettes;
    uint32_t npal, pal_stride;
    float vp[16];
    FrameBuffers fb;             // W, H, nbx, nby, own, counters
    const CompMat* comp;         // per-joint composites of every surviving instance (k_cull_instances), or nullptr: the
                                 // workgroup builds its instance's composites in LDS (a single model)
    uint16_t* work_mask;         // see GeomParams
    uint32_t keep_all;           // timing ablation (MTR_CULL_DEBUG=3): run the tests, keep everything
};
void mtr_launch_cull_chunks(const ChunkCullParams& p, hipStream_t s);
