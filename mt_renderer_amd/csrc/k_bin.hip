// k_bin.hip -- per-bin queue construction: exclusive scan of the (entries, segments) counts that
// k_geom accumulated, then an ordered fill.  A bin's queue is a set of SEGMENTS: each segment is
// the run of records one geometry wave contributed to that bin, written in submission order and
// tagged with the wave's global chunk id, so the tile kernel restores full submission order by
// sorting a few segment descriptors instead of every triangle reference.
#include "geom_common.h"

namespace mtr {

// single workgroup: nbins is a few thousand 16x16 bins (1080p: 8160, 4K: 32400)
__global__ __launch_bounds__(1024) void k_scan(FrameBuffers fb) {
    __shared__ uint32_t s_we[16], s_ws[16];
    __shared__ uint32_t s_carry_e, s_carry_s;
    const uint32_t nbins = fb.nbx * fb.nby;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_carry_e = 0; s_carry_s = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < nbins; base += 1024) {
        const uint32_t b = base + threadIdx.x;
        unsigned long long cnt = b < nbins ? fb.bin_count[b] : 0ull;
        uint32_t e = (uint32_t)cnt, s = (uint32_t)(cnt >> 32);
        // inclusive wave scan
        uint32_t ie = e, is = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t te = __shfl_up(ie, d), ts = __shfl_up(is, d);
            if ((int)lane >= d) { ie += te; is += ts; }
        }
        if (lane == 63) { s_we[wave] = ie; s_ws[wave] = is; }
        __syncthreads();
        uint32_t pe = s_carry_e, ps = s_carry_s;
        for (uint32_t w = 0; w < wave; w++) { pe += s_we[w]; ps += s_ws[w]; }
        if (b < nbins) {
            fb.bin_start[b] = pe + ie - e;
            fb.seg_start[b] = ps + is - s;
            fb.bin_fill[b] = 0ull;
        }
        __syncthreads();
        if (threadIdx.x == 1023) { s_carry_e = pe + ie; s_carry_s = ps + is; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        fb.bin_start[nbins] = s_carry_e;
        fb.seg_start[nbins] = s_carry_s;
        fb.counters[CTR_ENTRIES] = s_carry_e;
        fb.counters[CTR_SEGS] = s_carry_s;
        if (s_carry_e > fb.entry_cap || s_carry_s > fb.seg_cap) atomicOr(&fb.counters[CTR_OVERFLOW], 2u);
    }
}

// one wave per geometry chunk, replaying k_geom's bin grouping over the chunk's record headers
__global__ __launch_bounds__(256) void k_fill(FrameBuffers fb, uint32_t total_chunks) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t gid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gid >= total_chunks) return;
    if (fb.counters[CTR_OVERFLOW]) return;  // the host re-runs the frame with larger queues
    const ChunkInfo ci = fb.chunk_info[gid];
    for (uint32_t round = 0; round * 64 < ci.n; ++round) {
        const uint32_t j = round * 64 + lane;
        bool act = j < ci.n;
        RecHdr h = {0, 0, 0, 0};
        if (act) h = fb.rec_hdr[ci.base + j];
        act = act && h.bx0 <= h.bx1;  // a hole left by a guard-clipped fan (k_geom.hip)
        emit_bins<false>(fb, h, act, gid, round, lane);
    }
}

}  // namespace mtr

void mtr_launch_scan(const FrameBuffers& fb, hipStream_t s) {
    hipLaunchKernelGGL(mtr::k_scan, dim3(1), dim3(1024), 0, s, fb);
}

void mtr_launch_fill(const FrameBuffers& fb, uint32_t total_chunks, hipStream_t s) {
    if (total_chunks == 0) return;
    hipLaunchKernelGGL(mtr::k_fill, dim3((total_chunks + 3) / 4), dim3(256), 0, s, fb, total_chunks);
}
