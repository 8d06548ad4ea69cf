// k_shard.hip -- screen-space sharding helpers for the multi-GPU path: bins are dealt
// round-robin over the ranks (bin % world == rank).  pack: this rank's bins -> one contiguous,
// bin-major RGBA8 block (the all-gather send buffer); unpack: the gathered [rank][k][16][16] blocks
// -> the linear framebuffer.  Plain copies, one thread per pixel, 4 B/lane (64 B rows per bin row).
#include "mtr_internal.h"

namespace mtr {

__global__ __launch_bounds__(256) void k_pack_shard(const uint32_t* color, uint32_t* dst, uint32_t W, uint32_t H, uint32_t nbx,
                                                    uint32_t nbins, uint32_t rank, uint32_t world, uint32_t shard_bins) {
    const uint32_t k = blockIdx.x;  // k-th bin of this rank
    const uint32_t bin = k * world + rank;
    const uint32_t lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    uint32_t v = 0;
    if (bin < nbins) {
        const uint32_t x = (bin % nbx) * MTR_BIN + lx, y = (bin / nbx) * MTR_BIN + ly;
        if (x < W && y < H) v = color[(size_t)y * W + x];
    }
    if (k < shard_bins) dst[(size_t)k * 256 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void k_unpack_shards(const uint32_t* gathered, uint32_t* color, uint32_t W, uint32_t H,
                                                       uint32_t nbx, uint32_t nbins, uint32_t world, uint32_t shard_bins) {
    const uint32_t bin = blockIdx.x;
    if (bin >= nbins) return;
    const uint32_t rank = bin % world, k = bin / world;
    const uint32_t lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const uint32_t x = (bin % nbx) * MTR_BIN + lx, y = (bin / nbx) * MTR_BIN + ly;
    if (x < W && y < H) color[(size_t)y * W + x] = gathered[((size_t)rank * shard_bins + k) * 256 + threadIdx.x];
}

// W % 4 == 0: 16 bytes per thread (four pixels of a bin row), four bins per 256-thread block.  The scalar kernels
// above moved 4 bytes per thread and cost ~15 us each per 1080p frame; they remain for odd widths.
__global__ __launch_bounds__(256) void k_pack_shard_v4(const uint4* color, uint4* dst, uint32_t W4, uint32_t H, uint32_t nbx,
                                                       uint32_t nbins, uint32_t rank, uint32_t world, uint32_t shard_bins) {
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);  // k-th bin of this rank
    if (k >= shard_bins) return;
    const uint32_t bin = k * world + rank, t = threadIdx.x & 63;
    const uint32_t lx4 = t & 3, ly = t >> 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (bin < nbins) {
        const uint32_t x4 = (bin % nbx) * (MTR_BIN / 4) + lx4, y = (bin / nbx) * MTR_BIN + ly;
        if (x4 < W4 && y < H) v = color[(size_t)y * W4 + x4];
    }
    dst[(size_t)k * 64 + t] = v;
}

__global__ __launch_bounds__(256) void k_unpack_shards_v4(const uint4* gathered, uint4* color, uint32_t W4, uint32_t H,
                                                          uint32_t nbx, uint32_t nbins, uint32_t world, uint32_t shard_bins) {
    const uint32_t bin = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bin >= nbins) return;
    const uint32_t rank = bin % world, k = bin / world, t = threadIdx.x & 63;
    const uint32_t x4 = (bin % nbx) * (MTR_BIN / 4) + (t & 3), y = (bin / nbx) * MTR_BIN + (t >> 2);
    if (x4 < W4 && y < H) color[(size_t)y * W4 + x4] = gathered[((size_t)rank * shard_bins + k) * 64 + t];
}

}  // namespace mtr

void mtr_launch_pack_shard(const uint8_t* color, uint8_t* dst, uint32_t W, uint32_t H, uint32_t rank, uint32_t world,
                           hipStream_t s) {
    const uint32_t nbx = (W + MTR_BIN - 1) / MTR_BIN, nby = (H + MTR_BIN - 1) / MTR_BIN, nbins = nbx * nby;
    const uint32_t shard_bins = (nbins + world - 1) / world;
    if (W % 4 == 0 && ((uintptr_t)color | (uintptr_t)dst) % 16 == 0) {
        hipLaunchKernelGGL(mtr::k_pack_shard_v4, dim3((shard_bins + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(color),
                           reinterpret_cast<uint4*>(dst), W / 4, H, nbx, nbins, rank, world, shard_bins);
        return;
    }
    hipLaunchKernelGGL(mtr::k_pack_shard, dim3(shard_bins), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(color),
                       reinterpret_cast<uint32_t*>(dst), W, H, nbx, nbins, rank, world, shard_bins);
}

void mtr_launch_unpack_shards(const uint8_t* gathered, uint8_t* color, uint32_t W, uint32_t H, uint32_t world, hipStream_t s) {
    const uint32_t nbx = (W + MTR_BIN - 1) / MTR_BIN, nby = (H + MTR_BIN - 1) / MTR_BIN, nbins = nbx * nby;
    const uint32_t shard_bins = (nbins + world - 1) / world;
    if (W % 4 == 0 && ((uintptr_t)gathered | (uintptr_t)color) % 16 == 0) {
        hipLaunchKernelGGL(mtr::k_unpack_shards_v4, dim3((nbins + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(gathered),
                           reinterpret_cast<uint4*>(color), W / 4, H, nbx, nbins, world, shard_bins);
        return;
    }
    hipLaunchKernelGGL(mtr::k_unpack_shards, dim3(nbins), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(gathered),
                       reinterpret_cast<uint32_t*>(color), W, H, nbx, nbins, world, shard_bins);
}
