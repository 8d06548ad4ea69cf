// k_shard.hip -- screen-space sharding helpers of the multi-GPU path.  Which rank owns which 16x16 bin is
// the host's choice per frame (mtr_internal.h: Ownership); these kernels only see its tables.  pack: this rank's bins,
// in own_list order -> one contiguous, bin-major RGBA8 block (the all-gather send buffer, stride_bins bins long,
// zero-filled past the rank's share); unpack: the gathered [rank][k][16][16] blocks -> the linear framebuffer, bin b read
// from block src_of_bin[b].  Plain copies.
#include "mtr_internal.h"

namespace mtr {

__global__ __launch_bounds__(256) void k_pack_shard(const uint32_t* color, uint32_t* dst, uint32_t W, uint32_t H, uint32_t nbx,
                                                    const uint32_t* own_list, uint32_t own_count, uint32_t stride_bins) {
    const uint32_t k = blockIdx.x;  // k-th bin of this rank
    if (k >= stride_bins) return;
    const uint32_t lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    uint32_t v = 0;
    if (k < own_count) {
        const uint32_t bin = own_list[k];
        const uint32_t x = (bin % nbx) * MTR_BIN + lx, y = (bin / nbx) * MTR_BIN + ly;
        if (x < W && y < H) v = color[(size_t)y * W + x];
    }
    dst[(size_t)k * 256 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void k_unpack_shards(const uint32_t* gathered, uint32_t* color, uint32_t W, uint32_t H,
                                                       uint32_t nbx, uint32_t nbins, const uint32_t* src_of_bin) {
    const uint32_t bin = blockIdx.x;
    if (bin >= nbins) return;
    const uint32_t lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const uint32_t x = (bin % nbx) * MTR_BIN + lx, y = (bin / nbx) * MTR_BIN + ly;
    if (x < W && y < H) color[(size_t)y * W + x] = gathered[(size_t)src_of_bin[bin] * 256 + threadIdx.x];
}

// W % 4 == 0: 16 bytes per thread (four pixels of a bin row), four bins per 256-thread block.  The scalar kernels
// above move 4 bytes per thread and cost ~15 us each per 1080p frame; they remain for odd widths.
__global__ __launch_bounds__(256) void k_pack_shard_v4(const uint4* color, uint4* dst, uint32_t W4, uint32_t H, uint32_t nbx,
                                                       const uint32_t* own_list, uint32_t own_count, uint32_t stride_bins) {
    const uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6);  // k-th bin of this rank
    if (k >= stride_bins) return;
    const uint32_t t = threadIdx.x & 63;
    const uint32_t lx4 = t & 3, ly = t >> 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (k < own_count) {
        const uint32_t bin = own_list[k];
        const uint32_t x4 = (bin % nbx) * (MTR_BIN / 4) + lx4, y = (bin / nbx) * MTR_BIN + ly;
        if (x4 < W4 && y < H) v = color[(size_t)y * W4 + x4];
    }
    dst[(size_t)k * 64 + t] = v;
}

__global__ __launch_bounds__(256) void k_unpack_shards_v4(const uint4* gathered, uint4* color, uint32_t W4, uint32_t H,
                                                          uint32_t nbx, uint32_t nbins, const uint32_t* src_of_bin) {
    const uint32_t bin = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bin >= nbins) return;
    const uint32_t t = threadIdx.x & 63;
    const uint32_t x4 = (bin % nbx) * (MTR_BIN / 4) + (t & 3), y = (bin / nbx) * MTR_BIN + (t >> 2);
    if (x4 < W4 && y < H) color[(size_t)y * W4 + x4] = gathered[(size_t)src_of_bin[bin] * 64 + t];
}

}  // namespace mtr

void mtr_launch_pack_shard(const uint8_t* color, uint8_t* dst, uint32_t W, uint32_t H, const uint32_t* own_list, uint32_t own_count,
                           uint32_t stride_bins, hipStream_t s) {
    const uint32_t nbx = (W + MTR_BIN - 1) / MTR_BIN;
    if (stride_bins == 0) return;
    if (W % 4 == 0 && ((uintptr_t)color | (uintptr_t)dst) % 16 == 0) {
        hipLaunchKernelGGL(mtr::k_pack_shard_v4, dim3((stride_bins + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(color),
                           reinterpret_cast<uint4*>(dst), W / 4, H, nbx, own_list, own_count, stride_bins);
        return;
    }
    hipLaunchKernelGGL(mtr::k_pack_shard, dim3(stride_bins), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(color),
                       reinterpret_cast<uint32_t*>(dst), W, H, nbx, own_list, own_count, stride_bins);
}

void mtr_launch_unpack_shards(const uint8_t* gathered, uint8_t* color, uint32_t W, uint32_t H, const uint32_t* src_of_bin, hipStream_t s) {
    const uint32_t nbx = (W + MTR_BIN - 1) / MTR_BIN, nby = (H + MTR_BIN - 1) / MTR_BIN, nbins = nbx * nby;
    if (W % 4 == 0 && ((uintptr_t)gathered | (uintptr_t)color) % 16 == 0) {
        hipLaunchKernelGGL(mtr::k_unpack_shards_v4, dim3((nbins + 3) / 4), dim3(256), 0, s, reinterpret_cast<const uint4*>(gathered),
                           reinterpret_cast<uint4*>(color), W / 4, H, nbx, nbins, src_of_bin);
        return;
    }
    hipLaunchKernelGGL(mtr::k_unpack_shards, dim3(nbins), dim3(256), 0, s, reinterpret_cast<const uint32_t*>(gathered),
                       reinterpret_cast<uint32_t*>(color), W, H, nbx, nbins, src_of_bin);
}
