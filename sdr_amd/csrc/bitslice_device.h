// Device building blocks shared by the bit-sliced table kernels (size_bitslice.hip,
// allpairs_bitslice.hip): the transposed bit table of a batch and the register-staged copy of
// table tiles into LDS (the carry-save vertical counters live in vertical_counters.h).
#pragma once

#include "sdrb_internal.h"
#include "vertical_counters.h"

namespace sdrb {

namespace {

using u64 = unsigned long long;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// tuning constants (overridable at build time for experiments)
#ifndef SDRB_BS_BLOCK
#define SDRB_BS_BLOCK 256
#endif
#ifndef SDRB_BS_MIN_WAVES
#define SDRB_BS_MIN_WAVES 3
#endif
constexpr int kBlock = SDRB_BS_BLOCK;  // threads per block = rows of B per block
constexpr int kMinWaves = SDRB_BS_MIN_WAVES;  // waves per SIMD the register allocator must allow
constexpr int kTileRows = 64;  // rows of A per table tile = bits of one table word

// one wave per row, kBuildRows rows per block: launch with ceil(n_rows / kBuildRows) blocks of 64 * kBuildRows threads
constexpr int kBuildRows = 4;
__global__ __launch_bounds__(64 * kBuildRows) __attribute__((unused)) void k_build_table(const uint32_t* __restrict__ a_off,
                                                                  const int32_t* __restrict__ a_ids,
                                                                  uint32_t n_rows, int32_t width, uint32_t tile_words,
                                                                  u64* __restrict__ table,
                                                                  uint32_t* __restrict__ a_len) {
    const uint64_t i = (uint64_t)blockIdx.x * kBuildRows + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t beg = a_off[i];
    const uint32_t len = a_off[i + 1] - beg;
    if (lane == 0) a_len[i] = len;
    u64* tile = table + (i / kTileRows) * (uint64_t)tile_words;
    const u64 bit = 1ull << (i % kTileRows);
    for (uint32_t k = lane; k < len; k += 64) {
        const int32_t id = a_ids[beg + k];
        // ids outside B's declared width cannot match any id of B
        if ((uint32_t)id < (uint32_t)width) atomicOr(&tile[id], bit);
    }
}

template <int PF>
__device__ __forceinline__ void load_tile(u32x4 (&pf)[PF], const u64* __restrict__ tile, uint32_t tid) {
    const u32x4* src = reinterpret_cast<const u32x4*>(tile);
#pragma unroll
    for (int p = 0; p < PF; ++p) pf[p] = src[tid + p * kBlock];
}

template <int PF>
__device__ __forceinline__ void write_tile(const u32x4 (&pf)[PF], u64* buf, uint32_t tid) {
    u32x4* dst = reinterpret_cast<u32x4*>(buf);
#pragma unroll
    for (int p = 0; p < PF; ++p) dst[tid + p * kBlock] = pf[p];
}

}  // namespace

}  // namespace sdrb
