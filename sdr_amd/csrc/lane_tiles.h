// Tables of 32-row tiles of a batch A for the lane-per-row match kernels (lane_rows.hip, few_ande.hip):
//
//     mask[w]  (uint32)  bit q set  <=>  row 32 t + q of A contains id w       [wcap + 4 entries, the last four zero]
//     col[w]   (uint16)  where the tile's payloads of id w start (byte offset into the tile)   [wcap + 8 entries]
//     vals[]   (float)   the tile's payloads, column-compressed: by id, then by ascending row  [val_cap]
//     len[q]   (uint32)  length of row 32 t + q                                                [32]
//
// built once per call in global memory, one block per tile.
#pragma once

#include "sdrb_internal.h"

namespace sdrb {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;
constexpr int kTile = 32;            // rows of A per tile = bits of a mask word
constexpr int kMaxRowA = 128;        // longest row of A: a tile's payloads stay below 16 KB

constexpr uint32_t mask_bytes(uint32_t wcap) { return (wcap + 4) * 4; }   // entry [wcap] = 0: the sentinel id
constexpr uint32_t col_bytes(uint32_t wcap) { return (wcap + 8) * 2; }
constexpr uint32_t kLenBytes = kTile * 4;

// One block per tile of A: mask, col, the column-compressed payloads and the row lengths, written to the tile's
// slot in `tables`.  a_val == nullptr: A carries no payloads (only the ids and lengths are tabulated).
__global__ __launch_bounds__(kBlock) void k_build_lane_tiles(const uint32_t* __restrict__ a_off,
                                                             const int32_t* __restrict__ a_ids,
                                                             const float* __restrict__ a_val, uint32_t na,
                                                             uint32_t wcap, uint32_t val_cap, uint32_t tile_bytes,
                                                             unsigned char* __restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) uint32_t build_lds[];
    uint32_t* mask = build_lds;                                         // [wcap]
    uint16_t* col = reinterpret_cast<uint16_t*>(mask + wcap);           // [wcap]
    uint32_t* wave_total = reinterpret_cast<uint32_t*>(col + wcap);     // [kBlock / 64]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = blockIdx.x * kTile;
    const uint32_t rows = na - i0 < (uint32_t)kTile ? na - i0 : (uint32_t)kTile;
    unsigned char* tile = tables + (uint64_t)blockIdx.x * tile_bytes;
    uint32_t* g_mask = reinterpret_cast<uint32_t*>(tile);
    uint16_t* g_col = reinterpret_cast<uint16_t*>(tile + mask_bytes(wcap));
    const uint32_t vals_at = mask_bytes(wcap) + col_bytes(wcap);
    uint32_t* g_len = reinterpret_cast<uint32_t*>(tile + vals_at + val_cap * 4u);

    for (uint32_t w = tid; w < wcap; w += kBlock) mask[w] = 0;
    if (tid < (uint32_t)kTile) g_len[tid] = tid < rows ? a_off[i0 + tid + 1] - a_off[i0 + tid] : 0u;
    __syncthreads();
    for (uint32_t q = wave; q < rows; q += kBlock / 64) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < wcap) atomicOr(&mask[id], 1u << q);   // ids beyond B's table cannot match an id of B
        }
    }
    __syncthreads();
    // col: every thread counts a contiguous piece of the table, the pieces' totals are scanned over the block
    const uint32_t per = wcap / kBlock;
    const uint32_t w0 = tid * per;
    uint32_t mine = 0;
    for (uint32_t w = w0; w < w0 + per; ++w) mine += (uint32_t)__builtin_popcount(mask[w]);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
        if (lane >= (uint32_t)d) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    uint32_t run = incl - mine;
    for (uint32_t w = 0; w < wave; ++w) run += wave_total[w];
    for (uint32_t w = w0; w < w0 + per; ++w) {
        const uint32_t m = mask[w];
        col[w] = (uint16_t)run;
        g_mask[w] = m;
        g_col[w] = (uint16_t)(vals_at + run * 4u);
        run += (uint32_t)__builtin_popcount(m);
    }
    if (tid < 4) g_mask[wcap + tid] = 0;
    if (tid < 8) g_col[wcap + tid] = (uint16_t)vals_at;
    __syncthreads();
    if (a_val == nullptr) return;
    float* g_vals = reinterpret_cast<float*>(tile + vals_at);
    for (uint32_t q = wave; q < rows; q += kBlock / 64) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < wcap)
                g_vals[col[id] + (uint32_t)__builtin_popcount(mask[id] & ((1u << q) - 1u))] = a_val[beg + k];
        }
    }
}

// Builds the tables of all tiles of `a` over `wcap` ids in scratch slot 5 of the stream.  with_vals: A's payloads
// are tabulated too.
inline int build_lane_tiles(const sdrb_batch* a, int device, int wcap, bool with_vals, hipStream_t stream,
                            uint32_t* tile_bytes, uint32_t* ntiles, const unsigned char** tables_out) {
    *ntiles = (uint32_t)((a->n_rows + kTile - 1) / kTile);
    const uint32_t val_cap = with_vals ? ((uint32_t)kTile * (a->max_row_len ? a->max_row_len : 1u) + 3u) & ~3u : 0u;
    *tile_bytes = mask_bytes((uint32_t)wcap) + col_bytes((uint32_t)wcap) + val_cap * 4u + kLenBytes;
    unsigned char* tables = static_cast<unsigned char*>(workspace(device, stream, 5, (size_t)*ntiles * *tile_bytes));
    if (!tables) return SDRB_ERR_HIP;
    const size_t build_lds = (size_t)wcap * 6 + 64;
    hipLaunchKernelGGL(k_build_lane_tiles, dim3(*ntiles), dim3(kBlock), build_lds, stream, a->d_offsets, a->d_ids,
                       with_vals ? a->d_data : nullptr, (uint32_t)a->n_rows, (uint32_t)wcap, val_cap, *tile_bytes, tables);
    SDRB_HIP(hipGetLastError());
    *tables_out = tables;
    return SDRB_OK;
}

}  // namespace

}  // namespace sdrb
