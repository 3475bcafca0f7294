// Perf lab (not shipped): a lane-per-row ande fill for long rows that was measured and NOT adopted.
// Config 4 (64 x 128 Ki rows, 16384/328, UnitData): the fill takes 1.18 ms against 0.46 + 0.29 ms for the mask-table
// fill and its sort; sizes 0.37 ms against 0.23 ms.  Counters: 10.5 M scattered store instructions of ~10 active
// lanes at ~64 cycles each per compute unit, VALU 13 % busy -- every chunk's loads also wait for the stores of the
// chunk before (one counter orders loads and stores here), and nothing can be fetched ahead of them for 328-id rows.
// To build it again: add this file to sdr_amd/csrc/Makefile and declare its three entry points in sdrb_internal.h.
// Lane-per-row match kernel for rows of any length: ande (elements) and the payload-dependent size ops on
// batches whose rows do not fit a lane's registers (e.g. UnitData rows of 328 ids over a width of 16384).
//
// The reference finds the common ids of a pair with andv (SDR.hpp:958-997) and keeps a match when the combined
// payload is relevant (SDR.hpp:999-1025 with UnitData.hpp:22-51); every match is visited in ascending id order.
// Same plan as lane_rows.hip with the roles sized for long rows: the rows of A are cut into tiles of 16, a tile is
//
//     entry[w] (uint32)  low half: bit q set <=> row 16 t + q of A contains id w
//                        high half: where the tile's payloads of id w start (index into vals)
//     vals[]   (float)   the tile's payloads, column-compressed: by id, then by ascending row
//     len[q]   (uint32)  length of row 16 t + q
//
// built once per call in global memory and staged in LDS by a block of 16 waves.  A lane owns one row of B and
// STREAMS it: 32 elements at a time travel into registers (16-byte loads), each costs one LDS read, and every set
// bit of its mask is a match -- one payload of A from LDS, the payload rule, and an update of the lane's own
// counter (size ops) or output cursor (ande) of that row of A, kept in LDS as acc[q][lane]: nothing is shared
// between lanes, so there are no atomics, and because a lane walks its row in ascending id order the elements of
// every pair are written in id order -- no sort afterwards.  The loads of a chunk are all issued before its first
// store (on this hardware a wait for a load also waits for every store issued before it).
#include <type_traits>

#include "match_math.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 1024;         // rows of B per block: 16 waves share one tile's tables
constexpr int kTile = 16;            // rows of A per tile = bits of a mask half-word
constexpr int kChunk = 32;           // elements of a row a lane holds at a time
constexpr uint32_t kLenBytes = kTile * 4;

constexpr uint32_t entry_bytes(uint32_t width) { return ((width + 1 + 3) & ~3u) * 4; }   // entry [width] = 0: the sentinel

// One block per tile of A.
__global__ __launch_bounds__(256) void k_build_stream_tiles(const uint32_t* __restrict__ a_off,
                                                            const int32_t* __restrict__ a_ids,
                                                            const float* __restrict__ a_val, uint32_t na,
                                                            uint32_t width, uint32_t val_cap, uint32_t tile_bytes,
                                                            unsigned char* __restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) uint32_t stream_build_lds[];
    uint32_t* mask = stream_build_lds;                 // [width]
    uint32_t* col = mask + width;                      // [width]
    uint32_t* wave_total = col + width;                // [4]
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = blockIdx.x * kTile;
    const uint32_t rows = na - i0 < (uint32_t)kTile ? na - i0 : (uint32_t)kTile;
    unsigned char* tile = tables + (uint64_t)blockIdx.x * tile_bytes;
    uint32_t* g_entry = reinterpret_cast<uint32_t*>(tile);
    float* g_vals = reinterpret_cast<float*>(tile + entry_bytes(width));
    uint32_t* g_len = reinterpret_cast<uint32_t*>(tile + entry_bytes(width) + val_cap * 4u);

    for (uint32_t w = tid; w < width; w += 256) mask[w] = 0;
    if (tid < (uint32_t)kTile) g_len[tid] = tid < rows ? a_off[i0 + tid + 1] - a_off[i0 + tid] : 0u;
    __syncthreads();
    for (uint32_t q = wave; q < rows; q += 4) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) atomicOr(&mask[id], 1u << q);   // ids outside B's width never match
        }
    }
    __syncthreads();
    // col: every thread counts a contiguous piece of the table, the pieces' totals are scanned over the block
    const uint32_t per = (width + 255) / 256;
    const uint32_t w0 = tid * per < width ? tid * per : width, w1 = w0 + per < width ? w0 + per : width;
    uint32_t mine = 0;
    for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__builtin_popcount(mask[w]);
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
    for (uint32_t w = w0; w < w1; ++w) {
        const uint32_t m = mask[w];
        col[w] = run;
        g_entry[w] = m | (run << 16);
        run += (uint32_t)__builtin_popcount(m);
    }
    if (tid < 4 && width + tid < entry_bytes(width) / 4) g_entry[width + tid] = 0;
    __syncthreads();
    if (a_val == nullptr) return;
    for (uint32_t q = wave; q < rows; q += 4) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) g_vals[col[id] + (uint32_t)__builtin_popcount(mask[id] & ((1u << q) - 1u))] = a_val[beg + k];
        }
    }
}

// MATH : how the payloads of a match decide whether it is kept (match_math.h)
// FILL : write the kept matches (id, combined payload) at the pairs' output offsets; else write the sizes
template <int MATH, bool FILL>
__global__ __launch_bounds__(kBlock, 1) void k_lane_stream(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint32_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t width, uint32_t val_cap,
    uint32_t na, int op, uint32_t swap, uint32_t* __restrict__ out, uint64_t ld_out,
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, float* __restrict__ out_data) {
    constexpr bool NEED_A = MATH == kSelfA || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    constexpr bool NEED_B = MATH == kSelfB || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    // LDS: the tile's tables (tile_bytes) | acc[16][1024]
    extern __shared__ __attribute__((aligned(16))) unsigned char stream_lds[];
    const uint32_t* entry = reinterpret_cast<const uint32_t*>(stream_lds);
    const float* vals = reinterpret_cast<const float*>(stream_lds + entry_bytes(width));
    const uint32_t* la = reinterpret_cast<const uint32_t*>(stream_lds + entry_bytes(width) + val_cap * 4u);
    const uint32_t tid = threadIdx.x;
    uint32_t* acc_lane = reinterpret_cast<uint32_t*>(stream_lds + tile_bytes) + tid;

    const uint32_t tile = blockIdx.y, i0 = tile * kTile;
    const uint32_t rows_a = na - i0 < (uint32_t)kTile ? na - i0 : (uint32_t)kTile;
    {   // stage the tile's tables
        const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)tile * tile_bytes);
        const uint32_t nvec = tile_bytes / 16u;
        for (uint32_t v = tid; v < nvec; v += kBlock) reinterpret_cast<u32x4*>(stream_lds)[v] = src[v];
    }
    __syncthreads();

    const uint32_t n_groups = (nb + kBlock - 1) / kBlock;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kBlock, j = j0 + tid;
        uint32_t beg = 0, lb = 0;
        if (j < nb) {
            beg = b_off[j];
            lb = b_off[j + 1] - beg;
        }
        // bounds-checked buffers that start at the block's first element: loads are unconditional
        const uint32_t blk_beg = b_off[j0];
        const uint32_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
        const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(b_ids) + blk_beg, 0,
                                                                            (int)blk_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(b_val) + (NEED_B ? blk_beg : 0u), 0, NEED_B ? (int)blk_bytes : 0, 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
        // accumulators: size ops count from zero; the fill starts every pair's cursor at its output offset
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            uint32_t start = 0;
            if constexpr (FILL) {
                if ((uint32_t)q < rows_a && j < nb) start = out_off[(uint64_t)(i0 + q) * nb + j];
            }
            acc_lane[q * kBlock] = start;
        }
        // the longest row of the wave sets the number of chunks
        uint32_t longest = lb;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)longest, d, 64);
            longest = other > longest ? other : longest;
        }
        for (uint32_t k0 = 0; k0 < longest; k0 += kChunk) {
            uint32_t id[kChunk];
            float bv[kChunk];
#pragma unroll
            for (int k4 = 0; k4 < kChunk / 4; ++k4) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + (k0 + 4u * k4) * 4u), 0, 0);
                id[4 * k4 + 0] = v[0];
                id[4 * k4 + 1] = v[1];
                id[4 * k4 + 2] = v[2];
                id[4 * k4 + 3] = v[3];
            }
            if constexpr (NEED_B) {
#pragma unroll
                for (int k4 = 0; k4 < kChunk / 4; ++k4) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, (int)(rel + (k0 + 4u * k4) * 4u), 0, 0);
                    bv[4 * k4 + 0] = __uint_as_float(v[0]);
                    bv[4 * k4 + 1] = __uint_as_float(v[1]);
                    bv[4 * k4 + 2] = __uint_as_float(v[2]);
                    bv[4 * k4 + 3] = __uint_as_float(v[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const uint32_t w = k0 + (uint32_t)k < lb ? id[k] : width;   // past the row's end: the sentinel id
                const uint32_t e = entry[w];
                uint32_t mm = e & 0xffffu, cc = e >> 16;
                const float vb = NEED_B ? bv[k] : 0.0f;
                while (mm != 0) {
                    const uint32_t q = (uint32_t)__builtin_ctz(mm);
                    mm &= mm - 1;
                    float va = 0.0f;
                    if constexpr (NEED_A) va = vals[cc++];
                    float d;
                    const bool kept = match_kept_by<MATH>(va, vb, d);
                    uint32_t* slot = acc_lane + q * kBlock;
                    if constexpr (!FILL) {
                        *slot = *slot + (0x10000u | (kept ? 1u : 0u));
                    } else if (kept) {
                        const uint32_t p = *slot;
                        *slot = p + 1;
                        out_ids[p] = (int32_t)w;
                        if constexpr (MATH != kKeepAll) out_data[p] = d;
                    }
                }
            }
        }
        if constexpr (!FILL) {
            // sizes, one coalesced 256-byte segment per wave per row of A
            if (j < nb) {
#pragma unroll
                for (int q = 0; q < kTile; ++q) {
                    if ((uint32_t)q < rows_a) {
                        const uint32_t c = acc_lane[q * kBlock];
                        const uint32_t l_a = la[q];
                        out[(uint64_t)(i0 + q) * ld_out + j] = size_of(op, swap ? lb : l_a, swap ? l_a : lb, c >> 16, c & 0xffffu);
                    }
                }
            }
        }
    }
}

struct Plan {
    uint32_t width, val_cap, tile_bytes, ntiles;
    size_t lds;
    bool ok;
};

Plan make_plan(const sdrb_batch* a, const sdrb_batch* b, bool need_a) {
    Plan p{};
    p.ok = false;
    if (b->width <= 0 || a->n_rows == 0 || a->n_rows > 0xFFFF0ull || b->n_rows > 0xFFFFFF00ull) return p;
    p.width = (uint32_t)b->width;
    const uint64_t longest = a->max_row_len < p.width ? a->max_row_len : p.width;
    const uint64_t cap = need_a ? ((uint64_t)kTile * (longest ? longest : 1u) + 3u) & ~3ull : 0u;
    if (cap > 65535u) return p;     // the payload index of an entry is 16 bit
    if (b->max_row_len >= 65536u) return p;   // size counters hold matches in 16 bits
    p.val_cap = (uint32_t)cap;
    p.tile_bytes = entry_bytes(p.width) + p.val_cap * 4u + kLenBytes;
    p.ntiles = (uint32_t)((a->n_rows + kTile - 1) / kTile);
    p.lds = (size_t)p.tile_bytes + (size_t)kTile * kBlock * sizeof(uint32_t);
    p.ok = p.lds <= 160 * 1024 - 512 && (size_t)p.width * 8 + 64 <= 160 * 1024 - 512;
    return p;
}

template <int MATH, bool FILL>
int launch(const sdrb_batch* b, const Plan& p, const unsigned char* tables, uint32_t na, int op, bool swap, uint32_t* out,
           uint64_t ld_out, const uint32_t* out_off, int32_t* out_ids, float* out_data, hipStream_t stream) {
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lane_stream<MATH, FILL>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    const uint32_t groups = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    // one block per compute unit at a time: a block keeps its tile and strides over the groups of B
    uint32_t gx = (512u + p.ntiles - 1) / p.ntiles;
    if (gx > groups) gx = groups;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL((k_lane_stream<MATH, FILL>), dim3(gx, p.ntiles), dim3(kBlock), p.lds, stream, b->d_offsets, b->d_ids,
                       b->d_data, (uint32_t)b->n_rows, tables, p.tile_bytes, p.width, p.val_cap, na, op, swap ? 1u : 0u, out,
                       ld_out, out_off, out_ids, out_data);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <bool FILL>
int run(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
        const uint32_t* out_off, int32_t* out_ids, float* out_data, hipStream_t stream) {
    const int math = pick_math(a, b, op, swap);
    const bool need_a = math == kSelfA || math == kMul || math == kAbsDiff || math == kRmulA || math == kRmulB;
    const Plan p = make_plan(a, b, need_a);
    if (!p.ok) {
        set_error("lane-per-row streaming kernel: tables of this width / these rows do not fit LDS");
        return SDRB_ERR_UNSUPPORTED;
    }
    unsigned char* tables = static_cast<unsigned char*>(workspace(b->device, stream, 5, (size_t)p.ntiles * p.tile_bytes));
    if (!tables) return SDRB_ERR_HIP;
    const size_t build_lds = (size_t)p.width * 8 + 64;
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_stream_tiles),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)build_lds));
    hipLaunchKernelGGL(k_build_stream_tiles, dim3(p.ntiles), dim3(256), build_lds, stream, a->d_offsets, a->d_ids,
                       need_a ? a->d_data : nullptr, (uint32_t)a->n_rows, p.width, p.val_cap, p.tile_bytes, tables);
    SDRB_HIP(hipGetLastError());
    const uint32_t na = (uint32_t)a->n_rows;
    switch (math) {
    case kKeepAll: return launch<kKeepAll, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kDropAll: return launch<kDropAll, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kSelfA: return launch<kSelfA, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kSelfB: return launch<kSelfB, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kMul: return launch<kMul, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kAbsDiff: return launch<kAbsDiff, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    case kRmulA: return launch<kRmulA, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    default: return launch<kRmulB, FILL>(b, p, tables, na, op, swap, out, ld_out, out_off, out_ids, out_data, stream);
    }
}

}  // namespace

// `forced`: whether the kernel can serve the batches at all; else whether it is also the better choice (enough rows
// of B to give every compute unit a block of 1024, more rows of A than the few-rows kernels take)
bool lane_stream_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, bool forced) {
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) return false;
    const int math = pick_math(a, b, op, swap);
    const bool need_a = math == kSelfA || math == kMul || math == kAbsDiff || math == kRmulA || math == kRmulB;
    if (!make_plan(a, b, need_a).ok) return false;
    return forced || (a->n_rows > 16 && b->n_rows >= 64 * 1024);
}

int launch_size_lane_stream(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                            hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    return run<false>(a, b, op, swap, out, ld_out, nullptr, nullptr, nullptr, stream);
}

// ande for every pair: the kept matches of a pair, (id, combined payload), in id order
int launch_elem_lane_stream(const sdrb_batch* a, const sdrb_batch* b, bool swap, const uint32_t* out_offsets,
                            int32_t* out_ids, float* out_data, hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    return run<true>(a, b, SDRB_OP_AND, swap, nullptr, 0, out_offsets, out_ids, out_data, stream);
}

}  // namespace sdrb
