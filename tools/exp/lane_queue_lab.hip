// Perf lab (not shipped): lane_stream_lab.hip's ande fill with the stores taken off the walk.
// A lane owns one row of B and walks it in ascending id order against 16-row tables of A, so every pair's kept
// matches are met in id order and need no sort.  Instead of storing each kept match itself (one scattered store
// instruction with a handful of active lanes per turn of the bit loop), a wave pushes its kept matches into a ring
// in LDS -- (output position, id, payload), compacted over the lanes with a ballot -- and whenever the ring holds 64
// entries the whole wave stores them with all lanes active.  Cursors are 16 bit, relative to the output position of
// the wave's first pair of the same row of A, so that 1024 rows of B share a tile's tables.
// Build: tools/exp/build_queue_lab.sh   (product objects + this file -> tools/exp/libsdr_amd_queue.so)
//
// Measured (config 4: 64 x 128 Ki rows, 16384/328, UnitData, 37.2 M kept elements, results equal to the product's bit
// for bit; tools/exp/time_queue_lab.py) and NOT adopted -- the product's fill + sort take 0.46 + 0.29 ms:
//   rows of B read where they lie (16 bytes a lane at a row's stride)            1.23 ms
//   rows read from a copy dealt to the lanes (1 KiB contiguous a wave load)      1.09 ms  (+0.37 ms to make the copy)
//   the same with the ring never stored (-DLQ_NOFLUSH, timing only)              0.49 ms
// So full-wave store instructions do not help: 74 M scattered 4-byte lane-stores (id and payload of 37 M elements)
// cost 0.6 ms whether an instruction carries 10 lanes or 64 -- about 5 cycles of a compute unit per lane-store, each to
// a line of its own -- and the walk itself (two LDS round trips a match, ~10 LDS instructions a turn of the bit loop,
// 16 waves sharing one LDS pipe) takes as long as the product's fill.  What would beat both is an output staged in
// LDS in destination order and copied out as whole lines, which at this width the tables leave no room for.
// (A further timing-only switch that skipped the ring WRITES as well made the final flush store to whatever positions LDS
// held and faulted the GPU once; it was removed.  Timing-only builds here may drop stores and reads, never the writes
// that later addresses are read from.)
#include <type_traits>

#include "match_math.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#ifndef LQ_CHUNK
#define LQ_CHUNK 32
#endif
constexpr int kBlock = 1024;         // rows of B per block: 16 waves share one tile's tables
constexpr int kWaves = kBlock / 64;
constexpr int kTile = 16;            // rows of A per tile = bits of a mask half-word
constexpr int kChunk = LQ_CHUNK;     // elements of a row a lane holds at a time
constexpr int kQueue = 128;          // ring entries per wave
constexpr uint32_t kLenBytes = kTile * 4;

constexpr uint32_t entry_bytes(uint32_t width) { return ((width + 1 + 3) & ~3u) * 4; }   // entry [width] = 0: the sentinel

// One block per tile of A: entry[w] = mask of the tile's rows that hold id w | (start of their payloads << 16),
// payloads by id, then by ascending row.
__global__ __launch_bounds__(256) void k_build_queue_tiles(const uint32_t* __restrict__ a_off,
                                                           const int32_t* __restrict__ a_ids,
                                                           const float* __restrict__ a_val, uint32_t na,
                                                           uint32_t width, uint32_t val_cap, uint32_t tile_bytes,
                                                           unsigned char* __restrict__ tables) {
    extern __shared__ __attribute__((aligned(16))) uint32_t queue_build_lds[];
    uint32_t* mask = queue_build_lds;                  // [width]
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


// Rows of B dealt to the lanes of a wave: for each group of 64 consecutive rows, the four elements 4*k4 .. 4*k4+3 of
// row 64*g + lane sit at t_ids[(g * l4 + k4) * 64 + lane] (16 bytes a lane, 1 KiB contiguous a wave instruction);
// slots past a row's end hold the sentinel id `width` (payload 0).
__global__ __launch_bounds__(256) void k_deal_rows(const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids,
                                                   const float* __restrict__ b_val, uint32_t nb, uint32_t width,
                                                   uint32_t l4, u32x4* __restrict__ t_ids, u32x4* __restrict__ t_val) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t j = g * 64 + lane;
    uint32_t beg = 0, lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    for (uint32_t k4 = wave; k4 < l4; k4 += 4) {
        u32x4 vi, vv;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t k = 4 * k4 + c;
            vi[c] = k < lb ? (uint32_t)b_ids[beg + k] : width;
            vv[c] = (b_val != nullptr && k < lb) ? __float_as_uint(b_val[beg + k]) : 0u;
        }
        t_ids[((uint64_t)g * l4 + k4) * 64 + lane] = vi;
        if (b_val != nullptr) t_val[((uint64_t)g * l4 + k4) * 64 + lane] = vv;
    }
}

__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// MATH : how the payloads of a match decide whether it is kept (match_math.h)
template <int MATH, bool TR>
__global__ __launch_bounds__(kBlock, 1) void k_lane_queue_fill(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    const u32x4* __restrict__ t_ids, const u32x4* __restrict__ t_val, uint32_t l4, uint32_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t width, uint32_t val_cap,
    uint32_t na, const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, float* __restrict__ out_data) {
    constexpr bool NEED_A = MATH == kSelfA || MATH == kMul;
    constexpr bool NEED_B = MATH == kSelfB || MATH == kMul;
    // LDS: the tile's tables (tile_bytes) | cursors u16 [16][1024] | wave bases u32 [16 waves][16] | rings
    extern __shared__ __attribute__((aligned(16))) unsigned char queue_lds[];
    const uint32_t* entry = reinterpret_cast<const uint32_t*>(queue_lds);
    const float* vals = reinterpret_cast<const float*>(queue_lds + entry_bytes(width));
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* p = queue_lds + tile_bytes;
    uint16_t* cur_lane = reinterpret_cast<uint16_t*>(p) + tid;
    p += kTile * kBlock * 2;
    uint32_t* wbase = reinterpret_cast<uint32_t*>(p) + wave * kTile;
    p += kWaves * kTile * 4;
    uint32_t* qpos = reinterpret_cast<uint32_t*>(p) + wave * 3 * kQueue;
    uint32_t* qid = qpos + kQueue;
    float* qval = reinterpret_cast<float*>(qid + kQueue);

    const uint32_t tile = blockIdx.y, i0 = tile * kTile;
    const uint32_t rows_a = na - i0 < (uint32_t)kTile ? na - i0 : (uint32_t)kTile;
    {   // stage the tile's tables
        const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)tile * tile_bytes);
        const uint32_t nvec = tile_bytes / 16u;
        for (uint32_t v = tid; v < nvec; v += kBlock) reinterpret_cast<u32x4*>(queue_lds)[v] = src[v];
    }
    __syncthreads();

    uint32_t qhead = 0, qtail = 0;   // wave-uniform
    const uint32_t n_groups = (nb + kBlock - 1) / kBlock;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kBlock, j = j0 + tid;
        uint32_t beg = 0, lb = 0;
        if (j < nb) {
            beg = b_off[j];
            lb = b_off[j + 1] - beg;
        }
        const uint32_t blk_beg = b_off[j0];
        const uint32_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
        const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(b_ids) + blk_beg, 0,
                                                                            (int)blk_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(b_val) + (NEED_B ? blk_beg : 0u), 0, NEED_B ? (int)blk_bytes : 0, 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
        // cursors: a pair's output offset relative to the offset of the wave's first pair of that row of A
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            uint32_t start = 0;
            if ((uint32_t)q < rows_a && j < nb) start = out_off[(uint64_t)(i0 + q) * nb + j];
            const uint32_t base = (uint32_t)__builtin_amdgcn_readfirstlane((int)start);
            cur_lane[q * kBlock] = (uint16_t)(start - base);
            if (lane == 0) wbase[q] = base;
        }
        uint32_t longest = lb;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)longest, d, 64);
            longest = other > longest ? other : longest;
        }
        longest = (uint32_t)__builtin_amdgcn_readfirstlane((int)longest);
        if (TR && longest > 4 * l4) longest = 4 * l4;
        for (uint32_t k0 = 0; k0 < longest; k0 += kChunk) {
            uint32_t id[kChunk];
            float bv[kChunk];
#pragma unroll
            for (int k4 = 0; k4 < kChunk / 4; ++k4) {
                u32x4 v;
                if constexpr (TR) v = t_ids[((uint64_t)(j >> 6) * l4 + k0 / 4 + k4) * 64 + lane];
                else v = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + (k0 + 4u * k4) * 4u), 0, 0);
                id[4 * k4 + 0] = v[0];
                id[4 * k4 + 1] = v[1];
                id[4 * k4 + 2] = v[2];
                id[4 * k4 + 3] = v[3];
            }
            if constexpr (NEED_B) {
#pragma unroll
                for (int k4 = 0; k4 < kChunk / 4; ++k4) {
                    u32x4 v;
                    if constexpr (TR) v = t_val[((uint64_t)(j >> 6) * l4 + k0 / 4 + k4) * 64 + lane];
                    else v = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, (int)(rel + (k0 + 4u * k4) * 4u), 0, 0);
                    bv[4 * k4 + 0] = __uint_as_float(v[0]);
                    bv[4 * k4 + 1] = __uint_as_float(v[1]);
                    bv[4 * k4 + 2] = __uint_as_float(v[2]);
                    bv[4 * k4 + 3] = __uint_as_float(v[3]);
                }
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const uint32_t w = (TR || k0 + (uint32_t)k < lb) ? id[k] : width;   // past the row's end: the sentinel id
                const uint32_t e = entry[w];
                uint32_t mm = e & 0xffffu, cc = e >> 16;
                const float vb = NEED_B ? bv[k] : 0.0f;
                while (__builtin_amdgcn_ballot_w64(mm != 0) != 0) {   // every lane stays in the loop: the ring is the wave's
                    const bool act = mm != 0;
                    const uint32_t q = (uint32_t)__builtin_ctz(mm | 0x8000u);
                    mm &= mm - 1;
                    float va = 0.0f;
                    if constexpr (NEED_A) {
                        if (act) va = vals[cc];
                        ++cc;
                    }
                    float d;
                    const bool kept = match_kept_by<MATH>(va, vb, d) && act;
                    const uint64_t kb = __builtin_amdgcn_ballot_w64(kept);
                    if (kb != 0) {
                        if (kept) {
                            uint16_t* slot = cur_lane + q * kBlock;
                            const uint32_t s = *slot;
                            *slot = (uint16_t)(s + 1);
                            const uint32_t idx = (qtail + lanes_below(kb)) & (kQueue - 1);
                            qpos[idx] = wbase[q] + s;
                            qid[idx] = w;
                            if constexpr (MATH != kKeepAll) qval[idx] = d;
                        }
                        qtail += (uint32_t)__builtin_popcountll(kb);
#ifndef LQ_NOFLUSH
                        if (qtail - qhead >= 64u) {
#else   // timing only (results are wrong, every ring entry still holds a valid position): the walk without its stores
                        if (qtail - qhead >= 64u && qtail == 0xffffffffu) {
#endif
                            const uint32_t idx = (qhead + lane) & (kQueue - 1);
                            const uint32_t pos = qpos[idx];
                            out_ids[pos] = (int32_t)qid[idx];
                            if constexpr (MATH != kKeepAll) out_data[pos] = qval[idx];
                            qhead += 64u;
                        }
                    }
                }
            }
        }
    }
    if (lane < qtail - qhead) {
        const uint32_t idx = (qhead + lane) & (kQueue - 1);
        const uint32_t pos = qpos[idx];
        out_ids[pos] = (int32_t)qid[idx];
        if constexpr (MATH != kKeepAll) out_data[pos] = qval[idx];
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
    if (cap > 65535u) return p;               // the payload index of an entry is 16 bit
    if (b->max_row_len > 1023u) return p;     // 64 pairs' results must stay below 65536 elements: 16-bit cursors
    p.val_cap = (uint32_t)cap;
    p.tile_bytes = entry_bytes(p.width) + p.val_cap * 4u + kLenBytes;
    p.ntiles = (uint32_t)((a->n_rows + kTile - 1) / kTile);
    p.lds = (size_t)p.tile_bytes + (size_t)kTile * kBlock * 2 + (size_t)kWaves * kTile * 4 + (size_t)kWaves * 3 * kQueue * 4;
    p.ok = p.lds <= 160 * 1024 - 512 && (size_t)p.width * 8 + 64 <= 160 * 1024 - 512;
    return p;
}

template <int MATH, bool TR>
int launch(const sdrb_batch* b, const Plan& p, const unsigned char* tables, uint32_t na, const uint32_t* out_off,
           int32_t* out_ids, float* out_data, const u32x4* t_ids, const u32x4* t_val, uint32_t l4, hipStream_t stream) {
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lane_queue_fill<MATH, TR>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    const uint32_t groups = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gx = (512u + p.ntiles - 1) / p.ntiles;   // one block per compute unit at a time, two rounds
    if (gx > groups) gx = groups;
    if (gx == 0) gx = 1;
    hipLaunchKernelGGL((k_lane_queue_fill<MATH, TR>), dim3(gx, p.ntiles), dim3(kBlock), p.lds, stream, b->d_offsets, b->d_ids,
                       b->d_data, t_ids, t_val, l4, (uint32_t)b->n_rows, tables, p.tile_bytes, p.width, p.val_cap, na, out_off, out_ids, out_data);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace

}  // namespace sdrb

// ande for every pair (A = this, both UnitData or both EmptyData): the kept matches of a pair, (id, combined
// payload), in id order, written at the pairs' output offsets
// variant 0: rows of B read where they lie; 1: from a copy dealt to the lanes (built here on every call); 2: from
// the copy the last variant-1 call on this stream left (the walk alone, for timing)
extern "C" int sdrb_lab_lane_queue_fill(const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
                                        int32_t* out_ids, float* out_data, void* stream_, int variant) {
    using namespace sdrb;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const int math = pick_math(a, b, SDRB_OP_AND, false);
    if (math != kMul && math != kKeepAll) {
        set_error("lab: only UnitData x UnitData and EmptyData x any");
        return SDRB_ERR_UNSUPPORTED;
    }
    const bool need_a = math == kMul;
    const Plan p = make_plan(a, b, need_a);
    if (!p.ok) {
        set_error("lab: tables of this width / these rows do not fit LDS");
        return SDRB_ERR_UNSUPPORTED;
    }
    unsigned char* tables = static_cast<unsigned char*>(workspace(b->device, stream, 5, (size_t)p.ntiles * p.tile_bytes));
    if (!tables) return SDRB_ERR_HIP;
    const size_t build_lds = (size_t)p.width * 8 + 64;
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_build_queue_tiles),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)build_lds));
    hipLaunchKernelGGL(k_build_queue_tiles, dim3(p.ntiles), dim3(256), build_lds, stream, a->d_offsets, a->d_ids,
                       need_a ? a->d_data : nullptr, (uint32_t)a->n_rows, p.width, p.val_cap, p.tile_bytes, tables);
    SDRB_HIP(hipGetLastError());
    const uint32_t na = (uint32_t)a->n_rows;
    if (variant == 0) {
        if (math == kMul) return launch<kMul, false>(b, p, tables, na, out_offsets, out_ids, out_data, nullptr, nullptr, 0, stream);
        return launch<kKeepAll, false>(b, p, tables, na, out_offsets, out_ids, out_data, nullptr, nullptr, 0, stream);
    }
    const uint32_t chunk4 = kChunk / 4;
    const uint32_t l4 = ((b->max_row_len + 3) / 4 + chunk4 - 1) / chunk4 * chunk4;   // whole chunks
    const uint32_t groups64 = (uint32_t)((b->n_rows + 63) / 64);
    const size_t dealt = (size_t)groups64 * l4 * 64 * sizeof(u32x4);
    unsigned char* t = static_cast<unsigned char*>(workspace(b->device, stream, 12, dealt * (need_a ? 2 : 1)));
    if (!t) return SDRB_ERR_HIP;
    u32x4* t_ids = reinterpret_cast<u32x4*>(t);
    u32x4* t_val = need_a ? reinterpret_cast<u32x4*>(t + dealt) : nullptr;
    if (variant == 1) {
        hipLaunchKernelGGL(k_deal_rows, dim3(groups64), dim3(256), 0, stream, b->d_offsets, b->d_ids, need_a ? b->d_data : nullptr,
                           (uint32_t)b->n_rows, p.width, l4, t_ids, t_val);
        SDRB_HIP(hipGetLastError());
    }
    if (math == kMul) return launch<kMul, true>(b, p, tables, na, out_offsets, out_ids, out_data, t_ids, t_val, l4, stream);
    return launch<kKeepAll, true>(b, p, tables, na, out_offsets, out_ids, out_data, t_ids, t_val, l4, stream);
}
