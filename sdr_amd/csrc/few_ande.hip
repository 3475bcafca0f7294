// ande of a few rows of A (at most 32: one query, or a handful) with every row of a database B, in one call with no
// pass over per-pair offsets in between:  result(i, j) = the kept common ids of A_i and B_j, ascending, with the
// combined payload (SDR.hpp:999-1025; UnitData.hpp:30-51 for the payload rule).
//
// At SDR densities such a result is sparse -- 0.78 elements a pair at 2048/40 -- so the CSR offsets (4 bytes a pair)
// outweigh the elements, and a count -> scan -> fill pipeline over 16.7 M pairs spends more time on the offsets
// (write, scan, read again) than on finding the matches.  Here a lane owns one row of B (ids and payloads in
// registers) and walks it against the 32-row tile tables of A in LDS (lane_tiles.h), as in lane_rows.hip, and
//
//   (1) k_few_ande_count : per pair the number of kept matches (bit-sliced counters in registers) as ONE BYTE (a row
//                          of B holds at most 64 ids),
//                          per (row of A, 256-row group of B) their sum;
//   (2) k_few_ande_scan  : one workgroup scans the group sums (na x groups values) in pair order and leaves the
//                          element total in device memory;
//   (3) k_few_ande_fill  : reads its pairs' bytes, scans them over the lanes of the group (DPP wave scans), writes
//                          the offsets, walks the row again and writes every kept match at its pair's cursor -- a
//                          lane walks its row in ascending id order, so a pair's elements come out in id order.
//
// The caller sizes the element buffers by an estimate or a bound; elements beyond `capacity` are not written (the
// total tells the caller, who calls again with room).  Offsets are always complete.
#include <type_traits>

#include <algorithm>

#include "lane_tiles.h"
#include "match_math.h"
#include "sdrb_internal.h"
#include "vertical_counters.h"

namespace sdrb {

namespace {

constexpr int kWaves = kBlock / 64;

template <int MATH>
constexpr bool need_a() { return MATH == kSelfA || MATH == kMul; }
template <int MATH>
constexpr bool need_b() { return MATH == kSelfB || MATH == kMul; }

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// The lane's row of B: ids as byte offsets of their mask words (rows shorter than 8*NCH are padded with the
// sentinel id, whose mask is zero), payloads, and the row's length.
template <int NCH, bool BV>
__device__ __forceinline__ void load_row(const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids,
                                         const float* __restrict__ b_val, uint32_t nb, uint32_t j0, uint32_t tid,
                                         uint32_t wcap, uint32_t (&id)[8 * NCH], float (&val)[8 * NCH], uint32_t& lb) {
    const uint32_t j = j0 + tid;
    uint32_t beg = 0;
    lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    // bounds-checked buffers that start at the group's first element: loads are unconditional
    const uint32_t blk_beg = b_off[j0];
    const uint32_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
    const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(b_ids) + blk_beg, 0,
                                                                        (int)blk_bytes, 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
#pragma unroll
    for (int k4 = 0; k4 < 2 * NCH; ++k4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
        id[4 * k4 + 0] = v[0];
        id[4 * k4 + 1] = v[1];
        id[4 * k4 + 2] = v[2];
        id[4 * k4 + 3] = v[3];
    }
    if constexpr (BV) {
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_val) + blk_beg, 0,
                                                                            (int)blk_bytes, 0x00020000);
#pragma unroll
        for (int k4 = 0; k4 < 2 * NCH; ++k4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, (int)(rel + 16u * k4), 0, 0);
            val[4 * k4 + 0] = __uint_as_float(v[0]);
            val[4 * k4 + 1] = __uint_as_float(v[1]);
            val[4 * k4 + 2] = __uint_as_float(v[2]);
            val[4 * k4 + 3] = __uint_as_float(v[3]);
        }
    }
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k) {
        const bool live = (uint32_t)k < lb;
        id[k] = (live ? id[k] : wcap) * 4u;
        val[k] = (BV && live) ? val[k] : 0.0f;
    }
}

__device__ __forceinline__ void stage_tile(unsigned char* lds, const unsigned char* __restrict__ tables, uint32_t tile_bytes,
                                           uint32_t tid) {
    const u32x4* src = reinterpret_cast<const u32x4*>(tables);
    for (uint32_t v = tid; v < tile_bytes / 16u; v += kBlock) reinterpret_cast<u32x4*>(lds)[v] = src[v];
}

// Walks the lane's row: `hit(q, id, d)` for every kept match with row q of A, in ascending id order.
template <int MATH, int NCH, typename Hit>
__device__ __forceinline__ void walk(const unsigned char* tab, uint32_t col_at, const uint32_t (&id)[8 * NCH],
                                     const float (&val)[8 * NCH], Hit&& hit) {
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        uint32_t m[8], c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t o = id[ch * 8 + k];
            m[k] = *reinterpret_cast<const uint32_t*>(tab + o);
            c[k] = need_a<MATH>() ? *reinterpret_cast<const uint16_t*>(tab + col_at + (o >> 1)) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t mm = m[k], cc = c[k];
            while (mm != 0) {
                const uint32_t q = (uint32_t)__builtin_ctz(mm);
                mm &= mm - 1;
                float va = 0.0f;
                if constexpr (need_a<MATH>()) {
                    va = *reinterpret_cast<const float*>(tab + cc);
                    cc += 4;
                }
                float d;
                if (match_kept_by<MATH>(va, val[ch * 8 + k], d)) hit(q, id[ch * 8 + k] >> 2, d);
            }
        }
    }
}

// The number of kept matches of the lane's row with each of the tile's 32 rows, as bit-sliced counters: an id's
// kept matches are a word with a bit per row of A -- the mask itself where ids alone decide, else the lowest row tested
// without a branch and further rows in a second step, as in lane_rows.hip -- and eight words at a time go into the
// vertical counter.  Returns the counts as bytes: count of row q = byte q / 8 of word q % 8.
template <int MATH, int NCH>
__device__ __forceinline__ void count_rows(const unsigned char* tab, uint32_t col_at, const uint32_t (&id)[8 * NCH],
                                           const float (&val)[8 * NCH], uint32_t (&count_of)[8]) {
    constexpr int NPL = bits_for(8 * NCH);
    static_assert(NPL <= 8, "counts must fit one byte");
    uint32_t planes[NPL];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        uint32_t m[8], kp[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t o = id[ch * 8 + k];
            m[k] = *reinterpret_cast<const uint32_t*>(tab + o);
            kp[k] = need_a<MATH>() ? *reinterpret_cast<const uint16_t*>(tab + col_at + (o >> 1)) : 0u;
        }
        float d;
        if constexpr (MATH == kKeepAll) {
#pragma unroll
            for (int k = 0; k < 8; ++k) kp[k] = m[k];
        } else if constexpr (!need_a<MATH>()) {
#pragma unroll
            for (int k = 0; k < 8; ++k) kp[k] = match_kept_by<MATH>(0.0f, val[ch * 8 + k], d) ? m[k] : 0u;
        } else {
            uint32_t rest[8], cc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cc[k] = kp[k];
                kp[k] = m[k] & (0u - m[k]);
                rest[k] = m[k] ^ kp[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float a0 = *reinterpret_cast<const float*>(tab + cc[k]);
                kp[k] = match_kept_by<MATH>(a0, val[ch * 8 + k], d) ? kp[k] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t mm = rest[k];
                if (mm != 0) {
                    const float bv = val[ch * 8 + k];
                    uint32_t at = cc[k] + 4, word = kp[k];
                    while (mm != 0) {
                        const uint32_t low = mm & (0u - mm);
                        mm ^= low;
                        word |= match_kept_by<MATH>(*reinterpret_cast<const float*>(tab + at), bv, d) ? low : 0u;
                        at += 4;
                    }
                    kp[k] = word;
                }
            }
        }
        if (ch == 0) add8<NPL, true>(planes, kp);
        else add8<NPL, false>(planes, kp);
    }
    planes_to_bytes<NPL>(planes, count_of);
}

// (1) bytes per pair and sums per (row of A, group of B)
template <int MATH, int NCH>
__global__ __launch_bounds__(kBlock) void k_few_ande_count(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val, uint32_t nb,
    const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t wcap, uint32_t na,
    uint8_t* __restrict__ cnt8, uint32_t* __restrict__ group_total, uint32_t n_groups) {
    // LDS: the tile's tables | (acc[na][256], used by the fill only) | wave_part[na][4]
    extern __shared__ __attribute__((aligned(16))) unsigned char few_lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* wave_part = reinterpret_cast<uint32_t*>(few_lds + tile_bytes) + na * kBlock;
    stage_tile(few_lds, tables, tile_bytes, tid);
    __syncthreads();
    const uint32_t col_at = mask_bytes(wcap);
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kBlock, j = j0 + tid;
        uint32_t id[8 * NCH], lb;
        float val[8 * NCH];
        load_row<NCH, need_b<MATH>()>(b_off, b_ids, b_val, nb, j0, tid, wcap, id, val, lb);
        uint32_t count_of[8];
        count_rows<MATH, NCH>(few_lds, col_at, id, val, count_of);
        // (the loop stays rolled: the word and the byte of row q are picked by shifts of q)
        for (uint32_t q = 0; q < na; ++q) {
            uint32_t word = count_of[0];
#pragma unroll
            for (int r = 1; r < 8; ++r) word = (q & 7u) == (uint32_t)r ? count_of[r] : word;
            const uint32_t c = (word >> (8u * (q >> 3))) & 0xffu;
            if (j < nb) cnt8[(uint64_t)q * nb + j] = (uint8_t)c;
            const uint32_t incl = wave_inclusive_scan(c);
            if (lane == 63) wave_part[q * kWaves + wave] = incl;
        }
        __syncthreads();
        if (tid < na) {
            uint32_t s = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) s += wave_part[tid * kWaves + w];
            group_total[(uint64_t)tid * n_groups + g] = s;
        }
        __syncthreads();
    }
}

// (2) group sums -> where every (row of A, group of B) starts in the result, in pair order (row of A major); the
//     element total; the last offset.  One workgroup per row of A: it adds up the rows before its own (a few thousand
//     values each) and scans its own.  The starts go to an array of their own: the workgroups of later rows read the
//     sums of earlier rows while those rows' workgroups are at work, in no order the launch promises.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void k_few_ande_scan(const uint32_t* __restrict__ group_total,
                                                               uint32_t* __restrict__ group_base, uint32_t na,
                                                               uint32_t n_groups,
                                                               unsigned long long* __restrict__ total_out,
                                                               uint32_t* __restrict__ last_offset) {
    __shared__ unsigned long long part64[kScanThreads / 64];
    __shared__ uint32_t part32[kScanThreads / 64];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // elements of the rows of A before this one
    // (eight loads in flight per thread: one at a time, the 60 round trips of the last row's workgroup were the
    // kernel's whole duration)
    unsigned long long before = 0;
    const uint64_t n_before = (uint64_t)q * n_groups;
    uint64_t k = tid;
    for (; k + 7ull * kScanThreads < n_before; k += 8ull * kScanThreads) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = group_total[k + (uint64_t)u * kScanThreads];
#pragma unroll
        for (int u = 0; u < 8; ++u) before += v[u];
    }
    for (; k < n_before; k += kScanThreads) before += group_total[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) before += (unsigned long long)__shfl_xor((long long)before, d, 64);
    if (lane == 0) part64[wave] = before;
    __syncthreads();
    before = 0;
    for (int w = 0; w < kScanThreads / 64; ++w) before += part64[w];
    // this row: every thread owns a run of consecutive groups
    const uint32_t* row = group_total + (uint64_t)q * n_groups;
    uint32_t* row_base = group_base + (uint64_t)q * n_groups;
    const uint32_t per = (n_groups + kScanThreads - 1) / kScanThreads;
    const uint32_t g0 = tid * per < n_groups ? tid * per : n_groups, g1 = g0 + per < n_groups ? g0 + per : n_groups;
    uint32_t mine = 0;
    for (uint32_t g = g0; g < g1; ++g) mine += row[g];
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) part32[wave] = incl;
    __syncthreads();
    uint32_t run = (uint32_t)before + incl - mine;   // offsets are kept modulo 2^32; the 64-bit total says whether they are valid
    uint32_t row_sum = 0;
    for (uint32_t w = 0; w < kScanThreads / 64; ++w) {
        if (w < wave) run += part32[w];
        row_sum += part32[w];
    }
    for (uint32_t g = g0; g < g1; ++g) {
        row_base[g] = run;
        run += row[g];
    }
    if (q + 1 == na && tid == 0) {
        *total_out = before + row_sum;
        *last_offset = (uint32_t)(before + row_sum);
    }
}

// (3) offsets and elements.  (Collecting a group's matches in LDS and writing them out as coalesced runs was tried:
//     the 16 KB stage costs a resident block per compute unit and two more barriers a group, 151 against 118 us.)
template <int MATH, int NCH>
__global__ __launch_bounds__(kBlock) void k_few_ande_fill(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val, uint32_t nb,
    const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t wcap, uint32_t na,
    const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ group_base, uint32_t n_groups,
    uint32_t* __restrict__ out_offsets, int32_t* __restrict__ out_ids, float* __restrict__ out_data, uint64_t capacity) {
    // LDS: the tile's tables | acc[na][256] | wave_part[na][4]
    extern __shared__ __attribute__((aligned(16))) unsigned char few_lds[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* acc_lane = reinterpret_cast<uint32_t*>(few_lds + tile_bytes) + tid;
    uint32_t* wave_part = reinterpret_cast<uint32_t*>(few_lds + tile_bytes) + na * kBlock;
    stage_tile(few_lds, tables, tile_bytes, tid);
    __syncthreads();
    const uint32_t col_at = mask_bytes(wcap);
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kBlock, j = j0 + tid;
        uint32_t id[8 * NCH], lb;
        float val[8 * NCH];
        load_row<NCH, need_b<MATH>()>(b_off, b_ids, b_val, nb, j0, tid, wcap, id, val, lb);
        // this lane's pairs: exclusive prefix of their sizes over the wave, the waves' sums for the block
        for (uint32_t q = 0; q < na; ++q) {
            const uint32_t c = j < nb ? (uint32_t)cnt8[(uint64_t)q * nb + j] : 0u;
            const uint32_t incl = wave_inclusive_scan(c);
            acc_lane[q * kBlock] = incl - c;
            if (lane == 63) wave_part[q * kWaves + wave] = incl;
        }
        __syncthreads();
        for (uint32_t q = 0; q < na; ++q) {
            uint32_t base = group_base[(uint64_t)q * n_groups + g];
            for (uint32_t w = 0; w < wave; ++w) base += wave_part[q * kWaves + w];
            const uint32_t pos = base + acc_lane[q * kBlock];
            acc_lane[q * kBlock] = pos;
            if (j < nb) out_offsets[(uint64_t)q * nb + j] = pos;
        }
        walk<MATH, NCH>(few_lds, col_at, id, val, [&](uint32_t q, uint32_t w, float d) {
            uint32_t* slot = acc_lane + q * kBlock;
            const uint32_t p = *slot;
            *slot = p + 1u;
            if ((uint64_t)p < capacity) {
                out_ids[p] = (int32_t)w;
                if constexpr (MATH != kKeepAll) out_data[p] = d;
            }
        });
        __syncthreads();   // wave_part is rewritten by the next group
    }
}


// ---- at most eight rows of A of at most 64 ids whose kept matches do not depend on B's payloads (`this` EmptyData: every
// common id; `this` = A UnitData against EmptyData rows: the common ids whose own payload is relevant) ----
//
// A pair's result is then a subset of A_i's elements, and 64 bits say which: the count pass leaves that POSITION MASK
// beside the byte count (8 bytes a pair), and the fill expands masks against A_i instead of walking B a second time
// (164 bytes a row at 2048 / 40).  Every block builds its own table -- eight bytes an id: for each row of A whether it
// holds the id, and where -- from A's few hundred ids.
constexpr int kMaskRows = 8;

// ROWS: 1, 2, 4 or 8 -- the rows of A the lane loop is unrolled for (na rounded up)
template <int NCH, bool SELF_A, int ROWS>
__global__ __launch_bounds__(kBlock) void k_few_masks_count(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_val, uint32_t na,
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint32_t nb, uint32_t wcap,
    unsigned long long* __restrict__ pos_mask, uint8_t* __restrict__ cnt8, uint32_t* __restrict__ group_total, uint32_t n_groups) {
    // LDS: where[wcap + 1] 64-bit words, byte q of word w: 0x80 | position of id w in row q of A, 0 where the row does not
    //      hold it (the word at [wcap], the sentinel id's, stays zero) | wave_part[kMaskRows][4]
    // One 8-byte read an id answers all eight rows; per row the lane shifts the byte's top bit (there or not) by the
    // byte's low six bits (where) into the pair's mask -- no branch, no loop over matches, every lane the same work.
    extern __shared__ __attribute__((aligned(16))) unsigned char few_lds[];
    unsigned long long* where = reinterpret_cast<unsigned long long*>(few_lds);
    uint32_t* wave_part = reinterpret_cast<uint32_t*>(few_lds + ((size_t)wcap + 2) * 8);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t w = tid; w < wcap + 1; w += kBlock) where[w] = 0;
    __syncthreads();
    unsigned long long relevant[ROWS];   // positions of row q whose own payload is relevant (SELF_A), else all
#pragma unroll
    for (int q = 0; q < ROWS; ++q) relevant[q] = ~0ull;
#pragma unroll
    for (int q = 0; q < ROWS; ++q) {
        if ((uint32_t)q < na) {
            const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
            if (tid < len) {
                const uint32_t id = (uint32_t)a_ids[beg + tid];
                if (id < wcap) few_lds[(size_t)id * 8 + q] = (unsigned char)(0x80u | tid);   // ids outside B's table cannot match
            }
            if constexpr (SELF_A) {
                // a match keeps `this`'s payload (UnitData.hpp:43-46) and stays iff that is relevant (SDR.hpp:1017)
                const bool ok = lane < len && unit_relevant(a_val[beg + lane]);
                relevant[q] = __builtin_amdgcn_ballot_w64(ok);
            }
        }
    }
    __syncthreads();
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kBlock, j = j0 + tid;
        uint32_t id[8 * NCH], lb;
        float unused[8 * NCH];
        load_row<NCH, false>(b_off, b_ids, nullptr, nb, j0, tid, wcap, id, unused, lb);
        unsigned long long pm[ROWS];
#pragma unroll
        for (int q = 0; q < ROWS; ++q) pm[q] = 0;
#pragma unroll
        for (int k = 0; k < 8 * NCH; ++k) {
            const unsigned long long w = *reinterpret_cast<const unsigned long long*>(few_lds + id[k] * 2u);   // id[k] = 4 * id
#pragma unroll
            for (int q = 0; q < ROWS; ++q) {
                const uint32_t half = q < 4 ? (uint32_t)w : (uint32_t)(w >> 32);
                const uint32_t there = (half >> (8 * (q & 3) + 7)) & 1u, at = (half >> (8 * (q & 3))) & 63u;
                pm[q] |= (unsigned long long)there << at;
            }
        }
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if ((uint32_t)q < na) {   // uniform
                const unsigned long long mine = pm[q] & relevant[q];
                const uint32_t c = (uint32_t)__builtin_popcountll(mine);
                if (j < nb) {
                    pos_mask[(uint64_t)q * nb + j] = mine;
                    cnt8[(uint64_t)q * nb + j] = (uint8_t)c;
                }
                const uint32_t incl = wave_inclusive_scan(c);
                if (lane == 63) wave_part[q * kWaves + wave] = incl;
            }
        }
        __syncthreads();
        if (tid < na) {
            uint32_t sum = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) sum += wave_part[tid * kWaves + w];
            group_total[(uint64_t)tid * n_groups + g] = sum;
        }
        __syncthreads();
    }
}

// the fill of the same path: offsets from the bytes, elements from the position masks and A's rows (in LDS)
template <bool DATA>
__global__ __launch_bounds__(kBlock) void k_few_masks_fill(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_val, uint32_t na, uint32_t nb,
    const unsigned long long* __restrict__ pos_mask, const uint8_t* __restrict__ cnt8, const uint32_t* __restrict__ group_base,
    uint32_t n_groups, uint32_t* __restrict__ out_offsets, int32_t* __restrict__ out_ids, float* __restrict__ out_data,
    uint64_t capacity) {
    __shared__ int32_t row_ids[kMaskRows][64];
    __shared__ float row_val[kMaskRows][64];
    __shared__ uint32_t wave_part[kMaskRows][kWaves];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t e = tid; e < na * 64u; e += kBlock) {
        const uint32_t q = e >> 6, k = e & 63u, beg = a_off[q], len = a_off[q + 1] - beg;
        row_ids[q][k] = k < len ? a_ids[beg + k] : 0;
        if constexpr (DATA) row_val[q][k] = k < len ? a_val[beg + k] : 0.0f;
    }
    __syncthreads();
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j = g * kBlock + tid;
        uint32_t excl[kMaskRows];
        for (uint32_t q = 0; q < na; ++q) {
            const uint32_t c = j < nb ? (uint32_t)cnt8[(uint64_t)q * nb + j] : 0u;
            const uint32_t incl = wave_inclusive_scan(c);
#pragma unroll
            for (int r = 0; r < kMaskRows; ++r) excl[r] = q == (uint32_t)r ? incl - c : excl[r];
            if (lane == 63) wave_part[q][wave] = incl;
        }
        __syncthreads();
        for (uint32_t q = 0; q < na; ++q) {
            uint32_t p = group_base[(uint64_t)q * n_groups + g];
            for (uint32_t w = 0; w < wave; ++w) p += wave_part[q][w];
            uint32_t mine = excl[0];
#pragma unroll
            for (int r = 1; r < kMaskRows; ++r) mine = q == (uint32_t)r ? excl[r] : mine;
            p += mine;
            if (j < nb) {
                out_offsets[(uint64_t)q * nb + j] = p;
                unsigned long long m = pos_mask[(uint64_t)q * nb + j];
                while (m != 0) {
                    const uint32_t k = (uint32_t)__builtin_ctzll(m);
                    m &= m - 1;
                    if ((uint64_t)p < capacity) {
                        out_ids[p] = row_ids[q][k];
                        if constexpr (DATA) out_data[p] = row_val[q][k];
                    }
                    ++p;
                }
            }
        }
        __syncthreads();   // wave_part is rewritten by the next group
    }
}

struct FewPlan {
    bool masks;    // the position-mask path (few rows of A, kept matches independent of B's payloads)
    int wcap, nch, math;
    uint32_t tile_bytes, ntiles, n_groups, grid;
    size_t lds;
};

bool make_plan(const sdrb_batch* a, const sdrb_batch* b, bool swap, FewPlan* p) {
    if (a->id_bits != 32 || b->id_bits != 32 || a->n_rows == 0 || a->n_rows > (uint64_t)kTile || b->n_rows == 0) return false;
    if (b->n_rows > 0xFFFFFF00ull) return false;
    if (b->width <= 0 || b->width > 4096 || b->max_row_len > 64 || a->max_row_len > 1024) return false;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) return false;
    p->math = pick_math(a, b, SDRB_OP_AND, swap);
    p->wcap = b->width <= 2048 ? 2048 : 4096;
    p->nch = b->max_row_len <= 40 ? 5 : 8;
    p->masks = a->n_rows <= (uint64_t)kMaskRows && a->max_row_len <= 64 && (p->math == kKeepAll || p->math == kSelfA);
    if (p->masks) {
        p->tile_bytes = 0;
        p->ntiles = 1;
        p->lds = ((size_t)p->wcap + 2) * 8 + (size_t)kMaskRows * kWaves * 4u;
        p->n_groups = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
        const uint32_t per_cu = (uint32_t)((160u * 1024u) / (p->lds + 512));
        const uint32_t resident = 256u * (per_cu < 8u ? per_cu : 8u);
        p->grid = p->n_groups < resident ? p->n_groups : resident;
        return true;
    }
    const bool with_vals = p->math == kSelfA || p->math == kMul;
    const uint32_t val_cap = with_vals ? ((uint32_t)kTile * (a->max_row_len ? a->max_row_len : 1u) + 3u) & ~3u : 0u;
    // col[w] addresses a payload by a 16-bit byte offset into the tile: every payload the tile can hold must lie below 64 KiB
    const uint64_t most_vals = std::min<uint64_t>(a->nnz, (uint64_t)kTile * a->max_row_len);
    if (with_vals && mask_bytes((uint32_t)p->wcap) + col_bytes((uint32_t)p->wcap) + 4u * most_vals > 65535u) return false;
    p->tile_bytes = mask_bytes((uint32_t)p->wcap) + col_bytes((uint32_t)p->wcap) + val_cap * 4u + kLenBytes;
    p->lds = (size_t)p->tile_bytes + (size_t)a->n_rows * kBlock * 4u + (size_t)kTile * kWaves * 4u;
    if (p->lds > 160 * 1024 - 512) return false;
    p->n_groups = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    // blocks stay resident and stride over the groups, so the tables are staged once per block
    const uint32_t per_cu = (uint32_t)((160u * 1024u) / (p->lds + 512));
    const uint32_t resident = 256u * (per_cu < 8u ? per_cu : 8u);
    p->grid = p->n_groups < resident ? p->n_groups : resident;
    return true;
}

template <int MATH, int NCH>
int launch_all(const sdrb_batch* a, const sdrb_batch* b, const FewPlan& p, const unsigned char* tables, uint8_t* cnt8,
               uint32_t* group_total, uint32_t* out_offsets, int32_t* out_ids, float* out_data, uint64_t capacity,
               unsigned long long* total_dev, hipStream_t stream) {
    const uint32_t na = (uint32_t)a->n_rows, nb = (uint32_t)b->n_rows;
    uint32_t* group_base = group_total + (size_t)na * p.n_groups;
    if (p.lds > 64 * 1024) {   // dynamic LDS beyond 64 KiB has to be granted per kernel
        SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_few_ande_count<MATH, NCH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
        SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_few_ande_fill<MATH, NCH>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    }
    hipLaunchKernelGGL((k_few_ande_count<MATH, NCH>), dim3(p.grid), dim3(kBlock), p.lds, stream, b->d_offsets, b->d_ids,
                       b->d_data, nb, tables, p.tile_bytes, (uint32_t)p.wcap, na, cnt8, group_total, p.n_groups);
    SDRB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_few_ande_scan, dim3(na), dim3(kScanThreads), 0, stream, group_total, group_base, na, p.n_groups,
                       total_dev, out_offsets + (uint64_t)na * nb);
    SDRB_HIP(hipGetLastError());
    hipLaunchKernelGGL((k_few_ande_fill<MATH, NCH>), dim3(p.grid), dim3(kBlock), p.lds, stream, b->d_offsets, b->d_ids,
                       b->d_data, nb, tables, p.tile_bytes, (uint32_t)p.wcap, na, cnt8, group_base, p.n_groups, out_offsets,
                       out_ids, out_data, capacity);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <int MATH>
int launch_nch(const sdrb_batch* a, const sdrb_batch* b, const FewPlan& p, const unsigned char* tables, uint8_t* cnt8,
               uint32_t* group_total, uint32_t* out_offsets, int32_t* out_ids, float* out_data, uint64_t capacity,
               unsigned long long* total_dev, hipStream_t stream) {
    if (p.nch == 5)
        return launch_all<MATH, 5>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
    return launch_all<MATH, 8>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
}

}  // namespace

bool few_ande_applicable(const sdrb_batch* a, const sdrb_batch* b, bool swap) {
    FewPlan p;
    return make_plan(a, b, swap, &p);
}

// out_offsets[na*nb + 1], out_ids / out_data[capacity] (out_data only when `this` carries payloads), *total_dev: the
// element total (device memory).  Elements at positions >= capacity are not written.
int launch_few_ande(const sdrb_batch* a, const sdrb_batch* b, bool swap, uint32_t* out_offsets, int32_t* out_ids,
                    float* out_data, uint64_t capacity, unsigned long long* total_dev, hipStream_t stream) {
    FewPlan p;
    if (!make_plan(a, b, swap, &p)) {
        set_error("the fused few-rows ande serves at most 32 rows of a, a declared width <= 4096 and rows of b <= 64 ids");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (p.masks) {
        const uint32_t na = (uint32_t)a->n_rows, nb = (uint32_t)b->n_rows;
        const size_t pairs = (size_t)na * nb;
        const size_t mask_bytes_all = pairs * sizeof(unsigned long long), cnt_bytes = (pairs + 255) & ~(size_t)255;
        char* ws = static_cast<char*>(workspace(b->device, stream, 12, mask_bytes_all + cnt_bytes + 2 * (size_t)na * p.n_groups * sizeof(uint32_t)));
        if (!ws) return SDRB_ERR_HIP;
        unsigned long long* pos_mask = reinterpret_cast<unsigned long long*>(ws);
        uint8_t* cnt8 = reinterpret_cast<uint8_t*>(ws + mask_bytes_all);
        uint32_t* group_total = reinterpret_cast<uint32_t*>(ws + mask_bytes_all + cnt_bytes);
        uint32_t* group_base = group_total + (size_t)na * p.n_groups;
        const bool self_a = p.math == kSelfA;
#define SDRB_MASKS_COUNT_R(NCH, SELF, ROWS)                                                                             \
    do {                                                                                                                 \
        if (p.lds > 64 * 1024)                                                                                           \
            SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_few_masks_count<NCH, SELF, ROWS>),             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));                       \
        hipLaunchKernelGGL((k_few_masks_count<NCH, SELF, ROWS>), dim3(p.grid), dim3(kBlock), p.lds, stream, a->d_offsets, a->d_ids, \
                           a->d_data, na, b->d_offsets, b->d_ids, nb, (uint32_t)p.wcap, pos_mask, cnt8, group_total, p.n_groups); \
    } while (0)
#define SDRB_MASKS_COUNT(NCH, SELF)                    \
    do {                                               \
        if (na <= 1) SDRB_MASKS_COUNT_R(NCH, SELF, 1); \
        else if (na <= 2) SDRB_MASKS_COUNT_R(NCH, SELF, 2); \
        else if (na <= 4) SDRB_MASKS_COUNT_R(NCH, SELF, 4); \
        else SDRB_MASKS_COUNT_R(NCH, SELF, 8);         \
    } while (0)
        if (p.nch == 5) {
            if (self_a) SDRB_MASKS_COUNT(5, true);
            else SDRB_MASKS_COUNT(5, false);
        } else {
            if (self_a) SDRB_MASKS_COUNT(8, true);
            else SDRB_MASKS_COUNT(8, false);
        }
#undef SDRB_MASKS_COUNT
#undef SDRB_MASKS_COUNT_R
        SDRB_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_few_ande_scan, dim3(na), dim3(kScanThreads), 0, stream, group_total, group_base, na, p.n_groups,
                           total_dev, out_offsets + (uint64_t)na * nb);
        SDRB_HIP(hipGetLastError());
        if (self_a)
            hipLaunchKernelGGL((k_few_masks_fill<true>), dim3(p.grid), dim3(kBlock), 0, stream, a->d_offsets, a->d_ids, a->d_data, na,
                               nb, pos_mask, cnt8, group_base, p.n_groups, out_offsets, out_ids, out_data, capacity);
        else
            hipLaunchKernelGGL((k_few_masks_fill<false>), dim3(p.grid), dim3(kBlock), 0, stream, a->d_offsets, a->d_ids, a->d_data, na,
                               nb, pos_mask, cnt8, group_base, p.n_groups, out_offsets, out_ids, out_data, capacity);
        SDRB_HIP(hipGetLastError());
        return SDRB_OK;
    }
    const bool with_vals = p.math == kSelfA || p.math == kMul;
    const unsigned char* tables = nullptr;
    uint32_t tile_bytes = 0, ntiles = 0;
    int rc = build_lane_tiles(a, b->device, p.wcap, with_vals, stream, &tile_bytes, &ntiles, &tables);
    if (rc != SDRB_OK) return rc;
    const size_t cnt_bytes = ((size_t)a->n_rows * b->n_rows + 255) & ~(size_t)255;
    char* ws = static_cast<char*>(workspace(b->device, stream, 12, cnt_bytes + 2 * (size_t)a->n_rows * p.n_groups * sizeof(uint32_t)));
    if (!ws) return SDRB_ERR_HIP;
    uint8_t* cnt8 = reinterpret_cast<uint8_t*>(ws);
    uint32_t* group_total = reinterpret_cast<uint32_t*>(ws + cnt_bytes);
    switch (p.math) {
    case kKeepAll:
        return launch_nch<kKeepAll>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
    case kSelfA:
        return launch_nch<kSelfA>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
    case kSelfB:
        return launch_nch<kSelfB>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
    default:
        return launch_nch<kMul>(a, b, p, tables, cnt8, group_total, out_offsets, out_ids, out_data, capacity, total_dev, stream);
    }
}

}  // namespace sdrb
