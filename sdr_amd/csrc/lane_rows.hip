// Lane-per-row match kernels: a lane owns a row of B and meets the rows of A through per-tile tables, for
//   * sparse dot products, out(i, j) = A_i.inner(B_j) (SDR.hpp:1682-1692 with ArithData.hpp:37-43,59-63:
//     over the common ids in ascending order, acc = acc + (a * b), a rounded product and then a rounded sum):
//     k_lane_rows_inner, every id match visited once, in id order, and
//   * the size operations whose result depends on payloads (SDR.hpp:1083-1095, 1428-1433, 1572-1576 with
//     UnitData.hpp:22-28,48-51,94-97,117-120: a match counts when the combined payload is relevant):
//     k_lane_rows_size, counts in bit-sliced counters.
//
// At SDR densities nearly every (row of A, id of B_j) candidate is a miss -- 40 of 2048 ids are set -- so a dense
// table of A's values multiplies by zero 98 % of the time, and a wave that deals B's elements to its lanes spends
// its time in LDS atomics.  Here the rows of A are cut into tiles of 32; a tile is described by
//
//     mask[w]  (uint32)  bit q set  <=>  row 32 t + q of A contains id w
//     col[w]   (uint16)  where the tile's payloads of id w start (as a byte offset into the tile)
//     vals[]   (float)   the tile's payloads, column-compressed: by id, then by ascending row
//     len[q]   (uint32)  length of row 32 t + q
//
// built once per call in global memory (17 KB a tile for width 2048, 40 ids a row) and staged in LDS.
// The dot products: a lane owns one row of B, ids and payloads in registers, and walks it in ascending id order: an id
// costs two LDS reads (mask, col), and every set bit of its mask is one match -- one payload of A from
// LDS and one update of the lane's own accumulator of that row of A (acc[q][lane] in LDS: the lanes of a
// wave hit 64 different banks, and nothing is shared, so no atomics).  A pair's matches are visited in
// ascending id order, one after the other, as the reference does -- which is what makes the float sums
// bit-exact.  After a tile the lane stores its 32 results, one coalesced 256-byte segment per wave per
// row of A, and clears the accumulators.  The next tile's tables travel from L2 while a tile is walked.
#include "lane_tiles.h"
#include "match_math.h"
#include "vertical_counters.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

// NCH  : the lane keeps 8*NCH (id, payload) pairs of its row in registers; shorter rows are padded with the
//        sentinel id, whose mask is zero
// WCAP : table entries per tile
// BV   : B carries payloads (an EmptyData argument leaves the value of `this` unchanged, ArithData.hpp:39)
// PF   : 16-byte pieces of a tile a thread stages
template <int NCH, int WCAP, bool BV, int PF>
__global__ __launch_bounds__(kBlock, 3) void k_lane_rows_inner(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint64_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t ntiles,
    uint32_t tiles_per_block, uint32_t na, uint32_t* __restrict__ out, uint64_t ld_out) {
    constexpr uint32_t COL_AT = mask_bytes(WCAP);
    // LDS: the tile's tables (tile_bytes) | acc[32][256]
    extern __shared__ __attribute__((aligned(16))) unsigned char lane_lds[];
    unsigned char* tab = lane_lds;
    const uint32_t tid = threadIdx.x;
    uint32_t* acc_lane = reinterpret_cast<uint32_t*>(lane_lds + tile_bytes) + tid;

    const uint64_t j0 = (uint64_t)blockIdx.x * kBlock;
    const uint64_t j = j0 + tid;
    uint32_t beg = 0, lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    // bounds-checked buffers that start at the block's first element: loads are unconditional
    const uint32_t blk_beg = b_off[j0];
    const uint64_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
    const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(b_ids) + blk_beg, 0,
                                                                        (int)blk_bytes, 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
    uint32_t id[8 * NCH];
    float val[8 * NCH];
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
        id[k] = (live ? id[k] : (uint32_t)WCAP) * 4u;     // byte offset of the id's mask word
        val[k] = live ? (BV ? val[k] : 1.0f) : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kTile; ++q) acc_lane[q * kBlock] = 0u;

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    const uint32_t nvec = tile_bytes / 16u;
    const uint64_t row_bytes = ld_out * 4;
    const uint64_t rows_here = nb - j0 < (uint64_t)kBlock ? nb - j0 : (uint64_t)kBlock;
    const uint32_t nrec = (uint32_t)rows_here * 4;
    char* const out_block = reinterpret_cast<char*>(out) + j0 * 4;

    u32x4 piece[PF];
    if (t0 < t1) {
        const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)t0 * tile_bytes);
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (tid + p * kBlock < nvec) piece[p] = src[tid + p * kBlock];
    }
    for (uint32_t t = t0; t < t1; ++t) {
        __syncthreads();   // every wave is done with the previous tile's tables
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (tid + p * kBlock < nvec) reinterpret_cast<u32x4*>(tab)[tid + p * kBlock] = piece[p];
        __syncthreads();
        if (t + 1 < t1) {   // the next tile travels while this one is walked
            const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)(t + 1) * tile_bytes);
#pragma unroll
            for (int p = 0; p < PF; ++p)
                if (tid + p * kBlock < nvec) piece[p] = src[tid + p * kBlock];
        }

#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            uint32_t m[8], c[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                // the col address is derived here, per tile, from the one offset a lane keeps per id: hoisted out
                // of the tile loop it would cost a second register per id
                uint32_t o = id[ch * 8 + k];
                asm volatile("" : "+v"(o));
                m[k] = *reinterpret_cast<const uint32_t*>(tab + o);
                c[k] = *reinterpret_cast<const uint16_t*>(tab + COL_AT + (o >> 1));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t mm = m[k], cc = c[k];
                const float bv = val[ch * 8 + k];
                while (mm != 0) {
                    const uint32_t q = (uint32_t)__builtin_ctz(mm);
                    mm &= mm - 1;
                    const float a = *reinterpret_cast<const float*>(tab + cc);
                    cc += 4;
                    // acc = acc + a * b, the product and the sum rounded separately
                    uint32_t* slot = acc_lane + q * kBlock;
                    const float prod = a * bv;
                    *slot = __float_as_uint(__uint_as_float(*slot) + prod);
                }
            }
        }

        // the lane's results of this tile: store in row order of A and clear the accumulators
        char* row_ptr = out_block + (uint64_t)t * kTile * row_bytes;
        const uint32_t rows_left = na - t * kTile;
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            const uint32_t v = acc_lane[q * kBlock];
            acc_lane[q * kBlock] = 0u;
            const uint32_t n = (uint32_t)q < rows_left ? nrec : 0u;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)n, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)(tid * 4u), 0, 0);
            row_ptr += row_bytes;
        }
    }
}

// The size operations' own kernel.  Counts need no order, so nothing is accumulated per match: an id of the lane's
// row yields words with a bit per row of the tile -- the rows that hold the id (the mask itself) and the rows whose
// match is kept, or dropped -- and the words of eight ids at a time are added into one vertical counter held in
// registers (plane p = bit p of all 32 counts).  The lowest row that holds an id is tested without a branch, for
// eight ids at once (a lane whose id no row holds tests against an empty bit, which changes nothing); further rows
// -- at 40 of 2048 ids set, one id in eight has them -- take a second step of two rows and, rarely, a bit loop.
// What is counted, and what leaves the kernel (dropped = matches - kept):
//   kCountKept     ands = kept                                      (SDR.hpp:1083-1095)
//   kCountDropped  rms  = |this| - dropped                          (SDR.hpp:1572-1576)
//   kCountBoth     xors = |this| + |arg| - (matches + dropped)      (SDR.hpp:1428-1433): the masks counted in a pass of their own
enum Count { kCountKept, kCountDropped, kCountBoth };
constexpr uint32_t lane_tile_cap(uint32_t wcap) {   // the largest tile: rows of A of kMaxRowA ids
    return mask_bytes(wcap) + col_bytes(wcap) + (uint32_t)kTile * kMaxRowA * 4u + kLenBytes;
}
template <int MATH, int COUNT, int NCH, int WCAP, int PF>
__global__ __launch_bounds__(kBlock, 3) void k_lane_rows_size(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint64_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t ntiles,
    uint32_t tiles_per_block, uint32_t na, uint32_t la_coef, uint32_t lb_coef,
    uint32_t* __restrict__ out, uint64_t ld_out) {
    constexpr uint32_t COL_AT = mask_bytes(WCAP);
    constexpr bool kNeedA = MATH != kSelfB, kNeedB = MATH != kSelfA;
    constexpr int NPL = bits_for(8 * NCH);
    static_assert(NPL <= 8, "counts must fit one byte during unpacking");
    // a block of its own size (not the launch's): addresses into it are compile-time offsets
    __shared__ __attribute__((aligned(16))) unsigned char tab[lane_tile_cap(WCAP)];
    const uint32_t tid = threadIdx.x;

    const uint64_t j0 = (uint64_t)blockIdx.x * kBlock;
    const uint64_t j = j0 + tid;
    uint32_t beg = 0, lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    const uint32_t blk_beg = b_off[j0];
    const uint64_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
    const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t*>(b_ids) + blk_beg, 0,
                                                                        (int)blk_bytes, 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
    uint32_t id[8 * NCH];
    float val[8 * NCH];
#pragma unroll
    for (int k4 = 0; k4 < 2 * NCH; ++k4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
        id[4 * k4 + 0] = v[0];
        id[4 * k4 + 1] = v[1];
        id[4 * k4 + 2] = v[2];
        id[4 * k4 + 3] = v[3];
    }
    if constexpr (kNeedB) {
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
    // two ids a register, as the byte offsets of their mask words (below 2^16): the lane's row, its payloads and
    // the tile in flight would not fit the registers of three waves a SIMD otherwise
    uint32_t idp[4 * NCH];
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k) {
        const bool live = (uint32_t)k < lb;
        const uint32_t o = (live ? id[k] : (uint32_t)WCAP) * 4u;
        if (k & 1) idp[k >> 1] |= o << 16;
        else idp[k >> 1] = o;
        val[k] = kNeedB ? val[k] : 0.0f;
    }
    const uint32_t lb_term = lb * lb_coef;

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    const uint32_t nvec = tile_bytes / 16u;
    const uint32_t len_at = tile_bytes - kLenBytes;
    const uint64_t row_bytes = ld_out * 4;
    const uint64_t rows_here = nb - j0 < (uint64_t)kBlock ? nb - j0 : (uint64_t)kBlock;
    const uint32_t nrec = (uint32_t)rows_here * 4;
    char* const out_block = reinterpret_cast<char*>(out) + j0 * 4;

    u32x4 piece[PF];
    if (t0 < t1) {
        const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)t0 * tile_bytes);
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (tid + p * kBlock < nvec) piece[p] = src[tid + p * kBlock];
    }
    for (uint32_t t = t0; t < t1; ++t) {
        __syncthreads();   // every wave is done with the previous tile's tables
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (tid + p * kBlock < nvec) reinterpret_cast<u32x4*>(tab)[tid + p * kBlock] = piece[p];
        __syncthreads();
        if (t + 1 < t1) {   // the next tile travels while this one is walked
            const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)(t + 1) * tile_bytes);
#pragma unroll
            for (int p = 0; p < PF; ++p)
                if (tid + p * kBlock < nvec) piece[p] = src[tid + p * kBlock];
        }

        uint32_t planes[NPL];
        uint32_t match_of[8];   // xors only: the id matches of the 32 rows, four byte-wide counts a word
        if constexpr (COUNT == kCountBoth) {
            // the masks themselves are counted first, in a pass of their own, and unpacked; the fence keeps the compiler
            // from carrying all forty masks over to the payload pass in registers
            uint32_t mask_planes[NPL];
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                uint32_t m[8];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    uint32_t pair = idp[ch * 4 + k2];
                    asm volatile("" : "+v"(pair));
                    m[2 * k2] = *reinterpret_cast<const uint32_t*>(tab + (pair & 0xffffu));
                    m[2 * k2 + 1] = *reinterpret_cast<const uint32_t*>(tab + (pair >> 16));
                }
                if (ch == 0) add8<NPL, true>(mask_planes, m);
                else add8<NPL, false>(mask_planes, m);
            }
            planes_to_bytes<NPL>(mask_planes, match_of);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            uint32_t m[8], kp[8];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                uint32_t pair = idp[ch * 4 + k2];
                asm volatile("" : "+v"(pair));   // keeps the unpacked offsets out of the tile loop's live registers
                const uint32_t o0 = pair & 0xffffu, o1 = pair >> 16;
                m[2 * k2] = *reinterpret_cast<const uint32_t*>(tab + o0);
                m[2 * k2 + 1] = *reinterpret_cast<const uint32_t*>(tab + o1);
                kp[2 * k2] = kNeedA ? *reinterpret_cast<const uint16_t*>(tab + COL_AT + (o0 >> 1)) : 0u;
                kp[2 * k2 + 1] = kNeedA ? *reinterpret_cast<const uint16_t*>(tab + COL_AT + (o1 >> 1)) : 0u;
            }
            constexpr bool kWantKept = COUNT == kCountKept;   // the words collect kept matches, else dropped ones
            float d;
            if constexpr (!kNeedA) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    kp[k] = match_kept_by<MATH>(0.0f, val[ch * 8 + k], d) == kWantKept ? m[k] : 0u;
            } else {
                // first the lowest row of every id, for all eight ids at once and without a branch ...
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
                    kp[k] = match_kept_by<MATH>(a0, val[ch * 8 + k], d) == kWantKept ? kp[k] : 0u;
                }
                // ... then the ids that more rows of the tile hold (one in eight): two more rows in one step, the
                // loads kept apart (a column starts at any multiple of four bytes), and a loop for what is left
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    uint32_t mm = rest[k];
                    if (mm != 0) {
                        const float bv = val[ch * 8 + k];
                        uint32_t at = cc[k], word = kp[k];
                        uint32_t at2 = at + 8;
                        asm volatile("" : "+v"(at2));   // two loads of four bytes, not one of eight at an odd address
                        const float a1 = *reinterpret_cast<const float*>(tab + at + 4);
                        const float a2 = *reinterpret_cast<const float*>(tab + at2);
                        uint32_t low = mm & (0u - mm);
                        mm ^= low;
                        word |= match_kept_by<MATH>(a1, bv, d) == kWantKept ? low : 0u;
                        low = mm & (0u - mm);
                        mm ^= low;
                        word |= match_kept_by<MATH>(a2, bv, d) == kWantKept ? low : 0u;
                        at += 12;
                        while (mm != 0) {
                            low = mm & (0u - mm);
                            mm ^= low;
                            word |= match_kept_by<MATH>(*reinterpret_cast<const float*>(tab + at), bv, d) == kWantKept ? low : 0u;
                            at += 4;
                        }
                        kp[k] = word;
                    }
                }
            }
            if (ch == 0) add8<NPL, true>(planes, kp);
            else add8<NPL, false>(planes, kp);
        }

        // unpack: bits r, r+8, r+16, r+24 of plane p -> bit p of bytes 0..3 of word r; then one coalesced
        // 256-byte segment per wave per row of A, in row order
        uint32_t count_of[8];
        planes_to_bytes<NPL>(planes, count_of);
        if constexpr (COUNT == kCountBoth) {   // four byte-wide sums at once: both counts stay below 128
#pragma unroll
            for (int r = 0; r < 8; ++r) count_of[r] += match_of[r];
        }
        uint32_t la_vec = 0;
        if constexpr (COUNT != kCountKept)
            la_vec = *reinterpret_cast<const uint32_t*>(tab + len_at + (tid & 31u) * 4u) * la_coef;
        char* row_ptr = out_block + (uint64_t)t * kTile * row_bytes;
        const uint32_t rows_left = na - t * kTile;
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            uint32_t v = (count_of[q & 7] >> (8 * (q >> 3))) & 0xffu;
            if constexpr (COUNT != kCountKept) {
                const uint32_t la_term = (uint32_t)__builtin_amdgcn_readlane((int)la_vec, q);
                v = (la_term + lb_term) - v;
            }
            const uint32_t n = (uint32_t)q < rows_left ? nrec : 0u;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)n, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)(tid * 4u), 0, 0);
            row_ptr += row_bytes;
        }
    }
}

struct Args {
    const sdrb_batch* b;
    const unsigned char* tables;
    uint32_t tile_bytes, ntiles, tiles_per_block, na;
    int op;
    uint32_t swap;
    size_t lds;
    dim3 grid;
    uint32_t* out;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int NCH, int WCAP, bool BV, int PF>
void launch_inner_k(const Args& g) {
    if (g.lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lane_rows_inner<NCH, WCAP, BV, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    hipLaunchKernelGGL((k_lane_rows_inner<NCH, WCAP, BV, PF>), g.grid, dim3(kBlock), g.lds, g.stream, g.b->d_offsets,
                       g.b->d_ids, g.b->d_data, g.b->n_rows, g.tables, g.tile_bytes, g.ntiles, g.tiles_per_block, g.na,
                       g.out, g.ld_out);
}
template <int NCH, int WCAP, int PF>
void launch_inner_v(const Args& g) {
    if (g.b->d_data) launch_inner_k<NCH, WCAP, true, PF>(g);
    else launch_inner_k<NCH, WCAP, false, PF>(g);
}
template <int NCH>
void launch_inner_w(const Args& g, int wcap) {
    const uint32_t pieces = (g.tile_bytes / 16u + kBlock - 1) / kBlock;
    if (wcap == 2048) {
        if (pieces <= 5) launch_inner_v<NCH, 2048, 5>(g);
        else launch_inner_v<NCH, 2048, 8>(g);
    } else {
        launch_inner_v<NCH, 4096, 11>(g);
    }
}
void launch_inner_n(const Args& g, int wcap) {
    const uint32_t longest = g.b->max_row_len;
    if (longest <= 16) launch_inner_w<2>(g, wcap);
    else if (longest <= 32) launch_inner_w<4>(g, wcap);
    else if (longest <= 40) launch_inner_w<5>(g, wcap);
    else launch_inner_w<8>(g, wcap);
}

template <int MATH, int COUNT, int NCH, int WCAP, int PF>
void launch_size(const Args& g) {
    // this = A unless swapped: xors takes both lengths, rms the length of `this`, ands none
    const uint32_t la_coef = g.op == SDRB_OP_XOR || !g.swap ? 1u : 0u, lb_coef = g.op == SDRB_OP_XOR || g.swap ? 1u : 0u;
    hipLaunchKernelGGL((k_lane_rows_size<MATH, COUNT, NCH, WCAP, PF>), g.grid, dim3(kBlock), 0, g.stream,
                       g.b->d_offsets, g.b->d_ids, g.b->d_data, g.b->n_rows, g.tables, g.tile_bytes, g.ntiles,
                       g.tiles_per_block, g.na, la_coef, lb_coef, g.out, g.ld_out);
}
template <int MATH, int COUNT, int NCH>
void launch_size_w(const Args& g, int wcap) {
    const uint32_t pieces = (g.tile_bytes / 16u + kBlock - 1) / kBlock;
    if (wcap == 2048) {
        if (pieces <= 5) launch_size<MATH, COUNT, NCH, 2048, 5>(g);
        else launch_size<MATH, COUNT, NCH, 2048, 8>(g);
    } else {
        launch_size<MATH, COUNT, NCH, 4096, 11>(g);
    }
}
template <int MATH, int COUNT>
void launch_size_n(const Args& g, int wcap) {
    if (g.b->max_row_len <= 40) launch_size_w<MATH, COUNT, 5>(g, wcap);
    else launch_size_w<MATH, COUNT, 8>(g, wcap);
}
template <int MATH>
void launch_size_m(const Args& g, int wcap) {
    if (g.op == SDRB_OP_AND) launch_size_n<MATH, kCountKept>(g, wcap);
    else if (g.op == SDRB_OP_RM) launch_size_n<MATH, kCountDropped>(g, wcap);
    else launch_size_n<MATH, kCountBoth>(g, wcap);
}

bool shapes_fit(const sdrb_batch* a, const sdrb_batch* b) {
    if (b->width <= 0 || b->width > 4096 || b->max_row_len > 64) return false;
    return a->max_row_len <= (uint32_t)kMaxRowA && a->n_rows <= 0x7FFFFFE0ull;
}

// a match costs several times a dense table's multiply-add or a mask-table visit: only where most candidates are
// misses -- on average at most `tenths`/10 rows of a 32-row tile hold a given id of the declared width
bool sparse_enough(const sdrb_batch* a, const sdrb_batch* b, uint64_t tenths) {
    return 10ull * kTile * a->nnz <= tenths * a->n_rows * (uint64_t)b->width;
}

// tables of A in scratch, grid and LDS size; `with_vals`: A's payloads are tabulated
int prepare(const sdrb_batch* a, const sdrb_batch* b, bool with_vals, bool lds_acc, hipStream_t stream, Args* g) {
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    g->b = b;
    g->na = (uint32_t)a->n_rows;
    const unsigned char* tables = nullptr;
    const int rc = build_lane_tiles(a, b->device, wcap, with_vals, stream, &g->tile_bytes, &g->ntiles, &tables);
    if (rc != SDRB_OK) return rc;
    g->tables = tables;
    g->lds = (size_t)g->tile_bytes + (lds_acc ? (size_t)kTile * kBlock * sizeof(uint32_t) : 0);
    // Three blocks are resident per compute unit (768 in all).  A block walks every tile with its rows of B in
    // registers, so one block per 256 rows of B is the cheapest cut; with fewer than eight rounds of blocks the last,
    // partly filled round idles a noticeable part of the chip, and the tiles are shared out over blockIdx.y instead
    // (the rows of B are then fetched once per share, from L2; 4096 x 1 Mi: 8.03 -> 7.77 ms with two shares).
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 8u * 768u ? 1u : (8u * 768u + gx - 1) / gx;
    if (gy > g->ntiles) gy = g->ntiles;
    g->tiles_per_block = (g->ntiles + gy - 1) / gy;
    gy = (g->ntiles + g->tiles_per_block - 1) / g->tiles_per_block;
    g->grid = dim3(gx, gy);
    g->stream = stream;
    return SDRB_OK;
}

}  // namespace

// `forced`: whether the kernel can serve the batches at all; else whether it is also the faster choice
bool inner_sparse_applicable(const sdrb_batch* a, const sdrb_batch* b, bool forced) {
    if (!shapes_fit(a, b)) return false;
    // an id of A beyond the table would be dropped: fine for a match count, and for a dot product as well (it
    // cannot match an id of B, all of which lie below B's width).  Against the dense value table the match walk wins
    // up to several rows of a tile per id (measured at 256/40: 5 rows per id, 1.7 against 2.3 ms).
    return forced || sparse_enough(a, b, 60);
}

int launch_inner_sparse(const sdrb_batch* a, const sdrb_batch* b, float* out, uint64_t ld_out, hipStream_t stream) {
    Args g;
    const int rc = prepare(a, b, true, true, stream, &g);
    if (rc != SDRB_OK) return rc;
    g.op = 0;
    g.swap = 0;
    g.out = reinterpret_cast<uint32_t*>(out);
    g.ld_out = ld_out;
    launch_inner_n(g, b->width <= 2048 ? 2048 : 4096);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

// Size operations whose result depends on payloads (UnitData on the `this` side of ands/xors/rms)
bool size_lane_rows_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, bool forced) {
    if (!shapes_fit(a, b)) return false;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) return false;
    const int math = pick_math(a, b, op, swap);
    if (math == kKeepAll || math == kDropAll) return false;          // ids alone decide: the bit-sliced kernel's case
    // against the mask-table kernel: rows of B of at most 40 ids (longer rows leave a lane no registers for the tile
    // in flight; measured at 4096/64: 1.38 against 1.03 ms) -- then it wins from 0.16 to 5 rows of a tile per id
    // (1024 x 256 Ki rows of 40 ids at widths 4096 ... 256: 0.48 / 0.59, 0.48 / 0.81, 0.56 / 0.92, 0.72 / 1.10, 1.00 / 1.56 ms)
    // and against the few-rows kernel of mask_table.hip from two rows of A on (1 Mi rows of B, 2048/40, ands: 2 rows 78 / 84,
    // 8 rows 89 / 104, 16 rows 101 / 132 us; one row: 76 / 81 us for ands but 87 / 80 us for xors)
    return forced || (a->n_rows >= 2 && b->max_row_len <= 40 && sparse_enough(a, b, 50));
}

int launch_size_lane_rows(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                          hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const int math = pick_math(a, b, op, swap);
    Args g;
    const int rc = prepare(a, b, math != kSelfB, false, stream, &g);
    if (rc != SDRB_OK) return rc;
    g.op = op;
    g.swap = swap ? 1u : 0u;
    g.out = out;
    g.ld_out = ld_out;
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    if (g.tile_bytes > lane_tile_cap((uint32_t)wcap)) {   // the kernel's LDS array holds the largest tile shapes_fit admits
        set_error("lane-per-row size kernel: a tile of %u bytes exceeds the kernel's table", g.tile_bytes);
        return SDRB_ERR_UNSUPPORTED;
    }
    switch (math) {
    case kSelfA: launch_size_m<kSelfA>(g, wcap); break;
    case kSelfB: launch_size_m<kSelfB>(g, wcap); break;
    case kMul: launch_size_n<kMul, kCountKept>(g, wcap); break;
    case kAbsDiff: launch_size_n<kAbsDiff, kCountBoth>(g, wcap); break;
    case kRmulA: launch_size_n<kRmulA, kCountDropped>(g, wcap); break;
    default: launch_size_n<kRmulB, kCountDropped>(g, wcap); break;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
