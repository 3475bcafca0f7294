// Lane-per-row match kernels: every id match of every pair (row of A, row of B) is visited once, for
//   * sparse dot products, out(i, j) = A_i.inner(B_j) (SDR.hpp:1682-1692 with ArithData.hpp:37-43,59-63:
//     over the common ids in ascending order, acc = acc + (a * b), a rounded product and then a rounded sum), and
//   * the size operations whose result depends on payloads (SDR.hpp:1083-1095, 1428-1433, 1572-1576 with
//     UnitData.hpp:22-28,48-51,94-97,117-120: a match counts when the combined payload is relevant).
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
// A lane owns one row of B, ids and payloads in registers, and walks it in ascending id order: an id
// costs two LDS reads (mask, col), and every set bit of its mask is one match -- one payload of A from
// LDS and one update of the lane's own accumulator of that row of A (acc[q][lane] in LDS: the lanes of a
// wave hit 64 different banks, and nothing is shared, so no atomics).  A pair's matches are visited in
// ascending id order, one after the other, as the reference does -- which is what makes the float sums
// bit-exact.  After a tile the lane stores its 32 results, one coalesced 256-byte segment per wave per
// row of A, and clears the accumulators.  The next tile's tables travel from L2 while a tile is walked.
#include <type_traits>

#include "lane_tiles.h"
#include "match_math.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

// What a match does to the lane's accumulator of its pair, and what leaves the kernel.
struct InnerVisit {          // acc = acc + a * b, product and sum rounded separately
    static constexpr bool kNeedA = true;
    __device__ static __forceinline__ uint32_t visit(uint32_t acc, float a, float b) {
        const float prod = a * b;
        return __float_as_uint(__uint_as_float(acc) + prod);
    }
    __device__ static __forceinline__ uint32_t result(uint32_t acc, int, uint32_t, uint32_t, uint32_t) { return acc; }
};
template <int MATH>
struct SizeVisit {           // acc = matches << 16 | kept matches
    static constexpr bool kNeedA = MATH != kSelfB;
    static constexpr bool kNeedB = MATH != kSelfA;
    __device__ static __forceinline__ uint32_t visit(uint32_t acc, float a, float b) {
        float d;
        return acc + (0x10000u | (match_kept_by<MATH>(a, b, d) ? 1u : 0u));
    }
    __device__ static __forceinline__ uint32_t result(uint32_t acc, int op, uint32_t swap, uint32_t la, uint32_t lb) {
        return size_of(op, swap ? lb : la, swap ? la : lb, acc >> 16, acc & 0xffffu);
    }
};

// NCH  : the lane keeps 8*NCH (id, payload) pairs of its row in registers; shorter rows are padded with the
//        sentinel id, whose mask is zero
// WCAP : table entries per tile
// BV   : B carries payloads (an EmptyData argument leaves the value of `this` unchanged, ArithData.hpp:39)
// PF   : 16-byte pieces of a tile a thread stages
template <typename VISIT, int NCH, int WCAP, bool BV, int PF>
__global__ __launch_bounds__(kBlock, 3) void k_lane_rows(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint64_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t ntiles,
    uint32_t tiles_per_block, uint32_t na, int op, uint32_t swap, uint32_t* __restrict__ out, uint64_t ld_out) {
    constexpr uint32_t COL_AT = mask_bytes(WCAP);
    constexpr bool kSize = !std::is_same_v<VISIT, InnerVisit>;
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
                c[k] = VISIT::kNeedA ? *reinterpret_cast<const uint16_t*>(tab + COL_AT + (o >> 1)) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                uint32_t mm = m[k], cc = c[k];
                const float bv = val[ch * 8 + k];
                while (mm != 0) {
                    const uint32_t q = (uint32_t)__builtin_ctz(mm);
                    mm &= mm - 1;
                    float a = 0.0f;
                    if constexpr (VISIT::kNeedA) {
                        a = *reinterpret_cast<const float*>(tab + cc);
                        cc += 4;
                    }
                    uint32_t* slot = acc_lane + q * kBlock;
                    *slot = VISIT::visit(*slot, a, bv);
                }
            }
        }

        // the lane's results of this tile: store in row order of A and clear the accumulators
        char* row_ptr = out_block + (uint64_t)t * kTile * row_bytes;
        const uint32_t rows_left = na - t * kTile;
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            const uint32_t x = acc_lane[q * kBlock];
            acc_lane[q * kBlock] = 0u;
            uint32_t la = 0;
            if constexpr (kSize) la = *reinterpret_cast<const uint32_t*>(tab + len_at + q * 4);
            const uint32_t v = VISIT::result(x, op, swap, la, lb);
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

template <typename VISIT, int NCH, int WCAP, bool BV, int PF>
void launch(const Args& g) {
    if (g.lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lane_rows<VISIT, NCH, WCAP, BV, PF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    hipLaunchKernelGGL((k_lane_rows<VISIT, NCH, WCAP, BV, PF>), g.grid, dim3(kBlock), g.lds, g.stream, g.b->d_offsets,
                       g.b->d_ids, g.b->d_data, g.b->n_rows, g.tables, g.tile_bytes, g.ntiles, g.tiles_per_block, g.na,
                       g.op, g.swap, g.out, g.ld_out);
}
template <typename VISIT, int NCH, int WCAP, int PF>
void launch_v(const Args& g) {
    if constexpr (std::is_same_v<VISIT, InnerVisit>) {
        if (g.b->d_data) launch<VISIT, NCH, WCAP, true, PF>(g);
        else launch<VISIT, NCH, WCAP, false, PF>(g);
    } else {   // a size rule either reads B's payloads (then B has them) or never looks at them
        launch<VISIT, NCH, WCAP, VISIT::kNeedB, PF>(g);
    }
}
template <typename VISIT, int NCH>
void launch_w(const Args& g, int wcap) {
    const uint32_t pieces = (g.tile_bytes / 16u + kBlock - 1) / kBlock;
    if (wcap == 2048) {
        if (pieces <= 5) launch_v<VISIT, NCH, 2048, 5>(g);
        else launch_v<VISIT, NCH, 2048, 8>(g);
    } else {
        launch_v<VISIT, NCH, 4096, 11>(g);
    }
}
template <typename VISIT>
void launch_n(const Args& g, int wcap) {
    const uint32_t longest = g.b->max_row_len;
    if constexpr (std::is_same_v<VISIT, InnerVisit>) {
        if (longest <= 16) return launch_w<VISIT, 2>(g, wcap);
        if (longest <= 32) return launch_w<VISIT, 4>(g, wcap);
    }
    if (longest <= 40) launch_w<VISIT, 5>(g, wcap);
    else launch_w<VISIT, 8>(g, wcap);
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
int prepare(const sdrb_batch* a, const sdrb_batch* b, bool with_vals, hipStream_t stream, Args* g) {
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    g->b = b;
    g->na = (uint32_t)a->n_rows;
    const unsigned char* tables = nullptr;
    const int rc = build_lane_tiles(a, b->device, wcap, with_vals, stream, &g->tile_bytes, &g->ntiles, &tables);
    if (rc != SDRB_OK) return rc;
    g->tables = tables;
    g->lds = (size_t)g->tile_bytes + (size_t)kTile * kBlock * sizeof(uint32_t);
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
    const int rc = prepare(a, b, true, stream, &g);
    if (rc != SDRB_OK) return rc;
    g.op = 0;
    g.swap = 0;
    g.out = reinterpret_cast<uint32_t*>(out);
    g.ld_out = ld_out;
    launch_n<InnerVisit>(g, b->width <= 2048 ? 2048 : 4096);
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
    // in flight; measured at 4096/64: 1.67 against 1.03 ms) -- then it wins from 0.16 to 5 rows of a tile per id
    // (1024 x 256 Ki rows of 40 ids at widths 4096 ... 256: 0.52 / 0.58, 0.65 / 0.81, 0.80 / 0.91, 1.03 / 1.09, 1.42 / 1.55 ms)
    return forced || (a->n_rows > 16 && b->max_row_len <= 40 && sparse_enough(a, b, 50));
}

int launch_size_lane_rows(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                          hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const int math = pick_math(a, b, op, swap);
    Args g;
    const int rc = prepare(a, b, math != kSelfB, stream, &g);
    if (rc != SDRB_OK) return rc;
    g.op = op;
    g.swap = swap ? 1u : 0u;
    g.out = out;
    g.ld_out = ld_out;
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    switch (math) {
    case kSelfA: launch_n<SizeVisit<kSelfA>>(g, wcap); break;
    case kSelfB: launch_n<SizeVisit<kSelfB>>(g, wcap); break;
    case kMul: launch_n<SizeVisit<kMul>>(g, wcap); break;
    case kAbsDiff: launch_n<SizeVisit<kAbsDiff>>(g, wcap); break;
    case kRmulA: launch_n<SizeVisit<kRmulA>>(g, wcap); break;
    default: launch_n<SizeVisit<kRmulB>>(g, wcap); break;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
