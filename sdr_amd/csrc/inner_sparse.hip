// Sparse dot products for every pair, visiting id matches only:  out(i, j) = A_i.inner(B_j).
//
// The reference walks the common ids of two rows in ascending order and accumulates
// acc = acc + (a * b), a rounded product and then a rounded sum, from 0 (SDR.hpp:1682-1692 with
// ArithData.hpp:37-43,59-63).  At SDR densities nearly every (row of A, id of B_j) candidate is a miss
// -- 40 of 2048 ids are set, so a dense table of A's values multiplies by zero 98 % of the time.  Here
// the rows of A are cut into tiles of 32; a tile is described by
//
//     mask[w]  (uint32)  bit q set  <=>  row 32 t + q of A contains id w
//     col[w]   (uint16)  where the tile's values of id w start (as a byte offset into the tile)
//     vals[]   (float)   the tile's values, column-compressed: by id, then by ascending row
//
// built once per call in global memory (17 KB a tile for width 2048, 40 ids a row) and staged in LDS.
// A lane owns one row of B, ids and values in registers, and walks it in ascending id order: an id
// costs two LDS reads (mask, col), and every set bit of its mask is one match -- one value of A from
// LDS, one multiply, one add into the lane's own accumulator of that row of A (acc[q][lane] in LDS:
// the lanes of a wave hit 64 different banks).  A pair's matches are therefore added in ascending id
// order, one after the other, as the reference does.  After a tile the lane stores its 32 sums, one
// coalesced 256-byte segment per wave per row of A, and clears them.
#include "sdrb_internal.h"

namespace sdrb {

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;
constexpr int kTile = 32;            // rows of A per tile = bits of a mask word
constexpr int kMaxRowA = 128;        // longest row of A: a tile's values stay below 16 KB
constexpr int kMaxStage = 8;         // 16-byte pieces of a tile a thread stages at most (32 KiB / 256 / 16)

constexpr uint32_t mask_bytes(uint32_t wcap) { return (wcap + 4) * 4; }   // entry [wcap] = 0: the sentinel id
constexpr uint32_t col_bytes(uint32_t wcap) { return (wcap + 8) * 2; }

// One block per tile of A: mask, col and the column-compressed values, written to the tile's slot in `tables`.
__global__ __launch_bounds__(kBlock) void k_build_sparse_tiles(const uint32_t* __restrict__ a_off,
                                                               const int32_t* __restrict__ a_ids,
                                                               const float* __restrict__ a_val, uint32_t na,
                                                               uint32_t wcap, uint32_t tile_bytes,
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

    for (uint32_t w = tid; w < wcap; w += kBlock) mask[w] = 0;
    __syncthreads();
    for (uint32_t q = wave; q < rows; q += kBlock / 64) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) atomicOr(&mask[(uint32_t)a_ids[beg + k]], 1u << q);
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
    float* g_vals = reinterpret_cast<float*>(tile + vals_at);
    for (uint32_t q = wave; q < rows; q += kBlock / 64) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            g_vals[col[id] + (uint32_t)__builtin_popcount(mask[id] & ((1u << q) - 1u))] = a_val[beg + k];
        }
    }
}

// NCH  : the lane keeps 8*NCH (id, value) pairs of its row in registers; shorter rows are padded with the
//        sentinel id, whose mask is zero
// WCAP : table entries per tile
// BV   : B carries values (an EmptyData argument leaves the value of `this` unchanged, ArithData.hpp:39)
template <int NCH, int WCAP, bool BV>
__global__ __launch_bounds__(kBlock, 3) void k_inner_sparse(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint64_t nb, const unsigned char* __restrict__ tables, uint32_t tile_bytes, uint32_t ntiles,
    uint32_t tiles_per_block, uint32_t na, float* __restrict__ out, uint64_t ld_out) {
    constexpr uint32_t MASK_B = mask_bytes(WCAP), COL_AT = MASK_B;
    // LDS: the tile's tables (tile_bytes) | acc[32][256]
    extern __shared__ __attribute__((aligned(16))) unsigned char sparse_lds[];
    unsigned char* tab = sparse_lds;
    const uint32_t tid = threadIdx.x;
    float* acc_lane = reinterpret_cast<float*>(sparse_lds + tile_bytes) + tid;

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
    for (int q = 0; q < kTile; ++q) acc_lane[q * kBlock] = 0.0f;

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    const uint32_t nvec = tile_bytes / 16u;
    const uint64_t row_bytes = ld_out * 4;
    const uint64_t rows_here = nb - j0 < (uint64_t)kBlock ? nb - j0 : (uint64_t)kBlock;
    const uint32_t nrec = (uint32_t)rows_here * 4;
    char* const out_block = reinterpret_cast<char*>(out) + j0 * 4;

    for (uint32_t t = t0; t < t1; ++t) {
        // stage the tile: global -> registers -> LDS, once every wave is done with the previous tile
        const u32x4* src = reinterpret_cast<const u32x4*>(tables + (uint64_t)t * tile_bytes);
        __syncthreads();
        {
            u32x4 piece[kMaxStage];
#pragma unroll
            for (int p = 0; p < kMaxStage; ++p)
                if (tid + p * kBlock < nvec) piece[p] = src[tid + p * kBlock];
#pragma unroll
            for (int p = 0; p < kMaxStage; ++p)
                if (tid + p * kBlock < nvec) reinterpret_cast<u32x4*>(tab)[tid + p * kBlock] = piece[p];
        }
        __syncthreads();

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
                    float* slot = acc_lane + q * kBlock;
                    const float prod = a * bv;
                    *slot = *slot + prod;
                }
            }
        }

        // the lane's sums of this tile: store in row order of A and clear
        char* row_ptr = out_block + (uint64_t)t * kTile * row_bytes;
        const uint32_t rows_left = na - t * kTile;
#pragma unroll
        for (int q = 0; q < kTile; ++q) {
            const float x = acc_lane[q * kBlock];
            acc_lane[q * kBlock] = 0.0f;
            const uint32_t n = (uint32_t)q < rows_left ? nrec : 0u;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)n, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, x), rsrc, (int)(tid * 4u), 0, 0);
            row_ptr += row_bytes;
        }
    }
}

struct Args {
    const sdrb_batch* b;
    const unsigned char* tables;
    uint32_t tile_bytes, ntiles, tiles_per_block, na;
    size_t lds;
    dim3 grid;
    float* out;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int NCH, int WCAP, bool BV>
void launch(const Args& g) {
    if (g.lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inner_sparse<NCH, WCAP, BV>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    hipLaunchKernelGGL((k_inner_sparse<NCH, WCAP, BV>), g.grid, dim3(kBlock), g.lds, g.stream, g.b->d_offsets, g.b->d_ids,
                       g.b->d_data, g.b->n_rows, g.tables, g.tile_bytes, g.ntiles, g.tiles_per_block, g.na, g.out,
                       g.ld_out);
}
template <int NCH, int WCAP>
void launch_v(const Args& g) {
    if (g.b->d_data) launch<NCH, WCAP, true>(g);
    else launch<NCH, WCAP, false>(g);
}
template <int NCH>
void launch_w(const Args& g, int wcap) {
    if (wcap == 2048) launch_v<NCH, 2048>(g);
    else launch_v<NCH, 4096>(g);
}

}  // namespace

// `forced`: whether the kernel can serve the batches at all; else whether it is also the faster choice
bool inner_sparse_applicable(const sdrb_batch* a, const sdrb_batch* b, bool forced) {
    if (b->width <= 0 || b->width > 4096 || b->max_row_len > 64) return false;
    const int64_t wcap = b->width <= 2048 ? 2048 : 4096;
    if (a->nnz && (a->min_id < 0 || a->max_id >= wcap)) return false;   // every id of A needs a table entry
    if (a->max_row_len > (uint32_t)kMaxRowA || a->n_rows > 0x7FFFFFE0ull) return false;
    if (forced) return true;
    // a match costs several times a dense table's multiply-add: only where most candidates are misses
    // (on average at most one row of a tile holds a given id)
    return (uint64_t)kTile * a->nnz <= (uint64_t)a->n_rows * (uint64_t)wcap;
}

int launch_inner_sparse(const sdrb_batch* a, const sdrb_batch* b, float* out, uint64_t ld_out, hipStream_t stream) {
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    const int nch = b->max_row_len <= 16 ? 2 : b->max_row_len <= 32 ? 4 : b->max_row_len <= 40 ? 5 : 8;
    Args g;
    g.b = b;
    g.na = (uint32_t)a->n_rows;
    g.ntiles = (uint32_t)((a->n_rows + kTile - 1) / kTile);
    const uint32_t val_cap = ((uint32_t)kTile * (a->max_row_len ? a->max_row_len : 1u) + 3u) & ~3u;
    g.tile_bytes = mask_bytes((uint32_t)wcap) + col_bytes((uint32_t)wcap) + val_cap * 4u;
    if (g.tile_bytes > (uint32_t)kMaxStage * kBlock * 16u) {
        set_error("sdrb_inner: tile tables of %u bytes exceed the staging registers", g.tile_bytes);
        return SDRB_ERR_UNSUPPORTED;
    }
    const size_t table_bytes = (size_t)g.ntiles * g.tile_bytes;
    unsigned char* tables = static_cast<unsigned char*>(workspace(b->device, stream, 5, table_bytes));
    if (!tables) return SDRB_ERR_HIP;
    const size_t build_lds = (size_t)wcap * 6 + 64;
    hipLaunchKernelGGL(k_build_sparse_tiles, dim3(g.ntiles), dim3(kBlock), build_lds, stream, a->d_offsets, a->d_ids,
                       a->d_data, (uint32_t)a->n_rows, (uint32_t)wcap, g.tile_bytes, tables);
    SDRB_HIP(hipGetLastError());
    g.tables = tables;
    g.lds = (size_t)g.tile_bytes + (size_t)kTile * kBlock * sizeof(float);
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 768 ? 1u : (768u + gx - 1) / gx;
    if (gy > g.ntiles) gy = g.ntiles;
    g.tiles_per_block = (g.ntiles + gy - 1) / gy;
    gy = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    g.grid = dim3(gx, gy);
    g.out = out;
    g.ld_out = ld_out;
    g.stream = stream;
    switch (nch) {
    case 2: launch_w<2>(g, wcap); break;
    case 4: launch_w<4>(g, wcap); break;
    case 5: launch_w<5>(g, wcap); break;
    default: launch_w<8>(g, wcap); break;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
