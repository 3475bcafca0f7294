// Sparse dot products for every pair: out(i, j) = A_i.inner(B_j) on ArithData<float> rows.
//
// The reference walks the id matches of the two rows with andv (SDR.hpp:958-997) in ascending
// id order and accumulates  acc = acc + (a * b)  -- a rounded product, then a rounded sum
// (SDR.hpp:1682-1692 with ArithData.hpp:37-43,59-63), starting from 0.  The same numbers are the
// entries of diff_mul (SDR.hpp:1886-1923) and of row_major_mul_vec (SDR.hpp:1729-1750).  Float
// addition is not associative, so both kernels here add the matches of a pair in ascending id
// order, one after the other, with separate multiply and add (-ffp-contract=off).
//
// k_inner_table: for declared widths <= 4096 and rows of B <= 64 ids.  The rows of A are expanded
// once per call into a dense table, four rows interleaved: table[i/4][w] = float4 of the values of
// rows 4(i/4)..+3 at id w (0 where a row lacks the id).  A lane owns one row of B (ids as LDS byte
// offsets and values in registers), a tile of four rows of A sits in LDS, and each id of the row
// costs one ds_read_b128 and four multiply-adds -- four dot products advance together.  A term of
// an absent id is b * 0 = +-0, and adding +-0 never changes a sum that started at +0 (payloads
// are finite: uploads of ArithData batches reject NaN and infinities), so no match test is needed
// and the result is bit-identical to adding the matches alone.
//
// k_inner_merge: any ids, any row lengths; one lane per pair walks both rows with two pointers.
#include "sdrb_internal.h"

namespace sdrb {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;

// ---------------------------------------------------------------- dense interleaved table of A
__global__ __launch_bounds__(64) void k_build_value_table(const uint32_t* __restrict__ a_off,
                                                           const int32_t* __restrict__ a_ids,
                                                           const float* __restrict__ a_val, uint32_t wcap,
                                                           float* __restrict__ table) {
    const uint64_t i = blockIdx.x;
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    float* tile = table + (i / 4) * (uint64_t)wcap * 4 + (i % 4);
    for (uint32_t k = threadIdx.x; k < len; k += 64) tile[(uint64_t)(uint32_t)a_ids[beg + k] * 4] = a_val[beg + k];
}

template <int PF>
__device__ __forceinline__ void load_tile(f32x4 (&pf)[PF], const f32x4* __restrict__ tile, uint32_t tid) {
#pragma unroll
    for (int p = 0; p < PF; ++p) pf[p] = tile[tid + p * kBlock];
}
template <int PF>
__device__ __forceinline__ void write_tile(const f32x4 (&pf)[PF], f32x4* buf, uint32_t tid) {
#pragma unroll
    for (int p = 0; p < PF; ++p) buf[tid + p * kBlock] = pf[p];
}

// NCH  : the lane keeps 8*NCH (id, value) pairs of its row in registers; shorter rows are padded
//        with the sentinel offset of an all-zero table entry and value 0
// WCAP : table entries per tile; one tile (four rows of A) fills one LDS buffer, two alternate
template <int NCH, int WCAP, bool B_HAS_VALUES>
__global__ __launch_bounds__(kBlock, WCAP <= 2048 ? 2 : 1) void k_inner_table(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_val,
    uint64_t nb, const f32x4* __restrict__ table, uint32_t ntiles, uint32_t tiles_per_block, uint32_t na,
    float* __restrict__ out, uint64_t ld_out) {
    constexpr int BUF = WCAP + 1;          // entries per LDS buffer (zero sentinel at [WCAP])
    constexpr int PF = WCAP / kBlock;      // 16-byte table loads per thread per tile
    __shared__ f32x4 lds[2 * BUF];

    const uint32_t tid = threadIdx.x;
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
    uint32_t off[8 * NCH];
    float val[8 * NCH];
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k)
        off[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ids_rsrc, (int)(rel + 4u * k), 0, 0);
    if constexpr (B_HAS_VALUES) {
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_val) + blk_beg, 0,
                                                                            (int)blk_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8 * NCH; ++k)
            val[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, (int)(rel + 4u * k), 0, 0));
    }
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k) {
        const bool live = (uint32_t)k < lb;
        off[k] = live ? off[k] * 16u : WCAP * 16u;
        // an EmptyData argument leaves the value of `this` unchanged (ArithData.hpp:39): a * 1 = a exactly
        val[k] = live ? (B_HAS_VALUES ? val[k] : 1.0f) : 0.0f;
    }

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    if (t0 >= t1) return;
    if (tid == 0) {
        lds[WCAP] = f32x4{0.f, 0.f, 0.f, 0.f};
        lds[BUF + WCAP] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 pf[PF];
    load_tile(pf, table + (uint64_t)t0 * WCAP, tid);
    write_tile(pf, lds, tid);
    __syncthreads();

    for (uint32_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) load_tile(pf, table + (uint64_t)(t + 1) * WCAP, tid);
        const char* base = reinterpret_cast<const char*>(lds + ((t - t0) & 1) * BUF);
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll
        for (int k = 0; k < 8 * NCH; ++k) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(base + off[k]);
            acc0 = acc0 + val[k] * a[0];
            acc1 = acc1 + val[k] * a[1];
            acc2 = acc2 + val[k] * a[2];
            acc3 = acc3 + val[k] * a[3];
        }
        if (j < nb) {
            const uint64_t i = (uint64_t)t * 4;
            float* o = out + i * ld_out + j;
            o[0] = acc0;
            if (i + 1 < na) o[ld_out] = acc1;
            if (i + 2 < na) o[2 * ld_out] = acc2;
            if (i + 3 < na) o[3 * ld_out] = acc3;
        }
        if (more) write_tile(pf, lds + (((t - t0) & 1) ^ 1) * BUF, tid);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- general per-pair merge
// ---------------------------------------------------------------- few rows of A (matrix x vector)
// Up to eight rows of A against rows of B of at most 64 ids (row_major_mul_vec is the one-row case):
// a lane owns a row of B, fetched with 16-byte loads that are all issued up front, and walks it in
// id order, so the matches of each of its pairs are added in ascending id order by construction.
// LDS holds one byte per id (bit q: row q of A contains it), and the values of A column-compressed:
// col[w] says where the values of id w start, in ascending row order.  One launch, B read once.
constexpr int kFewRows = 8;
template <int NQ, bool B_HAS_VALUES>
__global__ __launch_bounds__(kBlock) void k_inner_few(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_val,
    uint32_t na, const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids,
    const float* __restrict__ b_val, uint64_t nb, uint32_t width, uint32_t val_cap,
    float* __restrict__ out, uint64_t ld_out) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // LDS: vals[val_cap] | bits[(width + 4) / 4 words] (entry [width] stays zero) | col[width + 1]
    extern __shared__ __attribute__((aligned(16))) unsigned char few_raw[];
    float* vals = reinterpret_cast<float*>(few_raw);
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(vals + val_cap);
    const unsigned char* bits = reinterpret_cast<const unsigned char*>(bits32);
    const uint32_t words = (width + 4) / 4;
    uint16_t* col = reinterpret_cast<uint16_t*>(bits32 + words);
    __shared__ uint32_t wave_tot[kBlock / 64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t w = tid; w < words; w += kBlock) bits32[w] = 0;
    __syncthreads();
    for (uint32_t q = 0; q < na; ++q) {
        const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
        for (uint32_t k = tid; k < len; k += kBlock) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) atomicOr(&bits32[id >> 2], (1u << q) << (8 * (id & 3)));   // ids outside B's width never match
        }
    }
    __syncthreads();
    {
        const uint32_t per = (width + kBlock) / kBlock;
        const uint32_t w0 = tid * per < width + 1 ? tid * per : width + 1;
        const uint32_t w1 = w0 + per < width + 1 ? w0 + per : width + 1;
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__builtin_popcount(bits[w]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) run += wave_tot[w];
        for (uint32_t w = w0; w < w1; ++w) {
            col[w] = (uint16_t)run;
            run += (uint32_t)__builtin_popcount(bits[w]);
        }
    }
    __syncthreads();
    for (uint32_t q = 0; q < na; ++q) {
        const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
        for (uint32_t k = tid; k < len; k += kBlock) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) vals[col[id] + (uint32_t)__builtin_popcount(bits[id] & ((1u << q) - 1u))] = a_val[beg + k];
        }
    }
    __syncthreads();

    const uint64_t pieces = (nb + kBlock - 1) / kBlock;
    for (uint64_t piece = blockIdx.x; piece < pieces; piece += gridDim.x) {
        const uint64_t j0 = piece * kBlock, j = j0 + tid;
        uint32_t beg = 0, lb = 0;
        if (j < nb) {
            beg = b_off[j];
            lb = b_off[j + 1] - beg;
        }
        const uint32_t blk_beg = b_off[j0];                                   // uniform
        const uint64_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
        const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;          // uniform
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)blk_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(b_val) + (B_HAS_VALUES ? blk_beg : 0u), 0, B_HAS_VALUES ? (int)blk_bytes : 0, 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
        u32x4 ids[NQ], vbs[B_HAS_VALUES ? NQ : 1];
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4) {
            ids[k4] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
            if constexpr (B_HAS_VALUES) vbs[k4] = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, (int)(rel + 16u * k4), 0, 0);
        }
        float acc[kFewRows];
#pragma unroll
        for (int q = 0; q < kFewRows; ++q) acc[q] = 0.0f;
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t id = (uint32_t)(4 * k4 + c) < lb ? ids[k4][c] : width;
                const uint32_t m = bits[id];
                if (m != 0) {
                    uint32_t at = col[id];
#pragma unroll
                    for (int q = 0; q < kFewRows; ++q) {
                        if (m & (1u << q)) {
                            const float va = vals[at++];
                            const float term = B_HAS_VALUES ? va * __builtin_bit_cast(float, (uint32_t)vbs[k4][c]) : va;
                            acc[q] = acc[q] + term;
                        }
                    }
                }
            }
        }
        float* dst = out + j;
#pragma unroll
        for (int q = 0; q < kFewRows; ++q) {
            if ((uint32_t)q < na && j < nb) __builtin_nontemporal_store(acc[q], dst);
            dst += ld_out;
        }
    }
}

template <bool B_HAS_VALUES>
__global__ __launch_bounds__(kBlock) void k_inner_merge(const uint32_t* __restrict__ a_off,
                                                         const int32_t* __restrict__ a_ids,
                                                         const float* __restrict__ a_val,
                                                         const uint32_t* __restrict__ b_off,
                                                         const int32_t* __restrict__ b_ids,
                                                         const float* __restrict__ b_val, uint64_t nb,
                                                         float* __restrict__ out, uint64_t ld_out) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t i = blockIdx.y;
    if (j >= nb) return;
    uint32_t p = a_off[i], q = b_off[j];
    const uint32_t pe = a_off[i + 1], qe = b_off[j + 1];
    float acc = 0.0f;
    while (p < pe && q < qe) {
        const int32_t x = a_ids[p], y = b_ids[q];
        if (x == y) {
            const float term = B_HAS_VALUES ? a_val[p] * b_val[q] : a_val[p];
            acc = acc + term;
            ++p;
            ++q;
        } else if (x < y) {
            ++p;
        } else {
            ++q;
        }
    }
    out[i * ld_out + j] = acc;
}

struct Args {
    const sdrb_batch* b;
    const f32x4* table;
    uint32_t ntiles, tiles_per_block, na;
    dim3 grid;
    float* out;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int NCH, int WCAP, bool BV>
void launch(const Args& g) {
    hipLaunchKernelGGL((k_inner_table<NCH, WCAP, BV>), g.grid, dim3(kBlock), 0, g.stream, g.b->d_offsets, g.b->d_ids,
                       g.b->d_data, g.b->n_rows, g.table, g.ntiles, g.tiles_per_block, g.na, g.out, g.ld_out);
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

bool table_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    if (b->width <= 0 || b->width > 4096 || b->max_row_len > 64) return false;
    // every id of A must have a table entry
    if (a->nnz && (a->min_id < 0 || a->max_id >= (b->width <= 2048 ? 2048 : 4096))) return false;
    return true;
}

}  // namespace

int launch_inner(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, float* out, uint64_t ld_out,
                 hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const bool want_sparse = (flags & SDRB_FLAG_FORCE_SPARSE) != 0, want_dense = (flags & SDRB_FLAG_FORCE_ROWTABLE) != 0;
    const bool want_merge = (flags & SDRB_FLAG_FORCE_MERGE) != 0;
    if (want_sparse) {
        if (!inner_sparse_applicable(a, b, true)) {
            set_error("sdrb_inner: the match-visiting kernel needs a declared width <= 4096, rows of b <= 64 ids and rows of a <= 128 ids");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_inner_sparse(a, b, out, ld_out, stream);
    }
    if (want_dense && !table_applicable(a, b)) {
        set_error("sdrb_inner: the value-table kernel needs a declared width <= 4096 and rows of b <= 64 ids");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (!want_dense && !want_merge && a->n_rows <= kFewRows && b->width > 0 && b->max_row_len <= 64 && a->max_row_len < 8192) {
        const uint32_t width = (uint32_t)b->width;
        const uint32_t longest = a->max_row_len < width ? a->max_row_len : width;
        const uint32_t val_cap = (uint32_t)a->n_rows * (longest ? longest : 1u);   // < 65536: col[] is 16 bit
        const size_t lds = (size_t)val_cap * 4 + (size_t)((width + 4) / 4) * 4 + (size_t)(width + 2) * 2;
        if (lds <= 64 * 1024 - 256) {
            const uint32_t quads = (b->max_row_len + 3) / 4;
            const uint64_t pieces = ((uint64_t)b->n_rows + kBlock - 1) / kBlock;
            const uint32_t per_cu = lds <= 8 * 1024 ? 8u : lds <= 16 * 1024 ? 6u : lds <= 32 * 1024 ? 4u : 2u;
            const uint32_t grid = (uint32_t)(pieces < 256ull * per_cu ? pieces : 256ull * per_cu);
#define SDRB_INNER_FEW_NV(N, V)                                                                                    \
    hipLaunchKernelGGL((k_inner_few<N, V>), dim3(grid), dim3(kBlock), lds, stream, a->d_offsets, a->d_ids, a->d_data, \
                       (uint32_t)a->n_rows, b->d_offsets, b->d_ids, b->d_data, (uint64_t)b->n_rows, width, val_cap,   \
                       out, ld_out)
#define SDRB_INNER_FEW(N)                         \
    do {                                          \
        if (b->d_data) SDRB_INNER_FEW_NV(N, true); \
        else SDRB_INNER_FEW_NV(N, false);          \
    } while (0)
            if (quads <= 4) SDRB_INNER_FEW(4);
            else if (quads <= 10) SDRB_INNER_FEW(10);
            else SDRB_INNER_FEW(16);
#undef SDRB_INNER_FEW
#undef SDRB_INNER_FEW_NV
            SDRB_HIP(hipGetLastError());
            return SDRB_OK;
        }
    }
    if (!want_dense && !want_merge && inner_sparse_applicable(a, b, false)) return launch_inner_sparse(a, b, out, ld_out, stream);
    if (want_merge || !table_applicable(a, b)) {
        const dim3 grid((uint32_t)((b->n_rows + kBlock - 1) / kBlock), (uint32_t)a->n_rows);
        if (a->n_rows > 65535) {
            set_error("sdrb_inner: more than 65535 rows of A need a declared width <= 4096 and rows of B <= 64 ids");
            return SDRB_ERR_UNSUPPORTED;
        }
        if (b->d_data)
            hipLaunchKernelGGL(k_inner_merge<true>, grid, dim3(kBlock), 0, stream, a->d_offsets, a->d_ids, a->d_data,
                               b->d_offsets, b->d_ids, b->d_data, b->n_rows, out, ld_out);
        else
            hipLaunchKernelGGL(k_inner_merge<false>, grid, dim3(kBlock), 0, stream, a->d_offsets, a->d_ids, a->d_data,
                               b->d_offsets, b->d_ids, b->d_data, b->n_rows, out, ld_out);
        SDRB_HIP(hipGetLastError());
        return SDRB_OK;
    }
    const int wcap = b->width <= 2048 ? 2048 : 4096;
    const int nch = b->max_row_len <= 16 ? 2 : b->max_row_len <= 32 ? 4 : b->max_row_len <= 40 ? 5 : 8;
    Args g;
    g.b = b;
    g.na = (uint32_t)a->n_rows;
    g.ntiles = (uint32_t)((a->n_rows + 3) / 4);
    const size_t table_bytes = (size_t)g.ntiles * wcap * sizeof(f32x4);
    float* table = static_cast<float*>(workspace(b->device, stream, 5, table_bytes));
    if (!table) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(table, 0, table_bytes, stream));
    hipLaunchKernelGGL(k_build_value_table, dim3((uint32_t)a->n_rows), dim3(64), 0, stream, a->d_offsets, a->d_ids,
                       a->d_data, (uint32_t)wcap, table);
    SDRB_HIP(hipGetLastError());
    g.table = reinterpret_cast<const f32x4*>(table);
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 1024 ? 1u : (1024u + gx - 1) / gx;
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
