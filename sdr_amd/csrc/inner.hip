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

int launch_inner(const sdrb_batch* a, const sdrb_batch* b, float* out, uint64_t ld_out, hipStream_t stream) {
    if (!table_applicable(a, b)) {
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
    float* table = static_cast<float*>(workspace(b->device, 5, table_bytes));
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
