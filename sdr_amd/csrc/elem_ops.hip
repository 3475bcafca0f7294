// Element operations ande/ore/xore/rme over every pair of two batches:
// count (a size op) -> exclusive scan -> fill.  This file holds the scan (rocPRIM's single-pass
// device scan, in place) and the general fill.
//
// Fill: one lane owns pair (i, j) and walks row i of A against row j of B with the
// three-way merge of the reference's union visitor orv (SDR.hpp:1099-1150), emitting
//   ande: kept matches only                                  (SDR.hpp:1015-1021)
//   ore : this-only, converted arg-only, every match          (SDR.hpp:1207-1217)
//   xore: this-only, converted arg-only, kept matches         (SDR.hpp:1332-1344)
//   rme : this-only and kept matches                          (SDR.hpp:1540-1560)
// in ascending id order at the offset the scan assigned to the pair.  An arg-only element
// is converted to the result's element type (SDRElem.hpp:29-30): an EmptyData argument
// becomes UnitData 1.0 (UnitData.hpp:13-14), a UnitData argument loses its payload in an
// EmptyData result (EmptyData.hpp:25-27).
#include "sdrb_device.h"
#include <rocprim/device/device_scan.hpp>

#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;

// VISIT (ids only, ande / ore): instead of elements, the positions a visitor of andv / orv would be called
// with -- out_ids (as uint32) gets the position in `this`, out_data (as uint32) the position in `arg`,
// 0xFFFFFFFF for the absent side of a single.
template <typename ID, int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT = false, bool ARITH = false>
__global__ __launch_bounds__(kBlock) void k_elem_fill(
    const uint32_t* __restrict__ a_off, const ID* __restrict__ a_ids,
    const float* __restrict__ a_data, uint64_t na, const uint32_t* __restrict__ b_off,
    const ID* __restrict__ b_ids, const float* __restrict__ b_data, uint64_t nb,
    const uint32_t* __restrict__ out_off, ID* __restrict__ out_ids,
    float* __restrict__ out_data, uint32_t rowwise) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool KEEP_THIS_ONLY = OP != SDRB_OP_AND;
    constexpr bool KEEP_ARG_ONLY = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nb) return;
    const uint32_t b_beg = b_off[j];
    const uint32_t lb = b_off[j + 1] - b_beg;
    const ID* bi = b_ids + b_beg;
    const float* bd = B_UNIT ? b_data + b_beg : nullptr;

    // all pairs, or row-wise: row j of A only (result j)
    for (uint64_t i = rowwise ? j : blockIdx.y; i < na; i += rowwise ? na : gridDim.y) {
        const uint32_t a_beg = a_off[i];
        const uint32_t la = a_off[i + 1] - a_beg;
        const ID* ai = a_ids + a_beg;
        const float* ad = A_UNIT ? a_data + a_beg : nullptr;
        uint32_t w = out_off[rowwise ? j : i * nb + j];
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        // a visitor's call: positions in A_i and B_j (NONE = absent), stored as (this, arg)
        auto emit_pos = [&](uint32_t pa, uint32_t pb) {
            reinterpret_cast<uint32_t*>(out_ids)[w] = SWAP ? pb : pa;
            reinterpret_cast<uint32_t*>(out_data)[w] = SWAP ? pa : pb;
            ++w;
        };
        auto emit = [&](ID id, float d) {
            out_ids[w] = id;
            if constexpr (THIS_UNIT) out_data[w] = d;
            ++w;
        };
        // an element present only in A / only in B
        auto only_a = [&](uint32_t pa) {
            if constexpr (VISIT) {   // orv's this/arg visitor; andv has none
                if constexpr (OP == SDRB_OP_OR) emit_pos(pa, NONE);
                return;
            }
            const float va = A_UNIT ? ad[pa] : 0.0f;
            if constexpr (!SWAP) {
                if constexpr (KEEP_THIS_ONLY) emit(ai[pa], va);
            } else {
                if constexpr (KEEP_ARG_ONLY) emit(ai[pa], converted_arg<ARG_UNIT, ARITH>(va));
            }
        };
        auto only_b = [&](uint32_t pb) {
            if constexpr (VISIT) {
                if constexpr (OP == SDRB_OP_OR) emit_pos(NONE, pb);
                return;
            }
            const float vb = B_UNIT ? bd[pb] : 0.0f;
            if constexpr (SWAP) {
                if constexpr (KEEP_THIS_ONLY) emit(bi[pb], vb);
            } else {
                if constexpr (KEEP_ARG_ONLY) emit(bi[pb], converted_arg<ARG_UNIT, ARITH>(vb));
            }
        };
        uint32_t pa = 0, pb = 0;
        while (pa < la && pb < lb) {
            const ID x = ai[pa];
            const ID y = bi[pb];
            if (x == y) {
                if constexpr (VISIT) {
                    emit_pos(pa, pb);
                } else {
                    const float va = A_UNIT ? ad[pa] : 0.0f;
                    const float vb = B_UNIT ? bd[pb] : 0.0f;
                    const float d = SWAP ? match_data<OP, ARG_UNIT, ARITH>(vb, va) : match_data<OP, ARG_UNIT, ARITH>(va, vb);
                    if (match_kept<OP, THIS_UNIT, ARITH>(d)) emit(x, d);
                }
                ++pa;
                ++pb;
            } else if (x < y) {
                only_a(pa++);
            } else {
                only_b(pb++);
            }
        }
        if constexpr (SWAP ? KEEP_ARG_ONLY : KEEP_THIS_ONLY) {
            while (pa < la) only_a(pa++);
        }
        if constexpr (SWAP ? KEEP_THIS_ONLY : KEEP_ARG_ONLY) {
            while (pb < lb) only_b(pb++);
        }
    }
}

struct FillArgs {
    const sdrb_batch* a;
    const sdrb_batch* b;
    const uint32_t* out_off;
    void* out_ids;
    float* out_data;
    bool rowwise;
    hipStream_t stream;
};

// ArithData<float> payloads (32-bit ids; `this` carries the payloads -- an EmptyData `this` is served as ids only)
template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
void launch_arith(const FillArgs& g) {
    const uint32_t gx = (uint32_t)((g.b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = g.rowwise ? 1u : (uint32_t)(g.a->n_rows < 65535 ? g.a->n_rows : 65535);
    hipLaunchKernelGGL((k_elem_fill<int32_t, OP, SWAP, A_UNIT, B_UNIT, false, true>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                       g.a->d_offsets, g.a->d_ids, g.a->d_data, g.a->n_rows, g.b->d_offsets, g.b->d_ids,
                       g.b->d_data, g.b->n_rows, g.out_off, static_cast<int32_t*>(g.out_ids), g.out_data,
                       g.rowwise ? 1u : 0u);
}

template <int OP>
void launch_arith_kinds(const FillArgs& g, bool swap) {
    const bool ap = g.a->data_kind != SDRB_DATA_EMPTY, bp = g.b->data_kind != SDRB_DATA_EMPTY;
    if (ap && bp) {
        if (swap) launch_arith<OP, true, true, true>(g);
        else launch_arith<OP, false, true, true>(g);
    } else if (ap) {
        launch_arith<OP, false, true, false>(g);   // this = A (payloads), arg = B (ids only)
    } else {
        launch_arith<OP, true, false, true>(g);    // this = B (payloads), arg = A (ids only)
    }
}

template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
void launch(const FillArgs& g) {
    const uint32_t gx = (uint32_t)((g.b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = g.rowwise ? 1u : (uint32_t)(g.a->n_rows < 65535 ? g.a->n_rows : 65535);
    if (g.a->id_bits == 64)
        hipLaunchKernelGGL((k_elem_fill<long long, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                           g.a->d_offsets, reinterpret_cast<const long long*>(g.a->d_ids), g.a->d_data, g.a->n_rows,
                           g.b->d_offsets, reinterpret_cast<const long long*>(g.b->d_ids), g.b->d_data, g.b->n_rows,
                           g.out_off, static_cast<long long*>(g.out_ids), g.out_data, g.rowwise ? 1u : 0u);
    else
        hipLaunchKernelGGL((k_elem_fill<int32_t, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                           g.a->d_offsets, g.a->d_ids, g.a->d_data, g.a->n_rows, g.b->d_offsets, g.b->d_ids,
                           g.b->d_data, g.b->n_rows, g.out_off, static_cast<int32_t*>(g.out_ids), g.out_data,
                           g.rowwise ? 1u : 0u);
}

template <int OP, bool SWAP>
void launch_kinds(const FillArgs& g) {
    const bool au = g.a->data_kind == SDRB_DATA_UNIT, bu = g.b->data_kind == SDRB_DATA_UNIT;
    if (au && bu) launch<OP, SWAP, true, true>(g);
    else if (au) launch<OP, SWAP, true, false>(g);
    else if (bu) launch<OP, SWAP, false, true>(g);
    else launch<OP, SWAP, false, false>(g);
}

template <int OP>
void launch_swap(const FillArgs& g, bool swap) {
    if (swap) launch_kinds<OP, true>(g);
    else launch_kinds<OP, false>(g);
}

}  // namespace

namespace {
// 64-bit sum of 32-bit counts (grid-stride, one atomic per block)
__global__ __launch_bounds__(kBlock) void k_sum_u32(const uint32_t* __restrict__ data, uint64_t n,
                                                    unsigned long long* __restrict__ total) {
    __shared__ unsigned long long wave_sum[kBlock / 64];
    unsigned long long mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) mine += data[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long block = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) block += wave_sum[w];
        atomicAdd(total, block);
    }
}
}  // namespace

// exact total of `n` counts, on the host (synchronises the stream)
int sum_u32_on_host(int device, const uint32_t* data, uint64_t n, uint64_t* total, hipStream_t stream) {
    *total = 0;
    if (n == 0) return SDRB_OK;
    unsigned long long* acc = static_cast<unsigned long long*>(workspace(device, 1, sizeof(unsigned long long)));
    if (!acc) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(acc, 0, sizeof(unsigned long long), stream));
    const uint64_t blocks = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_sum_u32, dim3((uint32_t)(blocks < 4096 ? blocks : 4096)), dim3(kBlock), 0, stream, data, n, acc);
    SDRB_HIP(hipGetLastError());
    unsigned long long host = 0;
    SDRB_HIP(hipMemcpyAsync(&host, acc, sizeof(host), hipMemcpyDeviceToHost, stream));
    SDRB_HIP(hipStreamSynchronize(stream));
    *total = host;
    return SDRB_OK;
}

int launch_exclusive_scan_u32(int device, uint32_t* data, uint64_t n, hipStream_t stream) {
    if (n == 0) return SDRB_OK;
    // single-pass scan (decoupled look-back, rocPRIM): the array is read once and written once, in place
    // 32 items a thread: measured ahead of the library's default on 16.7 M offsets (the scan inside a 16-query
    // ande over 1 Mi rows: 304 -> 282 us for the whole operation)
    using Cfg = rocprim::scan_config<256, 32, rocprim::block_load_method::block_load_transpose,
                                     rocprim::block_store_method::block_store_transpose,
                                     rocprim::block_scan_algorithm::using_warp_scan>;
    size_t temp_bytes = 0;
    SDRB_HIP(rocprim::exclusive_scan<Cfg>(nullptr, temp_bytes, data, data, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    void* temp = workspace(device, 1, temp_bytes ? temp_bytes : 1);
    if (!temp) return SDRB_ERR_HIP;
    SDRB_HIP(rocprim::exclusive_scan<Cfg>(temp, temp_bytes, data, data, 0u, (size_t)n, rocprim::plus<uint32_t>(), stream));
    return SDRB_OK;
}

namespace {
template <typename ID, int OP, bool SWAP>
void launch_visit(const FillArgs& g, uint32_t gx, uint32_t gy) {
    hipLaunchKernelGGL((k_elem_fill<ID, OP, SWAP, false, false, true>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                       g.a->d_offsets, reinterpret_cast<const ID*>(g.a->d_ids), nullptr, g.a->n_rows, g.b->d_offsets,
                       reinterpret_cast<const ID*>(g.b->d_ids), nullptr, g.b->n_rows, g.out_off,
                       static_cast<ID*>(g.out_ids), g.out_data, g.rowwise ? 1u : 0u);
}
}  // namespace

// op: SDRB_OP_AND (andv) or SDRB_OP_OR (orv); this_pos / arg_pos: uint32 positions, any id width
int launch_elem_visit(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                      uint32_t* this_pos, uint32_t* arg_pos, bool rowwise, hipStream_t stream) {
    // the kernel indexes its id output by the id type: for 64-bit ids positions are written through a
    // 32-bit view, which emit_pos does itself, so the output pointers are passed untyped
    FillArgs g{a, b, out_offsets, this_pos, reinterpret_cast<float*>(arg_pos), rowwise, stream};
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = rowwise ? 1u : (uint32_t)(a->n_rows < 65535 ? a->n_rows : 65535);
    const bool wide = a->id_bits == 64, is_and = op == SDRB_OP_AND;
    if (wide) {
        if (is_and) { if (swap) launch_visit<long long, SDRB_OP_AND, true>(g, gx, gy); else launch_visit<long long, SDRB_OP_AND, false>(g, gx, gy); }
        else { if (swap) launch_visit<long long, SDRB_OP_OR, true>(g, gx, gy); else launch_visit<long long, SDRB_OP_OR, false>(g, gx, gy); }
    } else {
        if (is_and) { if (swap) launch_visit<int32_t, SDRB_OP_AND, true>(g, gx, gy); else launch_visit<int32_t, SDRB_OP_AND, false>(g, gx, gy); }
        else { if (swap) launch_visit<int32_t, SDRB_OP_OR, true>(g, gx, gy); else launch_visit<int32_t, SDRB_OP_OR, false>(g, gx, gy); }
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

int launch_elem_fill(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                     const uint32_t* out_offsets, void* out_ids, float* out_data, bool rowwise,
                     hipStream_t stream) {
    FillArgs g{a, b, out_offsets, out_ids, out_data, rowwise, stream};
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) {   // caller: `this` is the ArithData side
        switch (op) {
        case SDRB_OP_AND: launch_arith_kinds<SDRB_OP_AND>(g, swap); break;
        case SDRB_OP_OR: launch_arith_kinds<SDRB_OP_OR>(g, swap); break;
        case SDRB_OP_RM: launch_arith_kinds<SDRB_OP_RM>(g, swap); break;
        default: set_error("ArithData has no xor (ArithData.hpp:110-114)"); return SDRB_ERR_UNSUPPORTED;
        }
        SDRB_HIP(hipGetLastError());
        return SDRB_OK;
    }
    switch (op) {
    case SDRB_OP_AND: launch_swap<SDRB_OP_AND>(g, swap); break;
    case SDRB_OP_OR: launch_swap<SDRB_OP_OR>(g, swap); break;
    case SDRB_OP_XOR: launch_swap<SDRB_OP_XOR>(g, swap); break;
    case SDRB_OP_RM: launch_swap<SDRB_OP_RM>(g, swap); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
