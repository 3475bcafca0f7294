// Per-pair merge kernel for the size operations ands/ors/xors/rms.
//
// General path: any signed ids, any row lengths, EmptyData or UnitData on either side.
// One lane owns one row j of the streamed batch B and walks it against row i of batch A
// with a two-pointer merge.  It visits exactly the id matches that the reference's
// intersection visitor andv (SDR.hpp:958-997) visits, in the same ascending order, and
// applies the size visitors of SDR.hpp:1087-1091, 1298-1301, 1428-1431, 1572-1574.
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;

template <typename ID, int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
__global__ __launch_bounds__(kBlock) void k_size_merge(
    const uint32_t* __restrict__ a_off, const ID* __restrict__ a_ids,
    const float* __restrict__ a_data, uint64_t na, const uint32_t* __restrict__ b_off,
    const ID* __restrict__ b_ids, const float* __restrict__ b_data, uint64_t nb,
    uint32_t* __restrict__ out, uint64_t ld_out, uint32_t rowwise) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nb) return;
    const uint32_t b_beg = b_off[j];
    const uint32_t lb = b_off[j + 1] - b_beg;
    const ID* bi = b_ids + b_beg;
    const float* bd = B_UNIT ? b_data + b_beg : nullptr;

    // all pairs: this lane's row against every row of A the block column visits;
    // row-wise: against row j of A only (result j)
    for (uint64_t i = rowwise ? j : blockIdx.y; i < na; i += rowwise ? na : gridDim.y) {
        const uint32_t a_beg = a_off[i];
        const uint32_t la = a_off[i + 1] - a_beg;
        const ID* ai = a_ids + a_beg;
        const float* ad = A_UNIT ? a_data + a_beg : nullptr;
        uint32_t pa = 0, pb = 0, m = 0, k = 0;
        while (pa < la && pb < lb) {
            const ID x = ai[pa];
            const ID y = bi[pb];
            if (x == y) {
                const float va = A_UNIT ? ad[pa] : 0.0f;
                const float vb = B_UNIT ? bd[pb] : 0.0f;
                const float d = SWAP ? match_data<OP, ARG_UNIT>(vb, va) : match_data<OP, ARG_UNIT>(va, vb);
                ++m;
                k += match_kept<OP, THIS_UNIT>(d) ? 1u : 0u;
                ++pa;
                ++pb;
            } else if (x < y) {
                ++pa;
            } else {
                ++pb;
            }
        }
        const uint32_t l_this = SWAP ? lb : la;
        const uint32_t l_arg = SWAP ? la : lb;
        out[(rowwise ? 0 : i * ld_out) + j] = size_from_counts<OP>(l_this, l_arg, m, k);
    }
}

// SDR::operator== (SDR.hpp:470-483) for every pair: equal lengths, equal ids position by position, and --
// only when both sides carry payloads -- equal payloads (float ==, UnitData.hpp:122-124, ArithData.hpp:90-92;
// a payload always equals EmptyData, UnitData.hpp:130-132).  out = 1 / 0.
template <typename ID, bool PAYLOADS>
__global__ __launch_bounds__(kBlock) void k_equal(const uint32_t* __restrict__ a_off, const ID* __restrict__ a_ids,
                                                  const float* __restrict__ a_data, uint64_t na,
                                                  const uint32_t* __restrict__ b_off, const ID* __restrict__ b_ids,
                                                  const float* __restrict__ b_data, uint64_t nb,
                                                  uint32_t* __restrict__ out, uint64_t ld_out, uint32_t rowwise) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nb) return;
    const uint32_t b_beg = b_off[j];
    const uint32_t lb = b_off[j + 1] - b_beg;
    for (uint64_t i = rowwise ? j : blockIdx.y; i < na; i += rowwise ? na : gridDim.y) {
        const uint32_t a_beg = a_off[i];
        bool same = a_off[i + 1] - a_beg == lb;
        for (uint32_t k = 0; same && k < lb; ++k) {
            same = a_ids[a_beg + k] == b_ids[b_beg + k];
            if constexpr (PAYLOADS) same = same && a_data[a_beg + k] == b_data[b_beg + k];
        }
        out[(rowwise ? 0 : i * ld_out) + j] = same ? 1u : 0u;
    }
}

template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
void launch(const sdrb_batch* a, const sdrb_batch* b, uint32_t* out, uint64_t ld_out, bool rowwise,
            hipStream_t stream) {
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = rowwise ? 1u : (uint32_t)(a->n_rows < 65535 ? a->n_rows : 65535);
    if (a->id_bits == 64)   // ids of the reference's `long` element type (src/fuzz.cpp:413-423)
        hipLaunchKernelGGL((k_size_merge<long long, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, stream,
                           a->d_offsets, reinterpret_cast<const long long*>(a->d_ids), a->d_data, a->n_rows,
                           b->d_offsets, reinterpret_cast<const long long*>(b->d_ids), b->d_data, b->n_rows, out,
                           ld_out, rowwise ? 1u : 0u);
    else
        hipLaunchKernelGGL((k_size_merge<int32_t, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, stream,
                           a->d_offsets, a->d_ids, a->d_data, a->n_rows, b->d_offsets, b->d_ids, b->d_data,
                           b->n_rows, out, ld_out, rowwise ? 1u : 0u);
}

template <int OP, bool SWAP>
void launch_kinds(const sdrb_batch* a, const sdrb_batch* b, uint32_t* out, uint64_t ld_out, bool rowwise,
                  hipStream_t stream) {
    const bool au = a->data_kind == SDRB_DATA_UNIT, bu = b->data_kind == SDRB_DATA_UNIT;
    if (au && bu) launch<OP, SWAP, true, true>(a, b, out, ld_out, rowwise, stream);
    else if (au) launch<OP, SWAP, true, false>(a, b, out, ld_out, rowwise, stream);
    else if (bu) launch<OP, SWAP, false, true>(a, b, out, ld_out, rowwise, stream);
    else launch<OP, SWAP, false, false>(a, b, out, ld_out, rowwise, stream);
}

template <int OP>
void launch_swap(const sdrb_batch* a, const sdrb_batch* b, bool swap, uint32_t* out,
                 uint64_t ld_out, bool rowwise, hipStream_t stream) {
    if (swap) launch_kinds<OP, true>(a, b, out, ld_out, rowwise, stream);
    else launch_kinds<OP, false>(a, b, out, ld_out, rowwise, stream);
}

}  // namespace

int launch_equal(const sdrb_batch* a, const sdrb_batch* b, uint32_t* out, uint64_t ld_out, bool rowwise,
                 hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const dim3 grid((uint32_t)((b->n_rows + kBlock - 1) / kBlock),
                    rowwise ? 1u : (uint32_t)(a->n_rows < 65535 ? a->n_rows : 65535));
    const bool payloads = has_payload(a) && has_payload(b);
#define SDRB_EQ(ID, P)                                                                                                    \
    hipLaunchKernelGGL((k_equal<ID, P>), grid, dim3(kBlock), 0, stream, a->d_offsets, reinterpret_cast<const ID*>(a->d_ids), \
                       a->d_data, a->n_rows, b->d_offsets, reinterpret_cast<const ID*>(b->d_ids), b->d_data, b->n_rows, out, \
                       ld_out, rowwise ? 1u : 0u)
    if (a->id_bits == 64) {
        if (payloads) SDRB_EQ(long long, true);
        else SDRB_EQ(long long, false);
    } else {
        if (payloads) SDRB_EQ(int32_t, true);
        else SDRB_EQ(int32_t, false);
    }
#undef SDRB_EQ
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

int launch_size_merge(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out,
                      uint64_t ld_out, bool rowwise, hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    switch (op) {
    case SDRB_OP_AND: launch_swap<SDRB_OP_AND>(a, b, swap, out, ld_out, rowwise, stream); break;
    case SDRB_OP_OR: launch_swap<SDRB_OP_OR>(a, b, swap, out, ld_out, rowwise, stream); break;
    case SDRB_OP_XOR: launch_swap<SDRB_OP_XOR>(a, b, swap, out, ld_out, rowwise, stream); break;
    case SDRB_OP_RM: launch_swap<SDRB_OP_RM>(a, b, swap, out, ld_out, rowwise, stream); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
