// How a match's payloads decide whether it is kept, shared by the mask-table kernels (mask_table.hip) and the
// lane-per-row match kernels (lane_rows.hip).  A = the batch whose rows are tabulated, B = the streamed batch;
// which of them is `this` is folded into the mode (UnitData.hpp:30-120, EmptyData.hpp:44-80).
#pragma once

#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

enum Math { kKeepAll, kDropAll, kSelfA, kSelfB, kMul, kAbsDiff, kRmulA, kRmulB };

template <int MATH>
__device__ __forceinline__ bool match_kept_by(float va, float vb, float& d) {
    d = 0.0f;
    if constexpr (MATH == kKeepAll) return true;
    else if constexpr (MATH == kDropAll) return false;
    else {
        if constexpr (MATH == kSelfA) d = va;
        else if constexpr (MATH == kSelfB) d = vb;
        else if constexpr (MATH == kMul) d = va * vb;
        else if constexpr (MATH == kAbsDiff) d = fabsf(va - vb);
        else if constexpr (MATH == kRmulA) { const float om = 1.0f - vb; d = va * om; }
        else { const float om = 1.0f - va; d = vb * om; }
        return unit_relevant(d);
    }
}

__device__ __forceinline__ uint32_t size_of(int op, uint32_t l_this, uint32_t l_arg, uint32_t m, uint32_t k) {
    if (op == SDRB_OP_AND) return k;
    if (op == SDRB_OP_OR) return l_this + l_arg - m;
    if (op == SDRB_OP_XOR) return l_this + l_arg - 2 * m + k;
    return l_this - m + k;
}

// the mode of op for `this`/`arg` = A/B (swap: B/A)
inline int pick_math(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const bool au = a->data_kind == SDRB_DATA_UNIT, bu = b->data_kind == SDRB_DATA_UNIT;
    const bool this_unit = swap ? bu : au, arg_unit = swap ? au : bu;
    if (op == SDRB_OP_OR) return kKeepAll;
    if (!this_unit) return op == SDRB_OP_AND ? kKeepAll : kDropAll;
    if (!arg_unit) return swap ? kSelfB : kSelfA;
    if (op == SDRB_OP_AND) return kMul;
    if (op == SDRB_OP_XOR) return kAbsDiff;
    return swap ? kRmulB : kRmulA;
}

}  // namespace

}  // namespace sdrb
