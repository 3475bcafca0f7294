// Device-side payload semantics shared by the merge and element kernels.
//
// All arithmetic is single IEEE binary32 operations (compile with -ffp-contract=off, no
// fast-math), mirroring DataTypes/UnitData.hpp of the reference:
//   and: a*b (:37-41)   or: a>b?a:b (:60-64)   xor: |a-b| (:83-87)   rm: a*(1-b) (:106-110)
// and identity when the argument is EmptyData (:43-46,66-69,89-92,112-115).
// relevant()/rm_relevant() compare `value_ >= 0.1` in double (:22-28); since
// 0.1f > 0.1 > nextbelow(0.1f) that is exactly `value_ >= 0.1f` in binary32.
// EmptyData (DataTypes/EmptyData.hpp:29-37): relevant() always, rm_relevant() never.
// ArithData<float> (DataTypes/ArithData.hpp, template flag ARITH): and: a*b (:37-43)  or: a+b (:57-63)
// rm: a-b (:77-83), no xor (:110-114); relevant() and rm_relevant() always (:22-28); ArithData(EmptyData)
// is 0 (:15-16) where UnitData(EmptyData) is 1 (UnitData.hpp:13-14).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/sdr_amd.h"

namespace sdrb {

__device__ __forceinline__ bool unit_relevant(float v) { return v >= 0.1f; }

// Payload of a match for set operation OP, `this` payload t, `arg` payload a.
// ARG_UNIT: the argument carries a payload; ARITH: payloads are ArithData<float>, else UnitData.
template <int OP, bool ARG_UNIT, bool ARITH = false>
__device__ __forceinline__ float match_data(float t, float a) {
    if constexpr (!ARG_UNIT) {
        return t;
    } else if constexpr (OP == SDRB_OP_AND) {
        return t * a;
    } else if constexpr (OP == SDRB_OP_OR) {
        if constexpr (ARITH) return t + a;
        else return t > a ? t : a;
    } else if constexpr (OP == SDRB_OP_XOR) {
        return fabsf(t - a);   // UnitData only: the C ABI refuses xor on ArithData
    } else {
        if constexpr (ARITH) {
            return t - a;
        } else {
            float one_minus = 1.0f - a;
            return t * one_minus;
        }
    }
}

// Whether a matched id stays in the result of OP (SDR.hpp:1017, 1213-1217, 1340, 1545).
// THIS_UNIT: `this` carries a payload.
template <int OP, bool THIS_UNIT, bool ARITH = false>
__device__ __forceinline__ bool match_kept(float d) {
    if constexpr (OP == SDRB_OP_OR || (ARITH && THIS_UNIT)) {
        return true;
    } else if constexpr (OP == SDRB_OP_AND) {
        return THIS_UNIT ? unit_relevant(d) : true;
    } else {
        return THIS_UNIT ? unit_relevant(d) : false;
    }
}

// Payload of an arg-only element converted into a result that carries payloads (SDRElem.hpp:29-30).
template <bool ARG_UNIT, bool ARITH = false>
__device__ __forceinline__ float converted_arg(float a) {
    return ARG_UNIT ? a : (ARITH ? 0.0f : 1.0f);
}

// Result length of OP from |this|, |arg|, id matches m and kept matches k
// (SDR.hpp:1094, 1303, 1433, 1576).
template <int OP>
__device__ __forceinline__ uint32_t size_from_counts(uint32_t l_this, uint32_t l_arg, uint32_t m,
                                                     uint32_t k) {
    if constexpr (OP == SDRB_OP_AND) return k;
    else if constexpr (OP == SDRB_OP_OR) return l_this + l_arg - m;
    else if constexpr (OP == SDRB_OP_XOR) return l_this + l_arg - 2 * m + k;
    else return l_this - m + k;
}

}  // namespace sdrb
