// Vertical (bit-sliced) counters: a counter per bit column of a 32-bit word, plane p holding bit p of all 32
// counters, and the carry-save adders that add one-bit-per-column words into them.  Shared by the bit-sliced
// table kernels (bitslice_device.h) and the lane-per-row size kernel (lane_rows.hip).
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace sdrb {

namespace {

constexpr int bits_for(int n) {  // bits needed to hold values 0..n
    int b = 0;
    while (n > 0) { ++b; n >>= 1; }
    return b;
}

// full adder over 32 independent 1-bit columns: carry = majority, sum = parity
__device__ __forceinline__ void csa(uint32_t& carry, uint32_t& sum, uint32_t a, uint32_t b, uint32_t c) {
    carry = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
    sum = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ void half_add(uint32_t& carry, uint32_t& sum, uint32_t a, uint32_t b) {
    carry = a & b;
    sum = a ^ b;
}

__device__ __forceinline__ uint32_t rotr32(uint32_t x, int s) {
    return __builtin_amdgcn_alignbit(x, x, (uint32_t)(s & 31));
}

// Planes -> bytes.  acc[r] gets, in byte k, the count of column 8k + r: bit p of that byte is bit 8k + r of plane p.
// That is the transpose of an 8 x 8 bit matrix (planes down, columns of a byte across) in each of the word's four bytes
// at once: three rounds of block swaps between registers 4, 2 and 1 apart (planes past NPL are zero, and what only
// moves zeros folds away).  About 50 instructions for six planes, against two a (plane, r) term -- 96 -- when every
// term is shifted and masked on its own.
__device__ __forceinline__ void swap_bits(uint32_t& a, uint32_t& b, int d, uint32_t m) {
    const uint32_t t = ((a >> d) ^ b) & m;
    b ^= t;
    a ^= t << d;
}
template <int NPL>
__device__ __forceinline__ void planes_to_bytes(const uint32_t (&pl)[NPL], uint32_t (&acc)[8]) {
    static_assert(NPL <= 8, "a byte holds eight planes");
#pragma unroll
    for (int p = 0; p < 8; ++p) acc[p] = p < NPL ? pl[p] : 0u;
#pragma unroll
    for (int p = 0; p < 4; ++p) swap_bits(acc[p], acc[p + 4], 4, 0x0F0F0F0Fu);
#pragma unroll
    for (int p = 0; p < 8; ++p)
        if ((p & 2) == 0) swap_bits(acc[p], acc[p + 2], 2, 0x33333333u);
#pragma unroll
    for (int p = 0; p < 8; p += 2) swap_bits(acc[p], acc[p + 1], 1, 0x55555555u);
}

// Adds eight 32-column bit words into the vertical counter P (Harley-Seal step):
// P[0..2] keep count mod 8, the weight-8 carry ripples into P[3..].
template <int NPL, bool FIRST>
__device__ __forceinline__ void add8(uint32_t (&P)[NPL], const uint32_t (&x)[8]) {
    uint32_t t2a, t2b, t4a, t4b, e8;
    if constexpr (FIRST) {  // P is all zero
        half_add(t2a, P[0], x[0], x[1]);
        csa(t2b, P[0], P[0], x[2], x[3]);
        half_add(t4a, P[1], t2a, t2b);
        csa(t2a, P[0], P[0], x[4], x[5]);
        csa(t2b, P[0], P[0], x[6], x[7]);
        csa(t4b, P[1], P[1], t2a, t2b);
        half_add(e8, P[2], t4a, t4b);
        if constexpr (NPL > 3) P[3] = e8;
    } else {
        csa(t2a, P[0], P[0], x[0], x[1]);
        csa(t2b, P[0], P[0], x[2], x[3]);
        csa(t4a, P[1], P[1], t2a, t2b);
        csa(t2a, P[0], P[0], x[4], x[5]);
        csa(t2b, P[0], P[0], x[6], x[7]);
        csa(t4b, P[1], P[1], t2a, t2b);
        csa(e8, P[2], P[2], t4a, t4b);
#pragma unroll
        for (int p = 3; p < NPL; ++p) {
            const uint32_t nc = P[p] & e8;
            P[p] ^= e8;
            e8 = nc;
        }
    }
}

}  // namespace

}  // namespace sdrb
