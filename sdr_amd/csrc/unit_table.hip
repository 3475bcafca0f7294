// Wave-per-row table kernels for operations that need payloads (UnitData on the `this` side)
// or rows too long / widths too large for the bit-sliced and bitset kernels.
//
// The reference finds the common ids of a pair with andv (SDR.hpp:958-997) and hands each
// match to a visitor that combines the two payloads (SDR.hpp:1015-1021, 1087-1091,
// 1428-1431, 1540-1548, 1572-1574).  Here a block keeps, for ONE row i of A at a time, a
// position table in LDS
//
//     pos[w] (uint16) = k+1 if ids_A[i][k] == w, else 0          (width <= 32768 -> <= 64 KiB)
//
// plus that row's payloads, and each wave owns a few rows of B with one lane per element.
// Rows of up to 512 elements are loaded ONCE into registers (ids and payloads) and stay there
// while the block streams over the rows of A, so the inner loop touches only LDS: one lookup
// per lane, a payload read for the few lanes that match, the single IEEE multiply / subtract
// of the op (sdrb_device.h), and wave ballots that turn keep-flags into counts (size ops) or
// write positions (ande; rme with the row of B as `this`).  The table is switched from row i
// to row i+1 of A by clearing and setting only that row's entries.  Longer rows are streamed
// from memory by the same code (NIT = 0).
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxLaLds = 2048;  // payloads of A's current row kept in LDS up to this length

// How a match's payload decides whether it is kept; `t` is the payload of `this`.
enum Math {
    kKeepAll,   // no payload involved, every id match kept      (ande/ands with `this` EmptyData, ors)
    kDropAll,   // no payload involved, every id match dropped   (xor/rm with `this` EmptyData)
    kSelfA,     // arg is EmptyData: d = t, t from A             (UnitData.hpp:43-46,89-92,112-115)
    kSelfB,     // arg is EmptyData: d = t, t from B
    kMul,       // d = t * a                                     (UnitData.hpp:37-41)
    kAbsDiff,   // d = |t - a|                                   (UnitData.hpp:83-87)
    kRmulA,     // d = t * (1 - a), t from A                     (UnitData.hpp:106-110)
    kRmulB,     // d = t * (1 - a), t from B
};

// rows of B per wave for a given number of 64-element register chunks per row
__host__ __device__ constexpr int rows_per_wave(int nit) { return nit == 1 ? 8 : nit == 2 ? 8 : 4; }

template <int MATH>
__device__ __forceinline__ bool match_result(float va, float vb, float& d) {
    if constexpr (MATH == kKeepAll) { d = 0.0f; return true; }
    else if constexpr (MATH == kDropAll) { d = 0.0f; return false; }
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

// FILL == false: size ops.  FILL == true with RM_ROWS == false: ande; RM_ROWS == true: rme with
// the rows of B as `this` (unmatched elements pass through).
template <bool FILL, bool RM_ROWS, int MATH, int NIT>
__global__ __launch_bounds__(kBlock) void k_unit_table(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids,
    const float* __restrict__ a_data, uint32_t na, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, const float* __restrict__ b_data, uint32_t nb, uint32_t nnz_b,
    uint32_t width, uint32_t rows_a_per_block, int op, uint32_t swap,
    uint32_t* __restrict__ out_counts, uint64_t ld_out,          // size ops
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids,
    float* __restrict__ out_data) {                               // element ops
    constexpr int RW = rows_per_wave(NIT);
    constexpr int IT = NIT == 0 ? 1 : NIT;
    constexpr bool NEED_A = MATH == kSelfA || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    constexpr bool NEED_B = MATH == kSelfB || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB ||
                            (RM_ROWS && MATH != kDropAll && MATH != kKeepAll);
    constexpr bool OUT_DATA = FILL && MATH != kKeepAll && MATH != kDropAll;
    // LDS: qdata[kMaxLaLds] floats, then pos[width + 2] (pos[width] stays 0: the sentinel)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* qdata = reinterpret_cast<float*>(lds_raw);
    uint16_t* pos = reinterpret_cast<uint16_t*>(lds_raw + kMaxLaLds * sizeof(float));

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t j_wave = (blockIdx.x * kWaves + wave) * RW;
    const uint32_t i0 = blockIdx.y * rows_a_per_block;
    const uint32_t i1 = i0 + rows_a_per_block < na ? i0 + rows_a_per_block : na;

    for (uint32_t w = tid * 2; w < width + 2; w += kBlock * 2) *reinterpret_cast<uint32_t*>(&pos[w]) = 0;

    // this wave's rows of B -> registers, through per-row bounds-checked buffers: lanes past
    // the end of a row read nothing and get the sentinel id `width` (never in the table)
    uint32_t b_beg[RW], lb[RW];
    int32_t rid[RW][IT];
    float rval[RW][IT];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const uint32_t j = j_wave + r;
        const uint32_t beg = j < nb ? b_off[j] : nnz_b;
        const uint32_t end = j < nb ? b_off[j + 1] : nnz_b;
        b_beg[r] = __builtin_amdgcn_readfirstlane(beg);
        lb[r] = __builtin_amdgcn_readfirstlane(end - beg);
    }
    if constexpr (NIT > 0) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<int32_t*>(b_ids) + b_beg[r], 0, (int)(lb[r] * 4u), 0x00020000);
            __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(b_data) + (NEED_B ? b_beg[r] : 0u), 0, NEED_B ? (int)(lb[r] * 4u) : 0, 0x00020000);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const uint32_t e = it * 64 + lane;
                rid[r][it] = __builtin_amdgcn_raw_buffer_load_b32(ids_rsrc, (int)(e * 4u), 0, 0);
                rval[r][it] = 0.0f;
                if constexpr (NEED_B)
                    rval[r][it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, (int)(e * 4u), 0, 0));
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                if ((uint32_t)(it * 64) + lane >= lb[r]) rid[r][it] = (int32_t)width;
    }
    __syncthreads();

    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t a_beg = a_off[i], la = a_off[i + 1] - a_beg;
        const bool q_in_lds = NEED_A && la <= (uint32_t)kMaxLaLds;
        for (uint32_t k = tid; k < la; k += kBlock) {
            const uint32_t id = (uint32_t)a_ids[a_beg + k];
            if (id < width) pos[id] = (uint16_t)(k + 1);   // ids outside B's width never match
            if constexpr (NEED_A) {
                if (q_in_lds) qdata[k] = a_data[a_beg + k];
            }
        }
        uint32_t base[RW];
        if constexpr (FILL) {
#pragma unroll
            for (int r = 0; r < RW; ++r) base[r] = j_wave + r < nb ? out_off[(uint64_t)i * nb + j_wave + r] : 0u;
        }
        __syncthreads();

#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const uint32_t j = j_wave + r;
            uint32_t m = 0, kept_n = 0, run = 0;
            auto visit = [&](int32_t id, float vb) {
                const uint32_t p = pos[id];
                const bool matched = p != 0;
                float va = 0.0f;
                if constexpr (NEED_A) {
                    const uint32_t k = matched ? p - 1 : 0u;
                    va = q_in_lds ? qdata[k] : (matched ? a_data[a_beg + k] : 0.0f);
                }
                float d;
                const bool kept = match_result<MATH>(va, vb, d) && matched;
                if constexpr (!FILL) {
                    m += (uint32_t)__builtin_popcountll(__ballot(matched));
                    kept_n += (uint32_t)__builtin_popcountll(__ballot(kept));
                } else {
                    const bool valid = (uint32_t)id != width;
                    const bool emit = RM_ROWS ? (valid && (!matched || kept)) : kept;
                    const float value = (!RM_ROWS || matched) ? d : vb;
                    const unsigned long long mask = __ballot(emit);
                    if (emit) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                            (uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
                        out_ids[base[r] + run + rank] = id;
                        if constexpr (OUT_DATA) out_data[base[r] + run + rank] = value;
                    }
                    run += (uint32_t)__builtin_popcountll(mask);
                }
            };
            if constexpr (NIT > 0) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) visit(rid[r][it], rval[r][it]);
            } else {
                for (uint32_t e0 = 0; e0 < lb[r]; e0 += 64) {
                    const uint32_t e = e0 + lane;
                    int32_t id = (int32_t)width;
                    float vb = 0.0f;
                    if (e < lb[r]) {
                        id = b_ids[b_beg[r] + e];
                        if constexpr (NEED_B) vb = b_data[b_beg[r] + e];
                    }
                    visit(id, vb);
                }
            }
            if constexpr (!FILL) {
                if (lane == 0 && j < nb) {
                    const uint32_t l_this = swap ? lb[r] : la, l_arg = swap ? la : lb[r];
                    out_counts[(uint64_t)i * ld_out + j] = size_of(op, l_this, l_arg, m, kept_n);
                }
            }
        }
        __syncthreads();
        for (uint32_t k = tid; k < la; k += kBlock) {
            const uint32_t id = (uint32_t)a_ids[a_beg + k];
            if (id < width) pos[id] = 0;
        }
        __syncthreads();
    }
}

struct Args {
    const sdrb_batch* a;
    const sdrb_batch* b;
    int op;
    bool swap;
    uint32_t* out_counts;
    uint64_t ld_out;
    const uint32_t* out_off;
    int32_t* out_ids;
    float* out_data;
    hipStream_t stream;
};

template <bool FILL, bool RM_ROWS, int MATH, int NIT>
void launch(const Args& g) {
    constexpr int rows_per_block = rows_per_wave(NIT) * kWaves;
    const uint32_t nb = (uint32_t)g.b->n_rows, na = (uint32_t)g.a->n_rows;
    const uint32_t gx = (nb + rows_per_block - 1) / rows_per_block;
    // split A over grid.y only as far as needed to fill the chip
    uint32_t gy = gx >= 2048 ? 1u : (2048u + gx - 1) / gx;
    if (gy > na) gy = na;
    if (gy > 65535u) gy = 65535u;
    const uint32_t per = (na + gy - 1) / gy;
    gy = (na + per - 1) / per;
    const uint32_t width_even = ((uint32_t)g.b->width + 1u) & ~1u;
    const size_t lds = kMaxLaLds * sizeof(float) + (width_even + 2) * sizeof(uint16_t);
    hipLaunchKernelGGL((k_unit_table<FILL, RM_ROWS, MATH, NIT>), dim3(gx, gy), dim3(kBlock), lds, g.stream,
                       g.a->d_offsets, g.a->d_ids, g.a->d_data, na, g.b->d_offsets, g.b->d_ids,
                       g.b->d_data, nb, (uint32_t)g.b->nnz, width_even, per, g.op, g.swap ? 1u : 0u,
                       g.out_counts, g.ld_out, g.out_off, g.out_ids, g.out_data);
}

template <bool FILL, bool RM_ROWS, int MATH>
void launch_nit(const Args& g) {
    const uint32_t len = g.b->max_row_len;
    if (len <= 64) launch<FILL, RM_ROWS, MATH, 1>(g);
    else if (len <= 128) launch<FILL, RM_ROWS, MATH, 2>(g);
    else if (len <= 384) launch<FILL, RM_ROWS, MATH, 6>(g);
    else launch<FILL, RM_ROWS, MATH, 0>(g);
}

// Which payload rule a pair of batches needs for `op` (after the swap flag).
int pick_math(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const bool au = a->data_kind == SDRB_DATA_UNIT, bu = b->data_kind == SDRB_DATA_UNIT;
    const bool this_unit = swap ? bu : au, arg_unit = swap ? au : bu;
    if (op == SDRB_OP_OR) return kKeepAll;
    if (!this_unit) return op == SDRB_OP_AND ? kKeepAll : kDropAll;
    if (!arg_unit) return swap ? kSelfB : kSelfA;
    if (op == SDRB_OP_AND) return kMul;
    if (op == SDRB_OP_XOR) return kAbsDiff;
    return swap ? kRmulB : kRmulA;
}

template <bool FILL, bool RM_ROWS>
void launch_math(const Args& g, int math) {
    switch (math) {
    case kKeepAll: if constexpr (!RM_ROWS) launch_nit<FILL, RM_ROWS, kKeepAll>(g); break;
    case kDropAll: if constexpr (!FILL || RM_ROWS) launch_nit<FILL, RM_ROWS, kDropAll>(g); break;
    case kSelfA: if constexpr (!RM_ROWS) launch_nit<FILL, RM_ROWS, kSelfA>(g); break;
    case kSelfB: launch_nit<FILL, RM_ROWS, kSelfB>(g); break;
    case kMul: if constexpr (!RM_ROWS) launch_nit<FILL, RM_ROWS, kMul>(g); break;
    case kAbsDiff: if constexpr (!FILL) launch_nit<FILL, RM_ROWS, kAbsDiff>(g); break;
    case kRmulA: if constexpr (!FILL) launch_nit<FILL, RM_ROWS, kRmulA>(g); break;
    default: if constexpr (!FILL || RM_ROWS) launch_nit<FILL, RM_ROWS, kRmulB>(g); break;
    }
}

}  // namespace

bool unit_table_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    // 8 KiB of payloads + the position table must fit a 64 KiB dynamic LDS allocation
    return b->width > 0 && b->width <= 28000 && a->max_row_len <= 65534;
}

bool unit_table_fill_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    if (!unit_table_applicable(a, b)) return false;
    return op == SDRB_OP_AND || (op == SDRB_OP_RM && swap);
}

int launch_size_unit_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                           uint32_t* out, uint64_t ld_out, hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    Args g{a, b, op, swap, out, ld_out, nullptr, nullptr, nullptr, stream};
    launch_math<false, false>(g, pick_math(a, b, op, swap));
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

int launch_elem_unit_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                           const uint32_t* out_offsets, int32_t* out_ids, float* out_data,
                           hipStream_t stream) {
    Args g{a, b, op, swap, nullptr, 0, out_offsets, out_ids, out_data, stream};
    if (op == SDRB_OP_AND) {
        launch_math<true, false>(g, pick_math(a, b, op, swap));
    } else if (op == SDRB_OP_RM && swap) {
        launch_math<true, true>(g, pick_math(a, b, op, swap));
    } else {
        set_error("table fill kernel does not cover this op");
        return SDRB_ERR_UNSUPPORTED;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
