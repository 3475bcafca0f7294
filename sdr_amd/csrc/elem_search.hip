// Element operations with one WAVE per pair for ids no table can index: arbitrary signed 32-bit ids, 64-bit ids --
// whatever their values, as in the reference, whose merge never looks at what an id is (SDRElem.hpp:15,56-75,
// SDR.hpp:958-997; `long` ids: src/fuzz.cpp:413-423).
//
// The formulation is the rank-table kernel's (elem_rank.hip): the 64 lanes of a wave walk `this` in id order, 64
// elements a turn, look each id up in `arg`, decide what the reference's merge decides and write the pair's result
// together, every output position computed from ranks:
//   this-only -> copied (not for ande)                         SDR.hpp:1207-1209,1332-1334,1541-1543
//   match     -> payload op, kept as the op says               SDR.hpp:1015-1021,1213-1217,1338-1344,1544-1548
//   pos(t)          = #emitted `this` elements before t  +  rank_arg(t) - #matches before t          (unions)
//   pos(g, arg-only) = rank_this(g) - #dropped matches before g  +  #arg-only elements before g       (unions)
// Without a width there is no bitset to take a rank from, so rank(x) = #{ids < x} is a binary search over the other
// row, which lies in LDS: log2(length) + 2 reads a lane, the same for all 64 (the search is branch-free and its trip
// count depends on the row's length alone).  A lane-per-pair merge of the same rows walks both of them out of global
// memory with one dependent load a step; here a block stages its rows of A once, a wave its row of B once per row.
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

using u64 = unsigned long long;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxRowsOfA = 16;   // rows of A a block stages (their output offsets ride in lanes 0..15)

template <typename T>
__device__ __forceinline__ void lds_store(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <typename T>
__device__ __forceinline__ T lds_load(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// lanes below mine whose flag is set, and the wave's total
__device__ __forceinline__ uint32_t prefix_of(bool flag, uint32_t& total) {
    const u64 m = __builtin_amdgcn_ballot_w64(flag);
    total = (uint32_t)__builtin_popcountll(m);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// rank of x in the ascending row ids[0, len): the number of ids below x; `present` says whether x is in the row (its
// rank is then its position).  Every lane of the wave runs the same number of steps.
template <typename Id>
__device__ __forceinline__ uint32_t rank_in(const Id* ids, uint32_t len, Id x, bool& present) {
    uint32_t base = 0, n = len;
    while (n > 1) {
        const uint32_t half = n >> 1;
        base = lds_load(&ids[base + half - 1]) < x ? base + half : base;
        n -= half;
    }
    present = false;
    if (len == 0) return 0;
    const Id at = lds_load(&ids[base]);
    present = at == x;
    return base + (at < x ? 1u : 0u);
}

// VISIT: instead of elements, the positions a visitor would be called with -- out_ids (as uint32) gets the position in
// `this` and out_data (as uint32) the position in `arg`, 0xFFFFFFFF for the absent side of a single.
template <typename Id, int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT, bool ARITH>
__global__ __launch_bounds__(kBlock) void k_elem_search(
    const uint32_t* __restrict__ a_off, const Id* __restrict__ a_ids, const float* __restrict__ a_data, uint32_t na,
    const uint32_t* __restrict__ b_off, const Id* __restrict__ b_ids, const float* __restrict__ b_data, uint32_t nb,
    const uint32_t* __restrict__ out_off, void* __restrict__ out_ids_any, float* __restrict__ out_data, uint32_t qc,
    uint32_t cap_a, uint32_t cap_b, uint32_t rows_per_wave) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool UNION = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    // matches of a xore can be dropped one by one only when `this` carries payloads
    constexpr bool DROP_BY_VALUE = OP == SDRB_OP_XOR && THIS_UNIT;
    // LDS: ids of the block's rows of A | ids of the waves' rows of B | payloads of A's rows | payloads of B's rows
    //      (cap_a / cap_b elements a row)
    extern __shared__ __attribute__((aligned(16))) unsigned char search_lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Id* a_ids_all = reinterpret_cast<Id*>(search_lds);
    Id* b_ids_l = a_ids_all + (size_t)qc * cap_a + (size_t)wave * cap_b;
    float* a_val_all = reinterpret_cast<float*>(a_ids_all + (size_t)qc * cap_a + (size_t)kWaves * cap_b);
    float* b_val_l = a_val_all + (size_t)qc * cap_a + (size_t)wave * cap_b;
    Id* out_ids = static_cast<Id*>(out_ids_any);
    uint32_t* out_pos = static_cast<uint32_t*>(out_ids_any);

    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t i_begin = blockIdx.y * qc;
    const uint32_t ni = (i_begin + qc < na ? i_begin + qc : na) - i_begin;
    // row offsets of the block's rows of A, one per lane (lanes 0..ni)
    const int a_off_reg = (int)a_off[i_begin + (lane <= ni ? lane : ni)];

    for (uint32_t ii = 0; ii < ni; ++ii) {
        const uint32_t beg = a_off[i_begin + ii], len = a_off[i_begin + ii + 1] - beg;   // len <= cap_a (host)
        for (uint32_t l = threadIdx.x; l < len; l += kBlock) {
            a_ids_all[ii * cap_a + l] = a_ids[beg + l];
            if constexpr (A_UNIT) a_val_all[ii * cap_a + l] = a_data[beg + l];
        }
    }
    __syncthreads();

    for (uint64_t j = j_begin; j < j_end; ++j) {
        const uint32_t b_beg = b_off[j], lb = b_off[j + 1] - b_beg;   // <= cap_b (host)
        // where the pairs of this row of B start, and how long their results are (the offsets hold pairs + 1 entries):
        // a pair whose result is empty -- nearly half the intersections at SDR densities -- is skipped unsearched
        const int base_reg = lane < ni ? (int)out_off[(uint64_t)(i_begin + lane) * nb + j] : 0;
        const int size_reg = lane < ni ? (int)(out_off[(uint64_t)(i_begin + lane) * nb + j + 1] - (uint32_t)base_reg) : 0;
        for (uint32_t k = lane; k < lb; k += 64) {
            lds_store(&b_ids_l[k], b_ids[b_beg + k]);
            if constexpr (B_UNIT) lds_store(&b_val_l[k], b_data[b_beg + k]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        for (uint32_t ii = 0; ii < ni; ++ii) {
            const uint32_t la = (uint32_t)(__builtin_amdgcn_readlane(a_off_reg, (int)ii + 1) -
                                           __builtin_amdgcn_readlane(a_off_reg, (int)ii));
            if (__builtin_amdgcn_readlane(size_reg, (int)ii) == 0) continue;
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane(base_reg, (int)ii);
            const Id* a_ids_l = a_ids_all + (size_t)ii * cap_a;
            const float* a_val_l = a_val_all + (size_t)ii * cap_a;
            // `this` and `arg` views of the pair
            const Id* t_ids = SWAP ? b_ids_l : a_ids_l;
            const float* t_val = SWAP ? b_val_l : a_val_l;
            const uint32_t lt = SWAP ? lb : la;
            const Id* g_ids = SWAP ? a_ids_l : b_ids_l;
            const float* g_val = SWAP ? a_val_l : b_val_l;
            const uint32_t lg = SWAP ? la : lb;

            // walk `this`: this-only elements and kept matches, 64 per turn
            uint32_t emitted = 0, matched = 0;
            for (uint32_t k0 = 0; k0 < lt; k0 += 64) {
                const bool live = k0 + lane < lt;
                const Id x = live ? lds_load(&t_ids[k0 + lane]) : (Id)0;
                float v = (THIS_UNIT && live) ? lds_load(&t_val[k0 + lane]) : 0.0f;
                bool hit = false;
                const uint32_t rg = rank_in<Id>(g_ids, lg, x, hit);
                hit = hit && live;
                bool keep = live && OP != SDRB_OP_AND;
                if (hit) {
                    const float gv = ARG_UNIT ? lds_load(&g_val[rg]) : 0.0f;   // a present id's rank is its position
                    v = match_data<OP, ARG_UNIT, ARITH>(v, gv);
                    keep = match_kept<OP, THIS_UNIT, ARITH>(v);
                }
                uint32_t n_keep, n_hit = 0;
                uint32_t pos = emitted + prefix_of(keep, n_keep);
                if constexpr (UNION) pos += rg - (matched + prefix_of(hit, n_hit));   // arg-only elements below x
                if (keep) {
                    if constexpr (VISIT) {
                        out_pos[base + pos] = k0 + lane;
                        reinterpret_cast<uint32_t*>(out_data)[base + pos] = hit ? rg : 0xFFFFFFFFu;
                    } else {
                        out_ids[base + pos] = x;
                        if constexpr (THIS_UNIT) out_data[base + pos] = v;
                    }
                }
                emitted += n_keep;
                matched += n_hit;
            }
            if constexpr (UNION) {
                // walk `arg`: its unmatched elements, converted to the result's element type
                uint32_t singles = 0, dropped = 0;
                for (uint32_t k0 = 0; k0 < lg; k0 += 64) {
                    const bool live = k0 + lane < lg;
                    const Id x = live ? lds_load(&g_ids[k0 + lane]) : (Id)0;
                    const float v = (ARG_UNIT && live) ? lds_load(&g_val[k0 + lane]) : 0.0f;
                    bool hit = false;
                    const uint32_t rt = rank_in<Id>(t_ids, lt, x, hit);
                    hit = hit && live;
                    const bool single = live && !hit;
                    bool drop = OP == SDRB_OP_XOR && hit;   // id-only xore drops every match
                    if constexpr (DROP_BY_VALUE) {
                        if (hit) {
                            const float tv = lds_load(&t_val[rt]);
                            drop = !match_kept<OP, THIS_UNIT>(match_data<OP, ARG_UNIT>(tv, ARG_UNIT ? v : 0.0f));
                        }
                    }
                    uint32_t n_single, n_drop = 0;
                    uint32_t pos = rt + singles + prefix_of(single, n_single);
                    if constexpr (OP == SDRB_OP_XOR) pos -= dropped + prefix_of(drop, n_drop);
                    if (single) {
                        if constexpr (VISIT) {
                            out_pos[base + pos] = 0xFFFFFFFFu;
                            reinterpret_cast<uint32_t*>(out_data)[base + pos] = k0 + lane;
                        } else {
                            out_ids[base + pos] = x;
                            if constexpr (THIS_UNIT) out_data[base + pos] = converted_arg<ARG_UNIT, ARITH>(v);
                        }
                    }
                    singles += n_single;
                    dropped += n_drop;
                }
            }
        }
        // the next row of B overwrites this wave's stage: every lane has read what it needed of it
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

struct Args {
    const sdrb_batch *a, *b;
    const uint32_t* out_off;
    void* out_ids;
    float* out_data;
    bool visit;
    uint32_t qc, cap_a, cap_b, rows_per_wave;
    size_t lds;
    dim3 grid;
    hipStream_t stream;
};

constexpr size_t kLdsWanted = 52 * 1024;    // three blocks a compute unit
constexpr size_t kLdsLimit = 150 * 1024;    // what one block may take (one a compute unit: very long rows)

struct Shape {
    uint32_t qc, cap_a, cap_b;
    size_t lds;
    bool ok;
};

Shape shape_for(const sdrb_batch* a, const sdrb_batch* b) {
    Shape s;
    const size_t each = (a->id_bits == 64 ? 8u : 4u) + 4u;   // an id and a payload
    s.cap_a = (a->max_row_len + 7) / 8 * 8;
    s.cap_b = (b->max_row_len + 7) / 8 * 8;
    if (s.cap_a == 0) s.cap_a = 8;
    if (s.cap_b == 0) s.cap_b = 8;
    s.qc = kMaxRowsOfA;
    auto bytes = [&](uint32_t qc) { return ((size_t)qc * s.cap_a + (size_t)kWaves * s.cap_b) * each; };
    while (s.qc > 1 && bytes(s.qc) > kLdsWanted) s.qc >>= 1;
    s.lds = bytes(s.qc);
    s.ok = s.lds <= kLdsLimit;
    return s;
}

template <typename Id, int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT = false, bool ARITH = false>
void launch(const Args& g) {
    if (g.lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_elem_search<Id, OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds);
    hipLaunchKernelGGL((k_elem_search<Id, OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH>), g.grid, dim3(kBlock), g.lds, g.stream,
                       g.a->d_offsets, reinterpret_cast<const Id*>(g.a->d_ids), g.a->d_data, (uint32_t)g.a->n_rows, g.b->d_offsets,
                       reinterpret_cast<const Id*>(g.b->d_ids), g.b->d_data, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data,
                       g.qc, g.cap_a, g.cap_b, g.rows_per_wave);
}

template <typename Id, int OP, bool SWAP>
void launch_kinds(const Args& g) {
    if constexpr (OP == SDRB_OP_AND || OP == SDRB_OP_OR) {
        if (g.visit) {   // positions only: payloads play no part
            launch<Id, OP, SWAP, false, false, true>(g);
            return;
        }
    }
    if constexpr (OP != SDRB_OP_XOR && sizeof(Id) == 4) {
        if (g.a->data_kind == SDRB_DATA_ARITH || g.b->data_kind == SDRB_DATA_ARITH) {
            // ArithData<float>: `this` is the side with payloads (the caller serves an EmptyData `this` as ids only)
            const bool ap = g.a->data_kind != SDRB_DATA_EMPTY, bp = g.b->data_kind != SDRB_DATA_EMPTY;
            if (ap && bp) launch<Id, OP, SWAP, true, true, false, true>(g);
            else if (ap) { if constexpr (!SWAP) launch<Id, OP, false, true, false, false, true>(g); }
            else { if constexpr (SWAP) launch<Id, OP, true, false, true, false, true>(g); }
            return;
        }
    }
    const bool au = g.a->data_kind == SDRB_DATA_UNIT, bu = g.b->data_kind == SDRB_DATA_UNIT;
    if (au && bu) launch<Id, OP, SWAP, true, true>(g);
    else if (au) launch<Id, OP, SWAP, true, false>(g);
    else if (bu) launch<Id, OP, SWAP, false, true>(g);
    else launch<Id, OP, SWAP, false, false>(g);
}

template <typename Id>
int launch_ops(const Args& g, int op, bool swap) {
#define SDRB_SEARCH_OP(OP)                          \
    do {                                            \
        if (swap) launch_kinds<Id, OP, true>(g);    \
        else launch_kinds<Id, OP, false>(g);        \
    } while (0)
    switch (op) {
    case SDRB_OP_AND: SDRB_SEARCH_OP(SDRB_OP_AND); break;
    case SDRB_OP_OR: SDRB_SEARCH_OP(SDRB_OP_OR); break;
    case SDRB_OP_XOR: SDRB_SEARCH_OP(SDRB_OP_XOR); break;
    case SDRB_OP_RM: SDRB_SEARCH_OP(SDRB_OP_RM); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
#undef SDRB_SEARCH_OP
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace

// All-pairs evaluation of two batches whose ids have one width (32 or 64 bits), rows short enough for LDS.  ArithData
// payloads: 32-bit ids (as everywhere).
bool elem_search_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    if (a->id_bits != b->id_bits || a->n_rows == 0 || b->n_rows == 0) return false;
    if (a->id_bits == 64 && (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH)) return false;
    const Shape s = shape_for(a, b);
    if (!s.ok) return false;
    return a->n_rows <= 65535ull * s.qc && b->n_rows <= 0xffffffffull;   // grid.y limit
}

// out_ids: ids of the batches' own width, or (visit) uint32 positions in `this`, out_data then the positions in `arg`
int launch_elem_search(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets, void* out_ids,
                       float* out_data, hipStream_t stream, bool visit) {
    const Shape shape = shape_for(a, b);
    Args g;
    g.a = a;
    g.b = b;
    g.out_off = out_offsets;
    g.out_ids = out_ids;
    g.out_data = out_data;
    g.visit = visit;
    g.qc = shape.qc;
    g.cap_a = shape.cap_a;
    g.cap_b = shape.cap_b;
    g.lds = shape.lds;
    g.stream = stream;
    // enough waves to fill the chip, few enough that each block amortises the rows of A it stages
    const uint64_t nb = b->n_rows;
    const uint64_t ychunks = (a->n_rows + g.qc - 1) / g.qc;
    uint32_t rpw = 16u;
    while (rpw > 1 && (nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves) * ychunks < 4096) rpw >>= 1;
    g.rows_per_wave = rpw;
    g.grid = dim3((uint32_t)((nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves)), (uint32_t)ychunks);
    if (a->id_bits == 64) return launch_ops<long long>(g, op, swap);
    return launch_ops<int32_t>(g, op, swap);
}

}  // namespace sdrb
