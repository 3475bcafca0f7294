// Element operations with one WAVE per pair, for results that carry payloads or rows too wide for
// the dense-bitset kernel: every output position is computed from ranks, so the 64 lanes of a wave
// write a pair's result together instead of one lane walking the merge.
//
// A sorted duplicate-free row over a declared width W is a W-bit set; with the exclusive popcount
// prefix of its 32-bit words stored beside them, rank(x) = #{ids < x} costs one 64-bit table read and
// a popcount, and for an id that is present rank(x) is its position in the row, i.e. where its payload
// lives.  Each wave builds this table for its row of B in LDS; the tables of all rows of A are built
// once per call in global memory and staged in LDS by the blocks that use them.
//
// Per pair (this = T, arg = G; SDRB_FLAG_SWAP decides which of A_i / B_j is T) the lanes walk T in
// id order, 64 elements per turn, look each id up in G and decide what the reference's merge decides:
//   this-only -> copied (not for ande)                         SDR.hpp:1207-1209,1332-1334,1541-1543
//   match     -> payload op, kept as the op says               SDR.hpp:1015-1021,1213-1217,1338-1344,1544-1548
// For ande / rme nothing else is emitted, so the output position of an element is the number of
// emitted elements before it: a ballot prefix.  For ore / xore the arg-only elements interleave, so
//   pos(t)          = #emitted T elements before t  +  rank_G(t) - #matches before t
//   pos(g, G-only)  = rank_T(g) - #dropped matches before g  +  #G-only elements before g
// and a second walk over G emits the converted arg-only elements (SDR.hpp:1210-1212,1335-1337;
// an EmptyData argument becomes UnitData 1.0, UnitData.hpp:13-14).  "Dropped" is evaluated with the
// same single-rounding payload operation on both walks, so the two agree.
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

using u64 = unsigned long long;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kQueryChunk = 16;   // rows of A a wave visits per row of B (grid.y splits A)

template <typename T>
__device__ __forceinline__ void lds_store(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <typename T>
__device__ __forceinline__ T lds_load(const T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// lanes below mine whose flag is set, and the wave's total
__device__ __forceinline__ uint32_t prefix_of(bool flag, uint32_t& total) {
    const u64 m = __builtin_amdgcn_ballot_w64(flag);
    total = (uint32_t)__builtin_popcountll(m);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Rank table of a row: entry w = {bits of ids 32w .. 32w+31, number of ids below 32w}, so that one
// 64-bit read serves a lookup.
struct alignas(8) RankEntry {
    uint32_t bits;
    uint32_t below;
};

// rank of id x in a row with table `tab` (LDS); `present` says whether x is in the row
__device__ __forceinline__ uint32_t rank_in(const RankEntry* tab, uint32_t x, bool& present) {
    const u64 raw = lds_load(reinterpret_cast<const u64*>(tab + (x >> 5)));
    const uint32_t bits = (uint32_t)raw, below = (uint32_t)(raw >> 32);
    const uint32_t sh = x & 31;
    present = (bits >> sh) & 1;
    return below + (uint32_t)__builtin_popcount(bits & ((1u << sh) - 1u));
}

// Rank tables of every row of A (one block per row; nw 32-bit words per row).
__global__ __launch_bounds__(kBlock) void k_build_rank_tables(const uint32_t* __restrict__ a_off,
                                                              const int32_t* __restrict__ a_ids, uint32_t nw,
                                                              RankEntry* __restrict__ tab) {
    extern __shared__ uint32_t s_bits[];
    __shared__ uint32_t s_wave_tot[kWaves];
    const uint64_t i = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t w = threadIdx.x; w < nw; w += kBlock) s_bits[w] = 0;
    __syncthreads();
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    for (uint32_t k = threadIdx.x; k < len; k += kBlock) {
        const uint32_t id = (uint32_t)a_ids[beg + k];   // host checked 0 <= id < 32 * nw
        atomicOr(&s_bits[id >> 5], 1u << (id & 31));
    }
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w0 = 0; w0 < nw; w0 += kBlock) {
        const uint32_t w = w0 + threadIdx.x;
        const uint32_t word = w < nw ? s_bits[w] : 0u;
        const uint32_t c = (uint32_t)__builtin_popcount(word);
        const uint32_t incl = wave_inclusive_scan(c);
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wave_base = 0, total = 0;
#pragma unroll
        for (int v = 0; v < kWaves; ++v) {
            if (v < (int)wave) wave_base += s_wave_tot[v];
            total += s_wave_tot[v];
        }
        if (w < nw) tab[i * nw + w] = RankEntry{word, base + wave_base + incl - c};
        base += total;
        __syncthreads();
    }
}

// What bounds a wave here is waiting, not arithmetic: on this hardware a wait for a load also waits
// for every store issued before it, so a pair must not touch global memory except to store.  The
// block therefore stages its (at most kQueryChunk) rows of A once -- rank tables, ids and payloads --
// in LDS; a wave stages its row of B per row, keeps the first 64 elements of it in registers, and
// fetches the output offsets of all its pairs with one load per row of B.
// VISIT: instead of elements, the positions a visitor would be called with -- out_ids gets the position in
// `this` and out_data (as uint32) the position in `arg`, 0xFFFFFFFF for the absent side of a single.
template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT = false, bool ARITH = false>
__global__ __launch_bounds__(kBlock) void k_elem_rank(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_data,
    uint32_t na, const RankEntry* __restrict__ a_tab, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, const float* __restrict__ b_data, uint32_t nb,
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, float* __restrict__ out_data,
    uint32_t nw, uint32_t qc, uint32_t cap_a, uint32_t cap_b, uint32_t rows_per_wave) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool UNION = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    // matches of a xore can be dropped one by one only when `this` carries payloads
    constexpr bool DROP_BY_VALUE = OP == SDRB_OP_XOR && THIS_UNIT;
    // the tables of A are looked up by the walk over `arg` of a union, and as `arg` when A is `arg`
    constexpr bool NEED_ATAB = UNION || SWAP;
    // ... and those of B likewise: by a union's second walk, and as `arg` when B is `arg`
    constexpr bool NEED_BTAB = UNION || !SWAP;
    // LDS: tables of the waves' rows of B | tables of the block's rows of A (if needed) | rows of A: ids,
    //      payloads | the waves' rows of B: ids, payloads   (cap_a / cap_b elements per row)
    extern __shared__ RankEntry smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    RankEntry* btab = smem + (size_t)wave * nw;
    RankEntry* atab_all = smem + (NEED_BTAB ? (size_t)kWaves * nw : (size_t)0);
    uint32_t* a_ids_all = reinterpret_cast<uint32_t*>(atab_all + (NEED_ATAB ? (size_t)qc * nw : (size_t)0));
    float* a_val_all = reinterpret_cast<float*>(a_ids_all + (size_t)qc * cap_a);
    uint32_t* b_ids_l = reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * cap_a) + (size_t)wave * cap_b;
    float* b_val_l = reinterpret_cast<float*>(reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * cap_a) +
                                              (size_t)kWaves * cap_b) + (size_t)wave * cap_b;

    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t i_begin = blockIdx.y * qc;
    const uint32_t ni = (i_begin + qc < na ? i_begin + qc : na) - i_begin;
    // row offsets of the block's rows of A, one per lane (lanes 0..ni)
    const int a_off_reg = (int)a_off[i_begin + (lane <= ni ? lane : ni)];

    if constexpr (NEED_ATAB)
        for (uint32_t idx = threadIdx.x; idx < ni * nw; idx += kBlock) atab_all[idx] = a_tab[(size_t)i_begin * nw + idx];
    for (uint32_t ii = 0; ii < ni; ++ii) {
        const uint32_t beg = a_off[i_begin + ii], len = a_off[i_begin + ii + 1] - beg;   // len <= cap_a (host)
        for (uint32_t l = threadIdx.x; l < len; l += kBlock) {
            a_ids_all[ii * cap_a + l] = (uint32_t)a_ids[beg + l];
            a_val_all[ii * cap_a + l] = A_UNIT ? a_data[beg + l] : 0.0f;
        }
    }
    __syncthreads();

    // A row of B and the output offsets of its pairs are requested one row ahead, into registers (rows of
    // up to 64 * kHeld elements; longer rows are fetched when their turn comes): a wait for a load
    // also waits for every store issued before it, so loads must not sit between a row's stores.
    constexpr int kHeld = 8;
    uint32_t held_id[kHeld];
    float held_val[kHeld];
    uint32_t held_beg = 0, held_len = 0;
    int held_base = 0;
    auto request_row = [&](uint64_t j) {
        held_beg = b_off[j];
        held_len = b_off[j + 1] - held_beg;   // <= cap_b (host)
        held_base = lane < ni ? (int)out_off[(uint64_t)(i_begin + lane) * nb + j] : 0;
#pragma unroll
        for (int c = 0; c < kHeld; ++c) {
            const uint32_t k = (uint32_t)c * 64u + lane;
            held_id[c] = 0;
            held_val[c] = 0.0f;
            if (k < held_len) {
                held_id[c] = (uint32_t)b_ids[held_beg + k];
                if constexpr (B_UNIT) held_val[c] = b_data[held_beg + k];
            }
        }
    };
    if (j_begin < j_end) request_row(j_begin);
    for (uint64_t j = j_begin; j < j_end; ++j) {
        const uint32_t b_beg = held_beg, lb = held_len;
        const int base_reg = held_base;
        // row j of B: copy it into LDS and build its table (zero, OR the ids in, prefix the popcounts)
        if constexpr (NEED_BTAB)
            for (uint32_t w = lane; w < nw; w += 64) lds_store(reinterpret_cast<u64*>(btab + w), (u64)0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const uint32_t b_x0 = held_id[0];
        const float b_v0 = held_val[0];
#pragma unroll
        for (int c = 0; c < kHeld; ++c) {
            const uint32_t k = (uint32_t)c * 64u + lane;
            if (k < lb) {
                lds_store(&b_ids_l[k], held_id[c]);
                lds_store(&b_val_l[k], held_val[c]);
                if constexpr (NEED_BTAB)
                    __hip_atomic_fetch_or(&btab[held_id[c] >> 5].bits, 1u << (held_id[c] & 31), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
        for (uint32_t k = (uint32_t)kHeld * 64u + lane; k < lb; k += 64) {
            const uint32_t id = (uint32_t)b_ids[b_beg + k];
            const float val = B_UNIT ? b_data[b_beg + k] : 0.0f;
            lds_store(&b_ids_l[k], id);
            lds_store(&b_val_l[k], val);
            if constexpr (NEED_BTAB)
                __hip_atomic_fetch_or(&btab[id >> 5].bits, 1u << (id & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        if (j + 1 < j_end) request_row(j + 1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t run = 0;
        for (uint32_t w0 = 0; NEED_BTAB && w0 < nw; w0 += 64) {
            const uint32_t w = w0 + lane;
            const uint32_t c = w < nw ? (uint32_t)__builtin_popcount(lds_load(&btab[w].bits)) : 0u;
            const uint32_t incl = wave_inclusive_scan(c);
            if (w < nw) lds_store(&btab[w].below, run + incl - c);
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        for (uint32_t ii = 0; ii < ni; ++ii) {
            const uint32_t la = (uint32_t)(__builtin_amdgcn_readlane(a_off_reg, (int)ii + 1) -
                                           __builtin_amdgcn_readlane(a_off_reg, (int)ii));
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane(base_reg, (int)ii);
            const RankEntry* atab = atab_all + (size_t)ii * nw;
            const uint32_t* a_ids_l = a_ids_all + (size_t)ii * cap_a;
            const float* a_val_l = a_val_all + (size_t)ii * cap_a;
            const uint32_t a_x0 = lane < la ? lds_load(&a_ids_l[lane]) : 0u;
            const float a_v0 = (A_UNIT && lane < la) ? lds_load(&a_val_l[lane]) : 0.0f;

            // `this` and `arg` views of the pair (rows and tables in LDS, first 64 elements in registers)
            const uint32_t* t_ids = SWAP ? b_ids_l : a_ids_l;
            const float* t_val = SWAP ? b_val_l : a_val_l;
            const uint32_t lt = SWAP ? lb : la;
            const uint32_t t_x0 = SWAP ? b_x0 : a_x0;
            const float t_v0 = SWAP ? b_v0 : a_v0;
            const RankEntry* t_tab = SWAP ? btab : atab;
            const uint32_t* g_ids = SWAP ? a_ids_l : b_ids_l;
            const float* g_val = SWAP ? a_val_l : b_val_l;
            const uint32_t lg = SWAP ? la : lb;
            const uint32_t g_x0 = SWAP ? a_x0 : b_x0;
            const float g_v0 = SWAP ? a_v0 : b_v0;
            const RankEntry* g_tab = SWAP ? atab : btab;

            // walk `this`: this-only elements and kept matches, 64 per turn
            uint32_t emitted = 0, matched = 0;
            for (uint32_t k0 = 0; k0 < lt; k0 += 64) {
                const bool live = k0 + lane < lt;
                uint32_t x = t_x0;
                float v = t_v0;
                if (k0 != 0) {
                    x = live ? lds_load(&t_ids[k0 + lane]) : 0u;
                    v = (THIS_UNIT && live) ? lds_load(&t_val[k0 + lane]) : 0.0f;
                }
                bool hit = false;
                const uint32_t rg = rank_in(g_tab, x, hit);
                hit = hit && live;
                bool keep = live && OP != SDRB_OP_AND;
                if (hit) {
                    const float gv = ARG_UNIT ? lds_load(&g_val[rg]) : 0.0f;   // a present id's rank is its position
                    v = match_data<OP, ARG_UNIT, ARITH>(v, gv);
                    keep = match_kept<OP, THIS_UNIT, ARITH>(v);
                }
                uint32_t n_keep, n_hit = 0;
                uint32_t pos = emitted + prefix_of(keep, n_keep);
                if constexpr (UNION) {
                    pos += rg - (matched + prefix_of(hit, n_hit));   // arg-only elements below x
                }
                if (keep) {
                    if constexpr (VISIT) {
                        out_ids[base + pos] = (int32_t)(k0 + lane);
                        reinterpret_cast<uint32_t*>(out_data)[base + pos] = hit ? rg : 0xFFFFFFFFu;
                    } else {
                        out_ids[base + pos] = (int32_t)x;
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
                    uint32_t x = g_x0;
                    float v = g_v0;
                    if (k0 != 0) {
                        x = live ? lds_load(&g_ids[k0 + lane]) : 0u;
                        v = (ARG_UNIT && live) ? lds_load(&g_val[k0 + lane]) : 0.0f;
                    }
                    bool hit = false;
                    const uint32_t rt = rank_in(t_tab, x, hit);
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
                            out_ids[base + pos] = -1;
                            reinterpret_cast<uint32_t*>(out_data)[base + pos] = k0 + lane;
                        } else {
                            out_ids[base + pos] = (int32_t)x;
                            if constexpr (THIS_UNIT) out_data[base + pos] = converted_arg<ARG_UNIT, ARITH>(v);
                        }
                    }
                    singles += n_single;
                    dropped += n_drop;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- short rows: four pairs per wave
// lanes of my group (LP consecutive lanes) below mine whose flag is set, and the group's total
template <int LP>
__device__ __forceinline__ uint32_t group_prefix(bool flag, uint32_t lane, uint32_t& total) {
    const u64 m = __builtin_amdgcn_ballot_w64(flag);
    const uint32_t gm = (uint32_t)(m >> (lane & ~(uint32_t)(LP - 1))) & ((1u << LP) - 1u);
    total = (uint32_t)__builtin_popcount(gm);
    return (uint32_t)__builtin_popcount(gm & ((1u << (lane & (LP - 1))) - 1u));
}

constexpr int kShortCap = 64;    // longest row the short-row kernel takes (either batch)
constexpr int kShortRows = 8;    // rows of B a wave stages and works through

// Rows of at most kShortCap elements on both sides: groups of 16 lanes work on four rows of A against the
// same row of B, so that 40-element rows still fill the wave, and EVERYTHING a wave reads -- its kShortRows
// rows of B and the output offsets of all its pairs, besides the block's rows of A -- is in LDS before its
// first store: between stores the wave never waits for a load (a wait for a load would also wait for every
// store issued before it).
// NCH: 16-element chunks per row (rows hold at most NCH * 16 elements).
template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT, bool ARITH, int NCH>
__global__ __launch_bounds__(kBlock) void k_elem_rank_short(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_data,
    uint32_t na, const RankEntry* __restrict__ a_tab, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, const float* __restrict__ b_data, uint32_t nb,
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, float* __restrict__ out_data,
    uint32_t nw, uint32_t qc, uint32_t rows_per_wave) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool UNION = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    constexpr bool DROP_BY_VALUE = OP == SDRB_OP_XOR && THIS_UNIT;
    constexpr int LP = 16, GROUPS = 64 / LP, CAP = kShortCap, ROWS = kShortRows;
    // LDS: tables of the waves' current rows of B | tables of the block's rows of A | rows of A: ids, payloads |
    //      the waves' rows of B: ids, payloads | output offsets of the waves' pairs
    extern __shared__ RankEntry smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t grp = lane / LP, gl = lane % LP;
    RankEntry* btab = smem + (size_t)wave * nw;
    RankEntry* atab_all = smem + (size_t)kWaves * nw;
    uint32_t* a_ids_all = reinterpret_cast<uint32_t*>(atab_all + (size_t)qc * nw);
    float* a_val_all = reinterpret_cast<float*>(a_ids_all + (size_t)qc * CAP);
    uint32_t* b_ids_all = reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * CAP) + (size_t)wave * ROWS * CAP;
    float* b_val_all = reinterpret_cast<float*>(reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * CAP) +
                                                (size_t)kWaves * ROWS * CAP) + (size_t)wave * ROWS * CAP;
    uint32_t* base_all = reinterpret_cast<uint32_t*>(reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * CAP) +
                                                     (size_t)2 * kWaves * ROWS * CAP) + (size_t)wave * ROWS * qc;

    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;   // rows_per_wave <= ROWS
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t nj = j_begin < j_end ? (uint32_t)(j_end - j_begin) : 0u;
    const uint32_t i_begin = blockIdx.y * qc;
    const uint32_t ni = (i_begin + qc < na ? i_begin + qc : na) - i_begin;
    // lengths of the block's rows of A, one per lane (lanes 0..ni-1)
    const uint32_t a_len_reg = lane < ni ? a_off[i_begin + lane + 1] - a_off[i_begin + lane] : 0u;

    for (uint32_t idx = threadIdx.x; idx < ni * nw; idx += kBlock) atab_all[idx] = a_tab[(size_t)i_begin * nw + idx];
    for (uint32_t idx = threadIdx.x; idx < ni * CAP; idx += kBlock) {
        const uint32_t beg = a_off[i_begin + idx / CAP], len = a_off[i_begin + idx / CAP + 1] - beg;   // len <= CAP (host)
        const uint32_t l = idx % CAP;
        a_ids_all[idx] = l < len ? (uint32_t)a_ids[beg + l] : 0u;
        a_val_all[idx] = (A_UNIT && l < len) ? a_data[beg + l] : 0.0f;
    }
    // the wave's rows of B (offsets one per lane, then all rows at once) and the output offsets of its pairs
    const uint32_t b_off_reg = nj ? b_off[j_begin + (lane <= nj ? lane : nj)] : 0u;
    uint32_t row_id[ROWS];
    float row_val[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t beg = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r + 1);
        const bool have = (uint32_t)r < nj && lane < end - beg;                  // end - beg <= CAP (host)
        row_id[r] = have ? (uint32_t)b_ids[beg + lane] : 0u;
        row_val[r] = (B_UNIT && have) ? b_data[beg + lane] : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        b_ids_all[r * CAP + lane] = row_id[r];
        b_val_all[r * CAP + lane] = row_val[r];
    }
    for (uint32_t idx = lane; idx < nj * ni; idx += 64)
        base_all[idx] = out_off[(uint64_t)(i_begin + idx % ni) * nb + j_begin + idx / ni];
    __syncthreads();

    for (uint32_t jr = 0; jr < nj; ++jr) {
        const uint32_t lb = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, (int)jr + 1) -
                            (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, (int)jr);
        const uint32_t* b_ids_l = b_ids_all + jr * CAP;
        const float* b_val_l = b_val_all + jr * CAP;
        // table of row jr of B: zero, OR the ids in, prefix the popcounts
        for (uint32_t w = lane; w < nw; w += 64) lds_store(reinterpret_cast<u64*>(btab + w), (u64)0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane < lb) {
            const uint32_t id = lds_load(&b_ids_l[lane]);
            __hip_atomic_fetch_or(&btab[id >> 5].bits, 1u << (id & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
            const uint32_t w = w0 + lane;
            const uint32_t c = w < nw ? (uint32_t)__builtin_popcount(lds_load(&btab[w].bits)) : 0u;
            const uint32_t incl = wave_inclusive_scan(c);
            if (w < nw) lds_store(&btab[w].below, run + incl - c);
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        for (uint32_t ii0 = 0; ii0 < ni; ii0 += GROUPS) {
            // my group's row of A (groups past the last row idle: their rows have no live elements)
            const bool valid = ii0 + grp < ni;
            const uint32_t ii = valid ? ii0 + grp : 0u;
            const uint32_t la = valid ? (uint32_t)__shfl((int)a_len_reg, (int)ii) : 0u;
            const uint32_t base = lds_load(&base_all[jr * ni + ii]);
            const RankEntry* atab = atab_all + (size_t)ii * nw;
            const uint32_t* a_ids_l = a_ids_all + (size_t)ii * CAP;
            const float* a_val_l = a_val_all + (size_t)ii * CAP;

            // `this` and `arg` views of the pair (rows and tables in LDS)
            const uint32_t* t_ids = SWAP ? b_ids_l : a_ids_l;
            const float* t_val = SWAP ? b_val_l : a_val_l;
            const uint32_t lt = SWAP ? (valid ? lb : 0u) : la;
            const RankEntry* t_tab = SWAP ? btab : atab;
            const uint32_t* g_ids = SWAP ? a_ids_l : b_ids_l;
            const float* g_val = SWAP ? a_val_l : b_val_l;
            const uint32_t lg = SWAP ? la : (valid ? lb : 0u);
            const RankEntry* g_tab = SWAP ? atab : btab;

            // Both walks, fully unrolled (rows hold at most NCH * 16 elements): all LDS reads of a stage are issued
            // together -- ids and payloads, then the table entries, then the matches' payloads -- so a turn waits
            // for LDS three times instead of three times per chunk.
            uint32_t xt[NCH], rg[NCH];
            float vt[NCH], gv[NCH];
            bool live_t[NCH], hit_t[NCH];
            uint32_t xg[NCH], rt[NCH];
            float vg[NCH], tv[NCH];
            bool live_g[NCH], hit_g[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const uint32_t k = c * LP + gl;
                live_t[c] = k < lt;
                xt[c] = live_t[c] ? lds_load(&t_ids[k]) : 0u;
                vt[c] = (THIS_UNIT && live_t[c]) ? lds_load(&t_val[k]) : 0.0f;
                if constexpr (UNION) {
                    live_g[c] = k < lg;
                    xg[c] = live_g[c] ? lds_load(&g_ids[k]) : 0u;
                    vg[c] = (ARG_UNIT && live_g[c]) ? lds_load(&g_val[k]) : 0.0f;
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                bool h = false;
                rg[c] = rank_in(g_tab, xt[c], h);
                hit_t[c] = h && live_t[c];
                if constexpr (UNION) {
                    bool hg = false;
                    rt[c] = rank_in(t_tab, xg[c], hg);
                    hit_g[c] = hg && live_g[c];
                }
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // a present id's rank is its position; an absent id's read is clamped into the row and ignored
                gv[c] = ARG_UNIT ? lds_load(&g_val[rg[c] < (uint32_t)CAP ? rg[c] : (uint32_t)CAP - 1]) : 0.0f;
                if constexpr (DROP_BY_VALUE) tv[c] = lds_load(&t_val[rt[c] < (uint32_t)CAP ? rt[c] : (uint32_t)CAP - 1]);
            }
            // walk `this`: this-only elements and kept matches
            uint32_t emitted = 0, matched = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float v = vt[c];
                bool keep = live_t[c] && OP != SDRB_OP_AND;
                if (hit_t[c]) {
                    v = match_data<OP, ARG_UNIT, ARITH>(v, gv[c]);
                    keep = match_kept<OP, THIS_UNIT, ARITH>(v);
                }
                uint32_t n_keep, n_hit = 0;
                uint32_t pos = emitted + group_prefix<LP>(keep, lane, n_keep);
                if constexpr (UNION) {
                    pos += rg[c] - (matched + group_prefix<LP>(hit_t[c], lane, n_hit));   // arg-only elements below x
                }
                if (keep) {
                    if constexpr (VISIT) {
                        out_ids[base + pos] = (int32_t)(c * LP + gl);
                        reinterpret_cast<uint32_t*>(out_data)[base + pos] = hit_t[c] ? rg[c] : 0xFFFFFFFFu;
                    } else {
                        out_ids[base + pos] = (int32_t)xt[c];
                        if constexpr (THIS_UNIT) out_data[base + pos] = v;
                    }
                }
                emitted += n_keep;
                matched += n_hit;
            }
            if constexpr (UNION) {
                // walk `arg`: its unmatched elements, converted to the result's element type
                uint32_t singles = 0, dropped = 0;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const bool single = live_g[c] && !hit_g[c];
                    bool drop = OP == SDRB_OP_XOR && hit_g[c];   // id-only xore drops every match
                    if constexpr (DROP_BY_VALUE) {
                        if (hit_g[c])
                            drop = !match_kept<OP, THIS_UNIT>(match_data<OP, ARG_UNIT>(tv[c], ARG_UNIT ? vg[c] : 0.0f));
                    }
                    uint32_t n_single, n_drop = 0;
                    uint32_t pos = rt[c] + singles + group_prefix<LP>(single, lane, n_single);
                    if constexpr (OP == SDRB_OP_XOR) pos -= dropped + group_prefix<LP>(drop, lane, n_drop);
                    if (single) {
                        if constexpr (VISIT) {
                            out_ids[base + pos] = -1;
                            reinterpret_cast<uint32_t*>(out_data)[base + pos] = c * LP + gl;
                        } else {
                            out_ids[base + pos] = (int32_t)xg[c];
                            if constexpr (THIS_UNIT) out_data[base + pos] = converted_arg<ARG_UNIT, ARITH>(vg[c]);
                        }
                    }
                    singles += n_single;
                    dropped += n_drop;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- short rows, narrow widths: coalesced runs
// Scattered 64-byte pieces are what bounds the short-row kernel (without its stores it runs almost three times
// as fast), so here the four lane groups of a turn work on four CONSECUTIVE rows of B against the same row of A:
// their results are neighbours in the output, are staged back to back in LDS and leave the wave as one
// coalesced run.  For that the tables of all of the wave's rows of B stay resident (kRunRows rows, so this
// kernel is for widths whose tables are small), and a pair's place in the run comes from the output offsets.
constexpr int kRunRows = 4;          // rows of B per wave = lane groups per turn
constexpr int kRunStage = 4 * 2 * kShortCap;   // elements a run can hold (four unions of two full rows)

template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT, bool ARITH, int NCH>
__global__ __launch_bounds__(kBlock) void k_elem_rank_runs(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_data,
    uint32_t na, const RankEntry* __restrict__ a_tab, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, const float* __restrict__ b_data, uint32_t nb,
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, float* __restrict__ out_data,
    uint32_t nw, uint32_t qc) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool UNION = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    constexpr bool DROP_BY_VALUE = OP == SDRB_OP_XOR && THIS_UNIT;
    constexpr bool SECOND = THIS_UNIT || VISIT;   // the run has a second array (payloads / arg positions)
    constexpr int LP = 16, CAP = kShortCap, ROWS = kRunRows;
    // LDS: tables of the block's rows of A | rows of A: ids, payloads | per wave: tables of its rows of B,
    //      those rows: ids, payloads | output offsets of its pairs ((ROWS + 1) per row of A) | the run: two arrays
    extern __shared__ RankEntry smem[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t grp = lane / LP, gl = lane % LP;
    RankEntry* atab_all = smem;
    uint32_t* a_ids_all = reinterpret_cast<uint32_t*>(atab_all + (size_t)qc * nw);
    float* a_val_all = reinterpret_cast<float*>(a_ids_all + (size_t)qc * CAP);
    const size_t per_wave_words = (size_t)ROWS * nw * 2 + (size_t)ROWS * CAP * 2 + (size_t)(ROWS + 1) * qc + 2 * kRunStage;
    uint32_t* mine = reinterpret_cast<uint32_t*>(a_val_all + (size_t)qc * CAP) + (size_t)wave * per_wave_words;
    RankEntry* btab_all = reinterpret_cast<RankEntry*>(mine);
    uint32_t* b_ids_all = mine + (size_t)ROWS * nw * 2;
    float* b_val_all = reinterpret_cast<float*>(b_ids_all + ROWS * CAP);
    uint32_t* base_all = b_ids_all + 2 * ROWS * CAP;        // [row of A][ROWS + 1]
    uint32_t* run_a = base_all + (ROWS + 1) * qc;           // ids / this positions
    uint32_t* run_b = run_a + kRunStage;                    // payload bits / arg positions

    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * ROWS;
    const uint64_t j_end = j_begin + ROWS < nb ? j_begin + ROWS : nb;
    const uint32_t nj = j_begin < j_end ? (uint32_t)(j_end - j_begin) : 0u;
    const uint32_t i_begin = blockIdx.y * qc;
    const uint32_t ni = (i_begin + qc < na ? i_begin + qc : na) - i_begin;

    for (uint32_t idx = threadIdx.x; idx < ni * nw; idx += kBlock) atab_all[idx] = a_tab[(size_t)i_begin * nw + idx];
    for (uint32_t idx = threadIdx.x; idx < ni * CAP; idx += kBlock) {
        const uint32_t beg = a_off[i_begin + idx / CAP], len = a_off[i_begin + idx / CAP + 1] - beg;   // len <= CAP (host)
        const uint32_t l = idx % CAP;
        a_ids_all[idx] = l < len ? (uint32_t)a_ids[beg + l] : 0u;
        a_val_all[idx] = (A_UNIT && l < len) ? a_data[beg + l] : 0.0f;
    }
    // lengths of the block's rows of A, one per lane; the wave's rows of B, their tables, its output offsets
    const uint32_t a_len_reg = lane < ni ? a_off[i_begin + lane + 1] - a_off[i_begin + lane] : 0u;
    const uint32_t b_off_reg = nj ? b_off[j_begin + (lane <= nj ? lane : nj)] : 0u;
    uint32_t row_id[ROWS];
    float row_val[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t beg = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r + 1);
        const bool have = (uint32_t)r < nj && lane < end - beg;                  // end - beg <= CAP (host)
        row_id[r] = have ? (uint32_t)b_ids[beg + lane] : 0u;
        row_val[r] = (B_UNIT && have) ? b_data[beg + lane] : 0.0f;
    }
    for (uint32_t idx = lane; idx < ni * (ROWS + 1); idx += 64) {
        const uint32_t ii = idx / (ROWS + 1), r = idx % (ROWS + 1);
        // offsets of the pairs (i, j_begin + r); entry nj closes the last pair's result
        base_all[idx] = r <= nj ? out_off[(uint64_t)(i_begin + ii) * nb + j_begin + r] : 0u;
    }
    for (uint32_t w = lane; w < ROWS * nw; w += 64) lds_store(reinterpret_cast<u64*>(btab_all + w), (u64)0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r + 1) -
                             (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r);
        b_ids_all[r * CAP + lane] = row_id[r];
        b_val_all[r * CAP + lane] = row_val[r];
        if ((uint32_t)r < nj && lane < len)
            __hip_atomic_fetch_or(&btab_all[r * nw + (row_id[r] >> 5)].bits, 1u << (row_id[r] & 31), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (uint32_t r = 0; r < (uint32_t)ROWS; ++r) {
        uint32_t run = 0;
        for (uint32_t w0 = 0; w0 < nw; w0 += 64) {
            const uint32_t w = w0 + lane;
            const uint32_t c = w < nw ? (uint32_t)__builtin_popcount(lds_load(&btab_all[r * nw + w].bits)) : 0u;
            const uint32_t incl = wave_inclusive_scan(c);
            if (w < nw) lds_store(&btab_all[r * nw + w].below, run + incl - c);
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
    }
    __syncthreads();
    if (nj == 0) return;

    // my group's row of B (groups past the last row idle: their rows have no live elements)
    const bool valid = grp < nj;
    const uint32_t jr = valid ? grp : 0u;
    const uint32_t lb = valid ? (uint32_t)__shfl((int)b_off_reg, (int)jr + 1) - (uint32_t)__shfl((int)b_off_reg, (int)jr) : 0u;
    const RankEntry* btab = btab_all + (size_t)jr * nw;
    const uint32_t* b_ids_l = b_ids_all + jr * CAP;
    const float* b_val_l = b_val_all + jr * CAP;

    for (uint32_t ii = 0; ii < ni; ++ii) {
        const uint32_t la = valid ? (uint32_t)__builtin_amdgcn_readlane((int)a_len_reg, (int)ii) : 0u;
        const uint32_t run_base = lds_load(&base_all[ii * (ROWS + 1)]);
        const uint32_t run_len = lds_load(&base_all[ii * (ROWS + 1) + nj]) - run_base;
        const uint32_t rel = lds_load(&base_all[ii * (ROWS + 1) + jr]) - run_base;   // my pair's place in the run
        const RankEntry* atab = atab_all + (size_t)ii * nw;
        const uint32_t* a_ids_l = a_ids_all + (size_t)ii * CAP;
        const float* a_val_l = a_val_all + (size_t)ii * CAP;

        // `this` and `arg` views of the pair (rows and tables in LDS)
        const uint32_t* t_ids = SWAP ? b_ids_l : a_ids_l;
        const float* t_val = SWAP ? b_val_l : a_val_l;
        const uint32_t lt = SWAP ? lb : la;
        const RankEntry* t_tab = SWAP ? btab : atab;
        const uint32_t* g_ids = SWAP ? a_ids_l : b_ids_l;
        const float* g_val = SWAP ? a_val_l : b_val_l;
        const uint32_t lg = SWAP ? la : lb;
        const RankEntry* g_tab = SWAP ? atab : btab;

        // Both walks, fully unrolled (rows hold at most NCH * 16 elements): all LDS reads of a stage are issued
        // together -- ids and payloads, then the table entries, then the matches' payloads.
        uint32_t xt[NCH], rg[NCH];
        float vt[NCH], gv[NCH];
        bool live_t[NCH], hit_t[NCH];
        uint32_t xg[NCH], rt[NCH];
        float vg[NCH], tv[NCH];
        bool live_g[NCH], hit_g[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const uint32_t k = c * LP + gl;
            live_t[c] = k < lt;
            xt[c] = live_t[c] ? lds_load(&t_ids[k]) : 0u;
            vt[c] = (THIS_UNIT && live_t[c]) ? lds_load(&t_val[k]) : 0.0f;
            if constexpr (UNION) {
                live_g[c] = k < lg;
                xg[c] = live_g[c] ? lds_load(&g_ids[k]) : 0u;
                vg[c] = (ARG_UNIT && live_g[c]) ? lds_load(&g_val[k]) : 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            bool h = false;
            rg[c] = rank_in(g_tab, xt[c], h);
            hit_t[c] = h && live_t[c];
            if constexpr (UNION) {
                bool hg = false;
                rt[c] = rank_in(t_tab, xg[c], hg);
                hit_g[c] = hg && live_g[c];
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            // a present id's rank is its position; an absent id's read is clamped into the row and ignored
            gv[c] = ARG_UNIT ? lds_load(&g_val[rg[c] < (uint32_t)CAP ? rg[c] : (uint32_t)CAP - 1]) : 0.0f;
            if constexpr (DROP_BY_VALUE) tv[c] = lds_load(&t_val[rt[c] < (uint32_t)CAP ? rt[c] : (uint32_t)CAP - 1]);
        }
        // walk `this`: this-only elements and kept matches
        uint32_t emitted = 0, matched = 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float v = vt[c];
            bool keep = live_t[c] && OP != SDRB_OP_AND;
            if (hit_t[c]) {
                v = match_data<OP, ARG_UNIT, ARITH>(v, gv[c]);
                keep = match_kept<OP, THIS_UNIT, ARITH>(v);
            }
            uint32_t n_keep, n_hit = 0;
            uint32_t pos = rel + emitted + group_prefix<LP>(keep, lane, n_keep);
            if constexpr (UNION) {
                pos += rg[c] - (matched + group_prefix<LP>(hit_t[c], lane, n_hit));   // arg-only elements below x
            }
            if (keep) {
                if constexpr (VISIT) {
                    lds_store(&run_a[pos], (uint32_t)(c * LP + gl));
                    lds_store(&run_b[pos], hit_t[c] ? rg[c] : 0xFFFFFFFFu);
                } else {
                    lds_store(&run_a[pos], xt[c]);
                    if constexpr (THIS_UNIT) lds_store(&run_b[pos], __float_as_uint(v));
                }
            }
            emitted += n_keep;
            matched += n_hit;
        }
        if constexpr (UNION) {
            // walk `arg`: its unmatched elements, converted to the result's element type
            uint32_t singles = 0, dropped = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const bool single = live_g[c] && !hit_g[c];
                bool drop = OP == SDRB_OP_XOR && hit_g[c];   // id-only xore drops every match
                if constexpr (DROP_BY_VALUE) {
                    if (hit_g[c])
                        drop = !match_kept<OP, THIS_UNIT>(match_data<OP, ARG_UNIT>(tv[c], ARG_UNIT ? vg[c] : 0.0f));
                }
                uint32_t n_single, n_drop = 0;
                uint32_t pos = rel + rt[c] + singles + group_prefix<LP>(single, lane, n_single);
                if constexpr (OP == SDRB_OP_XOR) pos -= dropped + group_prefix<LP>(drop, lane, n_drop);
                if (single) {
                    if constexpr (VISIT) {
                        lds_store(&run_a[pos], 0xFFFFFFFFu);
                        lds_store(&run_b[pos], (uint32_t)(c * LP + gl));
                    } else {
                        lds_store(&run_a[pos], xg[c]);
                        if constexpr (THIS_UNIT) lds_store(&run_b[pos], __float_as_uint(converted_arg<ARG_UNIT, ARITH>(vg[c])));
                    }
                }
                singles += n_single;
                dropped += n_drop;
            }
        }
        // the run leaves the wave: 256 contiguous bytes per store
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t t = lane; t < run_len; t += 64) {
            out_ids[run_base + t] = (int32_t)lds_load(&run_a[t]);
            if constexpr (SECOND) reinterpret_cast<uint32_t*>(out_data)[run_base + t] = lds_load(&run_b[t]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

struct Args {
    const sdrb_batch* a;
    const sdrb_batch* b;
    const RankEntry* a_tab;
    const uint32_t* out_off;
    int32_t* out_ids;
    float* out_data;
    uint32_t nw, qc, cap_a, cap_b, rows_per_wave;
    bool visit;
    bool short_rows;   // both batches hold rows of at most kShortCap elements: k_elem_rank_short
    bool runs;         // ... and the tables are small: k_elem_rank_runs (results leave as coalesced runs)
    dim3 grid;
    size_t lds;
    hipStream_t stream;
};

// Dynamic LDS beyond 64 KiB has to be granted per kernel before the launch.
template <typename K>
void allow_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT = false, bool ARITH = false>
void launch(const Args& g) {
    if (g.runs) {
        const uint32_t longest = g.a->max_row_len > g.b->max_row_len ? g.a->max_row_len : g.b->max_row_len;
        if (longest <= 48)
            hipLaunchKernelGGL((k_elem_rank_runs<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 3>), g.grid, dim3(kBlock), g.lds,
                               g.stream, g.a->d_offsets, g.a->d_ids, g.a->d_data, (uint32_t)g.a->n_rows, g.a_tab, g.b->d_offsets,
                               g.b->d_ids, g.b->d_data, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data, g.nw, g.qc);
        else
            hipLaunchKernelGGL((k_elem_rank_runs<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 4>), g.grid, dim3(kBlock), g.lds,
                               g.stream, g.a->d_offsets, g.a->d_ids, g.a->d_data, (uint32_t)g.a->n_rows, g.a_tab, g.b->d_offsets,
                               g.b->d_ids, g.b->d_data, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data, g.nw, g.qc);
        return;
    }
    if (g.short_rows) {
        const uint32_t longest = g.a->max_row_len > g.b->max_row_len ? g.a->max_row_len : g.b->max_row_len;
        allow_lds(&k_elem_rank_short<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 3>, g.lds);
        allow_lds(&k_elem_rank_short<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 4>, g.lds);
        if (longest <= 48)
            hipLaunchKernelGGL((k_elem_rank_short<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 3>), g.grid, dim3(kBlock), g.lds,
                               g.stream, g.a->d_offsets, g.a->d_ids, g.a->d_data, (uint32_t)g.a->n_rows, g.a_tab, g.b->d_offsets,
                               g.b->d_ids, g.b->d_data, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data, g.nw, g.qc,
                               g.rows_per_wave);
        else
            hipLaunchKernelGGL((k_elem_rank_short<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH, 4>), g.grid, dim3(kBlock), g.lds,
                               g.stream, g.a->d_offsets, g.a->d_ids, g.a->d_data, (uint32_t)g.a->n_rows, g.a_tab, g.b->d_offsets,
                               g.b->d_ids, g.b->d_data, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data, g.nw, g.qc,
                               g.rows_per_wave);
        return;
    }
    allow_lds(&k_elem_rank<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH>, g.lds);
    hipLaunchKernelGGL((k_elem_rank<OP, SWAP, A_UNIT, B_UNIT, VISIT, ARITH>), g.grid, dim3(kBlock), g.lds, g.stream, g.a->d_offsets,
                       g.a->d_ids, g.a->d_data, (uint32_t)g.a->n_rows, g.a_tab, g.b->d_offsets, g.b->d_ids, g.b->d_data,
                       (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.out_data, g.nw, g.qc, g.cap_a, g.cap_b, g.rows_per_wave);
}

template <int OP, bool SWAP>
void launch_kinds(const Args& g) {
    if constexpr (OP == SDRB_OP_AND || OP == SDRB_OP_OR) {
        if (g.visit) {   // positions only: payloads play no part
            launch<OP, SWAP, false, false, true>(g);
            return;
        }
    }
    if constexpr (OP != SDRB_OP_XOR) {
        if (g.a->data_kind == SDRB_DATA_ARITH || g.b->data_kind == SDRB_DATA_ARITH) {
            // ArithData<float>: `this` is the side with payloads (the caller serves an EmptyData `this` as ids only)
            const bool ap = g.a->data_kind != SDRB_DATA_EMPTY, bp = g.b->data_kind != SDRB_DATA_EMPTY;
            if (ap && bp) launch<OP, SWAP, true, true, false, true>(g);
            else if (ap) { if constexpr (!SWAP) launch<OP, false, true, false, false, true>(g); }
            else { if constexpr (SWAP) launch<OP, true, false, true, false, true>(g); }
            return;
        }
    }
    const bool au = g.a->data_kind == SDRB_DATA_UNIT, bu = g.b->data_kind == SDRB_DATA_UNIT;
    if (au && bu) launch<OP, SWAP, true, true>(g);
    else if (au) launch<OP, SWAP, true, false>(g);
    else if (bu) launch<OP, SWAP, false, true>(g);
    else launch<OP, SWAP, false, false>(g);
}

template <int OP>
void launch_swap(const Args& g, bool swap) {
    if (swap) launch_kinds<OP, true>(g);
    else launch_kinds<OP, false>(g);
}

constexpr int32_t kMaxWidth = 32768;           // 8 KiB of table per row
constexpr size_t kMaxTableBytes = 512u << 20;  // tables of A in global memory
constexpr size_t kLdsWanted = 80 * 1024;       // rows of A per block are halved until a block fits this (two blocks per CU)
constexpr size_t kLdsLimit = 80 * 1024;        // what a block may use at most
constexpr size_t kLdsThird = 53 * 1024;        // three blocks per compute unit
constexpr size_t kLdsShortWanted = 40 * 1024;  // short-row kernel: four blocks per compute unit
constexpr size_t kLdsRunsLimit = 52 * 1024;    // coalesced-run kernel: three blocks per compute unit

struct Shape {
    uint32_t nw, qc, cap_a, cap_b;
    size_t lds;
    bool ok;
    bool short_rows;
    bool runs;
};

size_t lds_bytes_runs(uint32_t nw, uint32_t qc) {
    const size_t per_wave_words = (size_t)kRunRows * nw * 2 + (size_t)kRunRows * kShortCap * 2 + (size_t)(kRunRows + 1) * qc +
                                  2 * (size_t)kRunStage;
    return (size_t)qc * nw * sizeof(RankEntry) + (size_t)qc * kShortCap * 8 + (size_t)kWaves * per_wave_words * 4;
}

size_t lds_bytes_short(uint32_t nw, uint32_t qc) {
    return ((size_t)kWaves * nw + (size_t)qc * nw) * sizeof(RankEntry) + (size_t)qc * kShortCap * 8 +
           (size_t)kWaves * kShortRows * kShortCap * 8 + (size_t)kWaves * kShortRows * qc * 4;
}

size_t lds_bytes(uint32_t nw, uint32_t qc, uint32_t cap_a, uint32_t cap_b, bool a_tables, bool b_tables) {
    return ((b_tables ? (size_t)kWaves * nw : (size_t)0) + (a_tables ? (size_t)qc * nw : (size_t)0)) * sizeof(RankEntry) +
           ((size_t)qc * cap_a + (size_t)kWaves * cap_b) * 8;
}

// rows of A per block: up to kQueryChunk, fewer when tables and rows are large (a row of B's table is
// rebuilt for every block that visits it, so more rows of A per block beat more blocks per compute unit)
Shape shape_for(const sdrb_batch* a, const sdrb_batch* b, bool a_tables = true, bool b_tables = true) {
    Shape s;
    s.nw = (uint32_t)(((size_t)b->width + 31) / 32);
    s.cap_a = (a->max_row_len + 7) / 8 * 8;
    s.cap_b = (b->max_row_len + 7) / 8 * 8;
    if (s.cap_a == 0) s.cap_a = 8;
    if (s.cap_b == 0) s.cap_b = 8;
    s.qc = kQueryChunk;
    // short rows on both sides and enough rows of A to fill the four lane groups: the short-row kernel
    s.short_rows = a->max_row_len <= (uint32_t)kShortCap && b->max_row_len <= (uint32_t)kShortCap && a->n_rows >= 4;
    s.runs = false;
    if (s.short_rows && lds_bytes_runs(s.nw, kQueryChunk) <= kLdsRunsLimit) {   // small tables: coalesced runs
        s.runs = true;
        s.lds = lds_bytes_runs(s.nw, s.qc);
        s.ok = true;
        return s;
    }
    if (s.short_rows) {
        while (s.qc > 4 && lds_bytes_short(s.nw, s.qc) > kLdsShortWanted) s.qc >>= 1;
        s.lds = lds_bytes_short(s.nw, s.qc);
        if (s.lds <= kLdsLimit) {
            s.ok = true;
            return s;
        }
        s.short_rows = false;
        s.qc = kQueryChunk;
    }
    // three blocks per compute unit if eight rows of A still fit a block then, else two blocks
    while (s.qc > 8 && lds_bytes(s.nw, s.qc, s.cap_a, s.cap_b, a_tables, b_tables) > kLdsThird) s.qc >>= 1;
    if (lds_bytes(s.nw, s.qc, s.cap_a, s.cap_b, a_tables, b_tables) > kLdsThird) {
        s.qc = kQueryChunk;
        while (s.qc > 1 && lds_bytes(s.nw, s.qc, s.cap_a, s.cap_b, a_tables, b_tables) > kLdsWanted) s.qc >>= 1;
    }
    s.lds = lds_bytes(s.nw, s.qc, s.cap_a, s.cap_b, a_tables, b_tables);
    s.ok = s.lds <= kLdsLimit;
    return s;
}

}  // namespace

bool elem_rank_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    if (a->id_bits != 32 || b->id_bits != 32) return false;
    if (b->width <= 0 || b->width > kMaxWidth) return false;
    const Shape s = shape_for(a, b);
    if (!s.ok) return false;                                                   // rows too long to stage in LDS
    if (a->n_rows > 65535ull * s.qc || b->n_rows > 0xffffffffull) return false;   // grid.y limit
    if (a->nnz && (a->min_id < 0 || a->max_id >= b->width)) return false;      // every id needs a bit
    return (size_t)a->n_rows * s.nw * sizeof(RankEntry) <= kMaxTableBytes;
}

int launch_elem_rank(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                     int32_t* out_ids, float* out_data, hipStream_t stream, bool visit) {
    // the long-row kernel stages / builds a batch's tables only where it looks them up (unions, or the
    // batch as `arg`)
    const bool is_union = op == SDRB_OP_OR || op == SDRB_OP_XOR;
    const Shape shape = shape_for(a, b, is_union || swap, is_union || !swap);
    const uint32_t nw = shape.nw;
    RankEntry* a_tab = static_cast<RankEntry*>(workspace(b->device, stream, 9, (size_t)a->n_rows * nw * sizeof(RankEntry)));
    if (!a_tab) return SDRB_ERR_HIP;
    hipLaunchKernelGGL(k_build_rank_tables, dim3((uint32_t)a->n_rows), dim3(kBlock), nw * sizeof(uint32_t), stream,
                       a->d_offsets, a->d_ids, nw, a_tab);
    SDRB_HIP(hipGetLastError());

    Args g;
    g.a = a;
    g.b = b;
    g.a_tab = a_tab;
    g.out_off = out_offsets;
    g.out_ids = out_ids;
    g.out_data = out_data;
    g.visit = visit;
    g.short_rows = shape.short_rows;
    g.runs = shape.runs;
    g.nw = nw;
    g.qc = shape.qc;
    g.cap_a = shape.cap_a;
    g.cap_b = shape.cap_b;
    g.lds = shape.lds;
    g.stream = stream;
    // enough waves to fill the chip, few enough that each amortises its tables
    const uint64_t nb = b->n_rows;
    const uint64_t ychunks = (a->n_rows + g.qc - 1) / g.qc;
    uint32_t rpw = shape.short_rows ? (uint32_t)kShortRows : 16u;
    while (rpw > 1 && (nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves) * ychunks < 4096) rpw >>= 1;
    if (shape.runs) rpw = (uint32_t)kRunRows;   // every wave takes exactly the rows of one turn
    g.rows_per_wave = rpw;
    g.grid = dim3((uint32_t)((nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves)), (uint32_t)ychunks);
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
