// Dense-bitset fill kernel for the element ops whose result carries ids only (`this` side
// EmptyData): ande/ore/xore/rme become AND / OR / XOR / AND-NOT of two bitsets.
//
// The reference produces these results with the merges andv/orv (SDR.hpp:958-997, 1099-1150)
// and pushes ids in ascending order (SDR.hpp:1015-1021, 1207-1217, 1332-1344, 1540-1560); with
// EmptyData every match is kept by ande/ore and dropped by xore/rme (EmptyData.hpp:29-37).
// A sorted id set over a declared width W is the same thing as a W-bit set, so here one wave
// owns one pair at a time: the 64 lanes hold the W bits of row i of A (precomputed once per
// call, 256 B per row at W = 2048) and of row j of B (scattered from its CSR ids through a
// per-wave LDS word array), combine them with one bitwise instruction, turn per-lane popcounts
// into write positions with a wave scan, and enumerate their set bits in ascending id order
// into the pair's slot of the output CSR.  Reads are coalesced (a row's ids are contiguous)
// and every wave writes one contiguous run.
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

template <typename WORD>
__global__ __launch_bounds__(64) void k_build_bitsets(const uint32_t* __restrict__ a_off,
                                                      const int32_t* __restrict__ a_ids,
                                                      WORD* __restrict__ bits) {
    constexpr int BITS = sizeof(WORD) * 8;
    const uint64_t i = blockIdx.x;
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    WORD* row = bits + i * 64;
    for (uint32_t k = threadIdx.x; k < len; k += 64) {
        const uint32_t id = (uint32_t)a_ids[beg + k];  // host checked 0 <= id < 64*BITS
        atomicOr(&row[id / BITS], (WORD)1 << (id % BITS));
    }
}

[[maybe_unused]] __device__ __forceinline__ uint32_t popc(uint32_t x) { return __builtin_popcount(x); }
__device__ __forceinline__ uint32_t popc(unsigned long long x) { return __builtin_popcountll(x); }
[[maybe_unused]] __device__ __forceinline__ uint32_t lowbit(uint32_t x) { return __builtin_ctz(x); }
__device__ __forceinline__ uint32_t lowbit(unsigned long long x) { return __builtin_ctzll(x); }

// Inclusive prefix sum over the 64 lanes of a wave with DPP moves (no LDS traffic): shifts inside
// rows of 16 lanes, then the last lane of row 0/2 is broadcast into row 1/3 and the last lane of the
// lower half into the upper half.  Masked-off rows / banks and lanes without a source add 0.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

constexpr int kStage = 256;   // ids a wave can stage in LDS for one coalesced copy-out
constexpr int kQueryChunk = 16;  // rows of A a block visits per row of B (grid.y splits A)

template <typename T>
__device__ __forceinline__ void lds_store(T* p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}
template <typename T>
__device__ __forceinline__ T lds_load(T* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

// The LDS executes one wave's operations in program order, so lanes of a wave exchange data
// through it without a hardware barrier; wavefront-scope atomics and fences only keep the
// compiler from treating a word as private to its lane.
template <typename WORD, int OP, bool SWAP>
__global__ __launch_bounds__(kBlock) void k_elem_bitmap(
    const WORD* __restrict__ a_bits, uint32_t na, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, uint32_t nb, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, uint32_t rows_per_wave) {
    constexpr int BITS = sizeof(WORD) * 8;
    __shared__ WORD lds_bits[kWaves][64];
    __shared__ int32_t lds_stage[kWaves][kStage];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t i_begin = blockIdx.y * kQueryChunk;
    const uint32_t i_end = i_begin + kQueryChunk < na ? i_begin + kQueryChunk : na;

    for (uint64_t j = j_begin; j < j_end; ++j) {
        // row j of B as a bitset: zero the wave's words, OR the ids in, read my word back
        const uint32_t beg = b_off[j], len = b_off[j + 1] - beg;
        lds_store(&lds_bits[wave][lane], (WORD)0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)b_ids[beg + k];
            __hip_atomic_fetch_or(&lds_bits[wave][id / BITS], (WORD)1 << (id % BITS), __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const WORD rbits = lds_load(&lds_bits[wave][lane]);

        for (uint32_t i = i_begin; i < i_end; ++i) {
            const WORD qbits = a_bits[(uint64_t)i * 64 + lane];
            const WORD self = SWAP ? rbits : qbits, arg = SWAP ? qbits : rbits;
            WORD res;
            if constexpr (OP == SDRB_OP_AND) res = self & arg;
            else if constexpr (OP == SDRB_OP_OR) res = self | arg;
            else if constexpr (OP == SDRB_OP_XOR) res = self ^ arg;
            else res = self & ~arg;
            // wave scan of the per-lane popcounts -> position of each lane's first id
            const uint32_t cnt = popc(res);
            const uint32_t incl = wave_inclusive_scan(cnt);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0) continue;
            const uint32_t base = out_off[(uint64_t)i * nb + j];
            uint32_t pos = incl - cnt;
            if (total <= (uint32_t)kStage) {
                // enumerate the set bits into LDS, then copy out with fully coalesced stores
                while (res) {
                    lds_store(&lds_stage[wave][pos++], (int32_t)(lane * BITS + lowbit(res)));
                    res &= res - 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t t = lane; t < total; t += 64) out_ids[base + t] = lds_load(&lds_stage[wave][t]);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            } else {
                pos += base;
                while (res) {
                    out_ids[pos++] = (int32_t)(lane * BITS + lowbit(res));
                    res &= res - 1;
                }
            }
        }
    }
}

// Several pairs per wave for widths <= 2048: LP lanes (32 or 16) hold the 2048 bits of a pair as 64-bit
// words (one or two per lane), so every wave instruction of the scan / enumerate / copy-out sequence
// serves 64 / LP pairs.  The wave's work items are the pairs (i, j) of its tile in j-major order; item
// p goes to lane group p % (64 / LP) (rows of A against the same row of B, or -- with a single row of
// A -- several rows of B).  a_bits is the same memory as for the 32-bit kernel (word k of 64 bits =
// ids 64 k .. 64 k + 63).
template <int LP>
__device__ __forceinline__ uint32_t group_inclusive_scan(uint32_t v) {   // prefix sums inside each group of LP lanes
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    if constexpr (LP == 32)
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    return v;
}

constexpr int kGroupStage = 128;   // ids a lane group can stage in LDS for one coalesced copy-out

template <int OP, bool SWAP, int LP>
__global__ __launch_bounds__(kBlock) void k_elem_bitmap_pairs(
    const unsigned long long* __restrict__ a_bits, uint32_t na, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, uint32_t nb, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, uint32_t rows_per_wave) {
    using u64 = unsigned long long;
    constexpr int PPW = 64 / LP;   // pairs per wave turn
    constexpr int WPL = 32 / LP;   // 64-bit words per lane
    __shared__ u64 lds_bits[kWaves][PPW][32];
    __shared__ int32_t lds_stage[kWaves][PPW][kGroupStage];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t grp = lane / LP, gl = lane % LP;
    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t i_begin = blockIdx.y * kQueryChunk;
    const uint32_t i_end = i_begin + kQueryChunk < na ? i_begin + kQueryChunk : na;
    if (j_begin >= j_end) return;
    const uint32_t ni = i_end - i_begin;
    const uint32_t nj = (uint32_t)(j_end - j_begin);

    u64* my_bits = lds_bits[wave][grp];
    int32_t* my_stage = lds_stage[wave][grp];
    uint64_t held_j = ~0ull;   // the row of B whose bitset this group currently holds
    u64 rbits[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) rbits[w] = 0;
    uint32_t jj = grp / ni, ii = grp % ni;   // this group's item, advanced by PPW items per turn
    for (uint32_t turn = 0, turns = (nj * ni + PPW - 1) / PPW; turn < turns; ++turn) {
        const bool valid = jj < nj;
        const uint64_t j = j_begin + (valid ? jj : 0);
        const uint32_t i = i_begin + (valid ? ii : 0);
        ii += PPW;
        while (ii >= ni) { ii -= ni; ++jj; }
        // (re)build the group's bitset of row j: zero the words, OR the ids in, read my words back
        if (__builtin_amdgcn_ballot_w64(valid && j != held_j) != 0) {
            if (valid && j != held_j) {
                const uint32_t beg = b_off[j], len = b_off[j + 1] - beg;
#pragma unroll
                for (int w = 0; w < WPL; ++w) lds_store(&my_bits[gl * WPL + w], (u64)0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                for (uint32_t k = gl; k < len; k += LP) {
                    const uint32_t id = (uint32_t)b_ids[beg + k];
                    __hip_atomic_fetch_or(&my_bits[id >> 6], (u64)1 << (id & 63), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int w = 0; w < WPL; ++w) rbits[w] = lds_load(&my_bits[gl * WPL + w]);
                held_j = j;
            }
        }
        u64 res[WPL];
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < WPL; ++w) {
            res[w] = 0;
            if (valid) {
                const u64 qbits = a_bits[(uint64_t)i * 32 + gl * WPL + w];
                const u64 self = SWAP ? rbits[w] : qbits, arg = SWAP ? qbits : rbits[w];
                if constexpr (OP == SDRB_OP_AND) res[w] = self & arg;
                else if constexpr (OP == SDRB_OP_OR) res[w] = self | arg;
                else if constexpr (OP == SDRB_OP_XOR) res[w] = self ^ arg;
                else res[w] = self & ~arg;
            }
            cnt += (uint32_t)__builtin_popcountll(res[w]);
        }
        const uint32_t incl = group_inclusive_scan<LP>(cnt);
        // the group's total sits in its last lane
        const uint32_t total = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane | (LP - 1)) << 2), (int)incl);
        if (__builtin_amdgcn_ballot_w64(total != 0) == 0) continue;
        const uint32_t base = valid ? out_off[(uint64_t)i * nb + j] : 0u;
        uint32_t pos = incl - cnt;
        // one row of A: the groups' pairs (i, j), (i, j+1), ... are neighbours in the output and leave the
        // wave as one coalesced run
        if (ni == 1 && __builtin_amdgcn_ballot_w64(total > (uint32_t)kGroupStage) == 0) {
            uint32_t run_before = 0, run_total = 0;
#pragma unroll
            for (int g = 0; g < PPW; ++g) {
                const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane((int)total, g * LP);
                if ((uint32_t)g < grp) run_before += tg;
                run_total += tg;
            }
            pos += run_before;
            int32_t* wave_stage = lds_stage[wave][0];
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    lds_store(&wave_stage[pos++], (int32_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r)));
                    r &= r - 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t run_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            for (uint32_t t = lane; t < run_total; t += 64) out_ids[run_base + t] = lds_load(&wave_stage[t]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else if (__builtin_amdgcn_ballot_w64(total > (uint32_t)kGroupStage) == 0) {
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    lds_store(&my_stage[pos++], (int32_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r)));
                    r &= r - 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t t = gl; t < total; t += LP) out_ids[base + t] = lds_load(&my_stage[t]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            pos += base;
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    out_ids[pos++] = (int32_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r));
                    r &= r - 1;
                }
            }
        }
    }
}

// The same kernel with everything a wave reads staged in LDS before its first store.  On this hardware a
// wait for a load also waits for every store issued before it, so a loop that loads (a row of B, a bitset
// of A, an output offset) and stores in every turn runs at the latency of a store.  Here the block copies
// the bitsets of its rows of A, and each wave the ids of its (at most kMaxRows) rows of B and the output
// offsets of its pairs, into LDS up front; the turns then touch global memory only to store.
// Needs rows of B of at most kRowCap ids (the launcher checks).
constexpr int kMaxRows = 16;
constexpr int kRowCap = 64;

template <int OP, bool SWAP, int LP>
__global__ __launch_bounds__(kBlock) void k_elem_bitmap_staged(
    const unsigned long long* __restrict__ a_bits, uint32_t na, const uint32_t* __restrict__ b_off,
    const int32_t* __restrict__ b_ids, uint32_t nb, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, uint32_t rows_per_wave) {
    using u64 = unsigned long long;
    constexpr int PPW = 64 / LP;   // pairs per wave turn
    constexpr int WPL = 32 / LP;   // 64-bit words per lane
    __shared__ u64 s_abits[kQueryChunk * 32];
    __shared__ u64 s_bits[kWaves][PPW][32];
    // ids are below 2048 here: 16 bits each keep the block's LDS at 24 KiB (six blocks per compute unit)
    __shared__ uint16_t s_stage[kWaves][PPW][kGroupStage];
    __shared__ uint16_t s_rows[kWaves][kMaxRows][kRowCap];
    __shared__ uint32_t s_base[kWaves][kMaxRows * kQueryChunk];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t grp = lane / LP, gl = lane % LP;
    const uint64_t j_begin = ((uint64_t)blockIdx.x * kWaves + wave) * rows_per_wave;   // rows_per_wave <= kMaxRows
    const uint64_t j_end = j_begin + rows_per_wave < nb ? j_begin + rows_per_wave : nb;
    const uint32_t i_begin = blockIdx.y * kQueryChunk;
    const uint32_t ni = (i_begin + kQueryChunk < na ? i_begin + kQueryChunk : na) - i_begin;
    const uint32_t nj = j_begin < j_end ? (uint32_t)(j_end - j_begin) : 0u;

    for (uint32_t idx = threadIdx.x; idx < ni * 32; idx += kBlock) s_abits[idx] = a_bits[(uint64_t)i_begin * 32 + idx];
    // offsets of the wave's rows of B, one per lane (lanes 0..nj); then all rows at once
    const uint32_t b_off_reg = nj ? b_off[j_begin + (lane <= nj ? lane : nj)] : 0u;
    int32_t row_id[kMaxRows];
#pragma unroll
    for (int r = 0; r < kMaxRows; ++r) {
        const uint32_t beg = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r);
        const uint32_t end = (uint32_t)__builtin_amdgcn_readlane((int)b_off_reg, r + 1 <= (int)nj ? r + 1 : (int)nj);
        row_id[r] = ((uint32_t)r < nj && lane < end - beg) ? b_ids[beg + lane] : 0;
    }
#pragma unroll
    for (int r = 0; r < kMaxRows; ++r) s_rows[wave][r][lane] = (uint16_t)row_id[r];
    for (uint32_t idx = lane; idx < nj * ni; idx += 64)
        s_base[wave][idx] = out_off[(uint64_t)(i_begin + idx % ni) * nb + j_begin + idx / ni];
    __syncthreads();
    if (nj == 0) return;

    u64* my_bits = s_bits[wave][grp];
    uint16_t* my_stage = s_stage[wave][grp];
    uint32_t held_jj = ~0u;   // the row of B whose bitset this group currently holds
    u64 rbits[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) rbits[w] = 0;
    // Items in i-major order: the groups of a turn take consecutive rows of B against the same row of A, so
    // their results are neighbours in the output and leave the wave as one coalesced run.
    uint32_t ii = grp / nj, jj = grp % nj;   // this group's item, advanced by PPW items per turn
    uint16_t* wave_stage = s_stage[wave][0];   // the groups' stages, back to back
    for (uint32_t turn = 0, turns = (nj * ni + PPW - 1) / PPW; turn < turns; ++turn) {
        const bool valid = ii < ni;
        const uint32_t cj = valid ? jj : 0u, ci = valid ? ii : 0u;
        jj += PPW;
        while (jj >= nj) { jj -= nj; ++ii; }
        // (re)build the group's bitset of row cj: zero the words, OR the ids in, read my words back
        if (__builtin_amdgcn_ballot_w64(valid && cj != held_jj) != 0) {
            if (valid && cj != held_jj) {
                // this lane's copy of the offsets is not its group's: fetch the row's bounds by index
                const uint32_t len = (uint32_t)__shfl((int)b_off_reg, (int)cj + 1) - (uint32_t)__shfl((int)b_off_reg, (int)cj);
#pragma unroll
                for (int w = 0; w < WPL; ++w) lds_store(&my_bits[gl * WPL + w], (u64)0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                for (uint32_t k = gl; k < len; k += LP) {
                    const uint32_t id = (uint32_t)lds_load(&s_rows[wave][cj][k]);
                    __hip_atomic_fetch_or(&my_bits[id >> 6], (u64)1 << (id & 63), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int w = 0; w < WPL; ++w) rbits[w] = lds_load(&my_bits[gl * WPL + w]);
                held_jj = cj;
            }
        }
        u64 res[WPL];
        uint32_t cnt = 0;
#pragma unroll
        for (int w = 0; w < WPL; ++w) {
            res[w] = 0;
            if (valid) {
                const u64 qbits = s_abits[ci * 32 + gl * WPL + w];
                const u64 self = SWAP ? rbits[w] : qbits, arg = SWAP ? qbits : rbits[w];
                if constexpr (OP == SDRB_OP_AND) res[w] = self & arg;
                else if constexpr (OP == SDRB_OP_OR) res[w] = self | arg;
                else if constexpr (OP == SDRB_OP_XOR) res[w] = self ^ arg;
                else res[w] = self & ~arg;
            }
            cnt += (uint32_t)__builtin_popcountll(res[w]);
        }
        const uint32_t incl = group_inclusive_scan<LP>(cnt);
        // the group's total sits in its last lane
        const uint32_t total = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((lane | (LP - 1)) << 2), (int)incl);
        if (__builtin_amdgcn_ballot_w64(total != 0) == 0) continue;
        const uint32_t base = lds_load(&s_base[wave][cj * ni + ci]);
        uint32_t pos = incl - cnt;
        const bool fits = __builtin_amdgcn_ballot_w64(total > (uint32_t)kGroupStage) == 0;
        // all groups on the same row of A: their results are contiguous in the output (pairs (i, j), (i, j+1), ...)
        const uint32_t first_i = (uint32_t)__builtin_amdgcn_readfirstlane((int)ci);
        const bool together = __builtin_amdgcn_ballot_w64(valid && ci != first_i) == 0;
        if (fits && together) {
            // totals of the groups (idle groups have none) -> where each group's ids start in the wave's run
            uint32_t run_before = 0, run_total = 0;
#pragma unroll
            for (int g = 0; g < PPW; ++g) {
                const uint32_t tg = (uint32_t)__builtin_amdgcn_readlane((int)total, g * LP);
                if ((uint32_t)g < grp) run_before += tg;
                run_total += tg;
            }
            pos += run_before;
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    lds_store(&wave_stage[pos++], (uint16_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r)));
                    r &= r - 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t run_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);   // group 0's pair comes first
            for (uint32_t t = lane; t < run_total; t += 64) out_ids[run_base + t] = (int32_t)lds_load(&wave_stage[t]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else if (fits) {
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    lds_store(&my_stage[pos++], (uint16_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r)));
                    r &= r - 1;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (uint32_t t = gl; t < total; t += LP) out_ids[base + t] = (int32_t)lds_load(&my_stage[t]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        } else {
            pos += base;
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                u64 r = res[w];
                while (r) {
                    out_ids[pos++] = (int32_t)((gl * WPL + w) * 64 + (uint32_t)__builtin_ctzll(r));
                    r &= r - 1;
                }
            }
        }
    }
}

struct Args {
    const void* a_bits;
    const sdrb_batch* a;
    const sdrb_batch* b;
    const uint32_t* out_off;
    int32_t* out_ids;
    uint32_t rows_per_wave;
    dim3 grid;
    hipStream_t stream;
};

template <typename WORD, int OP, bool SWAP>
void launch(const Args& g) {
    if constexpr (sizeof(WORD) == 4) {   // widths <= 2048: several pairs per wave on 64-bit words
        // 16 lanes per pair (four pairs per wave) measured ahead of 32 (config 3 ore: 2.85 vs 2.96 ms at 16 queries)
        // staging pays once a wave has a few rows of A per row of B to amortise it over
        if (g.b->max_row_len <= (uint32_t)kRowCap && g.rows_per_wave <= (uint32_t)kMaxRows && g.a->n_rows >= 4)
            hipLaunchKernelGGL((k_elem_bitmap_staged<OP, SWAP, 16>), g.grid, dim3(kBlock), 0, g.stream,
                               static_cast<const unsigned long long*>(g.a_bits), (uint32_t)g.a->n_rows, g.b->d_offsets,
                               g.b->d_ids, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.rows_per_wave);
        else
            hipLaunchKernelGGL((k_elem_bitmap_pairs<OP, SWAP, 16>), g.grid, dim3(kBlock), 0, g.stream,
                               static_cast<const unsigned long long*>(g.a_bits), (uint32_t)g.a->n_rows, g.b->d_offsets,
                               g.b->d_ids, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.rows_per_wave);
    } else {
        hipLaunchKernelGGL((k_elem_bitmap<WORD, OP, SWAP>), g.grid, dim3(kBlock), 0, g.stream,
                           static_cast<const WORD*>(g.a_bits), (uint32_t)g.a->n_rows, g.b->d_offsets,
                           g.b->d_ids, (uint32_t)g.b->n_rows, g.out_off, g.out_ids, g.rows_per_wave);
    }
}

template <typename WORD, int OP>
void launch_swap(const Args& g, bool swap) {
    if (swap) launch<WORD, OP, true>(g);
    else launch<WORD, OP, false>(g);
}

template <typename WORD>
int run(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_off,
        int32_t* out_ids, hipStream_t stream) {
    const size_t bytes = (size_t)a->n_rows * 64 * sizeof(WORD);
    WORD* bits = static_cast<WORD*>(workspace(b->device, stream, 3, bytes));
    if (!bits) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(bits, 0, bytes, stream));
    hipLaunchKernelGGL(k_build_bitsets<WORD>, dim3((uint32_t)a->n_rows), dim3(64), 0, stream,
                       a->d_offsets, a->d_ids, bits);
    SDRB_HIP(hipGetLastError());

    Args g;
    g.a_bits = bits;
    g.a = a;
    g.b = b;
    g.out_off = out_off;
    g.out_ids = out_ids;
    g.stream = stream;
    // enough waves to fill the chip, few enough that each amortises its setup
    const uint64_t nb = b->n_rows;
    uint32_t rpw = 16;
    const uint64_t ychunks = (a->n_rows + kQueryChunk - 1) / kQueryChunk;
    while (rpw > 1 && (nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves) * ychunks < 4096) rpw >>= 1;
    g.rows_per_wave = rpw;
    const uint32_t gx = (uint32_t)((nb + (uint64_t)rpw * kWaves - 1) / ((uint64_t)rpw * kWaves));
    const uint32_t gy = (uint32_t)((a->n_rows + kQueryChunk - 1) / kQueryChunk);
    g.grid = dim3(gx, gy);
    switch (op) {
    case SDRB_OP_AND: launch_swap<WORD, SDRB_OP_AND>(g, swap); break;
    case SDRB_OP_OR: launch_swap<WORD, SDRB_OP_OR>(g, swap); break;
    case SDRB_OP_XOR: launch_swap<WORD, SDRB_OP_XOR>(g, swap); break;
    case SDRB_OP_RM: launch_swap<WORD, SDRB_OP_RM>(g, swap); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace

bool elem_bitmap_applicable(const sdrb_batch* a, const sdrb_batch* b, bool swap) {
    const sdrb_batch* self = swap ? b : a;
    if (self->data_kind != SDRB_DATA_EMPTY) return false;       // results would carry payloads
    if (b->width <= 0 || b->width > 4096) return false;
    if (a->n_rows > 65535ull * kQueryChunk) return false;        // grid.y limit
    // every id of A must be representable in the bitset too (A-only ids appear in ore/xore)
    if (a->nnz && (a->min_id < 0 || a->max_id >= b->width)) return false;
    return true;
}

int launch_elem_bitmap(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                       const uint32_t* out_offsets, int32_t* out_ids, hipStream_t stream) {
    if (b->width <= 2048) return run<uint32_t>(a, b, op, swap, out_offsets, out_ids, stream);
    return run<unsigned long long>(a, b, op, swap, out_offsets, out_ids, stream);
}

}  // namespace sdrb
