// Query-mask table kernel for the size operations on long rows and/or payloads (e.g. UnitData rows
// of 328 ids over a width of 16384): the long batch B is read from HBM ONCE for 64 rows of A.
//
// The reference finds the common ids of a pair with andv (SDR.hpp:958-997) and hands each match to
// the op's size visitor (SDR.hpp:1087-1091, 1298-1301, 1428-1431, 1572-1574).  Here a block keeps,
// for a tile of 64 rows of A,
//
//     mask[w] (uint64, LDS)  bit q set  <=>  row 64*tile + q of A contains id w     (8 B x width)
//
// and streams rows of B, one wave per row, one lane per element: a lane looks up mask[id] and
// visits its set bits -- each one is an id match of the pair (q, j).  The payload of A for that
// match comes from a column-compressed copy of the tile's payloads in LDS (col[w] = where the
// payloads of id w start, in ascending row order -- the order the set bits are visited in) when
// that fits, else from a dense value matrix val[q][w] (global, L2-resident, built per call); the
// payload rule of the op is evaluated (single IEEE operations, sdrb_device.h) and the pair's
// counters in LDS get  (1 << 16) | kept  added.  After 64 rows of B the 64 x 64 counter tile is
// turned into sizes and written as coalesced rows.  B is read once per 64 rows of A instead of
// once per row; the table is rebuilt only when the block moves to another tile of A.
// The counter of pair (q, jl) lives at cnt[jl * 65 + q]: the lanes of a wave work on one jl and
// different q, and the write-out reads one q and 64 jl -- both land in 64 different banks.
#include <cstdlib>
#include <initializer_list>
#include <type_traits>

#include "sdrb_internal.h"
#include "match_math.h"

namespace sdrb {

namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / 64;
constexpr int kGroup = 64;                       // rows of B per counter tile
constexpr int kRowsPerWave = kGroup / kWaves;    // 4
// rows of A per mask table: 64, or 32 where that makes room for the tile's payloads in LDS;
// counter tile: [row of B][row of A], padded by one word against bank conflicts
constexpr int cnt_ld(int tile) { return tile + 1; }
// Lab switch (tools/exp/build_variant.sh ... -DSDRB_LAB_TILE_STEP=16): a 32-row tile takes only this many rows of A, so
// that B is streamed 32 / step times as often -- what narrower tiles would cost in passes, with everything else equal.
#ifndef SDRB_LAB_TILE_STEP
#define SDRB_LAB_TILE_STEP 0
#endif
constexpr int tile_step(int tile) { return SDRB_LAB_TILE_STEP && tile == 32 ? SDRB_LAB_TILE_STEP : tile; }

// dense payload matrix of A: val[i][w] = payload of id w in row i (only read where mask says so)
__global__ __launch_bounds__(64) void k_scatter_values(const uint32_t* __restrict__ a_off,
                                                        const int32_t* __restrict__ a_ids,
                                                        const float* __restrict__ a_val, uint32_t width,
                                                        float* __restrict__ val) {
    const uint64_t i = blockIdx.x;
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    for (uint32_t k = threadIdx.x; k < len; k += 64) {
        const uint32_t id = (uint32_t)a_ids[beg + k];
        if (id < width) val[i * width + id] = a_val[beg + k];
    }
}

// FILL == false: size ops (counters -> sizes).  FILL == true: ande -- the counter tile starts at the
// pairs' output offsets and every kept match takes the next slot of its pair; slots of one pair are
// claimed by concurrent lanes, so a pair's elements are put in id order afterwards (k_sort_pairs_staged).
// LIST (64-row tiles, flat loop, payloads of A needed): no mask at all.  ends[w + 1] (16 bit) is where the list of id w
// ends -- and the list of id w + 1 starts -- in two parallel arrays: the tile row (a byte) and the payload of every
// element of the tile that holds the id.  A lane walks the list of its element's id instead of the set bits of a
// mask.  Where a 64-row mask (8 bytes an id) leaves no room for the payloads and the 32-row mask makes B stream twice
// as often, the lists (2 bytes an id, 5 an element) fit 64 rows: config 4 streams B once instead of twice.
template <int MATH, int NIT, bool FILL, bool FLAT, int TILE, bool LIST = false>
__global__ __launch_bounds__(kBlock) void k_size_mask_table(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_val_dense,
    uint32_t na, const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids,
    const float* __restrict__ b_data, uint32_t nb, uint32_t width, int op, uint32_t swap,
    uint32_t* __restrict__ out, uint64_t ld_out, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, float* __restrict__ out_data, const float* __restrict__ a_val,
    uint32_t val_cap) {
    constexpr bool NEED_A = MATH == kSelfA || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    constexpr bool NEED_B = MATH == kSelfB || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    constexpr int IT = NIT == 0 ? 1 : NIT;
    static_assert(TILE == 64 || (TILE == 32 && FLAT), "32-row tiles exist for the flat loop only");
    static_assert(!LIST || (FLAT && TILE == 64 && NEED_A), "lists serve the flat loop over 64-row tiles with payloads of A");
    using mask_t = std::conditional_t<TILE == 64, unsigned long long, uint32_t>;
    constexpr int kTile = TILE, kCntLd = cnt_ld(TILE), kCntSize = kGroup * kCntLd;
    // LDS: mask[width + 1] (mask[width] = 0, the sentinel), cnt[64][TILE + 1], la[TILE], lb[64],
    //      and with FLAT vals[val_cap], col[width + 1]
    // LIST: ends[width + 2] in place of the mask, then cnt, la, lb, vals[max(val_cap, width + 2)], list_row[val_cap]
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    mask_t* mask = reinterpret_cast<mask_t*>(lds_raw);
    uint16_t* ends = reinterpret_cast<uint16_t*>(lds_raw);
    uint32_t* cnt = LIST ? reinterpret_cast<uint32_t*>(lds_raw + ((((size_t)width + 2) * 2 + 15) & ~(size_t)15))
                         : reinterpret_cast<uint32_t*>(mask + width + 1 + ((width + 1) & 1u));
    uint32_t* la = cnt + kCntSize;
    uint32_t* lb = la + kTile;
    float* vals = reinterpret_cast<float*>(lb + kGroup);
    const uint32_t vals_room = LIST && val_cap < width + 2 ? width + 2 : val_cap;
    uint16_t* col = reinterpret_cast<uint16_t*>(vals + vals_room);
    unsigned char* list_row = reinterpret_cast<unsigned char*>(vals + vals_room);

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = blockIdx.y, i0 = tile * (uint32_t)tile_step(TILE);
    const uint32_t rows_a = na - i0 < (uint32_t)tile_step(TILE) ? na - i0 : (uint32_t)tile_step(TILE);

    if constexpr (LIST) {
        // how many elements of the tile hold each id (32-bit counters where the payloads will be), their running sum as
        // the lists' starts, then every element takes the next place of its id's list: ends[w + 1] counts up from the
        // start of list w to its end, which is the start of list w + 1
        uint32_t* held = reinterpret_cast<uint32_t*>(vals);
        for (uint32_t w = tid; w < width + 1; w += kBlock) held[w] = 0;
        for (uint32_t e = tid; e < kCntSize; e += kBlock) cnt[e] = 0;
        if (tid < kTile) la[tid] = tid < rows_a ? a_off[i0 + tid + 1] - a_off[i0 + tid] : 0u;
        __syncthreads();
        for (uint32_t q = wave; q < rows_a; q += kWaves) {
            const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
            for (uint32_t k = lane; k < len; k += 64) {
                const uint32_t id = (uint32_t)a_ids[beg + k];
                if (id < width) atomicAdd(&held[id], 1u);   // ids outside B's width never match
            }
        }
        __syncthreads();
        const uint32_t per = (width + kBlock) / kBlock;
        const uint32_t w0 = tid * per < width + 1 ? tid * per : width + 1;
        const uint32_t w1 = w0 + per < width + 1 ? w0 + per : width + 1;
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += held[w];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) lb[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) run += lb[w];
        if (tid == 0) ends[0] = 0;
        for (uint32_t w = w0; w < w1; ++w) {
            ends[w + 1] = (uint16_t)run;
            run += held[w];
        }
        __syncthreads();
        uint32_t* ends_words = reinterpret_cast<uint32_t*>(ends);
        for (uint32_t q = wave; q < rows_a; q += kWaves) {
            const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
            for (uint32_t k = lane; k < len; k += 64) {
                const uint32_t id = (uint32_t)a_ids[beg + k];
                if (id < width) {
                    // a 16-bit counter inside a 32-bit word: it never passes 65535 (the host checked), so nothing carries
                    const uint32_t at = id + 1, hi = at & 1u;
                    const uint32_t old = atomicAdd(&ends_words[at >> 1], hi ? 0x10000u : 1u);
                    const uint32_t slot = hi ? old >> 16 : old & 0xffffu;
                    list_row[slot] = (unsigned char)q;
                    vals[slot] = a_val[beg + k];
                }
            }
        }
        __syncthreads();
    } else {
    for (uint32_t w = tid; w < width + 1; w += kBlock) mask[w] = 0;
    for (uint32_t e = tid; e < kCntSize; e += kBlock) cnt[e] = 0;
    if (tid < kTile) la[tid] = tid < rows_a ? a_off[i0 + tid + 1] - a_off[i0 + tid] : 0u;
    __syncthreads();
    // wave w sets the bits of rows w, w + 16, ... of the tile
    for (uint32_t q = wave; q < rows_a; q += kWaves) {
        const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
        for (uint32_t k = lane; k < len; k += 64) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) atomicOr(&mask[id], (mask_t)1 << q);   // ids outside B's width never match
        }
    }
    __syncthreads();
    }
    if constexpr (FLAT && NEED_A && !LIST) {
        // col[w] = number of tile elements with an id below w: every thread counts a contiguous piece
        // of the table, the pieces' totals are scanned over the block (lb[] holds the waves' totals)
        const uint32_t per = (width + kBlock) / kBlock;
        const uint32_t w0 = tid * per < width + 1 ? tid * per : width + 1;
        const uint32_t w1 = w0 + per < width + 1 ? w0 + per : width + 1;
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__builtin_popcountll((unsigned long long)mask[w]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) lb[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) run += lb[w];
        for (uint32_t w = w0; w < w1; ++w) {
            col[w] = (uint16_t)run;
            run += (uint32_t)__builtin_popcountll((unsigned long long)mask[w]);
        }
        __syncthreads();
        for (uint32_t q = wave; q < rows_a; q += kWaves) {
            const uint32_t beg = a_off[i0 + q], len = a_off[i0 + q + 1] - beg;
            for (uint32_t k = lane; k < len; k += 64) {
                const uint32_t id = (uint32_t)a_ids[beg + k];
                if (id < width)
                    vals[col[id] + (uint32_t)__builtin_popcountll((unsigned long long)(mask[id] & (((mask_t)1 << q) - 1)))] = a_val[beg + k];
            }
        }
        __syncthreads();
    }

    const float* val_tile = NEED_A && !FLAT ? a_val_dense + (uint64_t)i0 * width : nullptr;
    const uint32_t n_groups = (nb + kGroup - 1) / kGroup;
    // Fill with rows dealt 16 apart: the next group's row extents and output offsets are requested before the walk
    // of this one (before its first store), so that the next group starts with its rows' loads at once.
    constexpr int kInit = FILL ? kTile * kGroup / kBlock : 1;
    uint32_t next_off[kInit], next_beg[kRowsPerWave], next_len[kRowsPerWave];
    bool have_next = false;
    for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
        const uint32_t j0 = g * kGroup;
        // Fill: the tile of output positions of this group's pairs.  The offsets are requested here and put into
        // the tile by start_cursors(), which the walk calls once its own loads are on their way -- so that a group
        // waits once for memory, not once for the offsets and once more for its rows.
        uint32_t init_off[kInit];
        auto request_offsets = [&](uint32_t first_row, uint32_t (&dst)[kInit]) {
#pragma unroll
            for (int u = 0; u < kInit; ++u) {
                const uint32_t e = (uint32_t)u * kBlock + tid;
                const uint32_t q = e / kGroup, jl = e % kGroup;
                dst[u] = 0;
                if (q < rows_a && first_row + jl < nb) dst[u] = out_off[(uint64_t)(i0 + q) * nb + first_row + jl];
            }
        };
        if constexpr (FILL) {
            if (have_next) {
#pragma unroll
                for (int u = 0; u < kInit; ++u) init_off[u] = next_off[u];
            } else {
                request_offsets(j0, init_off);
            }
        }
        auto start_cursors = [&]() {
            if constexpr (FILL) {
#pragma unroll
                for (int u = 0; u < kInit; ++u) {
                    const uint32_t e = (uint32_t)u * kBlock + tid;
                    cnt[(e % kGroup) * kCntLd + e / kGroup] = init_off[u];
                }
                __syncthreads();
            }
        };
        if constexpr (FLAT) {
            // The four rows of a wave are one contiguous piece of B's element arrays: its elements are
            // dealt to the lanes, one lane per 32 bits of the mask (two lanes per element with 64-row
            // tiles), whatever row they belong to, so every lane has work and a lane's bit loop is short.
            constexpr uint32_t LPE = LIST ? 1 : TILE / 32;   // lanes per element
            const uint32_t r0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave * kRowsPerWave));
            uint32_t edge[kRowsPerWave + 1];
#pragma unroll
            for (int r = 0; r <= kRowsPerWave; ++r) {
                const uint32_t j = j0 + r0 + r;
                edge[r] = b_off[j < nb ? j : nb];
            }
            if (lane < (uint32_t)kRowsPerWave) {
                uint32_t len = 0;
#pragma unroll
                for (int r = 0; r < kRowsPerWave; ++r)
                    if (lane == (uint32_t)r) len = edge[r + 1] - edge[r];
                lb[r0 + lane] = len;
            }
            const uint32_t beg = edge[0], n = edge[kRowsPerWave] - edge[0];
            const uint32_t half = LPE == 2 ? lane & 1u : 0u;
            auto fetch = [&](uint32_t s0, uint32_t& id, float& vb) {
                const uint32_t e = (s0 + lane) / LPE;
                id = width;
                vb = 0.0f;
                if (e < n) {
                    id = (uint32_t)b_ids[beg + e];
                    if constexpr (NEED_B) vb = b_data[beg + e];
                }
            };
            auto visit_elem = [&](uint32_t id, float vb, uint32_t jl) {
                if constexpr (LIST) {
                    uint32_t* pair_cnt = cnt + jl * kCntLd;
                    const uint32_t end = ends[id + 1];
                    for (uint32_t at = ends[id]; at < end; ++at) {
                        const uint32_t q = list_row[at];
                        const float va = vals[at];
                        float d;
                        const bool kept = match_kept_by<MATH>(va, vb, d);
                        if constexpr (!FILL) {
                            atomicAdd(&pair_cnt[q], 0x10000u | (kept ? 1u : 0u));
                        } else if (kept) {
                            const uint32_t slot = atomicAdd(&pair_cnt[q], 1u);
                            out_ids[slot] = (int32_t)id;
                            if constexpr (MATH != kKeepAll) out_data[slot] = d;
                        }
                    }
                    return;
                }
                uint32_t m, below = 0;   // below: set bits of the element's mask that other lanes visit first
                if constexpr (LPE == 2) {
                    const uint2 both = *reinterpret_cast<const uint2*>(&mask[id]);
                    m = half ? both.y : both.x;
                    below = half ? (uint32_t)__builtin_popcount(both.x) : 0u;
                } else {
                    m = (uint32_t)mask[id];
                }
                uint32_t at = 0;   // where the payload of the next set bit is (set bits = ascending rows)
                if constexpr (NEED_A) at = (uint32_t)col[id] + below;
                uint32_t* pair_cnt = cnt + jl * kCntLd + half * 32u;
                while (m != 0) {
                    const uint32_t q = (uint32_t)__builtin_ctz(m);
                    m &= m - 1;
                    float va = 0.0f;
                    if constexpr (NEED_A) va = vals[at++];
                    float d;
                    const bool kept = match_kept_by<MATH>(va, vb, d);
                    if constexpr (!FILL) {
                        atomicAdd(&pair_cnt[q], 0x10000u | (kept ? 1u : 0u));
                    } else if (kept) {
                        const uint32_t slot = atomicAdd(&pair_cnt[q], 1u);
                        out_ids[slot] = (int32_t)id;
                        if constexpr (MATH != kKeepAll) out_data[slot] = d;
                    }
                }
            };
            auto visit_pass = [&](uint32_t s0, uint32_t id, float vb) {
                const uint32_t e = (s0 + lane) / LPE;
                uint32_t jl = r0;
#pragma unroll
                for (int r = 1; r < kRowsPerWave; ++r) jl += e >= edge[r] - beg ? 1u : 0u;
                visit_elem(id, vb, jl);
            };
            // Fill with 32-row tiles: the wave takes rows w, 16 + w, 32 + w, 48 + w of the group instead of four
            // neighbours, so that at any time the block's sixteen waves write the results of sixteen neighbouring rows
            // of B.  A 128-byte line of the result holds the elements of about seven neighbouring pairs; written this
            // way it is complete before L2 gives it up, and the kernel writes 0.31 GB to HBM at config 4 instead of
            // 1.06 GB in partly filled lines (the elements themselves are 0.30 GB).  Needs every row of the group to
            // fit PPR passes; the block votes, since all its waves must deal the rows the same way.
            bool stepped = false;
            if constexpr (FILL && LPE == 1) {
                constexpr int PPR = 6;   // passes a row: rows of up to 384 elements
                const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
                uint32_t rbeg[kRowsPerWave], rlen[kRowsPerWave];
                auto request_extents = [&](uint32_t first_row, uint32_t (&rb)[kRowsPerWave], uint32_t (&rl)[kRowsPerWave]) {
#pragma unroll
                    for (int r = 0; r < kRowsPerWave; ++r) {
                        const uint32_t j = first_row + (uint32_t)kWaves * r + wv;
                        rb[r] = b_off[j < nb ? j : nb];
                        rl[r] = j < nb ? b_off[j + 1] - rb[r] : 0u;
                    }
                };
                if (have_next) {
#pragma unroll
                    for (int r = 0; r < kRowsPerWave; ++r) {
                        rbeg[r] = next_beg[r];
                        rlen[r] = next_len[r];
                    }
                } else {
                    request_extents(j0, rbeg, rlen);
                }
                have_next = false;
                bool fits = true;
#pragma unroll
                for (int r = 0; r < kRowsPerWave; ++r) fits = fits && rlen[r] <= (uint32_t)PPR * 64u;
                if (__syncthreads_and(fits ? 1 : 0)) {   // every wave of the block takes the same rows
                    uint32_t held_id[kRowsPerWave * PPR];
                    float held_vb[kRowsPerWave * PPR];
#pragma unroll
                    for (int r = 0; r < kRowsPerWave; ++r) {
#pragma unroll
                        for (int k = 0; k < PPR; ++k) {
                            const uint32_t e = (uint32_t)k * 64u + lane;
                            held_id[r * PPR + k] = width;
                            held_vb[r * PPR + k] = 0.0f;
                            if (e < rlen[r]) {
                                held_id[r * PPR + k] = (uint32_t)b_ids[rbeg[r] + e];
                                if constexpr (NEED_B) held_vb[r * PPR + k] = b_data[rbeg[r] + e];
                            }
                        }
                    }
                    if (g + gridDim.x < n_groups) {   // uniform
                        request_extents((g + gridDim.x) * kGroup, next_beg, next_len);
                        request_offsets((g + gridDim.x) * kGroup, next_off);
                        have_next = true;
                    }
                    start_cursors();
#pragma unroll
                    for (int r = 0; r < kRowsPerWave; ++r) {
#pragma unroll
                        for (int k = 0; k < PPR; ++k)
                            if ((uint32_t)k * 64u < rlen[r]) visit_elem(held_id[r * PPR + k], held_vb[r * PPR + k], (uint32_t)kWaves * r + wv);
                    }
                    stepped = true;
                }
            }
            if (!stepped) start_cursors();
            // The fill stores inside the bit loop, and on this hardware a wait for a load also waits for
            // every store issued before it: when the wave's elements fit NP registers per lane they are
            // all fetched before the first store, so that nothing waits for a store until the group ends.
            constexpr int NP = FILL ? (LPE == 1 ? 24 : 8) : 0;
            if (stepped) {
                // the group's rows were dealt above
            } else if (FILL && LPE * n <= (uint32_t)NP * 64u) {
                uint32_t held_id[NP ? NP : 1];
                float held_vb[NP ? NP : 1];
#pragma unroll
                for (int pass = 0; pass < NP; ++pass) fetch((uint32_t)pass * 64u, held_id[pass], held_vb[pass]);
#pragma unroll
                for (int pass = 0; pass < NP; ++pass)
                    if ((uint32_t)pass * 64u < LPE * n) visit_pass((uint32_t)pass * 64u, held_id[pass], held_vb[pass]);
            } else if (!FILL) {
                // Size ops: the elements of four passes are requested ahead of the pass being walked.  With one
                // pass ahead a wave has 512 bytes in flight and the sixteen waves of a compute unit stream B at the
                // latency of memory (config 4: 0.22 ms for the loads alone, 0.11 ms this way).  The loads go
                // through bounds-checked buffers over the wave's piece and are issued unconditionally (slots past
                // the end read as zero): inside a branch the compiler would wait for ALL loads in flight at every
                // use instead of the oldest.
                constexpr int D = 4;
                __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<int32_t*>(b_ids) + beg, 0, (int)(n * 4u), 0x00020000);
                __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(b_data) + (NEED_B ? beg : 0u), 0, NEED_B ? (int)(n * 4u) : 0, 0x00020000);
                auto fetch_ahead = [&](uint32_t s0, uint32_t& id, float& vb) {
                    const uint32_t e = (s0 + lane) / LPE;
                    id = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ids_rsrc, (int)(e * 4u), 0, 0);
                    vb = 0.0f;
                    if constexpr (NEED_B)
                        vb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, (int)(e * 4u), 0, 0));
                };
                uint32_t ahead_id[D];
                float ahead_vb[D];
#pragma unroll
                for (int k = 0; k < D; ++k) fetch_ahead((uint32_t)k * 64u, ahead_id[k], ahead_vb[k]);
#pragma unroll 1
                for (uint32_t s0 = 0; s0 < LPE * n; s0 += D * 64u) {
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        const uint32_t s = s0 + (uint32_t)k * 64u;
                        const uint32_t id = (s + lane) / LPE < n ? ahead_id[k] : width;   // past the end: the sentinel id
                        const float vb = ahead_vb[k];
                        fetch_ahead(s + (uint32_t)D * 64u, ahead_id[k], ahead_vb[k]);
                        if (s < LPE * n) visit_pass(s, id, vb);
                    }
                }
            } else {
                uint32_t id, nxt_id = width;
                float vb, nxt_vb = 0.0f;
                fetch(0, id, vb);
                // the first elements are waited for here, not inside the loop, where the wait would also
                // cover the elements requested for the next turn
                asm volatile("" : "+v"(id), "+v"(vb));
#pragma unroll 1
                for (uint32_t s0 = 0; s0 < LPE * n; s0 += 64) {
                    if (s0 + 64 < LPE * n) fetch(s0 + 64, nxt_id, nxt_vb);
                    visit_pass(s0, id, vb);
                    id = nxt_id;
                    vb = nxt_vb;
                }
            }
        } else {
            start_cursors();
            // Rows of up to 64*NIT elements travel HBM -> registers whole (bounds-checked buffer loads, all
            // chunks issued back to back), and the next row is requested before the current one is
            // processed, so a wave keeps two rows in flight; NIT == 0 streams longer rows chunk by chunk.
            uint32_t cur_id[IT], nxt_id[IT];
            float cur_v[IT], nxt_v[IT];
            uint32_t cur_len = 0, nxt_len = 0;
            auto request = [&](uint32_t jl, uint32_t (&rid)[IT], float (&rv)[IT], uint32_t& rlen) {
                const uint32_t j = j0 + jl;
                const uint32_t beg = j < nb ? b_off[j] : 0u;
                rlen = j < nb ? b_off[j + 1] - beg : 0u;
                if constexpr (NIT > 0) {
                    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<int32_t*>(b_ids) + beg, 0, (int)(rlen * 4u), 0x00020000);
                    __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(b_data) + (NEED_B ? beg : 0u), 0, NEED_B ? (int)(rlen * 4u) : 0, 0x00020000);
    #pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const uint32_t e = it * 64 + lane;
                        rid[it] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ids_rsrc, (int)(e * 4u), 0, 0);
                        rv[it] = 0.0f;
                        if constexpr (NEED_B)
                            rv[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(val_rsrc, (int)(e * 4u), 0, 0));
                    }
                }
            };
            auto visit = [&](uint32_t id, float vb, uint32_t jl) {
                unsigned long long m = mask[id];
                    // every set bit is an id match of pair (q, j); four at a time so that the payload
                // gathers of a lane are in flight together
                while (__builtin_amdgcn_ballot_w64(m != 0) != 0) {
                    uint32_t q[4];
                    bool has[4];
                    float va[4];
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        has[u] = m != 0;
                        q[u] = has[u] ? (uint32_t)__builtin_ctzll(m) : 0u;
                        m &= m - 1;
                    }
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        va[u] = 0.0f;
                        if constexpr (NEED_A) {
                            if (has[u]) va[u] = val_tile[(uint64_t)q[u] * width + id];
                        }
                    }
    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (has[u]) {
                            float d;
                            const bool kept = match_kept_by<MATH>(va[u], vb, d);
                            if constexpr (!FILL) {
                                atomicAdd(&cnt[jl * kCntLd + q[u]], 0x10000u | (kept ? 1u : 0u));
                            } else if (kept) {
                                const uint32_t slot = atomicAdd(&cnt[jl * kCntLd + q[u]], 1u);
                                out_ids[slot] = (int32_t)id;
                                if constexpr (MATH != kKeepAll) out_data[slot] = d;
                            }
                        }
                    }
                }
            };
            request(wave * kRowsPerWave, cur_id, cur_v, cur_len);
    #pragma unroll 1
            for (int r = 0; r < kRowsPerWave; ++r) {
                const uint32_t jl = wave * kRowsPerWave + r;
                if (r + 1 < kRowsPerWave) request(jl + 1, nxt_id, nxt_v, nxt_len);
                if (lane == 0) lb[jl] = cur_len;
                if constexpr (NIT > 0) {
    #pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        // lanes past the row's end read nothing: give them the sentinel id (mask 0)
                        const uint32_t id = (uint32_t)(it * 64) + lane < cur_len ? cur_id[it] : width;
                        if ((uint32_t)(it * 64) < cur_len) visit(id, cur_v[it], jl);
                    }
                } else {
                    const uint32_t j = j0 + jl;
                    const uint32_t beg = j < nb ? b_off[j] : 0u;
                    for (uint32_t e0 = 0; e0 < cur_len; e0 += 64) {
                        const uint32_t e = e0 + lane;
                        uint32_t id = width;
                        float vb = 0.0f;
                        if (e < cur_len) {
                            id = (uint32_t)b_ids[beg + e];
                            if constexpr (NEED_B) vb = b_data[beg + e];
                        }
                        visit(id, vb, jl);
                    }
                }
    #pragma unroll
                for (int it = 0; it < IT; ++it) {
                    cur_id[it] = nxt_id[it];
                    cur_v[it] = nxt_v[it];
                }
                cur_len = nxt_len;
            }
        }
        __syncthreads();
        // counters -> sizes, one coalesced 256-byte row segment per wave instruction; reset the tile
        for (uint32_t e = tid; e < kTile * kGroup; e += kBlock) {
            if constexpr (FILL) break;
            const uint32_t q = e / kGroup, jl = e % kGroup;
            const uint32_t c = cnt[jl * kCntLd + q];
            cnt[jl * kCntLd + q] = 0;
            if (q < rows_a && j0 + jl < nb) {
                const uint32_t l_this = swap ? lb[jl] : la[q], l_arg = swap ? la[q] : lb[jl];
                out[(uint64_t)(i0 + q) * ld_out + j0 + jl] = size_of(op, l_this, l_arg, c >> 16, c & 0xffffu);
            }
        }
        __syncthreads();
    }
}

// Up to eight rows of A against a database whose rows hold at most 64 ids (one query, or a handful,
// with payloads): a lane owns a row of B, fetched with 16-byte loads that are all issued up front.
// LDS holds one BYTE per id (bit q: row q of A contains it), the column starts and the
// column-compressed payloads of A as in the flat loop above; a lane's match and kept counts of
// the eight pairs are bytes of two 32-bit registers each.  One launch, B read once, no atomics.
// FILL: ande -- a lane walks its row in id order, so it writes the kept matches of each of its pairs
// in order, starting at the pair's output offset (held in LDS, one column per lane).
// ROWS: 8 (a byte per id) or 16 (two bytes per id, four count registers instead of two).
constexpr int kFewRows = 16;
constexpr int kFewBlock = 256;
template <int MATH, int NQ, bool FILL, int ROWS>
__global__ __launch_bounds__(kFewBlock) void k_size_few_payload(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, const float* __restrict__ a_val,
    uint32_t na, const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids,
    const float* __restrict__ b_data, uint64_t nb, uint32_t width, int op, uint32_t swap, uint32_t val_cap,
    uint32_t* __restrict__ out, uint64_t ld_out, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, float* __restrict__ out_data) {
    constexpr bool NEED_A = MATH == kSelfA || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    constexpr bool NEED_B = MATH == kSelfB || MATH == kMul || MATH == kAbsDiff || MATH == kRmulA || MATH == kRmulB;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    static_assert(ROWS == 8 || ROWS == 16, "a byte or two per id");
    using bits_t = std::conditional_t<ROWS == 8, unsigned char, uint16_t>;
    constexpr uint32_t PER_WORD = 32 / ROWS;   // ids per 32-bit word of the bit table
    constexpr int CW = ROWS / 4;               // count registers: one byte per pair
    __shared__ uint32_t pair_base[FILL ? ROWS * kFewBlock : 1];   // [q][lane of the block]
    // LDS: vals[val_cap] | bits[width + 1 entries, rounded up to words] (entry [width] stays zero) | col[width + 1]
    extern __shared__ __attribute__((aligned(16))) unsigned char few_raw[];
    float* vals = reinterpret_cast<float*>(few_raw);
    uint32_t* bits32 = reinterpret_cast<uint32_t*>(vals + val_cap);
    const bits_t* bits = reinterpret_cast<const bits_t*>(bits32);
    const uint32_t words = (width + PER_WORD) / PER_WORD;
    uint16_t* col = reinterpret_cast<uint16_t*>(bits32 + words);
    __shared__ uint32_t wave_tot[kFewBlock / 64];

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t w = tid; w < words; w += kFewBlock) bits32[w] = 0;
    __syncthreads();
    for (uint32_t q = 0; q < na; ++q) {
        const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
        for (uint32_t k = tid; k < len; k += kFewBlock) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) atomicOr(&bits32[id / PER_WORD], (1u << q) << (ROWS * (id % PER_WORD)));   // ids outside B's width never match
        }
    }
    __syncthreads();
    if constexpr (NEED_A) {
        const uint32_t per = (width + kFewBlock) / kFewBlock;
        const uint32_t w0 = tid * per < width + 1 ? tid * per : width + 1;
        const uint32_t w1 = w0 + per < width + 1 ? w0 + per : width + 1;
        uint32_t mine = 0;
        for (uint32_t w = w0; w < w1; ++w) mine += (uint32_t)__builtin_popcount(bits[w]);
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t run = incl - mine;
        for (uint32_t w = 0; w < wave; ++w) run += wave_tot[w];
        for (uint32_t w = w0; w < w1; ++w) {
            col[w] = (uint16_t)run;
            run += (uint32_t)__builtin_popcount(bits[w]);
        }
        __syncthreads();
        for (uint32_t q = 0; q < na; ++q) {
            const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
            for (uint32_t k = tid; k < len; k += kFewBlock) {
                const uint32_t id = (uint32_t)a_ids[beg + k];
                if (id < width) vals[col[id] + (uint32_t)__builtin_popcount(bits[id] & ((1u << q) - 1u))] = a_val[beg + k];
            }
        }
        __syncthreads();
    }
    uint32_t la[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) la[q] = (uint32_t)q < na ? a_off[q + 1] - a_off[q] : 0u;

    const uint64_t pieces = (nb + kFewBlock - 1) / kFewBlock;
    for (uint64_t piece = blockIdx.x; piece < pieces; piece += gridDim.x) {
        const uint64_t j0 = piece * kFewBlock, j = j0 + tid;
        uint32_t beg = 0, lb = 0;
        if (j < nb) {
            beg = b_off[j];
            lb = b_off[j + 1] - beg;
        }
        const uint32_t blk_beg = b_off[j0];                                   // uniform
        const uint64_t blk_last = j0 + kFewBlock < nb ? j0 + kFewBlock : nb;
        const uint32_t blk_bytes = (b_off[blk_last] - blk_beg) * 4u;          // uniform
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)blk_bytes, 0x00020000);
        __amdgpu_buffer_rsrc_t val_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(b_data) + (NEED_B ? blk_beg : 0u), 0, NEED_B ? (int)blk_bytes : 0, 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
        u32x4 ids[NQ], vbs[NEED_B ? NQ : 1];
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4) {
            ids[k4] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
            if constexpr (NEED_B) vbs[k4] = __builtin_amdgcn_raw_buffer_load_b128(val_rsrc, (int)(rel + 16u * k4), 0, 0);
        }
        if constexpr (FILL) {
            for (uint32_t q = 0; q < na; ++q) pair_base[q * kFewBlock + tid] = j < nb ? out_off[(uint64_t)q * nb + j] : 0u;
        }
        uint32_t m_cnt[CW], k_cnt[CW];   // bytes: matches / kept matches of pairs 4w .. 4w + 3
#pragma unroll
        for (int w = 0; w < CW; ++w) m_cnt[w] = k_cnt[w] = 0;
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t id = (uint32_t)(4 * k4 + c) < lb ? ids[k4][c] : width;
                uint32_t m = bits[id];
                if (m != 0) {
                    float vb = 0.0f;
                    if constexpr (NEED_B) vb = __builtin_bit_cast(float, (uint32_t)vbs[k4][c]);
                    uint32_t at = 0;
                    if constexpr (NEED_A) at = col[id];
                    do {
                        const uint32_t q = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        float va = 0.0f;
                        if constexpr (NEED_A) va = vals[at++];
                        float d;
                        const uint32_t kept = match_kept_by<MATH>(va, vb, d) ? 1u : 0u;
                        const uint32_t sh = 8 * (q & 3);
                        uint32_t kept_before = 0;
#pragma unroll
                        for (int w = 0; w < CW; ++w) {
                            if ((q >> 2) == (uint32_t)w) {
                                kept_before = (k_cnt[w] >> sh) & 0xffu;
                                m_cnt[w] += 1u << sh;
                                k_cnt[w] += kept << sh;
                            }
                        }
                        if constexpr (FILL) {
                            if (kept) {
                                const uint32_t slot = pair_base[q * kFewBlock + tid] + kept_before;
                                out_ids[slot] = (int32_t)id;
                                if constexpr (MATH != kKeepAll) out_data[slot] = d;
                            }
                        }
                    } while (m != 0);
                }
            }
        }
        uint32_t* dst = out + j;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) {
            if constexpr (FILL) break;
            const uint32_t mq = (m_cnt[q / 4] >> (8 * (q & 3))) & 0xffu;
            const uint32_t kq = (k_cnt[q / 4] >> (8 * (q & 3))) & 0xffu;
            const uint32_t l_this = swap ? lb : la[q], l_arg = swap ? la[q] : lb;
            if ((uint32_t)q < na && j < nb) __builtin_nontemporal_store(size_of(op, l_this, l_arg, mq, kq), dst);
            dst += ld_out;
        }
    }
}

// Puts every pair's few elements in id order (payloads travel along), a block's pairs staged in LDS and a thread
// per ELEMENT: 256 * PPT consecutive pairs hold one contiguous piece of the result, which the block fetches with coalesced loads as (id, payload) words.  An
// element's place in its pair is the number of the pair's elements with a smaller id (ids of a pair are distinct),
// counted with independent LDS reads -- no lane waits for the longest pair of its wave to finish a serial insertion
// sort -- and only elements that move are written back.  A block whose piece exceeds the stage sorts in place in
// memory.  `stage`: elements a block can stage, a multiple of 4 (dynamic LDS: sort_stage_lds()).
template <bool DATA, int PPT>
__global__ __launch_bounds__(256) void k_sort_pairs_staged(const uint32_t* __restrict__ off, uint64_t n_pairs,
                                                           int32_t* __restrict__ ids, float* __restrict__ data,
                                                           uint32_t stage) {
    using kv_t = std::conditional_t<DATA, unsigned long long, uint32_t>;   // id (sign bit flipped) above the payload
    using own_t = std::conditional_t<PPT == 1, unsigned char, uint16_t>;
    constexpr uint32_t PPB = 256 * PPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    kv_t* kv = reinterpret_cast<kv_t*>(sort_lds);                    // [stage]
    uint32_t* edge = reinterpret_cast<uint32_t*>(kv + stage);        // [PPB + 1] pairs' offsets relative to the block's first element
    own_t* owner = reinterpret_cast<own_t*>(edge + PPB + 1);         // [stage] which of the block's pairs an element belongs to
    const uint32_t tid = threadIdx.x;
    const uint64_t p0 = (uint64_t)blockIdx.x * PPB;
    const uint32_t np = (uint32_t)(p0 + PPB < n_pairs ? PPB : n_pairs - p0);   // pairs of this block
    const uint32_t blk_lo = off[p0], span = off[p0 + np] - blk_lo;   // uniform
    // The first elements are requested together with the pairs' offsets, before it is known whether any pair of the
    // block needs sorting: one wait for memory instead of two (eight loads in flight a thread: with one, a block's
    // few thousand bytes arrive at the latency of memory).
    const bool staged = span <= stage;
    uint32_t key[8], val[8];
    auto request = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t e = base + u * 256 + tid;
            key[u] = e < span ? (uint32_t)ids[blk_lo + e] : 0u;
            val[u] = (DATA && e < span) ? __float_as_uint(data[blk_lo + e]) : 0u;
        }
    };
    auto stage_them = [&](uint32_t base) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t e = base + u * 256 + tid;
            if (e < span) {
                if constexpr (DATA) kv[e] = ((unsigned long long)(key[u] ^ 0x80000000u) << 32) | val[u];
                else kv[e] = key[u] ^ 0x80000000u;
            }
        }
    };
    if (staged) request(0);
    bool work = false;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const uint32_t i = u * 256 + tid;
        const uint32_t lo = off[p0 + (i < np ? i : np)], hi = off[p0 + (i + 1 < np ? i + 1 : np)];
        edge[i] = lo - blk_lo;
        work |= hi - lo >= 2u;
    }
    if (tid == 0) edge[PPB] = span;
    if (!__syncthreads_or(work)) return;
    if (span > stage) {
#pragma unroll 1
        for (int u = 0; u < PPT; ++u) {
            const uint32_t i = u * 256 + tid;
            const uint32_t lo = blk_lo + edge[i], hi = blk_lo + edge[i + 1];
            for (uint32_t k = lo + 1; k < hi; ++k) {
                const int32_t id = ids[k];
                const float d = DATA ? data[k] : 0.0f;
                uint32_t s = k;
                while (s > lo && ids[s - 1] > id) {
                    ids[s] = ids[s - 1];
                    if constexpr (DATA) data[s] = data[s - 1];
                    --s;
                }
                if (s != k) {
                    ids[s] = id;
                    if constexpr (DATA) data[s] = d;
                }
            }
        }
        return;
    }
    stage_them(0);
    for (uint32_t base = 8 * 256; base < span; base += 8 * 256) {
        request(base);
        stage_them(base);
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const uint32_t i = u * 256 + tid;
        for (uint32_t e = edge[i]; e < edge[i + 1]; ++e) owner[e] = (own_t)i;
    }
    __syncthreads();
    for (uint32_t e = tid; e < span; e += 256) {
        const uint32_t p = owner[e];
        const uint32_t l0 = edge[p], l1 = edge[p + 1];
        if (l1 - l0 < 2u) continue;
        const kv_t x = kv[e];
        uint32_t at = l0;
        for (uint32_t k = l0; k < l1; ++k) at += kv[k] < x ? 1u : 0u;
        if (at != e) {
            if constexpr (DATA) {
                ids[blk_lo + at] = (int32_t)((uint32_t)(x >> 32) ^ 0x80000000u);
                data[blk_lo + at] = __uint_as_float((uint32_t)x);
            } else {
                ids[blk_lo + at] = (int32_t)(x ^ 0x80000000u);
            }
        }
    }
}

size_t sort_stage_lds(uint32_t stage, bool with_data, int ppt) {
    return (size_t)stage * (with_data ? 8 : 4) + (size_t)(256 * ppt + 1) * 4 + (size_t)stage * (ppt == 1 ? 1 : 2);
}

size_t lds_bytes(uint32_t width, int tile = 64) {
    const size_t entries = (size_t)(width + 1) + ((width + 1) & 1u);
    return entries * (tile / 8) + (size_t)kGroup * cnt_ld(tile) * 4 + (tile + kGroup) * 4;
}

// with the tile's payloads in LDS: val_cap floats and a 16-bit column start per table entry
size_t lds_bytes_vals(uint32_t width, uint32_t val_cap, int tile) {
    return lds_bytes(width, tile) + (size_t)val_cap * 4 + (size_t)(width + 2) * 2;
}

// the lists of a 64-row tile instead of a mask: ends[width + 2], the counter tile, payloads (the build's counters lie in
// their place first: width + 2 words at least) and a row byte per element
size_t lds_bytes_lists(uint32_t width, uint32_t val_cap) {
    const size_t vals_room = val_cap < width + 2 ? width + 2 : val_cap;
    return ((((size_t)width + 2) * 2 + 15) & ~(size_t)15) + (size_t)kGroup * cnt_ld(64) * 4 + (64 + kGroup) * 4 + vals_room * 4 + val_cap;
}

}  // namespace

bool mask_table_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    // the mask table and the counter tile must fit the 160 KiB of LDS; counters hold row lengths in 16 bits
    return b->width > 0 && lds_bytes((uint32_t)b->width) <= 160 * 1024 - 512 && a->max_row_len < 65536 &&
           b->max_row_len < 65536 && (a->n_rows + 31) / 32 <= 65535;
}

namespace {

int run_mask_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, bool fill, uint32_t* out,
                   uint64_t ld_out, const uint32_t* out_off, int32_t* out_ids, float* out_data, hipStream_t stream) {
    const uint32_t width = (uint32_t)b->width, na = (uint32_t)a->n_rows, nb = (uint32_t)b->n_rows;
    const int math = pick_math(a, b, op, swap);
    const bool need_a = math == kSelfA || math == kMul || math == kAbsDiff || math == kRmulA || math == kRmulB;
    // the tile's payloads in LDS when they fit beside a 64-row table, else beside a 32-row table
    uint32_t val_cap = 0;
    int tile = 64;
    bool lists = false;   // 64-row tiles kept as per-id lists where the 64-row mask leaves no room for the payloads
    if (need_a && a->max_row_len < 1024) {
        const uint32_t longest = a->max_row_len < width ? a->max_row_len : width;
        static const bool no_lists = getenv("SDRB_MASKTABLE_NO_LISTS") != nullptr;   // measurements of the 32-row tiles
        for (int t : {64, 0, 32}) {   // 0: the lists
            const uint32_t cap = (uint32_t)(t ? t : 64) * (longest ? longest : 1u);
            if (t == 0 ? (!no_lists && cap <= 65535u && lds_bytes_lists(width, cap) <= 160 * 1024 - 512)
                       : lds_bytes_vals(width, cap, t) <= 160 * 1024 - 512) {
                val_cap = cap;
                tile = t ? t : 64;
                lists = t == 0;
                break;
            }
        }
    }
    // flat: B's elements dealt to lanes regardless of rows; needs the payloads of A (if any) in LDS.
    // Without payloads of A it pays where two blocks share a compute unit; with one block (wide
    // tables) whole rows prefetched into registers hide the latency better.
    const bool flat = need_a ? val_cap != 0 : lds_bytes(width) <= 80 * 1024 - 256;
    float* dense = nullptr;
    if (need_a && !flat) {
        const size_t bytes = (size_t)na * width * sizeof(float);
        dense = static_cast<float*>(workspace(b->device, stream, 6, bytes));
        if (!dense) return SDRB_ERR_HIP;
        // only entries the mask points at are ever read, so the matrix needs no clearing
        hipLaunchKernelGGL(k_scatter_values, dim3(na), dim3(64), 0, stream, a->d_offsets, a->d_ids, a->d_data, width,
                           dense);
        SDRB_HIP(hipGetLastError());
    }
    const size_t lds = lists ? lds_bytes_lists(width, val_cap) : val_cap ? lds_bytes_vals(width, val_cap, tile) : lds_bytes(width);
    const uint32_t tiles = (na + tile_step(tile) - 1) / tile_step(tile);
    const uint32_t groups = (nb + kGroup - 1) / kGroup;
    // one or two resident blocks per CU, shared among the tiles of A
    const uint32_t per_cu = lds <= 80 * 1024 - 256 ? 2u : 1u;
    uint32_t gx = (256u * per_cu + tiles - 1) / tiles;
    if (gx > groups) gx = groups;
    if (gx == 0) gx = 1;
    const dim3 grid(gx, tiles);
    const uint32_t len = b->max_row_len;
    const int nit = len <= 64 ? 1 : len <= 128 ? 2 : len <= 384 ? 6 : 0;
#define SDRB_MT_LAUNCH_NFL(M, N, F, L, T)                                                                             \
    do {                                                                                                             \
        SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_size_mask_table<M, N, F, L, T>),               \
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                         \
        hipLaunchKernelGGL((k_size_mask_table<M, N, F, L, T>), grid, dim3(kBlock), lds, stream, a->d_offsets,        \
                           a->d_ids,                                                                                \
                           dense, na, b->d_offsets, b->d_ids, b->d_data, nb, width, op, swap ? 1u : 0u, out, ld_out, \
                           out_off, out_ids, out_data, a->d_data, val_cap);                                          \
    } while (0)
#define SDRB_MT_LAUNCH_N(M, N)                          \
    do {                                                \
        if (fill) SDRB_MT_LAUNCH_NFL(M, N, true, false, 64);  \
        else SDRB_MT_LAUNCH_NFL(M, N, false, false, 64);      \
    } while (0)
#define SDRB_MT_LAUNCH_LISTS(M, F)                                                                                     \
    do {                                                                                                             \
        if constexpr (M == kSelfA || M == kMul || M == kAbsDiff || M == kRmulA || M == kRmulB) {                     \
            SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_size_mask_table<M, 0, F, true, 64, true>), \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
            hipLaunchKernelGGL((k_size_mask_table<M, 0, F, true, 64, true>), grid, dim3(kBlock), lds, stream,        \
                               a->d_offsets, a->d_ids, dense, na, b->d_offsets, b->d_ids, b->d_data, nb, width, op,  \
                               swap ? 1u : 0u, out, ld_out, out_off, out_ids, out_data, a->d_data, val_cap);         \
        }                                                                                                            \
    } while (0)
#define SDRB_MT_LAUNCH(M)                          \
    do {                                           \
        if (lists) {                               \
            if (fill) SDRB_MT_LAUNCH_LISTS(M, true);   \
            else SDRB_MT_LAUNCH_LISTS(M, false);       \
        } else if (flat && tile == 32) {           \
            if (fill) SDRB_MT_LAUNCH_NFL(M, 0, true, true, 32);  \
            else SDRB_MT_LAUNCH_NFL(M, 0, false, true, 32);      \
        } else if (flat) {                         \
            if (fill) SDRB_MT_LAUNCH_NFL(M, 0, true, true, 64);  \
            else SDRB_MT_LAUNCH_NFL(M, 0, false, true, 64);      \
        } else if (nit == 1) SDRB_MT_LAUNCH_N(M, 1);      \
        else if (nit == 2) SDRB_MT_LAUNCH_N(M, 2); \
        else if (nit == 6) SDRB_MT_LAUNCH_N(M, 6); \
        else SDRB_MT_LAUNCH_N(M, 0);               \
    } while (0)
    switch (math) {
    case kKeepAll: SDRB_MT_LAUNCH(kKeepAll); break;
    case kDropAll: SDRB_MT_LAUNCH(kDropAll); break;
    case kSelfA: SDRB_MT_LAUNCH(kSelfA); break;
    case kSelfB: SDRB_MT_LAUNCH(kSelfB); break;
    case kMul: SDRB_MT_LAUNCH(kMul); break;
    case kAbsDiff: SDRB_MT_LAUNCH(kAbsDiff); break;
    case kRmulA: SDRB_MT_LAUNCH(kRmulA); break;
    default: SDRB_MT_LAUNCH(kRmulB); break;
    }
#undef SDRB_MT_LAUNCH
#undef SDRB_MT_LAUNCH_LISTS
#undef SDRB_MT_LAUNCH_N
#undef SDRB_MT_LAUNCH_NFL
    SDRB_HIP(hipGetLastError());
    if (fill) {
        const uint64_t pairs = (uint64_t)na * nb;
        // expected matches per pair: sparse results take four times as many pairs a block
        const double expected = (double)a->nnz / (double)na * ((double)b->nnz / (double)nb) / (double)width;
        const int ppt = expected < 1.5 ? 4 : 1;
        const dim3 sort_grid((uint32_t)((pairs + 256ull * ppt - 1) / (256ull * ppt)));
        int32_t* const sort_ids = out_ids;
        const bool with_data = math != kKeepAll;
        float* const sort_data = with_data ? out_data : nullptr;
        // room for the expected id matches of a block's pairs and a sixth more (payload rules only drop matches; a
        // block that holds more still sorts, in place), 2048 or 4096 elements: a smaller stage lets more blocks share
        // a compute unit
        const uint32_t stage = expected * 300.0 * ppt <= 2048.0 ? 2048u : 4096u;
        const size_t sort_lds = sort_stage_lds(stage, with_data, ppt);
        if (ppt == 1 && !with_data)
            hipLaunchKernelGGL((k_sort_pairs_staged<false, 1>), sort_grid, dim3(256), sort_lds, stream, out_off, pairs, sort_ids, sort_data, stage);
        else if (ppt == 1)
            hipLaunchKernelGGL((k_sort_pairs_staged<true, 1>), sort_grid, dim3(256), sort_lds, stream, out_off, pairs, sort_ids, sort_data, stage);
        else if (!with_data)
            hipLaunchKernelGGL((k_sort_pairs_staged<false, 4>), sort_grid, dim3(256), sort_lds, stream, out_off, pairs, sort_ids, sort_data, stage);
        else
            hipLaunchKernelGGL((k_sort_pairs_staged<true, 4>), sort_grid, dim3(256), sort_lds, stream, out_off, pairs, sort_ids, sort_data, stage);
        SDRB_HIP(hipGetLastError());
    }
    return SDRB_OK;
}

}  // namespace

namespace {

size_t few_lds_bytes(uint32_t width, uint32_t val_cap, int rows) {
    const uint32_t per_word = 32u / (uint32_t)rows;
    return (size_t)val_cap * 4 + (size_t)((width + per_word) / per_word) * 4 + (size_t)(width + 2) * 2;
}

// the few-rows kernel takes the call when it can; *taken says whether it did
int run_few_payload(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, bool fill, uint32_t* out,
                    uint64_t ld_out, const uint32_t* out_off, int32_t* out_ids, float* out_data, hipStream_t stream,
                    bool* taken) {
    *taken = false;
    if (a->n_rows > kFewRows || b->max_row_len > 64 || a->max_row_len >= 8192) return SDRB_OK;
    const uint32_t width = (uint32_t)b->width;
    const uint32_t longest = a->max_row_len < width ? a->max_row_len : width;
    const uint32_t val_cap = (uint32_t)a->n_rows * (longest ? longest : 1u);
    if (val_cap > 65535u) return SDRB_OK;   // col[] is 16 bit
    const int rows = a->n_rows <= 8 ? 8 : 16;
    const size_t lds = few_lds_bytes(width, val_cap, rows);
    if (lds + (fill ? (size_t)rows * kFewBlock * 4 : 0) > 64 * 1024 - 256) return SDRB_OK;
    *taken = true;
    const int math = pick_math(a, b, op, swap);
    const uint32_t quads = (b->max_row_len + 3) / 4;
    const uint64_t pieces = ((uint64_t)b->n_rows + kFewBlock - 1) / kFewBlock;
    const uint32_t per_cu = lds <= 8 * 1024 ? 8u : lds <= 16 * 1024 ? 6u : lds <= 32 * 1024 ? 4u : 2u;
    const uint32_t grid = (uint32_t)(pieces < 256ull * per_cu ? pieces : 256ull * per_cu);
#define SDRB_FEWP_LAUNCH_NFR(M, N, F, R)                                                                             \
    hipLaunchKernelGGL((k_size_few_payload<M, N, F, R>), dim3(grid), dim3(kFewBlock), lds, stream, a->d_offsets,     \
                       a->d_ids, a->d_data, (uint32_t)a->n_rows, b->d_offsets, b->d_ids, b->d_data,                  \
                       (uint64_t)b->n_rows, width, op, swap ? 1u : 0u, val_cap, out, ld_out, out_off, out_ids,       \
                       out_data)
#define SDRB_FEWP_LAUNCH_NF(M, N, F)                     \
    do {                                                 \
        if (rows == 8) SDRB_FEWP_LAUNCH_NFR(M, N, F, 8); \
        else SDRB_FEWP_LAUNCH_NFR(M, N, F, 16);          \
    } while (0)
#define SDRB_FEWP_LAUNCH_N(M, N)                     \
    do {                                             \
        if (fill) SDRB_FEWP_LAUNCH_NF(M, N, true);   \
        else SDRB_FEWP_LAUNCH_NF(M, N, false);       \
    } while (0)
#define SDRB_FEWP_LAUNCH(M)                              \
    do {                                                 \
        if (quads <= 4) SDRB_FEWP_LAUNCH_N(M, 4);        \
        else if (quads <= 10) SDRB_FEWP_LAUNCH_N(M, 10); \
        else SDRB_FEWP_LAUNCH_N(M, 16);                  \
    } while (0)
    switch (math) {
    case kKeepAll: SDRB_FEWP_LAUNCH(kKeepAll); break;
    case kDropAll: SDRB_FEWP_LAUNCH(kDropAll); break;
    case kSelfA: SDRB_FEWP_LAUNCH(kSelfA); break;
    case kSelfB: SDRB_FEWP_LAUNCH(kSelfB); break;
    case kMul: SDRB_FEWP_LAUNCH(kMul); break;
    case kAbsDiff: SDRB_FEWP_LAUNCH(kAbsDiff); break;
    case kRmulA: SDRB_FEWP_LAUNCH(kRmulA); break;
    default: SDRB_FEWP_LAUNCH(kRmulB); break;
    }
#undef SDRB_FEWP_LAUNCH
#undef SDRB_FEWP_LAUNCH_N
#undef SDRB_FEWP_LAUNCH_NF
#undef SDRB_FEWP_LAUNCH_NFR
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace

int launch_size_mask_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out,
                           uint64_t ld_out, hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    bool taken = false;
    const int rc = run_few_payload(a, b, op, swap, false, out, ld_out, nullptr, nullptr, nullptr, stream, &taken);
    if (rc != SDRB_OK || taken) return rc;
    return run_mask_table(a, b, op, swap, false, out, ld_out, nullptr, nullptr, nullptr, stream);
}

// ande for every pair: the kept matches of a pair, (id, combined payload), in id order
int launch_elem_mask_table(const sdrb_batch* a, const sdrb_batch* b, bool swap, const uint32_t* out_offsets,
                           int32_t* out_ids, float* out_data, hipStream_t stream) {
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    bool taken = false;
    const int rc = run_few_payload(a, b, SDRB_OP_AND, swap, true, nullptr, 0, out_offsets, out_ids, out_data, stream, &taken);
    if (rc != SDRB_OK || taken) return rc;
    return run_mask_table(a, b, SDRB_OP_AND, swap, true, nullptr, 0, out_offsets, out_ids, out_data, stream);
}

}  // namespace sdrb
