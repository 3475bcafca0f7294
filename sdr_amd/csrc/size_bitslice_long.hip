// The bit-sliced table kernel for LONG rows and WIDE tables: id-only size operations (`this` EmptyData, or ors) when the
// rows of B hold more than 64 ids or the declared width exceeds 4096 (size_bitslice.hip serves the rest).
//
// Same formulation: table[t][w] has bit q set when row TILE * t + q of A contains id w; a lane owns a row of B, gathers
// the table words of its ids from LDS and adds them into vertical counters (plane p = bit p of the TILE match counts).
// What differs:
//   * a row no longer fits registers: the lane streams it, eight ids a turn (two 16-byte loads from its own row, the
//     next turn's ids requested before this turn's gathers), for every tile of A again -- from L2 after the first time;
//   * the table tile is as wide as the batch (dynamic LDS): 64 rows of A per tile (64-bit words) while that fits half
//     the LDS, else 32 rows (32-bit words) up to 160 KiB -- widths to about 40 000;
//   * counts may exceed a byte: the planes are unpacked two counts a register (16 bits each).
// Reference semantics as in size_bitslice.hip: SDR.hpp:1083-1095, 1294-1304, 1424-1434, 1568-1577 on EmptyData
// (EmptyData.hpp:50-51, 70-71, 79-80).
#include <type_traits>

#include "bitslice_device.h"

namespace sdrb {

namespace {

constexpr size_t kLdsBudget = 160 * 1024 - 512;

template <typename MASK>
__global__ __launch_bounds__(256) void k_build_table_long(const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids,
                                                          uint32_t n_rows, uint32_t width, uint32_t tile_words,
                                                          MASK* __restrict__ table, uint32_t* __restrict__ a_len) {
    constexpr uint32_t TILE = sizeof(MASK) * 8;
    const uint64_t i = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n_rows) return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    if (lane == 0) a_len[i] = len;
    MASK* tile = table + (i / TILE) * (uint64_t)tile_words;
    const MASK bit = (MASK)1 << (i % TILE);
    for (uint32_t k = lane; k < len; k += 64) {
        const uint32_t id = (uint32_t)a_ids[beg + k];
        if (id < width) atomicOr(&tile[id], bit);   // ids outside B's declared width cannot match any id of B
    }
}

// MODE : 0 -> out = m;  1 -> out = L - m;  2 -> out = L - 2m   (m = id matches, L = use_la*|A_i| + use_lb*|B_j|)
// MASK : uint32_t (32 rows of A per tile) or u64 (64 rows)
// NPL  : planes of the vertical counters = bits of the longest row of B
// TB   : threads per block = rows of B per block (512 when the tile leaves room for one block a compute unit only)
template <int MODE, typename MASK, int NPL, int TB>
__global__ __launch_bounds__(TB) void k_size_bitslice_long(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint64_t nb, const MASK* __restrict__ table,
    const uint32_t* __restrict__ a_len, uint32_t ntiles, uint32_t tiles_per_block, uint32_t na, uint32_t width,
    uint32_t tile_words, uint32_t use_la, uint32_t use_lb, uint32_t* __restrict__ out, uint64_t ld_out) {
    constexpr uint32_t TILE = sizeof(MASK) * 8;
    constexpr int HALVES = sizeof(MASK) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char long_lds[];
    MASK* tab = reinterpret_cast<MASK*>(long_lds);

    const uint32_t tid = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * TB;
    const uint64_t j = j0 + tid;
    uint32_t beg = 0, lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    const uint32_t blk_beg = b_off[j0];                                   // uniform
    const uint64_t blk_last = j0 + TB < nb ? j0 + TB : nb;
    const uint32_t blk_end = b_off[blk_last];                             // uniform
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)((blk_end - blk_beg) * 4u), 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
    // turns of the wave: the longest row of its 64 (shorter rows gather the zero word at [width])
    uint32_t longest = lb;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)longest, d, 64);
        longest = o > longest ? o : longest;
    }
    const uint32_t turns = (uint32_t)__builtin_amdgcn_readfirstlane((int)((longest + 7) / 8));
    const uint32_t lbterm = use_lb ? lb : 0u;
    const uint32_t zero_at = width * (uint32_t)sizeof(MASK);

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    const uint64_t row_bytes = ld_out * 4;
    const uint64_t rows_here = nb - j0 < (uint64_t)TB ? nb - j0 : (uint64_t)TB;
    const uint32_t nrec = (uint32_t)rows_here * 4;
    const uint32_t voff = tid * 4;
    char* const out_block = reinterpret_cast<char*>(out) + j0 * 4;
    const char* base = reinterpret_cast<const char*>(tab);

    for (uint32_t t = t0; t < t1; ++t) {
        __syncthreads();   // the previous tile is no longer read
        {
            const u32x4* src = reinterpret_cast<const u32x4*>(table + (uint64_t)t * tile_words);
            u32x4* dst = reinterpret_cast<u32x4*>(tab);
            const uint32_t vecs = tile_words * (uint32_t)sizeof(MASK) / 16;
            for (uint32_t v = tid; v < vecs; v += TB) dst[v] = src[v];
        }
        __syncthreads();
        uint32_t P[HALVES][NPL];
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int p = 0; p < NPL; ++p) P[h][p] = 0;
        // the ids of three turns are in flight: the loads are unconditional (bounds-checked: words past the block's
        // rows read as zero), so the compiler keeps their distance instead of waiting for all of them at every use
        u32x4 q0[3], q1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            q0[d] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 32u * d), 0, 0);
            q1[d] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 32u * d + 16u), 0, 0);
        }
        for (uint32_t c = 0; c < turns; ++c) {
            const u32x4 cur0 = q0[0], cur1 = q1[0];
            q0[0] = q0[1];
            q1[0] = q1[1];
            q0[1] = q0[2];
            q1[1] = q1[2];
            q0[2] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 32u * (c + 3)), 0, 0);
            q1[2] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 32u * (c + 3) + 16u), 0, 0);
            uint32_t x[HALVES][8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t id = k < 4 ? cur0[k & 3] : cur1[k & 3];
                const uint32_t at = c * 8 + k < lb ? id * (uint32_t)sizeof(MASK) : zero_at;
                if constexpr (HALVES == 2) {
                    const u64 w = *reinterpret_cast<const u64*>(base + at);
                    x[0][k] = (uint32_t)w;
                    x[1][k] = (uint32_t)(w >> 32);
                } else {
                    x[0][k] = *reinterpret_cast<const uint32_t*>(base + at);
                }
            }
#pragma unroll
            for (int h = 0; h < HALVES; ++h) add8<NPL, false>(P[h], x[h]);
        }
        uint32_t la_vec = 0;
        if constexpr (MODE != 0) {
            if (use_la && (tid & 63) < TILE) la_vec = a_len[(uint64_t)t * TILE + (tid & 63)];
        }
        const uint32_t rows_left = na - t * TILE;
        const uint32_t rows_in_tile = rows_left < TILE ? rows_left : TILE;
        char* row_ptr = out_block + (uint64_t)t * TILE * row_bytes;
#pragma unroll
        for (int h = 0; h < HALVES; ++h) {
            // rows r and r + 16 of this half: bit r of plane p -> bit p of the low half of acc[r], bit r + 16 -> high half
            uint32_t acc[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                uint32_t a = 0;
#pragma unroll
                for (int p = 0; p < NPL; ++p) a |= rotr32(P[h][p], r - p) & (0x00010001u << p);
                acc[r] = a;
            }
#pragma unroll
            for (int qq = 0; qq < 32; ++qq) {
                const int q = 32 * h + qq;
                uint32_t v = qq < 16 ? (acc[qq] & 0xffffu) : (acc[qq - 16] >> 16);
                if constexpr (MODE != 0) {
                    const uint32_t la = (uint32_t)__builtin_amdgcn_readlane((int)la_vec, q);
                    v = (la + lbterm) - (MODE == 2 ? 2u * v : v);
                }
                const uint32_t n = (uint32_t)q < rows_in_tile ? nrec : 0u;
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)n, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)voff, 0, 0);
                row_ptr += row_bytes;
            }
        }
    }
}

struct LongPlan {
    int mask_bytes;       // 4: 32 rows of A per tile, 8: 64 rows
    int npl;
    uint32_t tile_words;  // table words per tile: width + 1 (the zero word), rounded up to whole 16-byte vectors
    size_t lds;
};

bool make_long_plan(const sdrb_batch* a, const sdrb_batch* b, LongPlan* p) {
    (void)a;
    if (a->id_bits != 32 || b->id_bits != 32 || b->width <= 0 || b->max_row_len == 0 || b->max_row_len > 65535) return false;
    const uint32_t words = ((uint32_t)b->width + 1 + 3) & ~3u;
    // 64 rows per tile while two blocks fit a compute unit; the tile as wide as the LDS allows otherwise
    if ((size_t)words * 8 <= kLdsBudget / 2) p->mask_bytes = 8;
    else if ((size_t)words * 4 <= kLdsBudget) p->mask_bytes = 4;
    else return false;
    if (const char* env = getenv("SDRB_LONG_MASK_BYTES")) {      // experiments: force the tile height where it fits
        const int want = atoi(env);
        if ((want == 4 || want == 8) && (size_t)words * want <= kLdsBudget) p->mask_bytes = want;
    }
    p->tile_words = words;
    p->lds = (size_t)words * p->mask_bytes;
    const int bits = bits_for((int)b->max_row_len);
    p->npl = bits <= 7 ? 7 : bits <= 9 ? 9 : bits <= 12 ? 12 : 16;
    return true;
}

struct LongArgs {
    const sdrb_batch* a;
    const sdrb_batch* b;
    const void* table;
    const uint32_t* a_len;
    uint32_t ntiles, tiles_per_block, use_la, use_lb, tile_words;
    dim3 grid;
    int block;
    size_t lds;
    uint32_t* out;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int MODE, typename MASK, int NPL, int TB>
int launch_long_tb(const LongArgs& g) {
    if (g.lds > 64 * 1024)
        SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_size_bitslice_long<MODE, MASK, NPL, TB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
    hipLaunchKernelGGL((k_size_bitslice_long<MODE, MASK, NPL, TB>), g.grid, dim3(TB), g.lds, g.stream, g.b->d_offsets, g.b->d_ids,
                       g.b->n_rows, static_cast<const MASK*>(g.table), g.a_len, g.ntiles, g.tiles_per_block, (uint32_t)g.a->n_rows,
                       (uint32_t)g.b->width, g.tile_words, g.use_la, g.use_lb, g.out, g.ld_out);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <int MODE, typename MASK, int NPL>
int launch_long(const LongArgs& g) {
    if (g.block == 512) return launch_long_tb<MODE, MASK, NPL, 512>(g);
    return launch_long_tb<MODE, MASK, NPL, 256>(g);
}

template <int MODE, typename MASK>
int launch_long_npl(const LongArgs& g, int npl) {
    switch (npl) {
    case 7: return launch_long<MODE, MASK, 7>(g);
    case 9: return launch_long<MODE, MASK, 9>(g);
    case 12: return launch_long<MODE, MASK, 12>(g);
    default: return launch_long<MODE, MASK, 16>(g);
    }
}

template <int MODE>
int launch_long_mask(const LongArgs& g, const LongPlan& p) {
    if (p.mask_bytes == 8) return launch_long_npl<MODE, u64>(g, p.npl);
    return launch_long_npl<MODE, uint32_t>(g, p.npl);
}

}  // namespace

bool bitslice_long_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    LongPlan p;
    return make_long_plan(a, b, &p);
}

int launch_size_bitslice_long(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                              hipStream_t stream) {
    LongPlan plan;
    if (!make_long_plan(a, b, &plan)) {
        set_error("the bit-sliced kernel for long rows needs 32-bit ids and a declared width whose table fits the LDS (about 40 000)");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const uint32_t tile_rows = (uint32_t)plan.mask_bytes * 8;
    LongArgs g;
    g.a = a;
    g.b = b;
    g.ntiles = (uint32_t)((a->n_rows + tile_rows - 1) / tile_rows);
    g.tile_words = plan.tile_words;
    g.lds = plan.lds;
    const size_t table_bytes = (size_t)g.ntiles * plan.tile_words * plan.mask_bytes;
    const size_t len_bytes = (size_t)g.ntiles * tile_rows * sizeof(uint32_t);
    char* ws = static_cast<char*>(workspace(b->device, stream, 0, table_bytes + len_bytes));
    if (!ws) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(ws, 0, table_bytes + len_bytes, stream));
    uint32_t* a_len = reinterpret_cast<uint32_t*>(ws + table_bytes);
    const dim3 build_grid((uint32_t)((a->n_rows + 3) / 4));
    if (plan.mask_bytes == 8)
        hipLaunchKernelGGL((k_build_table_long<u64>), build_grid, dim3(256), 0, stream, a->d_offsets, a->d_ids, (uint32_t)a->n_rows,
                           (uint32_t)b->width, plan.tile_words, reinterpret_cast<u64*>(ws), a_len);
    else
        hipLaunchKernelGGL((k_build_table_long<uint32_t>), build_grid, dim3(256), 0, stream, a->d_offsets, a->d_ids, (uint32_t)a->n_rows,
                           (uint32_t)b->width, plan.tile_words, reinterpret_cast<uint32_t*>(ws), a_len);
    SDRB_HIP(hipGetLastError());
    g.table = ws;
    g.a_len = a_len;
    // a tile that leaves room for one block a compute unit is shared by eight waves instead of four
    g.block = plan.lds > kLdsBudget / 2 ? 512 : 256;
    if (const char* env = getenv("SDRB_LONG_BLOCK")) {
        const int want = atoi(env);
        if (want == 256 || want == 512) g.block = want;
    }
    const uint32_t gx = (uint32_t)((b->n_rows + g.block - 1) / g.block);
    uint32_t gy = gx >= 1024 ? 1u : (1024u + gx - 1) / gx;
    if (gy > g.ntiles) gy = g.ntiles;
    g.tiles_per_block = (g.ntiles + gy - 1) / gy;
    gy = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    g.grid = dim3(gx, gy);
    g.out = out;
    g.ld_out = ld_out;
    g.stream = stream;
    switch (op) {
    case SDRB_OP_AND:
        g.use_la = g.use_lb = 0;
        return launch_long_mask<0>(g, plan);
    case SDRB_OP_OR:
        g.use_la = g.use_lb = 1;
        return launch_long_mask<1>(g, plan);
    case SDRB_OP_XOR:
        g.use_la = g.use_lb = 1;
        return launch_long_mask<2>(g, plan);
    case SDRB_OP_RM:   // |this| - matches
        g.use_la = swap ? 0 : 1;
        g.use_lb = swap ? 1 : 0;
        return launch_long_mask<1>(g, plan);
    default:
        set_error("unknown op %d", op);
        return SDRB_ERR_INVALID_ARG;
    }
}

}  // namespace sdrb
