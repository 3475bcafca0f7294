// Bit-sliced table kernel for the id-only size operations (ands/ors/xors/rms whose result
// depends on ids alone: `this` side EmptyData, or ors on any payloads).
//
// For one pair the reference walks two sorted id lists with andv (SDR.hpp:958-997) and
// counts the matches (SDR.hpp:1083-1095); every size op is then a function of |this|,
// |arg| and that count (SDR.hpp:1303, 1433, 1576 with EmptyData.hpp:50-51,70-71,79-80).
// Here the count for 64 rows of A against one row of B is computed at once:
//
//   table[t][w]  (uint64)  bit q set  <=>  row 64*t+q of A contains id w
//
// is built once per call (1 MiB for 4096 rows of width 2048).  A lane owns one row j of B,
// keeps that row's ids in registers as LDS byte offsets, and for each tile t gathers the
// row's table words from LDS and adds them with carry-save adders into bit planes
// ("vertical counters"): plane p holds bit p of the 64 match counts.  The planes are then
// unpacked four counts at a time into bytes and stored as uint32, one 256-byte coalesced
// store per wave per A row.  Each row of B is read from HBM once; the output write
// (4 bytes per pair) is the only large stream, so the kernel is bound by HBM write bandwidth.
#include <mutex>
#include <type_traits>

#include "bitslice_device.h"

namespace sdrb {

namespace {

// Scratch of the entry point that allocates nothing (sdrb_size_op_static): the bit table of A and the
// lengths of its rows.  Every GPU the library runs on holds its own copy of this array until the process
// ends.  In host code the name is only a handle for hipGetSymbolAddress, never a device address.
constexpr size_t kStaticScratchBytes = (2u << 20) + (32u << 10);
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
std::mutex g_static_mutex;
void* g_static_ptr[kMaxDevices];   // device address of `scratch` on each device, resolved once

// One tile: gather the row's table words from LDS buffer `buf`, count, unpack, store.
// `rows_in_tile` < 64 only for the last tile of A; stores to the missing rows are switched
// off by a zero-sized buffer descriptor, lanes past the last row of B by `nrec`.
//
// OUTB = 4: a lane stores the uint32 count of its row of B for one row of A per instruction (256 B per wave and row).
// OUTB = 1 / 2 (counts as bytes / 16-bit words): the four lanes of a quad first exchange their registers -- a 4 x 4
// byte transpose in two steps of a DPP move and a byte permute -- so that a lane holds the counts of ONE row of A for
// the quad's four neighbouring rows of B as one dword (two for 16-bit counts); these go into a stage in LDS, and
// the block's waves carry the stage out a whole row piece (256 or 512 contiguous bytes) per instruction: a quarter
// of the store instructions for a quarter (half) of the bytes.  `narrow` carries what the narrow forms need per lane
// (see NarrowLane).
// rows of the LDS stage are a little longer than their 256 counts (4 bytes; 8 for 16-bit counts, whose accesses are
// 8 bytes wide): the four rows a quad writes then start in different banks
template <int OUTB>
constexpr uint32_t kStageStride = 256u * OUTB + 4u * OUTB;
template <int OUTB>
constexpr int kNarrowStages = OUTB == 1 ? 2 : 1;

struct NarrowLane {
    uint32_t sel_a, sel_b;     // byte selectors of the two transpose steps for this lane of its quad
    uint32_t row_in_quad;      // lane & 3: which of the four rows r, r+8, r+16, r+24 the lane holds after the transpose
    uint32_t lb_lo, lb_hi;     // |B_j| of the quad's four rows of B, packed as bytes (lb_lo) or 16-bit pairs (lb_lo, lb_hi)
    uint32_t voff;             // without a stage: byte offset of the lane's store inside a row group (row_in_quad * 8 rows + its
                               // columns); 0xF0000000 (past every buffer size, which end below 0xC8000000) when the quad's
                               // rows of B do not exist
    unsigned char* stage;      // the block's stage in LDS
    uint32_t stage_col;        // byte offset of the quad's four columns inside a stage row
    uint32_t lane_col;         // byte offset of the lane's piece of a row on the way out (stage row and output row alike)
    uint32_t wave;             // the lane's wave in the block (uniform)
    uint32_t cols_bytes;       // bytes of a row of the output that this block's rows of B cover, whole quads (uniform)
};

template <int NCH, int MODE, int OUTB, bool STAGED>
__device__ __forceinline__ void tile_body(const u64* buf, const uint32_t (&off)[8 * NCH],
                                          uint32_t rows_in_tile, uint32_t la_vec, uint32_t lbterm,
                                          char* tile_out, uint64_t row_bytes, uint32_t nrec, uint32_t voff,
                                          const NarrowLane& narrow) {
    constexpr int NPL = bits_for(8 * NCH);
    constexpr int SH = MODE == 2 ? 1 : 0;
    const char* base = reinterpret_cast<const char*>(buf);
    uint32_t lo[NPL], hi[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) lo[p] = hi[p] = 0;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        uint32_t xl[8], xh[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u64 w = *reinterpret_cast<const u64*>(base + off[ch * 8 + k]);
            xl[k] = (uint32_t)w;
            xh[k] = (uint32_t)(w >> 32);
        }
        if (ch == 0) {
            add8<NPL, true>(lo, xl);
            add8<NPL, true>(hi, xh);
        } else {
            add8<NPL, false>(lo, xl);
            add8<NPL, false>(hi, xh);
        }
    }
    // unpack one 32-row half at a time: bits r, r+8, r+16, r+24 of plane p -> bit p+SH of
    // bytes 0..3 of acc[r] (planes_to_bytes); then store in row order of A, advancing the row pointer
    char* row_ptr = tile_out;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t acc[8];
        planes_to_bytes<NPL>(h ? hi : lo, acc);
        if constexpr (SH != 0) {   // counts doubled (they stay below 128, nothing leaves its byte)
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] <<= SH;
        }
        if constexpr (OUTB != 4 && !STAGED) {
            // (tables of 4096 words leave no room for a stage beside two resident blocks: straight from the registers,
            // four rows a store instruction)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // transpose inside the quad: afterwards this lane holds row (32 h + r + 8 * row_in_quad) of A for the
                // quad's four rows of B, one count per byte
                const uint32_t x = acc[r];
                const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);   // lane ^ 1
                const uint32_t t = __builtin_amdgcn_perm(y, x, narrow.sel_a);
                const uint32_t z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, false);   // lane ^ 2
                uint32_t u = __builtin_amdgcn_perm(z, t, narrow.sel_b);
                const uint32_t q = 32u * h + r + 8u * narrow.row_in_quad;
                uint32_t la = 0;
                if constexpr (MODE != 0) la = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q * 4u), (int)la_vec);
                // Stores are unconditional: a branch around them would make every later wait for a load (the next tile
                // of the table) a wait for all stores before it.  What must not be written is cut off by the buffer's
                // size instead -- the rows of A past the tile's last (a uniform count of the row groups 8 apart that
                // exist from row 32 h + r on) -- and by an offset past every size for the lanes whose rows of B do not exist.
                char* group = row_ptr + (uint64_t)r * row_bytes;
                const int rows_left = (int)rows_in_tile - (32 * h + r);
                const uint32_t groups = rows_left <= 0 ? 0u : rows_left > 24 ? 4u : (uint32_t)(rows_left + 7) / 8u;
#ifdef SDRB_LAB_NO_NARROW_STORES   // timing only: everything but the stores (ld_out is a multiple of 4: never 6 bytes a row)
                const uint32_t size = groups && row_bytes == 6u ? (groups - 1u) * 8u * (uint32_t)row_bytes + narrow.cols_bytes : 0u;
#else
                const uint32_t size = groups ? (groups - 1u) * 8u * (uint32_t)row_bytes + narrow.cols_bytes : 0u;
#endif
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(group, 0, (int)size, 0x00020000);
                if constexpr (OUTB == 1) {
                    if constexpr (MODE != 0) u = la * 0x01010101u + narrow.lb_lo - u;
                    __builtin_amdgcn_raw_buffer_store_b32(u, rsrc, (int)narrow.voff, 0, 0);
                } else {
                    // bytes 0, 1 and 2, 3 widened to 16-bit pairs: one byte permute each (selector 0x0c yields a zero byte)
                    uint32_t lo2 = __builtin_amdgcn_perm(0u, u, 0x0c010c00u), hi2 = __builtin_amdgcn_perm(0u, u, 0x0c030c02u);
                    if constexpr (MODE != 0) {
                        lo2 = la * 0x00010001u + narrow.lb_lo - lo2;
                        hi2 = la * 0x00010001u + narrow.lb_hi - hi2;
                    }
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const u32x2 both = {lo2, hi2};
                    __builtin_amdgcn_raw_buffer_store_b64(both, rsrc, (int)narrow.voff, 0, 0);
                }
            }
            row_ptr += 32 * row_bytes;
            continue;
        }
        if constexpr (OUTB != 4) {
            // The half tile goes through a stage in LDS -- 32 rows of A by the block's 256 columns -- and leaves it a
            // whole row piece per wave instruction (256 bytes of byte counts, 512 of 16-bit counts): as four pieces of a
            // quarter of that per instruction, straight from the registers, the same matrix streams a fifth slower
            // (pure stores: 4.8 against 5.9 TB/s for 16-bit counts).  Byte counts alternate between two stages, so
            // that the second half's writes need not wait for the first half's reads; 16-bit counts share one.
            constexpr uint32_t kStride = kStageStride<OUTB>;
            unsigned char* stage_h = narrow.stage + (kNarrowStages<OUTB> == 2 ? h * 32u * kStride : 0u);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                // transpose inside the quad: afterwards this lane holds row (32 h + r + 8 * row_in_quad) of A for the
                // quad's four rows of B, one count per byte
                const uint32_t x = acc[r];
                const uint32_t y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false);   // lane ^ 1
                const uint32_t t = __builtin_amdgcn_perm(y, x, narrow.sel_a);
                const uint32_t z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x4E, 0xf, 0xf, false);   // lane ^ 2
                uint32_t u = __builtin_amdgcn_perm(z, t, narrow.sel_b);
                const uint32_t q = 32u * h + r + 8u * narrow.row_in_quad;
                uint32_t la = 0;
                if constexpr (MODE != 0) la = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(q * 4u), (int)la_vec);
                unsigned char* slot = stage_h + (r + 8u * narrow.row_in_quad) * kStride + narrow.stage_col;
                if constexpr (OUTB == 1) {
                    if constexpr (MODE != 0) u = la * 0x01010101u + narrow.lb_lo - u;
                    *reinterpret_cast<uint32_t*>(slot) = u;
                } else {
                    // bytes 0, 1 and 2, 3 widened to 16-bit pairs: one byte permute each (selector 0x0c yields a zero byte)
                    uint32_t lo2 = __builtin_amdgcn_perm(0u, u, 0x0c010c00u), hi2 = __builtin_amdgcn_perm(0u, u, 0x0c030c02u);
                    if constexpr (MODE != 0) {
                        lo2 = la * 0x00010001u + narrow.lb_lo - lo2;
                        hi2 = la * 0x00010001u + narrow.lb_hi - hi2;
                    }
                    *reinterpret_cast<uint2*>(slot) = make_uint2(lo2, hi2);
                }
            }
            __syncthreads();
            // wave w carries rows w, w + 4, ... of the half out.  Stores are unconditional (a branch around them would
            // make every later wait for a load -- the next tile of the table -- a wait for all stores before it): rows
            // past the tile's last get an empty buffer, columns past the block's rows of B lie beyond the buffer's size.
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t row = narrow.wave + 4u * (uint32_t)k;     // uniform
                const uint32_t size = 32u * h + row < rows_in_tile ? narrow.cols_bytes : 0u;
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr + (uint64_t)row * row_bytes, 0, (int)size, 0x00020000);
                const unsigned char* from = stage_h + row * kStride + narrow.lane_col;
                if constexpr (OUTB == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const uint32_t*>(from), rsrc, (int)narrow.lane_col, 0, 0);
                } else {
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    const uint2 v = *reinterpret_cast<const uint2*>(from);
                    const u32x2 both = {v.x, v.y};
                    __builtin_amdgcn_raw_buffer_store_b64(both, rsrc, (int)narrow.lane_col, 0, 0);
                }
            }
            if constexpr (kNarrowStages<OUTB> == 1) __syncthreads();   // the other half writes the same stage
            row_ptr += 32 * row_bytes;
            continue;
        }
#pragma unroll
        for (int qq = 0; qq < 32; ++qq) {
            const int q = 32 * h + qq;
            uint32_t v = (acc[qq & 7] >> (8 * (qq >> 3))) & 0xffu;
            if constexpr (MODE != 0) {
                const uint32_t la = (uint32_t)__builtin_amdgcn_readlane((int)la_vec, q);
                v = (la + lbterm) - v;
            }
            const uint32_t n = (uint32_t)q < rows_in_tile ? nrec : 0u;
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)n, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)voff, 0, 0);
            row_ptr += row_bytes;
        }
    }
}

// NCH  : the lane keeps 8*NCH ids of its row in registers (shorter rows are padded with a
//        sentinel offset that points at an all-zero table word)
// MODE : 0 -> out = m;  1 -> out = L - m;  2 -> out = L - 2m   (m = id matches,
//        L = use_la*|A_i| + use_lb*|B_j|)
// WCAP : table words per tile; one tile fills one LDS buffer, two buffers alternate
// OUTB : bytes per count in `out` (4: uint32, the reference's size_t narrowed; 1 / 2: see tile_body)
template <int NCH, int MODE, int WCAP, int OUTB>
#ifndef SDRB_BS_NARROW_WAVES
#define SDRB_BS_NARROW_WAVES kMinWaves
#endif
__global__ __launch_bounds__(kBlock, WCAP <= 2048 ? (OUTB == 4 ? kMinWaves : SDRB_BS_NARROW_WAVES) : 2) void k_size_bitslice(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint64_t nb,
    const u64* __restrict__ table, const uint32_t* __restrict__ a_len, uint32_t ntiles,
    uint32_t tiles_per_block, uint32_t na, uint32_t use_la, uint32_t use_lb,
    void* __restrict__ out, uint64_t ld_out) {
    constexpr int NPL = bits_for(8 * NCH);
    static_assert(NPL + (MODE == 2 ? 1 : 0) <= 8, "counts must fit one byte during unpacking");
    constexpr int BUF = WCAP + 2;          // words per LDS buffer (zero sentinel at [WCAP])
    constexpr int PF = WCAP / 2 / kBlock;  // uint4 table loads per thread per tile
    __shared__ __attribute__((aligned(16))) u64 lds[2 * BUF];
    __shared__ __attribute__((aligned(16))) unsigned char stage_lds[OUTB == 4 || WCAP > 2048 ? 16 : kNarrowStages<OUTB> * 32 * kStageStride<OUTB>];

    const uint32_t tid = threadIdx.x;
    const uint64_t j0 = (uint64_t)blockIdx.x * kBlock;
    const uint64_t j = j0 + tid;

    // this lane's row of B -> LDS byte offsets in registers.  The ids are fetched through a
    // bounds-checked buffer that starts at the block's first element, so every load can be
    // issued unconditionally (slots past the row's end are replaced by the sentinel).
    uint32_t beg = 0, lb = 0;
    if (j < nb) {
        beg = b_off[j];
        lb = b_off[j + 1] - beg;
    }
    const uint32_t blk_beg = b_off[j0];                                   // uniform
    const uint64_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
    const uint32_t blk_end = b_off[blk_last];                             // uniform
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)((blk_end - blk_beg) * 4u), 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
    uint32_t off[8 * NCH];
    // four ids per load: a lane's row is contiguous, and with 16 bytes per lane the wave asks for
    // each cache line four times instead of sixteen (dwords past the buffer's end read as zero
    // one by one, so a row that ends at the end of the array is still read whole)
#pragma unroll
    for (int k4 = 0; k4 < 2 * NCH; ++k4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
        off[4 * k4 + 0] = v[0];
        off[4 * k4 + 1] = v[1];
        off[4 * k4 + 2] = v[2];
        off[4 * k4 + 3] = v[3];
    }
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k) off[k] = (uint32_t)k < lb ? off[k] * 8u : WCAP * 8u;
    const uint32_t lbterm = use_lb ? lb : 0u;

    const uint32_t t0 = blockIdx.y * tiles_per_block;
    const uint32_t t1 = (t0 + tiles_per_block < ntiles) ? t0 + tiles_per_block : ntiles;
    if (t0 >= t1) return;

    if (tid == 0) {
        lds[WCAP] = 0;
        lds[BUF + WCAP] = 0;
    }
    u32x4 pf[PF];
    load_tile(pf, table + (uint64_t)t0 * WCAP, tid);
    write_tile(pf, lds, tid);
    __syncthreads();

    const uint64_t row_bytes = ld_out * OUTB;
    const uint64_t rows_here = nb - j0 < (uint64_t)kBlock ? nb - j0 : (uint64_t)kBlock;
    const uint32_t nrec = (uint32_t)rows_here * 4;
    const uint32_t voff = tid * 4;
    char* const out_block = reinterpret_cast<char*>(out) + j0 * OUTB;
    NarrowLane narrow{};
    if constexpr (OUTB != 4) {
        narrow.sel_a = (tid & 1) ? 0x03070105u : 0x06020400u;
        narrow.sel_b = (tid & 2) ? 0x03020706u : 0x05040100u;
        narrow.row_in_quad = tid & 3;
        const uint32_t l0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lbterm, 0x00, 0xf, 0xf, false);
        const uint32_t l1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lbterm, 0x55, 0xf, 0xf, false);
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lbterm, 0xAA, 0xf, 0xf, false);
        const uint32_t l3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lbterm, 0xFF, 0xf, 0xf, false);
        if constexpr (OUTB == 1) {
            narrow.lb_lo = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
        } else {
            narrow.lb_lo = l0 | (l1 << 16);
            narrow.lb_hi = l2 | (l3 << 16);
        }
        narrow.voff = (uint64_t)(tid & ~3u) < rows_here ? (uint32_t)((tid & 3) * 8u * row_bytes) + (tid & ~3u) * OUTB : 0xF0000000u;
        narrow.stage = stage_lds;
        narrow.stage_col = (tid >> 2) * 4u * OUTB;
        narrow.lane_col = (tid & 63u) * 4u * OUTB;
        narrow.wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
        narrow.cols_bytes = (((uint32_t)rows_here + 3u) & ~3u) * OUTB;
    }

    for (uint32_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) load_tile(pf, table + (uint64_t)(t + 1) * WCAP, tid);
        uint32_t la_vec = 0;
        if constexpr (MODE != 0) {
            if (use_la) la_vec = a_len[(uint64_t)t * kTileRows + (tid & 63)];
        }
        char* tile_out = out_block + (uint64_t)t * kTileRows * row_bytes;
        const uint32_t rows_left = na - t * kTileRows;
        const uint32_t rows_in_tile = rows_left < (uint32_t)kTileRows ? rows_left : (uint32_t)kTileRows;
        // one straight-line body per LDS buffer so that LDS addresses fold into instruction offsets
        if (((t - t0) & 1) == 0) {
            tile_body<NCH, MODE, OUTB, (WCAP <= 2048)>(lds, off, rows_in_tile, la_vec, lbterm, tile_out, row_bytes, nrec, voff, narrow);
            if (more) write_tile(pf, lds + BUF, tid);
        } else {
            tile_body<NCH, MODE, OUTB, (WCAP <= 2048)>(lds + BUF, off, rows_in_tile, la_vec, lbterm, tile_out, row_bytes, nrec, voff, narrow);
            if (more) write_tile(pf, lds, tid);
        }
        __syncthreads();
    }
}

// Up to eight rows of A (the one-query-against-a-database case): the table word of an id holds one
// BYTE per row of A, 1 where the row contains the id, so a lane adds the 64-bit words of its ids
// as two 32-bit sums and each byte ends up as a match count (rows of B hold at most 255 ids).
// Every block builds the 8 * width bytes of table itself (A is a few hundred ids), so the whole
// operation is one launch; blocks stride over the 256-row pieces of B.
// MODE, use_la, use_lb as in k_size_bitslice.
constexpr int kFewRows = 8;
// NQ: a lane fetches its row as NQ loads of four ids, all issued before the first is used
template <int MODE, int NQ>
__global__ __launch_bounds__(kBlock) void k_size_few(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, uint32_t na,
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint64_t nb, uint32_t width,
    uint32_t use_la, uint32_t use_lb, uint32_t* __restrict__ out, uint64_t ld_out) {
    extern __shared__ __attribute__((aligned(16))) u64 few_tab[];   // [width + 1], entry [width] stays zero
    const uint32_t tid = threadIdx.x;
    for (uint32_t w = tid; w <= width; w += kBlock) few_tab[w] = 0;
    __syncthreads();
    for (uint32_t q = 0; q < na; ++q) {
        const uint32_t beg = a_off[q], len = a_off[q + 1] - beg;
        for (uint32_t k = tid; k < len; k += kBlock) {
            const uint32_t id = (uint32_t)a_ids[beg + k];
            if (id < width) atomicOr(&few_tab[id], 1ull << (8 * q));   // ids outside B's width never match
        }
    }
    __syncthreads();
    uint32_t la[kFewRows];
#pragma unroll
    for (int q = 0; q < kFewRows; ++q) la[q] = (MODE != 0 && use_la && (uint32_t)q < na) ? a_off[q + 1] - a_off[q] : 0u;
    const char* base = reinterpret_cast<const char*>(few_tab);
    const uint64_t pieces = (nb + kBlock - 1) / kBlock;
    for (uint64_t piece = blockIdx.x; piece < pieces; piece += gridDim.x) {
        const uint64_t j0 = piece * kBlock, j = j0 + tid;
        uint32_t beg = 0, lb = 0;
        if (j < nb) {
            beg = b_off[j];
            lb = b_off[j + 1] - beg;
        }
        const uint32_t blk_beg = b_off[j0];                                   // uniform
        const uint64_t blk_last = j0 + kBlock < nb ? j0 + kBlock : nb;
        const uint32_t blk_end = b_off[blk_last];                             // uniform
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)((blk_end - blk_beg) * 4u), 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
        u32x4 ids[NQ];
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4)
            ids[k4] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * k4), 0, 0);
        uint32_t sum_lo = 0, sum_hi = 0;
#pragma unroll
        for (int k4 = 0; k4 < NQ; ++k4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t byte_off = (uint32_t)(4 * k4 + c) < lb ? ids[k4][c] * 8u : width * 8u;
                const u64 w = *reinterpret_cast<const u64*>(base + byte_off);
                sum_lo += (uint32_t)w;
                sum_hi += (uint32_t)(w >> 32);
            }
        }
        const uint32_t lbterm = (MODE != 0 && use_lb) ? lb : 0u;
        uint32_t* dst = out + j;
#pragma unroll
        for (int q = 0; q < kFewRows; ++q) {
            uint32_t v = ((q < 4 ? sum_lo : sum_hi) >> (8 * (q & 3))) & 0xffu;
            if constexpr (MODE == 1) v = la[q] + lbterm - v;
            if constexpr (MODE == 2) v = la[q] + lbterm - 2 * v;
            if ((uint32_t)q < na && j < nb) __builtin_nontemporal_store(v, dst);
            dst += ld_out;
        }
    }
}

struct Plan {
    int nch;
    int wcap;
};

bool make_plan(const sdrb_batch* a, const sdrb_batch* b, Plan* plan) {
    (void)a;
    if (b->width <= 0 || b->width > 4096) return false;
    if (b->max_row_len > 64) return false;
    plan->nch = b->max_row_len <= 16 ? 2 : b->max_row_len <= 32 ? 4 : b->max_row_len <= 40 ? 5 : 8;
    plan->wcap = b->width <= 2048 ? 2048 : 4096;
    return true;
}

struct Args {
    const sdrb_batch* a;
    const sdrb_batch* b;
    const u64* table;
    const uint32_t* a_len;
    uint32_t ntiles, tiles_per_block, use_la, use_lb;
    dim3 grid;
    void* out;
    int out_bytes;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int NCH, int MODE, int WCAP, int OUTB>
void launch(const Args& g) {
    hipLaunchKernelGGL((k_size_bitslice<NCH, MODE, WCAP, OUTB>), g.grid, dim3(kBlock), 0, g.stream,
                       g.b->d_offsets, g.b->d_ids, g.b->n_rows, g.table, g.a_len, g.ntiles,
                       g.tiles_per_block, (uint32_t)g.a->n_rows, g.use_la, g.use_lb, g.out, g.ld_out);
}

template <int NCH, int MODE, int WCAP>
void launch_o(const Args& g) {
    if (g.out_bytes == 1) launch<NCH, MODE, WCAP, 1>(g);
    else if (g.out_bytes == 2) launch<NCH, MODE, WCAP, 2>(g);
    else launch<NCH, MODE, WCAP, 4>(g);
}

template <int NCH, int MODE>
void launch_w(const Args& g, int wcap) {
    if (wcap == 2048) launch_o<NCH, MODE, 2048>(g);
    else launch_o<NCH, MODE, 4096>(g);
}

template <int MODE>
void launch_n(const Args& g, const Plan& p) {
    switch (p.nch) {
    case 2: launch_w<2, MODE>(g, p.wcap); break;
    case 4: launch_w<4, MODE>(g, p.wcap); break;
    case 5: launch_w<5, MODE>(g, p.wcap); break;
    default: launch_w<8, MODE>(g, p.wcap); break;
    }
}

}  // namespace

bool bitslice_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    Plan p;
    return make_plan(a, b, &p);
}

namespace {
// ws_fixed: caller-provided scratch of ws_fixed_bytes (the static array); null: the growing workspace
int launch_size_bitslice_ws(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, void* out_any, int out_bytes,
                            uint64_t ld_out, hipStream_t stream, char* ws_fixed, size_t ws_fixed_bytes) {
    uint32_t* out = static_cast<uint32_t*>(out_any);
    Plan plan;
    if (!make_plan(a, b, &plan)) {
        set_error("bit-sliced size kernel not applicable (needs 0 < width <= 4096 and rows <= 64 ids on the streamed batch)");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;

    if (a->n_rows <= kFewRows && out_bytes == 4) {
        const uint32_t quads = (b->max_row_len + 3) / 4;   // <= 16: rows of B hold at most 64 ids here
        const uint64_t pieces = ((uint64_t)b->n_rows + kBlock - 1) / kBlock;
        const uint32_t grid = (uint32_t)(pieces < 2048 ? pieces : 2048);   // eight blocks per compute unit
        const size_t lds = ((size_t)b->width + 1) * sizeof(u64);
        const uint32_t use_la = op == SDRB_OP_AND ? 0u : op == SDRB_OP_RM ? (swap ? 0u : 1u) : 1u;
        const uint32_t use_lb = op == SDRB_OP_AND ? 0u : op == SDRB_OP_RM ? (swap ? 1u : 0u) : 1u;
#define SDRB_FEW_LAUNCH_N(M, N)                                                                                 \
    hipLaunchKernelGGL((k_size_few<M, N>), dim3(grid), dim3(kBlock), lds, stream, a->d_offsets, a->d_ids,       \
                       (uint32_t)a->n_rows, b->d_offsets, b->d_ids, (uint64_t)b->n_rows, (uint32_t)b->width,    \
                       use_la, use_lb, out, ld_out)
#define SDRB_FEW_LAUNCH(M)                          \
    do {                                            \
        if (quads <= 4) SDRB_FEW_LAUNCH_N(M, 4);    \
        else if (quads <= 10) SDRB_FEW_LAUNCH_N(M, 10); \
        else SDRB_FEW_LAUNCH_N(M, 16);              \
    } while (0)
        switch (op) {
        case SDRB_OP_AND: SDRB_FEW_LAUNCH(0); break;
        case SDRB_OP_OR:
        case SDRB_OP_RM: SDRB_FEW_LAUNCH(1); break;
        case SDRB_OP_XOR: SDRB_FEW_LAUNCH(2); break;
        default:
            set_error("unknown op %d", op);
            return SDRB_ERR_INVALID_ARG;
        }
#undef SDRB_FEW_LAUNCH_N
#undef SDRB_FEW_LAUNCH
        SDRB_HIP(hipGetLastError());
        return SDRB_OK;
    }

    Args g;
    g.a = a;
    g.b = b;
    g.ntiles = (uint32_t)((a->n_rows + kTileRows - 1) / kTileRows);
    const size_t table_bytes = (size_t)g.ntiles * plan.wcap * sizeof(u64);
    const size_t len_bytes = (size_t)g.ntiles * kTileRows * sizeof(uint32_t);
    char* ws = ws_fixed;
    if (ws_fixed) {
        if (table_bytes + len_bytes > ws_fixed_bytes) {
            set_error("sdrb_size_op_static: %llu rows of A need %zu bytes of table, the static scratch holds %zu "
                      "(at most %zu rows at this width)", (unsigned long long)a->n_rows, table_bytes + len_bytes,
                      ws_fixed_bytes, ws_fixed_bytes / ((size_t)plan.wcap * sizeof(u64) + kTileRows * sizeof(uint32_t)) * kTileRows);
            return SDRB_ERR_UNSUPPORTED;
        }
    } else {
        ws = static_cast<char*>(workspace(b->device, stream, 0, table_bytes + len_bytes));
        if (!ws) return SDRB_ERR_HIP;
    }
    SDRB_HIP(hipMemsetAsync(ws, 0, table_bytes + len_bytes, stream));
    u64* table = reinterpret_cast<u64*>(ws);
    uint32_t* a_len = reinterpret_cast<uint32_t*>(ws + table_bytes);
    hipLaunchKernelGGL(k_build_table, dim3((uint32_t)((a->n_rows + kBuildRows - 1) / kBuildRows)), dim3(64 * kBuildRows), 0,
                       stream, a->d_offsets, a->d_ids, (uint32_t)a->n_rows, b->width, (uint32_t)plan.wcap, table, a_len);
    SDRB_HIP(hipGetLastError());
    g.table = table;
    g.a_len = a_len;

    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 1024 ? 1u : (1024u + gx - 1) / gx;
    if (gy > g.ntiles) gy = g.ntiles;
    g.tiles_per_block = (g.ntiles + gy - 1) / gy;
    gy = (g.ntiles + g.tiles_per_block - 1) / g.tiles_per_block;
    g.grid = dim3(gx, gy);
    g.out = out_any;
    g.out_bytes = out_bytes;
    g.ld_out = ld_out;
    g.stream = stream;

    switch (op) {
    case SDRB_OP_AND:
        g.use_la = g.use_lb = 0;
        launch_n<0>(g, plan);
        break;
    case SDRB_OP_OR:
        g.use_la = g.use_lb = 1;
        launch_n<1>(g, plan);
        break;
    case SDRB_OP_XOR:
        g.use_la = g.use_lb = 1;
        launch_n<2>(g, plan);
        break;
    case SDRB_OP_RM:  // |this| - matches
        g.use_la = swap ? 0 : 1;
        g.use_lb = swap ? 1 : 0;
        launch_n<1>(g, plan);
        break;
    default:
        set_error("unknown op %d", op);
        return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}
}  // namespace

int launch_size_bitslice(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                         uint32_t* out, uint64_t ld_out, hipStream_t stream) {
    return launch_size_bitslice_ws(a, b, op, swap, out, 4, ld_out, stream, nullptr, 0);
}

// The largest value `op` can yield for any pair of these batches, from the longest rows alone.
uint64_t size_op_bound(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const uint64_t la = a->max_row_len, lb = b->max_row_len;
    if (op == SDRB_OP_AND) return la < lb ? la : lb;
    if (op == SDRB_OP_RM) return swap ? lb : la;
    return la + lb;
}

int launch_size_bitslice_narrow(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, void* out, int out_bytes,
                                uint64_t ld_out, hipStream_t stream) {
    return launch_size_bitslice_ws(a, b, op, swap, out, out_bytes, ld_out, stream, nullptr, 0);
}

// Device address of the static scratch on `device`, which must be the current device.  The first call on a
// device asks the runtime for it (a lookup in the loaded code object, no allocation); later calls read a table.
int static_scratch_resolve(int device, void** ptr) {
    if (device < 0 || device >= kMaxDevices) {
        set_error("device %d is outside the %d devices the static scratch is tracked for", device, kMaxDevices);
        return SDRB_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_ptr[device]) {
        void* p = nullptr;
        SDRB_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        g_static_ptr[device] = p;
    }
    *ptr = g_static_ptr[device];
    return SDRB_OK;
}

int launch_size_bitslice_static(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out,
                                uint64_t ld_out, hipStream_t stream) {
    void* ptr = nullptr;
    const int rc = static_scratch_resolve(b->device, &ptr);
    if (rc != SDRB_OK) return rc;
    return launch_size_bitslice_ws(a, b, op, swap, out, 4, ld_out, stream, static_cast<char*>(ptr), kStaticScratchBytes);
}

}  // namespace sdrb
