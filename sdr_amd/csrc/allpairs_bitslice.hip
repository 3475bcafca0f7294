// All-pairs overlap + per-row top-k with the bit-sliced table formulation (no matrix cores).
//
// Each overlap is the value `ands` returns for EmptyData elements (SDR.hpp:1083-1095 over andv,
// SDR.hpp:958-997); the reference has no top-k, the order -- larger overlap first, then the
// lower row of b -- is this engine's own rule.
//
// The candidates (rows j of b) are transposed into the bit table of size_bitslice.hip:
// table[t][w] bit q  <=>  row 64 t + q of b contains id w.  A lane owns one row i of a, keeps
// its ids in registers as LDS byte offsets, and per tile gathers the row's table words and adds
// them with carry-save adders into bit planes: plane p holds bit p of the 64 overlaps of row i
// with rows 64 t .. 64 t + 63.  Nothing is unpacked or stored: a bit-sliced comparator marks
// the columns whose count reaches what the lane's list still accepts, and only those (a few per
// tile per wave once the lists have warmed up) are extracted and inserted into the lane's
// register-resident sorted list of packed keys.  At 2 % density this does ~1/8 of the work of
// the dense int8 product and needs no byte matrix; the table streams from L2 / Infinity Cache.
#include "bitslice_device.h"

namespace sdrb {

namespace {

constexpr int LIST = kTopkList;

__device__ __forceinline__ void list_insert(int (&list)[LIST], int key) {
#pragma unroll
    for (int s = 0; s < LIST; ++s) {
        const int hi = list[s] > key ? list[s] : key;
        key = list[s] > key ? key : list[s];
        list[s] = hi;
    }
}

// columns (of 32) whose vertical count P is >= need; nb[p] = bit p of need spread over all columns
template <int NPL>
__device__ __forceinline__ uint32_t columns_ge(const uint32_t (&P)[NPL], const uint32_t (&nb)[NPL]) {
    uint32_t gt = 0, eq = ~0u;
#pragma unroll
    for (int p = NPL - 1; p >= 0; --p) {
        gt |= eq & P[p] & ~nb[p];
        eq &= ~(P[p] ^ nb[p]);
    }
    return gt | eq;
}

// One tile: count the overlaps of this lane's row with the 64 candidate rows of the tile and
// feed the ones that can still enter the lane's list.
template <int NCH>
__device__ __forceinline__ void tile_topk(const u64* buf, const uint32_t (&off)[8 * NCH], int (&list)[LIST],
                                          uint32_t valid_lo, uint32_t valid_hi, int low_tile, int skip_q) {
    constexpr int NPL = bits_for(8 * NCH);
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
    // Candidates reach the lane in ascending j, so on equal counts the earlier one stays: a new one
    // must beat the count of the list's last entry (any count while the list is not full).
    const int last = list[LIST - 1];
    const int need = last == 0 ? 0 : (last >> kTopkKeyShift) + 1;
    uint32_t nb[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) nb[p] = (uint32_t)(-((need >> p) & 1));
    uint32_t ge_lo = columns_ge<NPL>(lo, nb) & valid_lo;
    uint32_t ge_hi = columns_ge<NPL>(hi, nb) & valid_hi;
    if (need >> NPL) ge_lo = ge_hi = 0;   // the list already holds counts no column can exceed
    // one insertion round per marked column of the busiest lane
    while (__builtin_amdgcn_ballot_w64((ge_lo | ge_hi) != 0) != 0) {
        int key = -1;
        if ((ge_lo | ge_hi) != 0) {
            const bool low_half = ge_lo != 0;
            const int q = low_half ? __builtin_ctz(ge_lo) : __builtin_ctz(ge_hi);
            if (low_half) ge_lo &= ge_lo - 1;
            else ge_hi &= ge_hi - 1;
            int count = 0;
#pragma unroll
            for (int p = 0; p < NPL; ++p) count |= (int)(((low_half ? lo[p] : hi[p]) >> q) & 1u) << p;
            const int col = low_half ? q : q + 32;
            if (col != skip_q) key = (count << kTopkKeyShift) + (low_tile - col);
        }
        list_insert(list, key);
    }
}

// NCH  : the lane keeps 8*NCH ids of its row in registers (shorter rows are padded with a
//        sentinel offset that points at an all-zero table word)
// WCAP : table words per tile; one tile fills one LDS buffer, two buffers alternate
template <int NCH, int WCAP>
__global__ __launch_bounds__(kBlock, WCAP <= 2048 ? kMinWaves : 2) void k_allpairs_bitslice(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, uint32_t na,
    const u64* __restrict__ table, uint32_t ntiles, uint32_t tiles_per_split, uint32_t nb, long long excl,
    int* __restrict__ partial) {
    constexpr int BUF = WCAP + 2;          // words per LDS buffer (zero sentinel at [WCAP])
    constexpr int PF = WCAP / 2 / kBlock;  // 16-byte table loads per thread per tile
    __shared__ __attribute__((aligned(16))) u64 lds[2 * BUF];

    const uint32_t tid = threadIdx.x;
    const uint32_t i0 = blockIdx.x * kBlock;
    const uint32_t i = i0 + tid;

    // this lane's row of a -> LDS byte offsets in registers, through a bounds-checked buffer that
    // starts at the block's first element so every load can be issued unconditionally
    uint32_t beg = 0, la = 0;
    if (i < na) {
        beg = a_off[i];
        la = a_off[i + 1] - beg;
    }
    const uint32_t blk_beg = a_off[i0];                                   // uniform
    const uint32_t blk_last = i0 + kBlock < na ? i0 + kBlock : na;
    const uint32_t blk_end = a_off[blk_last];                             // uniform
    __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<int32_t*>(a_ids) + blk_beg, 0, (int)((blk_end - blk_beg) * 4u), 0x00020000);
    const uint32_t rel = (beg - blk_beg) * 4u;
    uint32_t off[8 * NCH];
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k)
        off[k] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ids_rsrc, (int)(rel + 4u * k), 0, 0);
#pragma unroll
    for (int k = 0; k < 8 * NCH; ++k) off[k] = (uint32_t)k < la ? off[k] * 8u : WCAP * 8u;

    int list[LIST];
#pragma unroll
    for (int s = 0; s < LIST; ++s) list[s] = 0;

    const uint32_t split = blockIdx.y;
    const uint32_t t0 = split * tiles_per_split;
    const uint32_t t1 = (t0 + tiles_per_split < ntiles) ? t0 + tiles_per_split : ntiles;
    if (t0 < t1) {
        if (tid == 0) {
            lds[WCAP] = 0;
            lds[BUF + WCAP] = 0;
        }
        u32x4 pf[PF];
        load_tile(pf, table + (uint64_t)t0 * WCAP, tid);
        write_tile(pf, lds, tid);
        __syncthreads();

        // column of the excluded pair (i, i + excl) relative to the current tile, when it has one
        const long long skip_j = excl >= 0 ? (long long)i + excl : -1;
        for (uint32_t t = t0; t < t1; ++t) {
            const bool more = t + 1 < t1;
            if (more) load_tile(pf, table + (uint64_t)(t + 1) * WCAP, tid);
            // candidates past the last row of b (partial last tile) never qualify
            const uint32_t rows_left = nb - t * kTileRows;
            const uint32_t valid_lo = rows_left >= 32 ? ~0u : (1u << rows_left) - 1u;
            const uint32_t valid_hi = rows_left >= 64 ? ~0u : rows_left <= 32 ? 0u : (1u << (rows_left - 32)) - 1u;
            const int low_tile = kTopkLowMask - (int)((t - t0) * kTileRows);
            const long long d = skip_j - (long long)t * kTileRows;
            const int skip_q = (skip_j >= 0 && d >= 0 && d < kTileRows) ? (int)d : -1;
            // one straight-line body per LDS buffer so that LDS addresses fold into instruction offsets
            if (((t - t0) & 1) == 0) {
                tile_topk<NCH>(lds, off, list, valid_lo, valid_hi, low_tile, skip_q);
                if (more) write_tile(pf, lds + BUF, tid);
            } else {
                tile_topk<NCH>(lds + BUF, off, list, valid_lo, valid_hi, low_tile, skip_q);
                if (more) write_tile(pf, lds, tid);
            }
            __syncthreads();
        }
    }
    if (i < na) {
        int* dst = partial + ((uint64_t)i * gridDim.y + split) * LIST;
#pragma unroll
        for (int s = 0; s < LIST; ++s) dst[s] = list[s];
    }
}

struct Args {
    const sdrb_batch* a;
    const u64* table;
    uint32_t ntiles, tiles_per_split, nb;
    long long excl;
    int* partial;
    dim3 grid;
    hipStream_t stream;
};

template <int NCH, int WCAP>
void launch(const Args& g) {
    hipLaunchKernelGGL((k_allpairs_bitslice<NCH, WCAP>), g.grid, dim3(kBlock), 0, g.stream, g.a->d_offsets,
                       g.a->d_ids, (uint32_t)g.a->n_rows, g.table, g.ntiles, g.tiles_per_split, g.nb, g.excl,
                       g.partial);
}

template <int NCH>
void launch_w(const Args& g, int wcap) {
    if (wcap == 2048) launch<NCH, 2048>(g);
    else launch<NCH, 4096>(g);
}

int table_width(const sdrb_batch* a, const sdrb_batch* b) { return a->width > b->width ? a->width : b->width; }

}  // namespace

bool allpairs_bitslice_applicable(const sdrb_batch* a, const sdrb_batch* b, uint32_t k) {
    if (a->width <= 0 || b->width <= 0 || table_width(a, b) > 4096) return false;
    if (a->max_row_len > 64) return false;        // the lane keeps its whole row in registers
    if (k == 0 || k > (uint32_t)LIST) return false;
    if (a->n_rows > 0x7FFFFF00ull) return false;
    // the merge holds 32 splits of 16 keys per row, and a split's local row index must fit the key
    if (b->n_rows > 32ull * (kTopkMaxSplitRows / kTileRows) * kTileRows) return false;   // 16 775 168 candidate rows
    return true;
}

int launch_allpairs_bitslice(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                             uint32_t* out_counts, int32_t* out_rows, hipStream_t stream) {
    const int wcap = table_width(a, b) <= 2048 ? 2048 : 4096;
    const int nch = a->max_row_len <= 16 ? 2 : a->max_row_len <= 32 ? 4 : a->max_row_len <= 40 ? 5 : 8;

    Args g;
    g.a = a;
    g.nb = (uint32_t)b->n_rows;
    g.excl = exclude_offset;
    g.stream = stream;
    g.ntiles = (uint32_t)((b->n_rows + kTileRows - 1) / kTileRows);
    // the transposed bit table of the candidates (same table and scratch slot as the size kernel)
    const size_t table_bytes = (size_t)g.ntiles * wcap * sizeof(u64);
    const size_t len_bytes = (size_t)g.ntiles * kTileRows * sizeof(uint32_t);
    char* ws = static_cast<char*>(workspace(b->device, stream, 0, table_bytes + len_bytes));
    if (!ws) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(ws, 0, table_bytes + len_bytes, stream));
    if (b->n_rows) {
        hipLaunchKernelGGL(k_build_table, dim3((uint32_t)((b->n_rows + kBuildRows - 1) / kBuildRows)), dim3(64 * kBuildRows), 0,
                           stream, b->d_offsets, b->d_ids, (uint32_t)b->n_rows, (int32_t)wcap, (uint32_t)wcap,
                           reinterpret_cast<u64*>(ws),
                           reinterpret_cast<uint32_t*>(ws + table_bytes));
        SDRB_HIP(hipGetLastError());
    }
    g.table = reinterpret_cast<const u64*>(ws);

    // splits of the candidate tiles: enough blocks to fill the chip three deep, each split within
    // the key's row range; every split keeps its own lists, merged afterwards
    const uint32_t gx = (uint32_t)((a->n_rows + kBlock - 1) / kBlock);
    uint32_t splits = gx >= 768 ? 1u : (768u + gx - 1) / gx;
    if (splits > 32u) splits = 32u;              // the merge kernel holds 32 lists of 16 keys per row
    if (splits > g.ntiles) splits = g.ntiles ? g.ntiles : 1u;
    const uint32_t max_tiles = kTopkMaxSplitRows / kTileRows;
    if ((g.ntiles + splits - 1) / splits > max_tiles) splits = (g.ntiles + max_tiles - 1) / max_tiles;
    g.tiles_per_split = (g.ntiles + splits - 1) / splits;
    if (g.tiles_per_split == 0) g.tiles_per_split = 1;
    splits = g.ntiles ? (g.ntiles + g.tiles_per_split - 1) / g.tiles_per_split : 1u;
    g.grid = dim3(gx, splits);

    int* partial = static_cast<int*>(workspace(b->device, stream, 4, (size_t)a->n_rows * splits * LIST * sizeof(int)));
    if (!partial) return SDRB_ERR_HIP;
    g.partial = partial;
    switch (nch) {
    case 2: launch_w<2>(g, wcap); break;
    case 4: launch_w<4>(g, wcap); break;
    case 5: launch_w<5>(g, wcap); break;
    default: launch_w<8>(g, wcap); break;
    }
    SDRB_HIP(hipGetLastError());
    return launch_merge_topk(partial, (uint32_t)a->n_rows, splits, 1, (uint64_t)g.tiles_per_split * kTileRows, k,
                             out_counts, out_rows, stream);
}

}  // namespace sdrb
