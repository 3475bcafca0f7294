// All-pairs overlap as a dense int8 0/1 matrix product on the matrix cores, with the per-row
// top-k selection fused into the epilogue so the na x nb count matrix is never written.
//
// Entry (i, j) of the product of the two 0/1 row matrices is |a_i ∩ b_j|, the value the
// reference computes per pair with `ands` on EmptyData elements (SDR.hpp:1083-1095 over
// andv, SDR.hpp:958-997).  The reference has no top-k; the order used here -- larger overlap
// first, then the lower row of b -- is this engine's own rule.
//
// Layout.  A batch is expanded once (cached in its handle) into bytes in MFMA fragment order:
// rows are grouped in panels of 256, ids in chunks of 64, and the 16 KiB block of
// (panel P, chunk c) holds [g = row/32 % 8][s = id/32 % 2][h = id/16 % 2][r = row % 32][id % 16].
// A block copies such a 16 KiB piece linearly into LDS, and lane l = 32h + r of a wave reads
// the operand fragment of (g, s) -- 16 consecutive K bytes of row r -- with one conflict-free
// ds_read_b128 at [(g*2 + s)*64 + l].
//
// Kernel.  v_mfma_i32_32x32x32_i8 computes D[row][col] += sum_k A[row][k] * B[k][col]; the rows
// of b (candidates j) are the A operand and the rows of a (the rows i whose neighbours are
// wanted) the B operand, so a lane owns ONE i (col = lane & 31) and its 16 accumulator registers
// of a tile are 16 different j: the top-k lists live in registers, one sorted list of packed
// keys (count << 19 | reversed local j) per i-tile per lane, and need no cross-lane traffic.
// A block (4 waves as 2 x 2, wave tile 128 j x 128 i) keeps its panel of 256 rows of a and
// walks a contiguous range of j panels; at the end the 4 partial lists per i (2 waves x 2
// lane halves) go to HBM and k_merge_topk merges them over the j-splits.
#include "sdrb_internal.h"

namespace sdrb {

namespace {

using i32x4 = int __attribute__((ext_vector_type(4)));
using i32x16 = int __attribute__((ext_vector_type(16)));

constexpr int kPanel = 256;          // rows per panel
constexpr int kChunk = 64;           // ids per chunk
constexpr int kPieceVec = 1024;      // 16-byte vectors per (panel, chunk) piece
constexpr int kKeyShift = 19;        // key = count << 19 | (2^19 - 1 - local j)
constexpr int kLowMask = (1 << kKeyShift) - 1;
constexpr uint32_t kMaxSplitRows = (uint32_t)kLowMask - 1;   // local j < 2^19 - 1 keeps keys > 0
constexpr int kSubLists = 4;         // partial lists per (row i, j-split): 2 waves x 2 lane halves

__host__ __device__ inline uint64_t dense_offset(uint64_t row, uint32_t id, uint32_t n_chunks) {
    const uint64_t panel = row / kPanel;
    const uint32_t g = (uint32_t)(row % kPanel) / 32, r = (uint32_t)row % 32;
    const uint32_t c = id / kChunk, s = (id % kChunk) / 32, h = (id % 32) / 16, b = id % 16;
    return ((panel * n_chunks + c) * kPieceVec + ((g * 2 + s) * 2 + h) * 32 + r) * 16 + b;
}

// one wave per row: a byte 1 at every id of the row
__global__ __launch_bounds__(256) void k_densify(const uint32_t* __restrict__ off, const int32_t* __restrict__ ids,
                                                 uint32_t n_rows, uint32_t n_chunks, uint8_t* __restrict__ dense) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const uint32_t beg = off[row], len = off[row + 1] - beg;
    for (uint32_t k = lane; k < len; k += 64) dense[dense_offset(row, (uint32_t)ids[beg + k], n_chunks)] = 1;
}

// Sorted (descending) insertion into one lane's list, kept in LDS as list[slot * 256]: a lane
// touches only its own column, so no synchronisation is involved.  Returns the new last entry.
template <int LIST>
__device__ __forceinline__ int list_insert(int* list, int key) {
    int cur[LIST];
#pragma unroll
    for (int s = 0; s < LIST; ++s) cur[s] = list[s * 256];
#pragma unroll
    for (int s = 0; s < LIST; ++s) {
        const int hi = cur[s] > key ? cur[s] : key;
        key = cur[s] > key ? key : cur[s];
        cur[s] = hi;
    }
#pragma unroll
    for (int s = 0; s < LIST; ++s) list[s * 256] = cur[s];
    return cur[LIST - 1];
}

// Accumulator register `reg` of lane half h holds row (reg & 3) + 8 * (reg >> 2) + 4 * h of the 32 x 32 tile.

template <int LIST>
__global__ __launch_bounds__(256, 1) void k_allpairs_mfma(
    const uint4* __restrict__ dense_i, const uint4* __restrict__ dense_j, uint32_t n_ipanels, uint32_t n_jpanels,
    uint32_t n_chunks, uint32_t splits, uint32_t panels_per_split, uint32_t nb, long long excl,
    int* __restrict__ partial) {
    __shared__ uint4 lds[2][2][kPieceVec];   // [buffer][0 = j operand, 1 = i operand]
    __shared__ int lists[4][LIST][256];      // [i tile of the wave][slot][thread]: sorted keys, 0 = empty

    // block -> (i panel, j split): the 8 XCDs take blocks round-robin, so consecutive slots of one
    // XCD get the 8 splits of one i panel (they share that panel's chunks in the XCD's L2) and
    // different XCDs get different i panels.
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t split = slot % splits;
    const uint32_t ip = (slot / splits) * 8 + xcd;
    if (ip >= n_ipanels) return;
    const uint32_t jp_begin = split * panels_per_split;
    const uint32_t jp_end = jp_begin + panels_per_split < n_jpanels ? jp_begin + panels_per_split : n_jpanels;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1;     // wave tile: j rows [128 wm, +128), i rows [128 wn, +128)
    const uint32_t half = lane >> 5, col = lane & 31;

    int thr[4];   // last (smallest) entry of each list
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        thr[nt] = 0;
#pragma unroll
        for (int s = 0; s < LIST; ++s) lists[nt][s][tid] = 0;
    }

    const uint4* gi = dense_i + (uint64_t)ip * n_chunks * kPieceVec;
    const long long i_lo = (long long)ip * kPanel + wn * 128;   // first row i of this wave

    for (uint32_t jp = jp_begin; jp < jp_end; ++jp) {
        const uint4* gj = dense_j + (uint64_t)jp * n_chunks * kPieceVec;
        i32x16 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0;

        uint4 stage[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            stage[m] = gj[tid + 256 * m];
            stage[4 + m] = gi[tid + 256 * m];
        }
        __syncthreads();   // every wave is done reading both buffers of the previous panel
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            lds[0][0][tid + 256 * m] = stage[m];
            lds[0][1][tid + 256 * m] = stage[4 + m];
        }
        __syncthreads();

        for (uint32_t c = 0; c < n_chunks; ++c) {
            const uint32_t buf = c & 1;
            const bool more = c + 1 < n_chunks;
            if (more) {
                const uint4* nj = gj + (uint64_t)(c + 1) * kPieceVec;
                const uint4* ni = gi + (uint64_t)(c + 1) * kPieceVec;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    stage[m] = nj[tid + 256 * m];
                    stage[4 + m] = ni[tid + 256 * m];
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                i32x4 fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint4 va = lds[buf][0][((wm * 4 + t) * 2 + s) * 64 + lane];
                    const uint4 vb = lds[buf][1][((wn * 4 + t) * 2 + s) * 64 + lane];
                    fa[t] = i32x4{(int)va.x, (int)va.y, (int)va.z, (int)va.w};
                    fb[t] = i32x4{(int)vb.x, (int)vb.y, (int)vb.z, (int)vb.w};
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
            }
            if (more) {
                // buffer buf^1 was last read in iteration c-1; every wave has passed the barrier since
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    lds[buf ^ 1][0][tid + 256 * m] = stage[m];
                    lds[buf ^ 1][1][tid + 256 * m] = stage[4 + m];
                }
            }
            __syncthreads();
        }

        // ---- epilogue: fold this 128 x 128 wave tile into the per-lane top-k lists
        const uint32_t j_local0 = (jp - jp_begin) * kPanel + wm * 128 + 4 * half;   // local j of (mt 0, reg 0)
        const long long j_lo = (long long)jp * kPanel + wm * 128;                    // first global j of the wave tile
        const bool tail = j_lo + 127 >= (long long)nb;
        const bool diag = excl >= 0 && j_lo <= i_lo + 127 + excl && i_lo + excl <= j_lo + 127;
        // Padding rows of the last panel and the excluded pair are never candidates.  With r = row of
        // the wave tile that lane half 0 holds in a slot, both tests are against lane constants.
        const int pad_from = tail ? (int)((long long)nb - j_lo) - 4 * (int)half : (1 << 30);
        const int skip_base = diag ? (int)(i_lo + excl - j_lo) + (int)col - 4 * (int)half : -(1 << 30);
        const int low_base = kLowMask - (int)j_local0;
        // The operand buffers are idle until the barrier at the top of the next panel: each wave
        // parks one i tile's 64 counts per lane (4 j tiles x 16 registers) in its own quarter of them,
        // straight from the accumulators, and walks them in a rolled loop.  The counts therefore
        // never need VALU-visible registers while the matrix pipeline owns the register file.
        i32x4* park = reinterpret_cast<i32x4*>(&lds[0][0][0]) + wave * 1024 + lane;   // [slot/4][lane]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            __builtin_amdgcn_sched_barrier(0);   // keep each i tile's accumulator reads next to their stores
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    park[(mt * 4 + q) * 64] = i32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2],
                                                    acc[mt][nt][4 * q + 3]};
            __builtin_amdgcn_sched_barrier(0);
            const int skip_at = skip_base + nt * 32;
            int m = -1;
            if (tail || diag) {
#pragma nounroll
                for (int g = 0; g < 16; ++g) {   // slot group g: rows (g/4)*32 + 8*(g%4) + 0..3
                    i32x4 v = park[g * 64];
                    const int r0 = (g >> 2) * 32 + (g & 3) * 8;
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        if (r0 + x >= pad_from || r0 + x == skip_at) v[x] = -1;
                    park[g * 64] = v;
#pragma unroll
                    for (int x = 0; x < 4; ++x) m = v[x] > m ? v[x] : m;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const i32x4 v = park[g * 64];
#pragma unroll
                    for (int x = 0; x < 4; ++x) m = v[x] > m ? v[x] : m;
                }
            }
            // candidates reach a lane in ascending j, so on equal counts the earlier one stays
            const bool trigger = m >= 0 && (thr[nt] == 0 || m > (thr[nt] >> kKeyShift));
            if (__builtin_amdgcn_ballot_w64(trigger) == 0) continue;
#pragma nounroll
            for (int g = 0; g < 16; ++g) {
                const i32x4 v = park[g * 64];
                const int low = low_base - ((g >> 2) * 32 + (g & 3) * 8);
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const int key = v[x] < 0 ? -1 : (v[x] << kKeyShift) + (low - x);
                    if (__builtin_amdgcn_ballot_w64(key > thr[nt]) != 0) {
                        if (key > thr[nt]) thr[nt] = list_insert<LIST>(&lists[nt][0][tid], key);
                    }
                }
            }
        }
    }

    // partial lists: [row i][split][wm*2 + half][LIST]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const uint64_t i = (uint64_t)ip * kPanel + wn * 128 + nt * 32 + col;
        int* dst = partial + ((i * splits + split) * kSubLists + (wm * 2 + half)) * LIST;
#pragma unroll
        for (int s = 0; s < LIST; ++s) dst[s] = lists[nt][s][tid];
    }
}

// one thread per row i: k rounds of "largest key below the previous one" over all partial lists
template <int LIST>
__global__ __launch_bounds__(256) void k_merge_topk(const int* __restrict__ partial, uint32_t na, uint32_t splits,
                                                    uint32_t panels_per_split, uint32_t k,
                                                    uint32_t* __restrict__ out_counts, int32_t* __restrict__ out_rows) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= na) return;
    const int* mine = partial + (uint64_t)i * splits * kSubLists * LIST;
    unsigned long long prev = ~0ull;
    for (uint32_t t = 0; t < k; ++t) {
        unsigned long long best = 0;
        for (uint32_t sp = 0; sp < splits; ++sp) {
            const unsigned long long j_base = (unsigned long long)sp * panels_per_split * kPanel;
            for (int e = 0; e < kSubLists * LIST; ++e) {
                const int key = mine[sp * kSubLists * LIST + e];
                if (key <= 0) continue;
                const unsigned long long count = (unsigned long long)(key >> kKeyShift);
                const unsigned long long j = j_base + (unsigned long long)(kLowMask - (key & kLowMask));
                const unsigned long long wide = (count << 32) | (0xFFFFFFFFull - j);
                if (wide < prev && wide > best) best = wide;
            }
        }
        if (best == 0) {
            out_counts[(uint64_t)i * k + t] = 0;
            out_rows[(uint64_t)i * k + t] = -1;
        } else {
            out_counts[(uint64_t)i * k + t] = (uint32_t)(best >> 32);
            out_rows[(uint64_t)i * k + t] = (int32_t)(0xFFFFFFFFull - (best & 0xFFFFFFFFull));
            prev = best;
        }
    }
}

uint32_t chunks_of(const sdrb_batch* b) { return (uint32_t)((b->width + kChunk - 1) / kChunk); }

// builds (or reuses) the fragment-ordered byte matrix of a batch, `n_chunks` chunks wide
int ensure_dense(const sdrb_batch* b, uint32_t n_chunks, hipStream_t stream) {
    sdrb_batch* mb = const_cast<sdrb_batch*>(b);
    if (mb->d_dense8 && mb->dense_chunks == n_chunks) return SDRB_OK;
    if (mb->d_dense8) {
        SDRB_HIP(hipStreamSynchronize(stream));
        (void)hipFree(mb->d_dense8);
        mb->d_dense8 = nullptr;
    }
    const uint64_t panels = (b->n_rows + kPanel - 1) / kPanel;
    const size_t bytes = (size_t)panels * n_chunks * kPieceVec * 16;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&mb->d_dense8), bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) for the dense byte matrix failed: %s", bytes, hipGetErrorString(e));
        mb->d_dense8 = nullptr;
        return SDRB_ERR_HIP;
    }
    mb->dense_chunks = n_chunks;
    SDRB_HIP(hipMemsetAsync(mb->d_dense8, 0, bytes, stream));
    hipLaunchKernelGGL(k_densify, dim3((uint32_t)((b->n_rows + 3) / 4)), dim3(256), 0, stream, b->d_offsets,
                       b->d_ids, (uint32_t)b->n_rows, n_chunks, mb->d_dense8);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <int LIST>
int run(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t excl, uint32_t* out_counts,
        int32_t* out_rows, hipStream_t stream) {
    const uint32_t n_chunks = chunks_of(a) > chunks_of(b) ? chunks_of(a) : chunks_of(b);
    int rc = ensure_dense(a, n_chunks, stream);
    if (rc != SDRB_OK) return rc;
    rc = ensure_dense(b, n_chunks, stream);
    if (rc != SDRB_OK) return rc;
    const uint32_t n_ipanels = (uint32_t)((a->n_rows + kPanel - 1) / kPanel);
    const uint32_t n_jpanels = (uint32_t)((b->n_rows + kPanel - 1) / kPanel);
    // j-splits: enough blocks to fill 256 CUs, at most 8 (the L2 sharing pattern), each within the key range
    uint32_t splits = 1;
    while (splits < 8 && (uint64_t)n_ipanels * splits < 512 && splits * 2 <= n_jpanels) splits *= 2;
    uint32_t pps = (n_jpanels + splits - 1) / splits;
    while ((uint64_t)pps * kPanel > kMaxSplitRows) {
        splits *= 2;
        pps = (n_jpanels + splits - 1) / splits;
    }
    const uint32_t ipanels_pad = (n_ipanels + 7) / 8 * 8;
    const size_t pbytes = (size_t)n_ipanels * kPanel * splits * kSubLists * LIST * sizeof(int);
    int* partial = static_cast<int*>(workspace(b->device, 4, pbytes));
    if (!partial) return SDRB_ERR_HIP;
    hipLaunchKernelGGL(k_allpairs_mfma<LIST>, dim3(ipanels_pad * splits), dim3(256), 0, stream,
                       reinterpret_cast<const uint4*>(a->d_dense8), reinterpret_cast<const uint4*>(b->d_dense8),
                       n_ipanels, n_jpanels, n_chunks, splits, pps, (uint32_t)b->n_rows, (long long)excl, partial);
    SDRB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_merge_topk<LIST>, dim3((uint32_t)((a->n_rows + 255) / 256)), dim3(256), 0, stream, partial,
                       (uint32_t)a->n_rows, splits, pps, k, out_counts, out_rows);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace

bool allpairs_mfma_applicable(const sdrb_batch* a, const sdrb_batch* b, uint32_t k) {
    if (a->width <= 0 || b->width <= 0 || a->width > 16384 || b->width > 16384) return false;
    if (k == 0 || k > 16) return false;   // list length of the kernel
    // counts must fit the key: an overlap never exceeds the shorter row
    const uint32_t cap = a->max_row_len < b->max_row_len ? a->max_row_len : b->max_row_len;
    if (cap >= (1u << (31 - kKeyShift))) return false;
    if (a->n_rows > 0x7FFFFF00ull || b->n_rows > 0x7FFFFF00ull) return false;
    return true;
}

int launch_allpairs_mfma(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                         uint32_t* out_counts, int32_t* out_rows, hipStream_t stream) {
    return run<16>(a, b, k, exclude_offset, out_counts, out_rows, stream);
}

}  // namespace sdrb
