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
// native vectors (not the HIP uint4 struct) so staging arrays stay in registers

constexpr int kPanel = 256;          // rows per panel
constexpr int kChunk = 64;           // ids per chunk
constexpr int kPieceVec = 1024;      // 16-byte vectors per (panel, chunk) piece
constexpr int kStageVec = 2048;      // vectors per operand per pipeline stage: two chunks of one panel
constexpr int kKeyShift = kTopkKeyShift;
constexpr int kLowMask = kTopkLowMask;
constexpr uint32_t kMaxSplitRows = kTopkMaxSplitRows;
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

// Sorted (descending) insertion into one lane's register-resident list.  A key that does not beat
// the last entry falls through unchanged, so the call is harmless for lanes without a candidate.
template <int LIST>
__device__ __forceinline__ void list_insert(int (&list)[LIST], int key) {
#pragma unroll
    for (int s = 0; s < LIST; ++s) {
        const int hi = list[s] > key ? list[s] : key;
        key = list[s] > key ? key : list[s];
        list[s] = hi;
    }
}

// Accumulator register `reg` of lane half h holds row (reg & 3) + 8 * (reg >> 2) + 4 * h of the 32 x 32 tile.

// LDS-DMA of one stage (two consecutive 16 KiB pieces of a panel, 2048 vectors) for both operands:
// lane l of wave w moves vector 64 w + 256 m + l, the LDS side being wave-uniform base + 16 l.
__device__ __forceinline__ void stage_dma(const i32x4* __restrict__ src_j, const i32x4* __restrict__ src_i,
                                          i32x4* dst_j, i32x4* dst_i, uint32_t tid, uint32_t wave) {
    using gptr = const __attribute__((address_space(1))) void*;
    using lptr = __attribute__((address_space(3))) void*;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        __builtin_amdgcn_global_load_lds((gptr)(src_j + tid + 256 * m), (lptr)(dst_j + wave * 64 + 256 * m), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr)(src_i + tid + 256 * m), (lptr)(dst_i + wave * 64 + 256 * m), 16, 0, 0);
    }
}


template <int LIST>
__global__ __launch_bounds__(256, 1) void k_allpairs_mfma(
    const i32x4* __restrict__ dense_i, const i32x4* __restrict__ dense_j, uint32_t n_ipanels, uint32_t n_jpanels,
    uint32_t n_chunks, uint32_t splits, uint32_t panels_per_split, uint32_t nb, long long excl,
    int* __restrict__ partial) {
    __shared__ i32x4 lds[2][2][kStageVec];   // [buffer][0 = j operand, 1 = i operand], one stage = 128 ids deep
    // per row i of the panel: a count that 16 candidates of ONE of its lists already reach, so a
    // candidate with a smaller count cannot be among the block's best 16 (equal counts still can)
    __shared__ int shared_level[kPanel];

    // block -> (i panel, j split): the 8 XCDs take blocks round-robin; one XCD works on panels
    // ip = xcd, xcd + 8, ... and all their splits, so the blocks of an XCD stream the same j panels
    // through its L2 at about the same time.
    const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const uint32_t split = slot % splits;
    const uint32_t ip = (slot / splits) * 8 + xcd;
    if (ip >= n_ipanels) return;
    const uint32_t jp_begin = split * panels_per_split;
    const uint32_t jp_end = jp_begin + panels_per_split < n_jpanels ? jp_begin + panels_per_split : n_jpanels;

    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t wm = wave >> 1, wn = wave & 1;     // wave tile: j rows [128 wm, +128), i rows [128 wn, +128)
    const uint32_t half = lane >> 5, col = lane & 31;

    int list[4][LIST];   // per i tile of the wave: this lane's sorted keys, 0 = empty
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s = 0; s < LIST; ++s) list[nt][s] = 0;
    shared_level[tid] = 0;

    const i32x4* gi = dense_i + (uint64_t)ip * n_chunks * kPieceVec;
    const long long i_lo = (long long)ip * kPanel + wn * 128;   // first row i of this wave
    const uint32_t n_stages = n_chunks / 2;                     // the host pads the matrices to an even chunk count

    // One stream of (panel, stage) pieces: while a stage is multiplied out of one LDS buffer the next
    // one -- at the end of a panel, the first stage of the next j panel -- is copied HBM -> LDS by
    // the DMA path (no staging registers, no ds_write); __syncthreads() also waits for the copy.
    if (jp_begin < jp_end)
        stage_dma(dense_j + (uint64_t)jp_begin * n_chunks * kPieceVec, gi, &lds[0][0][0], &lds[0][1][0], tid, wave);
    __syncthreads();
    uint32_t buf = 0;

    for (uint32_t jp = jp_begin; jp < jp_end; ++jp) {
        i32x16 acc[4][4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0;
        const bool last_panel = jp + 1 == jp_end;
        for (uint32_t st = 0; st < n_stages; ++st) {
            const bool last_stage = st + 1 == n_stages;
            if (!(last_stage && last_panel)) {
                // the other buffer was last read (or used as parking space) before the previous barrier
                const uint32_t st_next = last_stage ? 0 : st + 1;
                stage_dma(dense_j + ((uint64_t)(last_stage ? jp + 1 : jp) * n_chunks + 2 * st_next) * kPieceVec,
                          gi + (uint64_t)(2 * st_next) * kPieceVec, &lds[buf ^ 1][0][0], &lds[buf ^ 1][1][0], tid, wave);
            }
            i32x4 fa[2][4], fb[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[0][t] = lds[buf][0][((wm * 4 + t) * 2) * 64 + lane];
                fb[0][t] = lds[buf][1][((wn * 4 + t) * 2) * 64 + lane];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {   // four 32-deep steps per stage; fragments of step s+1 load during step s
                if (s < 3) {
                    const int sn = s + 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        fa[sn & 1][t] = lds[buf][0][(sn >> 1) * kPieceVec + ((wm * 4 + t) * 2 + (sn & 1)) * 64 + lane];
                        fb[sn & 1][t] = lds[buf][1][(sn >> 1) * kPieceVec + ((wn * 4 + t) * 2 + (sn & 1)) * 64 + lane];
                    }
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[s & 1][mt], fb[s & 1][nt], acc[mt][nt], 0, 0, 0);
            }
            __syncthreads();
            buf ^= 1;
        }

        // ---- epilogue (once per j panel): fold this 128 x 128 wave tile into the per-lane top-k lists
        {
            const uint32_t j_local0 = (jp - jp_begin) * kPanel + wm * 128 + 4 * half;   // local j of (mt 0, reg 0)
            const long long j_lo = (long long)jp * kPanel + wm * 128;                    // first global j of the wave tile
            const bool tail = j_lo + 127 >= (long long)nb;
            const bool diag = excl >= 0 && j_lo <= i_lo + 127 + excl && i_lo + excl <= j_lo + 127;
            // Padding rows of the last panel and the excluded pair are never candidates.  With r = row of
            // the wave tile that lane half 0 holds in a slot, both tests are against lane constants.
            const int pad_from = tail ? (int)((long long)nb - j_lo) - 4 * (int)half : (1 << 30);
            const int skip_base = diag ? (int)(i_lo + excl - j_lo) + (int)col - 4 * (int)half : -(1 << 30);
            const int low_base = kLowMask - (int)j_local0;
            // The buffer that was just multiplied is idle until the next barrier: each wave parks one i
            // tile's 64 counts per lane (4 j tiles x 16 registers) in its own quarter of it, straight from
            // the accumulators, and walks them in a rolled loop.  The counts therefore never need
            // VALU-visible registers while the matrix pipeline owns the register file.
            i32x4* park = &lds[buf ^ 1][0][0] + wave * 1024 + lane;   // [slot / 4][lane]
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                __builtin_amdgcn_sched_barrier(0);   // keep these accumulator reads next to their stores
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const i32x16& t16 = acc[mt][nt];
                        park[(mt * 4 + q) * 64] = i32x4{t16[4 * q], t16[4 * q + 1], t16[4 * q + 2], t16[4 * q + 3]};
                    }
                __builtin_amdgcn_sched_barrier(0);
                const int skip_at = skip_base + nt * 32;
                int* level_slot = &shared_level[wn * 128 + nt * 32 + col];
                const int level = __hip_atomic_load(level_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
                // Candidates reach a lane in ascending j, so on equal counts the earlier one stays: a new
                // one must beat the count of the list's last entry, and reach the block's shared level.
                const int last = list[nt][LIST - 1];
                const int own = last == 0 ? 0 : (last >> kKeyShift) + 1;
                const int need = own > level ? own : level;
                if (__builtin_amdgcn_ballot_w64(m >= need) == 0) continue;
                // per lane: which of its 64 parked slots qualify (slot order = ascending j) ...
                uint32_t pass_lo = 0, pass_hi = 0;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const i32x4 v = park[g * 64];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        const uint32_t bit = v[x] >= need ? (1u << ((g * 4 + x) & 31)) : 0u;
                        if (g < 8) pass_lo |= bit;
                        else pass_hi |= bit;
                    }
                }
                // ... then one insertion round per set bit of the busiest lane (usually one or two)
                const int* park_dw = reinterpret_cast<const int*>(park);
                while (__builtin_amdgcn_ballot_w64((pass_lo | pass_hi) != 0) != 0) {
                    int key = -1;
                    if ((pass_lo | pass_hi) != 0) {
                        const int slot = pass_lo ? __builtin_ctz(pass_lo) : 32 + __builtin_ctz(pass_hi);
                        if (pass_lo) pass_lo &= pass_lo - 1;
                        else pass_hi &= pass_hi - 1;
                        const int g = slot >> 2, x = slot & 3;
                        const int count = park_dw[g * 256 + x];   // vector g of this lane, element x
                        key = (count << kKeyShift) + (low_base - ((g >> 2) * 32 + (g & 3) * 8 + x));
                    }
                    list_insert<LIST>(list[nt], key);
                }
                const int now = list[nt][LIST - 1];
                if (now != last && (now >> kKeyShift) > level) atomicMax(level_slot, now >> kKeyShift);
            }
        }
        __syncthreads();   // nobody may refill the parking buffer while a wave still walks its counts
    }

    // partial lists: [row i][split][wm*2 + half][LIST]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const uint64_t i = (uint64_t)ip * kPanel + wn * 128 + nt * 32 + col;
        int* dst = partial + ((i * splits + split) * kSubLists + (wm * 2 + half)) * LIST;
#pragma unroll
        for (int s = 0; s < LIST; ++s) dst[s] = list[nt][s];
    }
}

// One wave per row i: every lane keeps up to 8 of the row's partial-list keys (widened to
// count << 32 | ~j) in registers; k rounds of "largest key below the previous one", each a lane-local
// maximum followed by a wave-wide maximum.
constexpr int kMergeKeysPerLane = 8;   // 64 lanes x 8 = 512 keys = 8 splits x 4 lists x 16 entries at most

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_merge_topk(const int* __restrict__ partial, uint32_t na, uint32_t splits,
                                                    uint32_t per_split, unsigned long long rows_per_split, uint32_t k,
                                                    uint32_t* __restrict__ out_counts, int32_t* __restrict__ out_rows) {
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= na) return;
    const uint32_t n_keys = splits * per_split;
    const int* mine = partial + (uint64_t)i * n_keys;
    unsigned long long wide[kMergeKeysPerLane];
#pragma unroll
    for (int s = 0; s < kMergeKeysPerLane; ++s) {
        const uint32_t e = lane + 64u * s;
        const int key = e < n_keys ? mine[e] : 0;
        const unsigned long long j_base = (e / per_split) * rows_per_split;
        const unsigned long long count = (unsigned long long)(key >> kKeyShift);
        const unsigned long long j = j_base + (unsigned long long)(kLowMask - (key & kLowMask));
        wide[s] = key > 0 ? (count << 32) | (0xFFFFFFFFull - j) : 0ull;
    }
    unsigned long long prev = ~0ull;
    for (uint32_t t = 0; t < k; ++t) {
        unsigned long long best = 0;
#pragma unroll
        for (int s = 0; s < kMergeKeysPerLane; ++s)
            if (wide[s] < prev && wide[s] > best) best = wide[s];
        best = wave_max_u64(best);
        if (lane == 0) {
            out_counts[(uint64_t)i * k + t] = best ? (uint32_t)(best >> 32) : 0u;
            out_rows[(uint64_t)i * k + t] = best ? (int32_t)(0xFFFFFFFFull - (best & 0xFFFFFFFFull)) : -1;
        }
        if (best) prev = best;
        else prev = 0;   // nothing left: every further slot is empty
    }
}

uint32_t chunks_of(const sdrb_batch* b) { return (uint32_t)((b->width + kChunk - 1) / kChunk); }

// builds (or reuses) the fragment-ordered byte matrix of a batch, `n_chunks` chunks wide
int ensure_dense(const sdrb_batch* b, uint32_t n_chunks, hipStream_t stream) {
    sdrb_batch* mb = derived_owner(b);
    std::lock_guard<std::mutex> lock(derived_mutex());
    if (mb->d_dense8 && mb->dense_chunks == n_chunks) return derived_wait(mb->dense_ready, mb->dense_stream, stream);
    if (mb->d_dense8) {
        // another chunk count (the batch now meets a partner of another width): calls on any stream may still
        // read the old matrix, so the device is drained before it goes
        SDRB_HIP(hipDeviceSynchronize());
        (void)hipFree(mb->d_dense8);
        mb->d_dense8 = nullptr;
    }
    const uint64_t panels = (b->n_rows + kPanel - 1) / kPanel;
    const size_t bytes = (size_t)panels * n_chunks * kPieceVec * 16;
    uint8_t* dense = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dense), bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) for the dense byte matrix failed: %s", bytes, hipGetErrorString(e));
        return SDRB_ERR_HIP;
    }
    int rc = SDRB_OK;
    if (hipMemsetAsync(dense, 0, bytes, stream) != hipSuccess) rc = SDRB_ERR_HIP;
    if (rc == SDRB_OK) {
        hipLaunchKernelGGL(k_densify, dim3((uint32_t)((b->n_rows + 3) / 4)), dim3(256), 0, stream, b->d_offsets,
                           b->d_ids, (uint32_t)b->n_rows, n_chunks, dense);
        if (hipGetLastError() != hipSuccess) rc = SDRB_ERR_HIP;
    }
    if (rc == SDRB_OK) rc = derived_built(&mb->dense_ready, &mb->dense_stream, stream);
    if (rc != SDRB_OK) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(dense);
        set_error("building the dense byte matrix of a batch failed");
        return rc;
    }
    mb->dense_chunks = n_chunks;
    mb->d_dense8 = dense;   // published last: whoever sees the pointer also sees the event of its build
    return SDRB_OK;
}

template <int LIST>
int run(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t excl, uint32_t* out_counts,
        int32_t* out_rows, hipStream_t stream) {
    // common chunk count of both byte matrices, even (the kernel consumes two chunks per stage)
    const uint32_t n_chunks = ((chunks_of(a) > chunks_of(b) ? chunks_of(a) : chunks_of(b)) + 1) / 2 * 2;
    int rc = ensure_dense(a, n_chunks, stream);
    if (rc != SDRB_OK) return rc;
    rc = ensure_dense(b, n_chunks, stream);
    if (rc != SDRB_OK) return rc;
    const uint32_t n_ipanels = (uint32_t)((a->n_rows + kPanel - 1) / kPanel);
    const uint32_t n_jpanels = (uint32_t)((b->n_rows + kPanel - 1) / kPanel);
    // j-splits: as few as fill the 256 CUs (every split keeps its own candidate lists, whose warm-up
    // is the expensive part of the epilogue), each within the key range
    uint32_t splits = 1;
    while (splits < 8 && (uint64_t)((n_ipanels + 7) / 8 * 8) * splits < 256 && splits * 2 <= n_jpanels) splits *= 2;
    uint32_t pps = (n_jpanels + splits - 1) / splits;
    while ((uint64_t)pps * kPanel > kMaxSplitRows) {
        splits *= 2;
        pps = (n_jpanels + splits - 1) / splits;
    }
    const uint32_t ipanels_pad = (n_ipanels + 7) / 8 * 8;
    const size_t pbytes = (size_t)n_ipanels * kPanel * splits * kSubLists * LIST * sizeof(int);
    int* partial = static_cast<int*>(workspace(b->device, stream, 4, pbytes));
    if (!partial) return SDRB_ERR_HIP;
    hipLaunchKernelGGL(k_allpairs_mfma<LIST>, dim3(ipanels_pad * splits), dim3(256), 0, stream,
                       reinterpret_cast<const i32x4*>(derived_owner(a)->d_dense8), reinterpret_cast<const i32x4*>(derived_owner(b)->d_dense8),
                       n_ipanels, n_jpanels, n_chunks, splits, pps, (uint32_t)b->n_rows, (long long)excl, partial);
    SDRB_HIP(hipGetLastError());
    return launch_merge_topk(partial, (uint32_t)a->n_rows, splits, kSubLists, (uint64_t)pps * kPanel, k, out_counts,
                             out_rows, stream);
}

}  // namespace

int launch_merge_topk(const int* partial, uint32_t na, uint32_t splits, uint32_t lists_per_split,
                      uint64_t rows_per_split, uint32_t k, uint32_t* out_counts, int32_t* out_rows,
                      hipStream_t stream) {
    if ((uint64_t)splits * lists_per_split * kTopkList > 64u * kMergeKeysPerLane) {
        set_error("too many partial top-k lists per row (%u splits)", splits);
        return SDRB_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_merge_topk, dim3((na + 3) / 4), dim3(256), 0, stream, partial, na, splits,
                       lists_per_split * kTopkList, (unsigned long long)rows_per_split, k, out_counts, out_rows);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

bool allpairs_mfma_applicable(const sdrb_batch* a, const sdrb_batch* b, uint32_t k) {
    if (a->width <= 0 || b->width <= 0 || a->width > 16384 || b->width > 16384) return false;
    if (k == 0 || k > 16) return false;   // list length of the kernel
    // counts must fit the key: an overlap never exceeds the shorter row
    const uint32_t cap = a->max_row_len < b->max_row_len ? a->max_row_len : b->max_row_len;
    if (cap >= (1u << (31 - kKeyShift))) return false;
    if (a->n_rows > 0x7FFFFF00ull) return false;
    // the merge holds 8 splits x 4 lists of 16 keys per row, and a split's local row index must fit the key
    if (b->n_rows > 8ull * (kMaxSplitRows / kPanel) * kPanel) return false;      // 4 192 256 candidate rows
    return true;
}

int launch_allpairs_mfma(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                         uint32_t* out_counts, int32_t* out_rows, hipStream_t stream) {
    return run<16>(a, b, k, exclude_offset, out_counts, out_rows, stream);
}

}  // namespace sdrb

