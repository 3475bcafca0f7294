// Union-type element operations (ore / xore / rme) with id-only results (`this` side EmptyData) on short
// rows over a declared width of at most 4096: one LANE per pair, output positions from ranks instead of a
// serial merge, results staged in LDS and written as one coalesced run.
//
// The reference produces a pair's result with the three-way merge of orv (SDR.hpp:1099-1150): ore pushes every
// id of either row (SDR.hpp:1207-1217), xore the ids of exactly one row -- an EmptyData match is never
// rm_relevant (SDR.hpp:1332-1344, EmptyData.hpp:34-37) --, rme the ids of `this` that `arg` lacks
// (SDR.hpp:1540-1560), all in ascending order.  The place of an id in such a result is a count:
//     an id b of B_j that A_i lacks sits behind the ids of A_i below it and the B-only ids below it,
//     an id a of A_i             sits behind the ids of A_i below it and the B-only ids below it
// (for xore the matched ids of A_i are taken out of both counts, for rme only one side is emitted).
// A wave takes 64 consecutive rows of B, one row per lane, the row's ids in REGISTERS.  For row i of A
//   pass B: every lane walks its own ids b (a fully unrolled loop, no data-dependent control): one 8-byte
//           LDS read of A_i's rank table -- entry w = {bits of ids 32w..32w+31, number of ids below 32w} --
//           gives rank_A(b) = #{a < b} and whether b is in A_i; a B-only id goes straight to its place in the
//           wave's LDS stage and leaves, in a per-lane table over rank_A (bytes in LDS), the number of B-only
//           ids up to and including itself; a matched one sets bit rank_A(b) of a per-lane 64-bit mask;
//   pass A: the wave walks A_i's ids a_k (the same id for every lane: a `readlane` of a register that holds
//           the row), a lane takes the running maximum of that table's entries 0..k -- the B-only ids below
//           a_k -- and writes a_k to its place.
// Nothing in either pass waits for the result of an earlier LDS access of the same pass, so the reads of many
// ids are in flight together; a serial two-pointer merge pays an LDS round trip per id.
// The results of one row of A over 64 consecutive rows of B are neighbours in the output ([i][j] order, j
// fastest): the stage is one contiguous run that leaves the wave as 16-byte stores per lane.
//
// Memory operations of a wave complete in order, so a wait for a load also waits for every store issued
// before it.  Nothing is therefore loaded between a run's stores and the next wait: the offsets of the next
// row of A -- and, at a wave's last row, the next 64 rows of B -- are requested before the passes and claimed
// right after them, before the run's stores are issued.
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

#ifndef SDRB_LR_WAVES
#define SDRB_LR_WAVES 2
#endif
#ifndef SDRB_LR_TABLE_KB
#define SDRB_LR_TABLE_KB 4   // eight rows of A a block at width 2048: five blocks (ten waves) share a compute unit; with 8 KiB
                             // four do (config 3, 16 queries: ore 1.37 -> 1.27 ms, rme 1.11 -> 0.97 ms)
#endif
constexpr int kLrWaves = SDRB_LR_WAVES;      // waves per block (each works on its own 64 rows of B)
constexpr int kLrBlock = 64 * kLrWaves;
constexpr uint32_t kLrRowCap = 64;           // longest row on either side
constexpr uint32_t kLrTableLds = SDRB_LR_TABLE_KB * 1024;   // LDS for the rank tables of the block's chunk of rows of A

typedef uint32_t lr_u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t lr_i32x4 __attribute__((ext_vector_type(4)));

// vmcnt(0) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
__device__ __forceinline__ void lr_wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// OP: SDRB_OP_OR / XOR / RM.  B_IS_THIS (rme only): result = B_j.rme(A_i), i.e. ids of B_j that A_i lacks.
// NQ: a lane fetches its row of B as NQ loads of four ids (rows of B hold at most 4 NQ ids).
// Dynamic LDS: a_tab[chunk][nw] {bits, below} | a_ids[chunk][64] u16 | a_len[chunk] |
//              per wave: hist[la_cap + 1][64] bytes, stage[stage_cap] u16
template <int OP, bool B_IS_THIS, int NQ>
__global__ __launch_bounds__(kLrBlock) void k_elem_lanerank(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, uint32_t na, uint32_t a_chunk, uint32_t nw,
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint32_t nb, uint32_t stage_cap,
    const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids, uint32_t aligned, uint32_t la_cap) {
    constexpr bool NEED_HIST = OP != SDRB_OP_RM;                     // pass A counts the B-only ids below a_k
    constexpr bool NEED_MASK = OP == SDRB_OP_XOR || (OP == SDRB_OP_RM && !B_IS_THIS);   // pass A skips matched a_k
    constexpr bool PASS_A = !(OP == SDRB_OP_RM && B_IS_THIS);
    constexpr bool B_EMITS = !(OP == SDRB_OP_RM && !B_IS_THIS);
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // uniform, and known to be
    uint2* a_tab = reinterpret_cast<uint2*>(lr_raw);
    uint16_t* a_row_ids = reinterpret_cast<uint16_t*>(lr_raw + (size_t)a_chunk * nw * 8u);
    uint32_t* a_len = reinterpret_cast<uint32_t*>(a_row_ids + a_chunk * 64u);
    const uint32_t shared_bytes = (a_chunk * nw * 8u + a_chunk * 128u + a_chunk * 4u + 15u) & ~15u;
    const uint32_t hist_bytes = (la_cap + 1) * 64u;      // ranks run from 0 to the longest row of A
    const uint32_t wave_bytes = (hist_bytes + stage_cap * 2u + 15u) & ~15u;
    unsigned char* hist = lr_raw + shared_bytes + wave * wave_bytes;
    uint16_t* stage = reinterpret_cast<uint16_t*>(hist + hist_bytes);

    // rank tables, ids and lengths of this block's rows of A
    const uint32_t i0 = blockIdx.y * a_chunk;
    const uint32_t rows_a = na - i0 < a_chunk ? na - i0 : a_chunk;
    for (uint32_t e = tid; e < rows_a * nw; e += kLrBlock) a_tab[e] = make_uint2(0u, 0u);
    __syncthreads();
    for (uint32_t e = tid; e < rows_a * 64u; e += kLrBlock) {
        const uint32_t r = e >> 6, k = e & 63u;
        const uint32_t beg = a_off[i0 + r], len = a_off[i0 + r + 1] - beg;
        if (k == 0) a_len[r] = len;
        if (k < len) {
            const uint32_t id = (uint32_t)a_ids[beg + k];      // the host checked 0 <= id < 32 nw
            a_row_ids[e] = (uint16_t)id;
            atomicOr(&a_tab[r * nw + (id >> 5)].x, 1u << (id & 31u));
        }
    }
    __syncthreads();
    for (uint32_t e = tid; e < rows_a * nw; e += kLrBlock) {
        const uint32_t r = e / nw, w = e % nw;
        uint32_t below = 0;
        for (uint32_t v = 0; v < w; ++v) below += (uint32_t)__builtin_popcount(a_tab[r * nw + v].x);
        a_tab[e].y = below;
    }
    __syncthreads();

    const uint32_t n_tasks = (nb + 63) / 64;
    const uint32_t task_step = gridDim.x * kLrWaves;

    // where this lane's row of a task starts and ends: raw loads from a clamped (always loadable) position, so
    // that nothing has to wait for them here; span_len() turns them into the row's length once they have arrived
    auto fetch_span = [&](uint32_t task, uint32_t& beg, uint32_t& end) {
        const uint64_t j = (uint64_t)task * 64 + lane;
        const uint64_t jc = j < nb ? j : nb - 1;
        beg = b_off[jc];
        end = b_off[jc + 1];
    };
    auto span_len = [&](uint32_t task, uint32_t beg, uint32_t end) {
        return (uint64_t)task * 64 + lane < nb ? end - beg : 0u;
    };
    // a task's rows of B, one row per lane, as 16-byte loads through a bounds-checked buffer that covers
    // exactly the wave's rows (slots past a row's end are never looked at)
    auto fetch_rows = [&](uint32_t task, uint32_t beg, lr_u32x4 (&rows)[NQ]) {
        const uint32_t j0 = task * 64;
        const uint32_t rows_here = nb - j0 < 64u ? nb - j0 : 64u;
        const uint32_t blk_beg = b_off[j0];
        const uint32_t blk_end = b_off[j0 + rows_here];
        __amdgpu_buffer_rsrc_t ids_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<int32_t*>(b_ids) + blk_beg, 0, (int)((blk_end - blk_beg) * 4u), 0x00020000);
        const uint32_t rel = (beg - blk_beg) * 4u;
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) rows[q4] = __builtin_amdgcn_raw_buffer_load_b128(ids_rsrc, (int)(rel + 16u * q4), 0, 0);
    };
    // offsets of a wave's pairs for row i of the chunk: this lane's pair and the next one
    auto fetch_off = [&](uint32_t task, uint32_t i, uint32_t& mine, uint32_t& next) {
        const uint32_t j0 = task * 64;
        const uint32_t rows_here = nb - j0 < 64u ? nb - j0 : 64u;
        const uint64_t p = (uint64_t)(i0 + i) * nb + j0 + (lane < rows_here ? lane : rows_here - 1);
        mine = out_off[p];
        next = out_off[p + 1];
    };

    uint32_t task = blockIdx.x * kLrWaves + wave;
    if (task >= n_tasks) return;
    lr_u32x4 rows[NQ], nx_rows[NQ];
    // lb: length of this lane's row in the current task; nx_*: start and end of its row in the next task,
    // always known one task ahead so that the rows themselves can be requested without waiting for anything
    uint32_t lb, beg0, end0, nx_beg, nx_end, off_mine, off_next;
    fetch_span(task, beg0, end0);
    fetch_span(task + task_step, nx_beg, nx_end);
    fetch_off(task, 0, off_mine, off_next);
    lr_wait_vmem();
    lb = span_len(task, beg0, end0);
    fetch_rows(task, beg0, rows);
    lr_wait_vmem();
#pragma unroll
    for (int q4 = 0; q4 < NQ; ++q4) nx_rows[q4] = rows[q4];
    while (true) {
        const uint32_t j0 = task * 64;
        const uint32_t rows_here = nb - j0 < 64u ? nb - j0 : 64u;
        const bool valid_lane = lane < rows_here;
        uint32_t lb_wave = lb;      // the longest row of this wave
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)lb_wave, d, 64);
            lb_wave = o > lb_wave ? o : lb_wave;
        }
        lb_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)lb_wave);

        const uint32_t next_task = task + task_step;
        const bool more_tasks = next_task < n_tasks;
        for (uint32_t i = 0; i < rows_a; ++i) {
            const uint32_t run_beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)off_mine);
            const uint32_t run_end = (uint32_t)__builtin_amdgcn_readlane((int)off_next, (int)(rows_here - 1));
            const uint32_t shift = aligned ? (run_beg & 3u) : 0u;
            uint16_t* mine = stage + (off_mine - run_beg + shift);   // this pair's first slot in the stage
            // requests for what comes next: the offsets of the next row (of the next task's first row at the
            // wave's last row -- clamped to something loadable when there is none) and the next rows of B
            const bool last_row = i + 1 == rows_a;
            uint32_t nxt_mine, nxt_next;
            fetch_off(last_row ? (more_tasks ? next_task : task) : task, last_row ? 0u : i + 1, nxt_mine, nxt_next);
            uint32_t n2_beg = 0, n2_end = 0;
            if (last_row) {
                if (more_tasks) fetch_rows(next_task, nx_beg, nx_rows);
                fetch_span(next_task + task_step, n2_beg, n2_end);
            }

            const uint32_t la_i = a_len[i];
            const uint2* tab = a_tab + i * nw;
            const uint32_t a_vec = a_row_ids[i * 64u + lane];      // lane k holds a_k
            if constexpr (NEED_HIST) {
                // clear rows 0 .. la_i of the histogram (16 bytes per lane and store)
                for (uint32_t o = lane * 16u; o < (la_i + 1) * 64u; o += 1024u)
                    *reinterpret_cast<lr_u32x4*>(hist + o) = lr_u32x4{0u, 0u, 0u, 0u};
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }

            // pass B: this lane's ids
            uint32_t m = 0;                 // matches so far
            unsigned long long match_mask = 0;   // bit k: a_k is in B_j
#pragma unroll
            for (int l = 0; l < 4 * NQ; ++l) {
                if ((uint32_t)l < lb_wave && (uint32_t)l < lb) {    // the first test is uniform: whole turns are skipped
                    const uint32_t b = rows[l / 4][l % 4];
                    const uint2 ent = tab[b >> 5];
                    const uint32_t sh = b & 31u;
                    const uint32_t r = ent.y + (uint32_t)__builtin_popcount(ent.x & ((1u << sh) - 1u));
                    const uint32_t in_a = (ent.x >> sh) & 1u;
                    if (!in_a) {
                        if constexpr (B_EMITS) {
                            uint32_t pos;
                            if constexpr (OP == SDRB_OP_OR) pos = r + (uint32_t)l - m;
                            else if constexpr (OP == SDRB_OP_XOR) pos = r + (uint32_t)l - 2u * m;
                            else pos = (uint32_t)l - m;
                            mine[pos] = (uint16_t)b;
                        }
                        // B-only ids so far, this one included: ranks never decrease along the row, so the last
                        // value written under a rank is the number of B-only ids of that rank or lower
                        if constexpr (NEED_HIST) hist[r * 64u + lane] = (unsigned char)((uint32_t)l - m + 1u);
                    }
                    m += in_a;
                    if constexpr (NEED_MASK) match_mask |= (unsigned long long)in_a << (r & 63u);
                }
            }
            if constexpr (PASS_A) {
                if constexpr (NEED_HIST) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // pass A: the ids of A_i, the same id for every lane in every turn
                uint32_t below = 0;      // B-only ids below a_k (ore, xore)
                uint32_t gone = 0;       // matched ids of A_i below a_k (xore, rme)
#pragma unroll
                for (int k = 0; k < (int)kLrRowCap; ++k) {
                    if ((uint32_t)k < la_i) {     // uniform
                        const uint32_t a_k = (uint32_t)__builtin_amdgcn_readlane((int)a_vec, k);
                        if constexpr (NEED_HIST) {
                            const uint32_t upto = hist[(uint32_t)k * 64u + lane];   // 0 where no B-only id has rank k
                            below = upto > below ? upto : below;
                        }
                        uint32_t matched = 0;
                        if constexpr (NEED_MASK) matched = (uint32_t)(match_mask >> k) & 1u;
                        if (valid_lane && !matched) mine[(uint32_t)k - gone + below] = (uint16_t)a_k;
                        gone += matched;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // what was requested above has had both passes to arrive; claiming it here keeps the wait clear of
            // the stores that follow
            lr_wait_vmem();

            // The run leaves the wave four ids per lane: the stage is laid out so that slot 4c matches a 16-byte
            // boundary of the output (the pairs start `shift` slots in), a lane expands four staged ids and stores
            // them with one 16-byte store; only the run's first and last quad can be partial.
            const uint32_t run_len = run_end - run_beg;
            const uint32_t span = shift + run_len, quads_out = (span + 3) / 4;
            int32_t* dst = out_ids + (run_beg - shift);    // 16-byte aligned when out_ids is
#ifdef SDRB_LAB_LANERANK_NO_STORES   // timing only: everything but the run's stores (`aligned` is 0 or 1)
            if (aligned == 2u)
#endif
            for (uint32_t c = lane; c < quads_out; c += 64) {
                const uint2 packed = *reinterpret_cast<const uint2*>(stage + 4 * c);
                lr_i32x4 v;
                v[0] = (int32_t)(packed.x & 0xFFFFu);
                v[1] = (int32_t)(packed.x >> 16);
                v[2] = (int32_t)(packed.y & 0xFFFFu);
                v[3] = (int32_t)(packed.y >> 16);
                const uint32_t e = 4 * c;
                if (aligned && e >= shift && e + 4 <= span) {
                    *reinterpret_cast<lr_i32x4*>(dst + e) = v;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (e + k >= shift && e + k < span) dst[e + k] = v[k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            off_mine = nxt_mine;
            off_next = nxt_next;
            if (last_row) {
                lb = span_len(next_task, nx_beg, nx_end);
                nx_beg = n2_beg;
                nx_end = n2_end;
            }
        }
        if (!more_tasks) break;
        task = next_task;
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) rows[q4] = nx_rows[q4];
    }
}

struct LrPlan {
    int nq;
    uint32_t nw, stage_cap, a_chunk;
    size_t lds;
};

bool make_lr_plan(const sdrb_batch* a, const sdrb_batch* b, LrPlan* p) {
    if (a->id_bits != 32 || b->id_bits != 32 || a->n_rows == 0 || b->n_rows == 0) return false;
    if (b->width <= 0 || b->width > 4096) return false;
    // every id of A must have a bit in the rank tables (A-only ids appear in ore / xore / rme)
    if (a->nnz && (a->min_id < 0 || a->max_id >= b->width)) return false;
    if (a->max_row_len > kLrRowCap || b->max_row_len > kLrRowCap) return false;
    p->nq = b->max_row_len <= 16 ? 4 : b->max_row_len <= 40 ? 10 : 16;
    p->nw = ((uint32_t)b->width + 31u) / 32u;
    p->stage_cap = 64u * (a->max_row_len + b->max_row_len) + 8u;   // up to 3 slots of alignment shift, rounded for 8-byte reads
    uint32_t chunk = kLrTableLds / (p->nw * 8u);
    if (chunk > a->n_rows) chunk = (uint32_t)a->n_rows;
    if ((a->n_rows + chunk - 1) / chunk > 65535) return false;
    p->a_chunk = chunk;
    const size_t shared_bytes = ((size_t)chunk * p->nw * 8u + (size_t)chunk * 128u + (size_t)chunk * 4u + 15u) & ~(size_t)15u;
    const size_t wave_bytes = ((size_t)(a->max_row_len + 1) * 64u + (size_t)p->stage_cap * 2u + 15u) & ~(size_t)15u;
    p->lds = shared_bytes + (size_t)kLrWaves * wave_bytes;
    return p->lds <= 160 * 1024 - 512;
}

template <int OP, bool BT, int NQ>
int launch_lr(const LrPlan& p, dim3 grid, const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
              int32_t* out_ids, hipStream_t stream) {
    const uint32_t aligned = (reinterpret_cast<uintptr_t>(out_ids) & 15u) == 0 ? 1u : 0u;
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_elem_lanerank<OP, BT, NQ>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    hipLaunchKernelGGL((k_elem_lanerank<OP, BT, NQ>), grid, dim3(kLrBlock), p.lds, stream, a->d_offsets, a->d_ids,
                       (uint32_t)a->n_rows, p.a_chunk, p.nw, b->d_offsets, b->d_ids, (uint32_t)b->n_rows, p.stage_cap,
                       out_offsets, out_ids, aligned, a->max_row_len);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <int OP, bool BT>
int launch_lr_nq(const LrPlan& p, dim3 grid, const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
                 int32_t* out_ids, hipStream_t stream) {
    switch (p.nq) {
    case 4: return launch_lr<OP, BT, 4>(p, grid, a, b, out_offsets, out_ids, stream);
    case 10: return launch_lr<OP, BT, 10>(p, grid, a, b, out_offsets, out_ids, stream);
    default: return launch_lr<OP, BT, 16>(p, grid, a, b, out_offsets, out_ids, stream);
    }
}

}  // namespace

bool elem_lanerank_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const sdrb_batch* self = swap ? b : a;
    if (self->data_kind != SDRB_DATA_EMPTY) return false;   // results would carry payloads
    if (op == SDRB_OP_AND) return false;                    // intersections: the mask-table kernels
    if ((uint64_t)a->n_rows * b->n_rows >= 0xFFFFFFFFull) return false;
    LrPlan p;
    return make_lr_plan(a, b, &p);
}

int launch_elem_lanerank(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                         int32_t* out_ids, hipStream_t stream) {
    LrPlan p;
    if (!make_lr_plan(a, b, &p)) {
        set_error("lane-per-pair rank fill does not apply to these batches");
        return SDRB_ERR_UNSUPPORTED;
    }
    const uint32_t na = (uint32_t)a->n_rows, nb = (uint32_t)b->n_rows;
    const uint32_t n_tasks = (nb + 63) / 64;
    const uint32_t gy = (na + p.a_chunk - 1) / p.a_chunk;
    // blocks resident per compute unit by their LDS, a few turns of work for each
    const uint32_t per_cu = (uint32_t)((160u * 1024u) / (p.lds + 256));
    uint32_t gx = (n_tasks + kLrWaves - 1) / kLrWaves;
    const uint32_t cap = 256u * (per_cu ? per_cu : 1u) * 2u;
    if ((uint64_t)gx * gy > cap) gx = cap / gy ? cap / gy : 1u;
    const dim3 grid(gx, gy);
    switch (op) {
    case SDRB_OP_OR: return launch_lr_nq<SDRB_OP_OR, false>(p, grid, a, b, out_offsets, out_ids, stream);
    case SDRB_OP_XOR: return launch_lr_nq<SDRB_OP_XOR, false>(p, grid, a, b, out_offsets, out_ids, stream);
    case SDRB_OP_RM:
        if (swap) return launch_lr_nq<SDRB_OP_RM, true>(p, grid, a, b, out_offsets, out_ids, stream);
        return launch_lr_nq<SDRB_OP_RM, false>(p, grid, a, b, out_offsets, out_ids, stream);
    default:
        set_error("lane-per-pair rank fill serves ore / xore / rme");
        return SDRB_ERR_UNSUPPORTED;
    }
}

}  // namespace sdrb
