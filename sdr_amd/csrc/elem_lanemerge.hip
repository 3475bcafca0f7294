// Union-type element operations (ore / xore / rme) on SHORT rows whose results carry ids only
// (`this` side EmptyData): one LANE per pair, outputs staged in LDS and written as one coalesced run.
//
// The reference walks a pair with the three-way merge of orv (SDR.hpp:1099-1150) and pushes
//   ore : every id of either row                       (SDR.hpp:1207-1217)
//   xore: ids of exactly one row -- an EmptyData match is never rm_relevant (SDR.hpp:1332-1344, EmptyData.hpp:34-37)
//   rme : ids of `this` that `arg` lacks               (SDR.hpp:1540-1560)
// in ascending order.  Here a wave takes 64 consecutive rows of B and keeps them in LDS transposed
// ([position][lane], so a lane's reads never collide with another lane's); a lane then runs exactly that merge
// for its row against row i of A (a block keeps a chunk of A's rows in LDS).  A turn of the merge loop looks
// at the two leading ids of each row and emits the next TWO ids of the union, so the loop pays one LDS round
// trip per two outputs; lanes diverge only in that some finish early.  The results of one row of A over 64
// consecutive rows of B are neighbours in the output ([i][j] order, j fastest): every lane writes its ids into
// the wave's LDS stage at the place its pair's offset gives it, and the run leaves the wave as 16-byte stores
// per lane.  Ids are staged as 16-bit values when every id fits.
//
// Memory operations of a wave complete in order, so a wait for a load also waits for every store issued
// before it.  The loop is therefore arranged so that nothing is loaded between a run's stores and the next
// wait: the offsets of the next row of A -- and, at a wave's last row, the next 64 rows of B -- are requested
// BEFORE the merge and claimed right after it, before the run's stores are issued.
#include "sdrb_device.h"
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kLmWaves = 2;                 // waves per block (each works on its own 64 rows of B)
constexpr int kLmBlock = 64 * kLmWaves;
constexpr uint32_t kLmWaveLds = 36 * 1024;  // per-wave LDS the kernel may use (tile of B + stage)
constexpr uint32_t kLmChunkLds = 8 * 1024;  // LDS for the block's chunk of rows of A

template <typename T>
struct Sentinel;
template <>
struct Sentinel<uint16_t> {
    static constexpr uint32_t value = 0xFFFFu;
};
template <>
struct Sentinel<int32_t> {
    static constexpr int32_t value = 0x7FFFFFFF;
};

typedef uint32_t lm_u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t lm_i32x4 __attribute__((ext_vector_type(4)));

// vmcnt(0) alone (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
__device__ __forceinline__ void wait_vmem() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// OP: SDRB_OP_OR / XOR / RM.  B_IS_THIS (rme only): result = B_j.rme(A_i), i.e. ids of B_j that A_i lacks.
// T: uint16_t (all ids in [0, 65535)) or int32_t (all ids < INT32_MAX); the sentinel ends a row.
// NQ: a lane fetches its row of B as NQ loads of four ids (rows of B hold at most 4 NQ ids).
// Layout of the dynamic LDS: a_rows[chunk][la_cap + 2] | a_len[chunk] | per wave: b_tile[lb_cap + 2][64], stage[stage_cap]
template <int OP, bool B_IS_THIS, typename T, int NQ>
__global__ __launch_bounds__(kLmBlock) void k_elem_lanemerge(
    const uint32_t* __restrict__ a_off, const int32_t* __restrict__ a_ids, uint32_t na, uint32_t a_chunk,
    uint32_t la_cap, const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, uint32_t nb,
    uint32_t lb_cap, uint32_t stage_cap, const uint32_t* __restrict__ out_off, int32_t* __restrict__ out_ids,
    uint32_t aligned) {
    constexpr auto SENT = Sentinel<T>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char lm_raw[];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));   // uniform, and known to be
    const uint32_t a_stride = la_cap + 2;
    T* a_rows = reinterpret_cast<T*>(lm_raw);
    const uint32_t a_ids_bytes = (a_chunk * a_stride * (uint32_t)sizeof(T) + 15u) & ~15u;
    const uint32_t a_bytes = a_ids_bytes + ((a_chunk * 4u + 15u) & ~15u);
    uint32_t* a_len = reinterpret_cast<uint32_t*>(lm_raw + a_ids_bytes);
    const uint32_t wave_bytes = (((lb_cap + 2) * 64u + stage_cap) * (uint32_t)sizeof(T) + 15u) & ~15u;
    T* b_tile = reinterpret_cast<T*>(lm_raw + a_bytes + wave * wave_bytes);
    T* stage = b_tile + (lb_cap + 2) * 64u;

    // this block's rows of A, each closed by sentinels
    const uint32_t i0 = blockIdx.y * a_chunk;
    const uint32_t rows_a = na - i0 < a_chunk ? na - i0 : a_chunk;
    for (uint32_t e = tid; e < rows_a * a_stride; e += kLmBlock) {
        const uint32_t r = e / a_stride, k = e % a_stride;
        const uint32_t beg = a_off[i0 + r], len = a_off[i0 + r + 1] - beg;
        a_rows[e] = k < len ? (T)a_ids[beg + k] : (T)SENT;
        if (k == 0) a_len[r] = len;
    }
    __syncthreads();

    const uint32_t n_tasks = (nb + 63) / 64;
    const uint32_t task_step = gridDim.x * kLmWaves;

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
    // exactly the wave's rows (what lies past the buffer's end reads as zero and is replaced later)
    auto fetch_rows = [&](uint32_t task, uint32_t beg, lm_u32x4 (&rows)[NQ]) {
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

    uint32_t task = blockIdx.x * kLmWaves + wave;
    if (task >= n_tasks) return;
    lm_u32x4 rows[NQ];
    // lb: length of this lane's row in the current task; nx_*: start and end of its row in the next task,
    // always known one task ahead so that the rows themselves can be requested without waiting for anything
    uint32_t lb, beg0, end0, nx_beg, nx_end, off_mine, off_next;
    fetch_span(task, beg0, end0);
    fetch_span(task + task_step, nx_beg, nx_end);
    fetch_off(task, 0, off_mine, off_next);
    wait_vmem();
    lb = span_len(task, beg0, end0);
    fetch_rows(task, beg0, rows);
    wait_vmem();
    while (true) {
        const uint32_t j0 = task * 64;
        const uint32_t rows_here = nb - j0 < 64u ? nb - j0 : 64u;
        // the lane's row of B -> column `lane` of the tile; slots past the row's end hold the sentinel
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t k = 4 * q4 + c;
                if (k < lb_cap + 2) b_tile[k * 64 + lane] = k < lb ? (T)rows[q4][c] : (T)SENT;
            }
        }
        for (uint32_t k = 4 * NQ; k < lb_cap + 2; ++k) b_tile[k * 64 + lane] = (T)SENT;
        uint32_t lb_wave = lb;      // the longest row of this wave
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)lb_wave, d, 64);
            lb_wave = o > lb_wave ? o : lb_wave;
        }
        lb_wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)lb_wave);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

        const uint32_t next_task = task + task_step;
        const bool more_tasks = next_task < n_tasks;
        for (uint32_t i = 0; i < rows_a; ++i) {
            const uint32_t run_beg = (uint32_t)__builtin_amdgcn_readfirstlane((int)off_mine);
            const uint32_t run_end = (uint32_t)__builtin_amdgcn_readlane((int)off_next, (int)(rows_here - 1));
            const uint32_t shift = aligned ? (run_beg & 3u) : 0u;
            const uint32_t w = off_mine - run_beg + shift;        // this pair's first slot in the stage
            // requests for what comes next: the offsets of the next row (of the next task's first row at the
            // wave's last row -- clamped to something loadable when there is none) and the next rows of B
            const bool last_row = i + 1 == rows_a;
            uint32_t nxt_mine, nxt_next;
            fetch_off(last_row ? (more_tasks ? next_task : task) : task, last_row ? 0u : i + 1, nxt_mine, nxt_next);
            uint32_t n2_beg = 0, n2_end = 0;
            if (last_row) {
                if (more_tasks) fetch_rows(next_task, nx_beg, rows);
                fetch_span(next_task + task_step, n2_beg, n2_end);
            }

            // (lanes past the last row of B start at the terminator of A's row; their column of the tile holds
            // terminators only)
            const T* pa = a_rows + i * a_stride + (lane < rows_here ? 0u : la_cap);
            const T* pb = b_tile + lane;
            auto x0 = pa[0], x1 = pa[1];
            auto y0 = pb[0], y1 = pb[64];
            T* ws = stage + w;
            auto keeps = [](bool from_a, bool from_b) {
                if constexpr (OP == SDRB_OP_OR) return from_a || from_b;
                else if constexpr (OP == SDRB_OP_XOR) return from_a != from_b;
                else return B_IS_THIS ? (from_b && !from_a) : (from_a && !from_b);
            };
            // a pair needs at most |A_i| + |B_j| single steps; the wave runs the longest pair's count, two per turn
            const uint32_t la_i = a_len[i];
#pragma unroll 1
            for (uint32_t turn = (la_i + lb_wave + 1) / 2; turn > 0; --turn) {
                // first id: the smaller head; a finished row (head at the terminator) never advances
                const auto m1 = x0 < y0 ? x0 : y0;
                const bool fa1 = x0 != (T)SENT && x0 <= y0, fb1 = y0 != (T)SENT && y0 <= x0;
                const auto xh = fa1 ? x1 : x0;
                const auto yh = fb1 ? y1 : y0;
                // second id, from the heads the first step leaves
                const auto m2 = xh < yh ? xh : yh;
                const bool fa2 = xh != (T)SENT && xh <= yh, fb2 = yh != (T)SENT && yh <= xh;
                const bool e1 = keeps(fa1, fb1), e2 = keeps(fa2, fb2);
                pa += (fa1 ? 1 : 0) + (fa2 ? 1 : 0);
                pb += (fb1 ? 64 : 0) + (fb2 ? 64 : 0);
                // the next heads are requested before this turn's ids are staged: the stage may alias the
                // tiles as far as the compiler can tell, and a read placed behind the stores would be repeated
                x0 = pa[0];
                x1 = pa[1];
                y0 = pb[0];
                y1 = pb[64];
                if (e1) ws[0] = (T)m1;
                ws += e1 ? 1 : 0;
                if (e2) ws[0] = (T)m2;
                ws += e2 ? 1 : 0;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // what was requested above has had the whole merge to arrive; claiming it here keeps the wait clear
            // of the stores that follow
            wait_vmem();

            // The run leaves the wave four ids per lane: the stage is laid out so that slot 4c matches a 16-byte
            // boundary of the output (the pairs start `shift` slots in), a lane expands four staged ids and stores
            // them with one 16-byte store; only the run's first and last quad can be partial.
            const uint32_t run_len = run_end - run_beg;
            const uint32_t span = shift + run_len, quads_out = (span + 3) / 4;
            int32_t* dst = out_ids + (run_beg - shift);    // 16-byte aligned when out_ids is
            for (uint32_t c = lane; c < quads_out; c += 64) {
                lm_i32x4 v;
                if constexpr (sizeof(T) == 2) {
                    const uint2 packed = *reinterpret_cast<const uint2*>(stage + 4 * c);
                    v[0] = (int32_t)(packed.x & 0xFFFFu);
                    v[1] = (int32_t)(packed.x >> 16);
                    v[2] = (int32_t)(packed.y & 0xFFFFu);
                    v[3] = (int32_t)(packed.y >> 16);
                } else {
                    v = *reinterpret_cast<const lm_i32x4*>(stage + 4 * c);
                }
                const uint32_t e = 4 * c;
                if (aligned && e >= shift && e + 4 <= span) {
                    *reinterpret_cast<lm_i32x4*>(dst + e) = v;
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
    }
}

struct LmPlan {
    bool narrow;          // 16-bit staging
    int nq;
    uint32_t la_cap, lb_cap, stage_cap, a_chunk;
    size_t lds;
};

bool make_lm_plan(const sdrb_batch* a, const sdrb_batch* b, LmPlan* p) {
    if (a->id_bits != 32 || b->id_bits != 32 || a->n_rows == 0 || b->n_rows == 0) return false;
    // the sentinel must exceed every id
    const int32_t hi = a->max_id > b->max_id ? a->max_id : b->max_id;
    const int32_t lo = a->min_id < b->min_id ? a->min_id : b->min_id;
    if (hi == 0x7FFFFFFF) return false;
    p->narrow = lo >= 0 && hi < 0xFFFF;
    const uint32_t sz = p->narrow ? 2u : 4u;
    p->la_cap = a->max_row_len;
    p->lb_cap = b->max_row_len;
    if (p->lb_cap > 96) return false;
    p->nq = p->lb_cap <= 16 ? 4 : p->lb_cap <= 40 ? 10 : p->lb_cap <= 64 ? 16 : 24;
    p->stage_cap = 64u * (p->la_cap + p->lb_cap) + 8u;   // up to 3 slots of alignment shift, rounded for 16-byte reads
    const uint64_t wave_bytes = ((uint64_t)((p->lb_cap + 2) * 64u + p->stage_cap) * sz + 15u) & ~15ull;
    if (wave_bytes > kLmWaveLds) return false;
    const uint32_t a_row_bytes = (p->la_cap + 2) * sz;
    uint32_t chunk = kLmChunkLds / a_row_bytes;
    if (chunk == 0) return false;
    if (chunk > a->n_rows) chunk = (uint32_t)a->n_rows;
    if ((a->n_rows + chunk - 1) / chunk > 65535) return false;
    p->a_chunk = chunk;
    p->lds = (((size_t)chunk * a_row_bytes + 15u) & ~(size_t)15u) + (((size_t)chunk * 4u + 15u) & ~(size_t)15u) +
             (size_t)kLmWaves * wave_bytes;
    return true;
}

template <int OP, bool BT, typename T, int NQ>
int launch_lm(const LmPlan& p, dim3 grid, const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
              int32_t* out_ids, hipStream_t stream) {
    SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_elem_lanemerge<OP, BT, T, NQ>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    const uint32_t aligned = (reinterpret_cast<uintptr_t>(out_ids) & 15u) == 0 ? 1u : 0u;
    hipLaunchKernelGGL((k_elem_lanemerge<OP, BT, T, NQ>), grid, dim3(kLmBlock), p.lds, stream, a->d_offsets, a->d_ids,
                       (uint32_t)a->n_rows, p.a_chunk, p.la_cap, b->d_offsets, b->d_ids, (uint32_t)b->n_rows, p.lb_cap,
                       p.stage_cap, out_offsets, out_ids, aligned);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <int OP, bool BT, typename T>
int launch_lm_nq(const LmPlan& p, dim3 grid, const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
                 int32_t* out_ids, hipStream_t stream) {
    switch (p.nq) {
    case 4: return launch_lm<OP, BT, T, 4>(p, grid, a, b, out_offsets, out_ids, stream);
    case 10: return launch_lm<OP, BT, T, 10>(p, grid, a, b, out_offsets, out_ids, stream);
    case 16: return launch_lm<OP, BT, T, 16>(p, grid, a, b, out_offsets, out_ids, stream);
    default: return launch_lm<OP, BT, T, 24>(p, grid, a, b, out_offsets, out_ids, stream);
    }
}

template <int OP, bool BT>
int launch_lm_t(const LmPlan& p, dim3 grid, const sdrb_batch* a, const sdrb_batch* b, const uint32_t* out_offsets,
                int32_t* out_ids, hipStream_t stream) {
    if (p.narrow) return launch_lm_nq<OP, BT, uint16_t>(p, grid, a, b, out_offsets, out_ids, stream);
    return launch_lm_nq<OP, BT, int32_t>(p, grid, a, b, out_offsets, out_ids, stream);
}

}  // namespace

bool elem_lanemerge_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const sdrb_batch* self = swap ? b : a;
    if (self->data_kind != SDRB_DATA_EMPTY) return false;   // results would carry payloads
    if (op == SDRB_OP_AND) return false;                    // intersections: the mask-table kernels
    if ((uint64_t)a->n_rows * b->n_rows >= 0xFFFFFFFFull) return false;
    LmPlan p;
    return make_lm_plan(a, b, &p);
}

int launch_elem_lanemerge(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                          int32_t* out_ids, hipStream_t stream) {
    LmPlan p;
    if (!make_lm_plan(a, b, &p)) {
        set_error("lane-per-pair merge fill does not apply to these batches");
        return SDRB_ERR_UNSUPPORTED;
    }
    const uint32_t na = (uint32_t)a->n_rows, nb = (uint32_t)b->n_rows;
    const uint32_t n_tasks = (nb + 63) / 64;
    const uint32_t gy = (na + p.a_chunk - 1) / p.a_chunk;
    // blocks resident per compute unit by their LDS, a few turns of work for each
    const uint32_t per_cu = (uint32_t)((160u * 1024u) / (p.lds + 256));
    uint32_t gx = (n_tasks + kLmWaves - 1) / kLmWaves;
    const uint32_t cap = 256u * (per_cu ? per_cu : 1u) * 2u;
    if ((uint64_t)gx * gy > cap) gx = cap / gy ? cap / gy : 1u;
    const dim3 grid(gx, gy);
    switch (op) {
    case SDRB_OP_OR: return launch_lm_t<SDRB_OP_OR, false>(p, grid, a, b, out_offsets, out_ids, stream);
    case SDRB_OP_XOR: return launch_lm_t<SDRB_OP_XOR, false>(p, grid, a, b, out_offsets, out_ids, stream);
    case SDRB_OP_RM:
        if (swap) return launch_lm_t<SDRB_OP_RM, true>(p, grid, a, b, out_offsets, out_ids, stream);
        return launch_lm_t<SDRB_OP_RM, false>(p, grid, a, b, out_offsets, out_ids, stream);
    default:
        set_error("lane-per-pair merge fill serves ore / xore / rme");
        return SDRB_ERR_UNSUPPORTED;
    }
}

}  // namespace sdrb
