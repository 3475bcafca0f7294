// Size operations for ids WITHOUT a usable width: arbitrary signed ids, 64-bit ids that do not fit 32 bits, widths
// beyond every table kernel.  The reference's merge does not care what the ids are (SDRElem.hpp:15,73-75,
// SDR.hpp:958-997); the table kernels here do, because they index LDS tables by id.  A result size depends only on
// WHICH ids two rows share (and on the payloads at those ids), never on the ids' values, so the rows are given small
// ids first:
//
//   1. the rows of A are cut into chunks of R rows, R * (longest row of A) below the widest table that serves the op;
//   2. k_label_a enters the ids of a chunk into an open-addressing table in global memory whose slots hold POSITIONS
//      in A's id array (a slot is claimed with one compare-and-swap; the key of a claimed slot is the id at that
//      position, which is input data, so no reader ever sees a half-written key).  The label of an id is the position
//      of the first element that claimed it, relative to the chunk: equal ids get equal labels, labels are < R * longest;
//   3. k_label_b looks every id of B up in the finished table: the label, or the chunk's sentinel label (one past the
//      largest) for ids no row of the chunk holds -- a table word that is zero in every kernel;
//   4. the op runs on views of the two batches whose id arrays are the labels and whose declared width is the number
//      of labels: every table kernel applies (bit-sliced, lane-per-row, mask table, position table), payload rules
//      included, since payloads stay where their elements are.  Rows of labels are not ascending and rows of B may
//      hold the sentinel several times; the size kernels only ask which table words an id selects.
//
// B is read and its labels written once per chunk of A (and read again by the op), which is what this path costs over
// a batch with a declared width.  Chunks whose table fits LDS are looked up there (k_label_b_lds).
#include <algorithm>
#include <cstdlib>

#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kEmptySlot = 0xFFFFFFFFu;

template <typename Id>
__device__ __forceinline__ uint32_t slot_of(Id id, uint32_t log_slots) {
    const unsigned long long h = (unsigned long long)(long long)id * 0x9E3779B97F4A7C15ull;
    return (uint32_t)(h >> (64 - log_slots));
}

// elements [e0, e1) of A (the chunk's rows): label[k] = position (relative to e0) of the first element with that id
template <typename Id>
__global__ __launch_bounds__(kBlock) void k_label_a(const uint32_t* __restrict__ a_off, const Id* __restrict__ a_ids,
                                                    uint32_t row0, uint32_t row1, uint32_t* __restrict__ table,
                                                    uint32_t log_slots, int32_t* __restrict__ label) {
    const uint32_t e0 = a_off[row0], e1 = a_off[row1];
    const uint32_t mask = (1u << log_slots) - 1u;
    for (uint32_t k = e0 + blockIdx.x * kBlock + threadIdx.x; k < e1; k += gridDim.x * kBlock) {
        const Id id = a_ids[k];
        uint32_t s = slot_of(id, log_slots);
        for (;;) {
            uint32_t cur = table[s];
            if (cur == kEmptySlot) cur = atomicCAS(&table[s], kEmptySlot, k);
            if (cur == kEmptySlot) {          // claimed: this element stands for its id
                label[k] = (int32_t)(k - e0);
                break;
            }
            if (a_ids[cur] == id) {
                label[k] = (int32_t)(cur - e0);
                break;
            }
            s = (s + 1) & mask;               // the table holds at most half as many ids as slots: an empty one comes
        }
    }
}

// every element of B: the label of its id in the chunk's table, or `absent`
template <typename Id>
__global__ __launch_bounds__(kBlock) void k_label_b(const Id* __restrict__ b_ids, uint64_t nnz_b, const uint32_t* __restrict__ a_off,
                                                    const Id* __restrict__ a_ids, uint32_t row0, const uint32_t* __restrict__ table,
                                                    uint32_t log_slots, int32_t absent, int32_t* __restrict__ label) {
    const uint32_t e0 = a_off[row0];
    const uint32_t mask = (1u << log_slots) - 1u;
    for (uint64_t k = (uint64_t)blockIdx.x * kBlock + threadIdx.x; k < nnz_b; k += (uint64_t)gridDim.x * kBlock) {
        const Id id = b_ids[k];
        uint32_t s = slot_of(id, log_slots);
        int32_t found = absent;
        for (;;) {
            const uint32_t cur = table[s];
            if (cur == kEmptySlot) break;
            if (a_ids[cur] == id) {
                found = (int32_t)(cur - e0);
                break;
            }
            s = (s + 1) & mask;
        }
        label[k] = found;
    }
}

// The same look-up with the chunk's table in LDS: a block copies the table once -- the KEY of every claimed slot (the id
// at the position the slot holds) and its label as 16 bits, 0xFFFF for an empty slot -- and then streams its share of B
// with 16-byte loads and stores, so that labelling B costs about one read and one write of it; the probes never leave
// the compute unit.  For tables of up to kLdsTableBytes (6 bytes a slot for 32-bit ids, 10 for 64-bit ids).
constexpr int kLdsBlock = 1024;
constexpr size_t kLdsTableBytes = 100 * 1024;
constexpr uint32_t kNoLabel = 0xFFFFu;

template <typename Id>
__global__ __launch_bounds__(kLdsBlock) void k_label_b_lds(const Id* __restrict__ b_ids, uint64_t nnz_b,
                                                            const uint32_t* __restrict__ a_off, const Id* __restrict__ a_ids,
                                                            uint32_t row0, const uint32_t* __restrict__ table, uint32_t log_slots,
                                                            int32_t absent, int32_t* __restrict__ label) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_table[];
    const uint32_t slots = 1u << log_slots, mask = slots - 1u;
    Id* key = reinterpret_cast<Id*>(lds_table);
    uint16_t* lab = reinterpret_cast<uint16_t*>(key + slots);
    const uint32_t e0 = a_off[row0];
    for (uint32_t s = threadIdx.x; s < slots; s += kLdsBlock) {
        const uint32_t cur = table[s];
        Id k = 0;
        uint32_t l = kNoLabel;
        if (cur != kEmptySlot) {
            k = a_ids[cur];
            l = cur - e0;
        }
        key[s] = k;
        lab[s] = (uint16_t)l;
    }
    __syncthreads();
    auto label_of = [&](Id id) -> int32_t {
        uint32_t s = slot_of(id, log_slots);
        for (;;) {
            const uint32_t l = lab[s];
            if (l == kNoLabel) return absent;
            if (key[s] == id) return (int32_t)l;
            s = (s + 1) & mask;
        }
    };
    // four ids a lane and turn; the array's last ids (fewer than four) one by one
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    const uint64_t quads = nnz_b / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * kLdsBlock + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * kLdsBlock) {
        Id ids[4];
        if constexpr (sizeof(Id) == 4) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(b_ids + 4 * q);
            ids[0] = v[0], ids[1] = v[1], ids[2] = v[2], ids[3] = v[3];
        } else {
            typedef long long i64x2 __attribute__((ext_vector_type(2)));
            const i64x2 lo = *reinterpret_cast<const i64x2*>(b_ids + 4 * q), hi = *reinterpret_cast<const i64x2*>(b_ids + 4 * q + 2);
            ids[0] = lo[0], ids[1] = lo[1], ids[2] = hi[0], ids[3] = hi[1];
        }
        i32x4 out;
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = label_of(ids[c]);
        *reinterpret_cast<i32x4*>(label + 4 * q) = out;
    }
    if (blockIdx.x == 0 && threadIdx.x < (nnz_b & 3u)) {
        const uint64_t k = quads * 4 + threadIdx.x;
        label[k] = label_of(b_ids[k]);
    }
}

// the widest table a size op of these batches can use, by the kernels' own limits (see their *_applicable)
uint32_t widest_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const sdrb_batch* self = swap ? b : a;
    const bool id_only = op == SDRB_OP_OR || self->data_kind == SDRB_DATA_EMPTY;
    if (b->max_row_len <= 64) return 4096;   // the bit-sliced kernel (ids alone decide), the lane-per-row kernels (payloads)
    if (id_only) return 32768;               // the bit-sliced kernel for long rows (its LDS table ends near 40 000 ids)
    return 16384;                            // the mask-table kernel (its LDS table ends near 17 800 ids)
}

template <typename Id>
int run(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out, uint64_t ld_out, hipStream_t s) {
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    const uint32_t longest = a->max_row_len ? a->max_row_len : 1u;
    const uint32_t widest = widest_table(a, b, op, swap);
    uint64_t chunk_rows = (widest - 1) / longest;
    // A table that fits LDS (k_label_b_lds) labels B about 3.5 times faster than one in global memory, and a pass of the
    // op over B costs about 1.3 such label passes: smaller chunks -- more passes, each cheaper to prepare -- are taken
    // where that sum is lower.  Config 4's shape (64 x 128 Ki rows of 328 ids): `ands` 2 chunks of 49 rows 0.87 ms, 3 of
    // 24 rows 0.68 ms; `ors`, whose table takes all 64 rows at once, stays at one chunk (0.48 against 0.57 ms).
    static const char* cap_env = getenv("SDRB_RELABEL_CHUNK_LABELS");
    const uint64_t lds_labels = cap_env ? strtoull(cap_env, nullptr, 10) : (sizeof(Id) == 4 ? 8191u : 4095u);
    if (chunk_rows * longest > lds_labels && lds_labels / longest >= 8 && b->nnz >= (1u << 20)) {
        const uint64_t rows_lds = lds_labels / longest;
        const uint64_t n_global = (a->n_rows + chunk_rows - 1) / chunk_rows, n_lds = (a->n_rows + rows_lds - 1) / rows_lds;
        if (23 * n_lds < 48 * n_global) chunk_rows = rows_lds;
    }
    if (chunk_rows >= 64) chunk_rows &= ~63ull;          // whole tiles of the table kernels
    if (chunk_rows > a->n_rows) chunk_rows = a->n_rows;
    const uint32_t labels = (uint32_t)(chunk_rows * longest);          // labels of a chunk lie in [0, labels); sentinel = labels
    uint32_t log_slots = 10;
    while ((1u << log_slots) < 2u * labels) ++log_slots;
    const size_t table_bytes = (size_t)4 << log_slots;
    const size_t la_bytes = ((size_t)a->nnz * 4 + 255) & ~(size_t)255, lb_bytes = ((size_t)b->nnz * 4 + 255) & ~(size_t)255;
    char* ws = static_cast<char*>(workspace(b->device, s, 13, table_bytes + la_bytes + lb_bytes));
    if (!ws) return SDRB_ERR_HIP;
    uint32_t* table = reinterpret_cast<uint32_t*>(ws);
    int32_t* label_a = reinterpret_cast<int32_t*>(ws + table_bytes);
    int32_t* label_b = reinterpret_cast<int32_t*>(ws + table_bytes + la_bytes);
    const Id* a_ids = reinterpret_cast<const Id*>(a->d_ids);
    const Id* b_ids = reinterpret_cast<const Id*>(b->d_ids);

    sdrb_batch va = *a, vb = *b;
    for (sdrb_batch* v : {&va, &vb}) {
        v->id_bits = 32;
        v->width = (int32_t)labels + 1;
        v->min_id = 0;
        v->max_id = (int32_t)labels;
        v->owns_storage = false;
        v->narrow = nullptr;
        v->d_narrow_ids = nullptr;
        v->derived_home = nullptr;
        v->d_bitplanes = nullptr;
        v->d_dense8 = nullptr;
        v->bitplanes_ready = v->dense_ready = nullptr;
    }
    va.d_ids = label_a;
    vb.d_ids = label_b;
    int rc = SDRB_OK;
    for (uint64_t row0 = 0; row0 < a->n_rows && rc == SDRB_OK; row0 += chunk_rows) {
        const uint64_t row1 = std::min<uint64_t>(a->n_rows, row0 + chunk_rows);
        SDRB_HIP(hipMemsetAsync(table, 0xFF, table_bytes, s));
        const uint32_t ga = (uint32_t)std::min<uint64_t>(1024, ((row1 - row0) * longest + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((k_label_a<Id>), dim3(ga ? ga : 1), dim3(kBlock), 0, s, a->d_offsets, a_ids, (uint32_t)row0, (uint32_t)row1,
                           table, log_slots, label_a);
        SDRB_HIP(hipGetLastError());
        const size_t lds_table = ((size_t)sizeof(Id) + 2) << log_slots;
        static const bool no_lds = getenv("SDRB_RELABEL_NO_LDS") != nullptr;   // measurements of the table in global memory
        if (!no_lds && lds_table <= kLdsTableBytes && labels < kNoLabel && (reinterpret_cast<uintptr_t>(b_ids) & 15u) == 0) {
            SDRB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_label_b_lds<Id>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_table));
            const uint32_t gb = (uint32_t)std::min<uint64_t>(256, (b->nnz / 4 + kLdsBlock - 1) / kLdsBlock);
            hipLaunchKernelGGL((k_label_b_lds<Id>), dim3(gb ? gb : 1), dim3(kLdsBlock), lds_table, s, b_ids, b->nnz, a->d_offsets, a_ids,
                               (uint32_t)row0, table, log_slots, (int32_t)labels, label_b);
        } else {
            const uint32_t gb = (uint32_t)std::min<uint64_t>(8192, (b->nnz + kBlock * 4 - 1) / (kBlock * 4));
            hipLaunchKernelGGL((k_label_b<Id>), dim3(gb ? gb : 1), dim3(kBlock), 0, s, b_ids, b->nnz, a->d_offsets, a_ids, (uint32_t)row0, table,
                               log_slots, (int32_t)labels, label_b);
        }
        SDRB_HIP(hipGetLastError());
        // the chunk's rows as a batch of their own: offsets stay absolute (they index label_a as they indexed A's ids)
        va.d_offsets = a->d_offsets + row0;
        va.n_rows = row1 - row0;
        rc = sdrb_size_op(&va, &vb, op, (flags & SDRB_FLAG_SWAP) | kFlagLabelled, out + row0 * ld_out, ld_out, s);
        // a form of B derived inside the call (bit planes of the labels) belongs to this chunk's labels only
        if (vb.d_bitplanes || vb.d_dense8 || va.d_bitplanes || va.d_dense8) {
            (void)hipStreamSynchronize(s);
            for (sdrb_batch* v : {&va, &vb}) {
                if (v->d_bitplanes) (void)hipFree(v->d_bitplanes);
                if (v->d_dense8) (void)hipFree(v->d_dense8);
                if (v->bitplanes_ready) (void)hipEventDestroy(v->bitplanes_ready);
                if (v->dense_ready) (void)hipEventDestroy(v->dense_ready);
                v->d_bitplanes = nullptr;
                v->d_dense8 = nullptr;
                v->bitplanes_ready = v->dense_ready = nullptr;
            }
        }
    }
    return rc;
}

}  // namespace

// All-pairs evaluation of batches whose ids no table kernel takes as they are; rows of A must be short enough for a
// chunk of at least one row to fit the widest table.
bool size_relabel_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    if (a->id_bits != b->id_bits || a->n_rows == 0 || b->n_rows == 0 || a->nnz == 0 || b->nnz == 0) return false;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) return false;   // sizes of ArithData rows go through id views first
    if (b->max_row_len >= 65536) return false;
    return (uint64_t)a->max_row_len <= widest_table(a, b, op, swap) - 1;
}

int launch_size_relabel(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out, uint64_t ld_out,
                        hipStream_t stream) {
    if (a->id_bits == 64) return run<long long>(a, b, op, flags, out, ld_out, stream);
    return run<int32_t>(a, b, op, flags, out, ld_out, stream);
}

}  // namespace sdrb
