// Internal declarations shared by the C-ABI layer and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <exception>
#include <mutex>
#include <new>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include <rocprofiler-sdk-roctx/roctx.h>

#include "../../include/sdr_amd.h"

struct sdrb_batch {
    int device = 0;
    int32_t width = 0;     // what the kernels take as the width: the declared one, or, for a batch created without one,
                           // largest id + 1 once rows without a negative id have been uploaded (settle_width); else 0
    bool width_declared = false;   // `width` was given at creation (ids are checked against it, callers see it)
    int data_kind = SDRB_DATA_EMPTY;
    int id_bits = 32;      // 64: d_ids points at int64 ids (min_id/max_id unused); see `narrow`
    uint64_t n_rows = 0;
    uint64_t nnz = 0;
    uint32_t max_row_len = 0;
    int32_t min_id = 0, max_id = -1;  // over all rows (max_id < min_id when empty)
    int64_t min_id64 = 0, max_id64 = -1;   // a batch of 64-bit ids: their span, where an upload has looked at them (else max < min)
    uint32_t* d_offsets = nullptr;    // [n_rows + 1]
    int32_t* d_ids = nullptr;         // [nnz]
    float* d_data = nullptr;          // [nnz] or null
    bool owns_storage = true;            // false: d_offsets/d_ids/d_data are views into caller memory
    uint64_t cap_rows = 0, cap_nnz = 0;  // allocated capacity (rows, elements); storage is reused across uploads
    uint32_t* d_bitplanes = nullptr;  // derived, built on demand: word w of row j at [w*n_rows + j]
    uint8_t* d_dense8 = nullptr;      // derived, built on demand: 0/1 bytes in MFMA fragment order (allpairs_mfma.hip)
    uint32_t dense_chunks = 0;        // width of d_dense8 in 64-id chunks
    // A derived form is built on the stream of the call that first needs it; the event marks the end of that
    // build, and a call on any other stream waits for it before its kernels read the form.
    hipEvent_t bitplanes_ready = nullptr, dense_ready = nullptr;
    hipStream_t bitplanes_stream = nullptr, dense_stream = nullptr;
    // A temporary view of another batch (the same rows seen without payloads) keeps no derived forms of its own:
    // they are built in, and read from, the batch it views.
    sdrb_batch* derived_home = nullptr;
    // A batch of 64-bit ids whose ids all fit 32 bits also keeps them narrowed: `narrow` is a 32-bit batch that
    // views this batch's offsets and payloads and owns only d_narrow_ids (declared width = largest id + 1 when no
    // id is negative), so that two such batches are served by every kernel the 32-bit path has.
    sdrb_batch* narrow = nullptr;
    int32_t* d_narrow_ids = nullptr;
};

namespace sdrb {

// A batch created without a width whose ids are all non-negative is served as if largest id + 1 had been declared:
// every table kernel then applies to it (ids of the other batch beyond that width never match, as with any declared
// width).  Called by the uploads that have looked at every id; a negative id leaves the width at 0.
inline void settle_width(sdrb_batch* b) {
    if (b->width_declared) return;
    b->width = (b->nnz != 0 && b->min_id >= 0 && b->max_id < 0x7FFFFFFF) ? b->max_id + 1 : 0;
}
inline int32_t declared_width(const sdrb_batch* b) { return b->width_declared ? b->width : 0; }

// does the batch carry a float per element (UnitData or ArithData<float>)?
inline bool has_payload(const sdrb_batch* b) { return b->data_kind != SDRB_DATA_EMPTY; }

void set_error(const char* fmt, ...);

// A named range around the host side of an entry point, for `rocprofv3 --marker-trace`: the kernels an entry
// point enqueues show up under its name.  Without a profiler attached the two calls do nothing.
struct TraceRange {
    explicit TraceRange(const char* name) { (void)roctxRangePushA(name); }
    ~TraceRange() { (void)roctxRangePop(); }
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
};

// makes `device` current for the scope and restores the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Runs `body` (a callable returning a status code) so that nothing is thrown through the C ABI: a failed
// host allocation or any other C++ exception becomes a status code and a message.
template <typename F>
int guarded(const char* what, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        set_error("%s: out of host memory", what);
        return SDRB_ERR_NO_MEMORY;
    } catch (const std::exception& e) {
        set_error("%s: %s", what, e.what());
        return SDRB_ERR_INVALID_ARG;
    } catch (...) {
        set_error("%s: unexpected exception", what);
        return SDRB_ERR_INVALID_ARG;
    }
}

// Scratch per (device, stream, slot) that grows on demand and lives until process exit: calls on different
// streams never share a block, and a block is reused only by later work on the same stream.
// slot 0: query bit-table; slot 1: scan block sums; slot 2: pack statistics; slot 3: bitsets of A;
// slot 4: partial top-k lists; slot 5: interleaved value table of A (dot products);
// slot 6: dense payload matrix of A (mask-table kernel); slot 7: staged range / point queries;
// slot 8: sort buffers of the transpose; slot 9: rank tables of A (wave-per-pair element fill).
// *fresh (optional) tells whether the block has just been allocated, i.e. holds nothing of an earlier call.
void* workspace(int device, hipStream_t stream, int slot, size_t bytes, bool* fresh = nullptr);

// capi.cpp: room for n_rows rows / nnz elements in the batch (the current allocation is kept when it is large
// enough; derived forms of the previous content are dropped).  The batch's device must be current.
int reserve_batch_storage(sdrb_batch* b, uint64_t n_rows, uint64_t nnz);
// the batch holds no rows any more (its storage stays allocated for the next upload)
void clear_batch_rows(sdrb_batch* b);

// ingest.cpp: the checks of an upload on host CSR (offsets monotone, rows strictly ascending, ids inside `width`
// when it is > 0, payloads in range), by a team of threads when the arrays are large; row_offsets[0] == 0 and
// non-null arrays are the caller's business.  Status code and message of the first finding.
int check_host_csr(uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids, const float* data, int32_t width,
                   int data_kind);

#define SDRB_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            sdrb::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),       \
                            __FILE__, __LINE__);                                         \
            return SDRB_ERR_HIP;                                                         \
        }                                                                                \
    } while (0)

// the batch whose handle caches the derived forms of b's rows (b itself unless b is a temporary view)
inline sdrb_batch* derived_owner(const sdrb_batch* b) {
    return b->derived_home ? b->derived_home : const_cast<sdrb_batch*>(b);
}
// Handles may be read by any number of threads at once, and derived forms are built lazily inside such reads:
// check, build and publication of a derived form happen under this lock (defined in capi.cpp), the pointer is
// published after the event that marks the end of the build has been recorded.
std::mutex& derived_mutex();

// after the build of a derived form has been enqueued on `stream`
inline int derived_built(hipEvent_t* ready, hipStream_t* built_on, hipStream_t stream) {
    if (!*ready) SDRB_HIP(hipEventCreateWithFlags(ready, hipEventDisableTiming));
    SDRB_HIP(hipEventRecord(*ready, stream));
    *built_on = stream;
    return SDRB_OK;
}
// before kernels on `stream` read a derived form that another stream may still be building
inline int derived_wait(hipEvent_t ready, hipStream_t built_on, hipStream_t stream) {
    if (ready && built_on != stream) SDRB_HIP(hipStreamWaitEvent(stream, ready, 0));
    return SDRB_OK;
}

// How a pair's result depends on payloads, after SDRB_FLAG_SWAP has been applied:
// `this_unit` / `arg_unit` say whether the `this` / `arg` side carries UnitData.
struct PairKinds {
    bool this_unit;
    bool arg_unit;
};

// size_bitslice.hip ---------------------------------------------------------------
// True when the id-only bit-sliced kernel can serve (a as tile side, b as lane side).
bool bitslice_applicable(const sdrb_batch* a, const sdrb_batch* b);
int launch_size_bitslice(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                         uint32_t* out, uint64_t ld_out, hipStream_t stream);

// size_bitslice_long.hip: the same formulation for rows of B beyond 64 ids and widths beyond 4096 (to about 40 000)
bool bitslice_long_applicable(const sdrb_batch* a, const sdrb_batch* b);
int launch_size_bitslice_long(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                              hipStream_t stream);

// The same kernel writing counts as bytes (out_bytes 1) or 16-bit words (2): out[i * ld_out + j], ld_out a multiple of 4.
// size_op_bound: the largest value the op can yield for these batches (from the longest rows).
uint64_t size_op_bound(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap);
int launch_size_bitslice_narrow(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, void* out, int out_bytes,
                                uint64_t ld_out, hipStream_t stream);

// The same kernels with the bit table in a file-scope __device__ array instead of the growing workspace
// (no hipMalloc anywhere on the path); static_scratch_resolve looks the array's address up once per device,
// with that device current.
int static_scratch_resolve(int device, void** ptr);
int launch_size_bitslice_static(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out,
                                uint64_t ld_out, hipStream_t stream);

// relabel.hip ---------------------------------------------------------------------
// Size ops of batches whose ids no table kernel takes as they are (no declared width, a width beyond every table,
// 64-bit ids): the ids are replaced by small labels, chunk of A by chunk, and the op runs on the labelled views.
// kFlagLabelled marks the inner calls (never set by callers): such a call does not label again.
constexpr unsigned kFlagLabelled = 1u << 30;
bool size_relabel_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap);
int launch_size_relabel(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out, uint64_t ld_out,
                        hipStream_t stream);

// size_merge.hip ------------------------------------------------------------------
// rowwise: result j = A_j.op(B_j) for two batches of equal length, out[j]
// (both batches hold ids of the same width: 32 or 64 bits)
int launch_size_merge(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                      uint32_t* out, uint64_t ld_out, bool rowwise, hipStream_t stream);

// SDR::operator== for every pair (or row-wise): out = 1 / 0
int launch_equal(const sdrb_batch* a, const sdrb_batch* b, uint32_t* out, uint64_t ld_out, bool rowwise,
                 hipStream_t stream);

// elem_ops.hip --------------------------------------------------------------------
// out[k] = in[k] widened to 64 bits (results of the 32-bit kernels for batches of 64-bit ids)
int launch_widen_ids(const int32_t* in, int64_t* out, uint64_t n, hipStream_t stream);
int launch_narrow_ids(const int64_t* in, int32_t* out, uint64_t n, hipStream_t stream);
int launch_shift_ids(void* ids, int id_bits, uint64_t n, int64_t amount, hipStream_t stream);
// out[k] = (int32)(in[k] - base): ids moved to start at zero (in: 32- or 64-bit ids whose distance from base fits 31 bits)
int launch_rebase_ids(const void* in, int id_bits, int32_t* out, uint64_t n, int64_t base, hipStream_t stream);
// ids[k] += base for k < *n_dev (n_dev: device memory -- the last offset of a result); 64-bit: from the 32-bit ids in `in`
int launch_unbase_ids32(int32_t* ids, const uint32_t* n_dev, int32_t base, hipStream_t stream);
int launch_unbase_ids64(const int32_t* in, int64_t* out, const uint32_t* n_dev, int64_t base, hipStream_t stream);
// In-place exclusive prefix sum of data[0..n) in one pass; entries at index >= n_valid count as zero (so a
// scan over n = pairs + 1, n_valid = pairs leaves the total in data[pairs] without clearing it first).
// Offsets are kept modulo 2^32; the exact 64-bit sum goes to *total_dev (device memory, may be null).
int launch_exclusive_scan_u32(int device, uint32_t* data, uint64_t n, uint64_t n_valid,
                              unsigned long long* total_dev, hipStream_t stream);
// out_ids: int32 ids, or int64 ids when the batches hold 64-bit ids
int launch_elem_fill(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                     const uint32_t* out_offsets, void* out_ids, float* out_data, bool rowwise,
                     hipStream_t stream);

// visitor positions of andv (SDRB_OP_AND) / orv (SDRB_OP_OR) by the per-pair merge, any id width
int launch_elem_visit(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                      uint32_t* this_pos, uint32_t* arg_pos, bool rowwise, hipStream_t stream);

// elem_search.hip ------------------------------------------------------------------
// Wave-per-pair element fill whose ranks are binary searches over the other row in LDS: ids of any value, 32 or 64
// bits wide (both batches alike), all-pairs evaluation, rows short enough to stage.  out_ids holds ids of the batches'
// width; visit: out_ids = positions in `this` (uint32), out_data = positions in `arg` (uint32), 0xFFFFFFFF where absent.
bool elem_search_applicable(const sdrb_batch* a, const sdrb_batch* b);
int launch_elem_search(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets, void* out_ids,
                       float* out_data, hipStream_t stream, bool visit = false);

// elem_bitmap.hip ------------------------------------------------------------------
// Dense-bitset fill for id-only results (`this` side EmptyData, declared width <= 4096).
bool elem_bitmap_applicable(const sdrb_batch* a, const sdrb_batch* b, bool swap);
int launch_elem_bitmap(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                       const uint32_t* out_offsets, int32_t* out_ids, hipStream_t stream);

// elem_lanemerge.hip ---------------------------------------------------------------
// ore / xore / rme with id-only results on short rows: a lane per pair, outputs staged in LDS per wave.
bool elem_lanemerge_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap);
int launch_elem_lanemerge(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                          int32_t* out_ids, hipStream_t stream);

// elem_lanerank.hip ----------------------------------------------------------------
// the same results for a declared width <= 4096 and rows <= 64 ids: a lane per pair, places from rank tables
bool elem_lanerank_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap);
int launch_elem_lanerank(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                         int32_t* out_ids, hipStream_t stream);

// elem_rank.hip --------------------------------------------------------------------
// Wave-per-pair fill from rank tables (bitset + popcount prefix per row): any payload kinds, all four
// ops, declared width <= 32768.
bool elem_rank_applicable(const sdrb_batch* a, const sdrb_batch* b);
// visit: write visitor positions instead of elements (op AND / OR; out_ids = this positions, out_data =
// arg positions as uint32)
int launch_elem_rank(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                     int32_t* out_ids, float* out_data, hipStream_t stream, bool visit = false);

// size_popcount.hip ------------------------------------------------------------------
// Dense-bitset popcount size ops (id-only results, declared width <= 2048).
bool popcount_applicable(const sdrb_batch* a, const sdrb_batch* b);
int launch_size_popcount(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                         uint32_t* out, uint64_t ld_out, hipStream_t stream);

// allpairs_mfma.hip, allpairs_bitslice.hip ------------------------------------------------
// For every row i of a: the k rows j of b with the largest id overlap (count desc, row asc).
// Both kernels keep candidates as packed keys  count << 19 | (2^19 - 1 - local j)  (0 = empty,
// so a valid key is > 0), in sorted lists of kTopkList entries; the rows of b are cut into
// `splits` contiguous ranges of at most kTopkMaxSplitRows rows so that local j fits the key.
constexpr int kTopkList = 16;
constexpr int kTopkKeyShift = 19;
constexpr int kTopkLowMask = (1 << kTopkKeyShift) - 1;
constexpr uint32_t kTopkMaxSplitRows = (uint32_t)kTopkLowMask - 1;
bool allpairs_mfma_applicable(const sdrb_batch* a, const sdrb_batch* b, uint32_t k);
int launch_allpairs_mfma(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                         uint32_t* out_counts, int32_t* out_rows, hipStream_t stream);
bool allpairs_bitslice_applicable(const sdrb_batch* a, const sdrb_batch* b, uint32_t k);
int launch_allpairs_bitslice(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                             uint32_t* out_counts, int32_t* out_rows, hipStream_t stream);
// partial: [row i][split][list][kTopkList] keys -> the best k per row (count, row), (0, -1) padded
int launch_merge_topk(const int* partial, uint32_t na, uint32_t splits, uint32_t lists_per_split,
                      uint64_t rows_per_split, uint32_t k, uint32_t* out_counts, int32_t* out_rows,
                      hipStream_t stream);

// few_ande.hip: ande of at most 32 rows of a with every row of b, offsets and elements in one enqueue (bytes per pair,
// a scan of per-group sums, then the fill); elements beyond `capacity` are not written
bool few_ande_applicable(const sdrb_batch* a, const sdrb_batch* b, bool swap);
int launch_few_ande(const sdrb_batch* a, const sdrb_batch* b, bool swap, uint32_t* out_offsets, int32_t* out_ids,
                    float* out_data, uint64_t capacity, unsigned long long* total_dev, hipStream_t stream);

// mask_table.hip ---------------------------------------------------------------------------
// Size ops with a 64-row query-mask table in LDS: B is read once per 64 rows of A (width <= ~17800).
bool mask_table_applicable(const sdrb_batch* a, const sdrb_batch* b);
int launch_size_mask_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                           uint32_t* out, uint64_t ld_out, hipStream_t stream);
int launch_elem_mask_table(const sdrb_batch* a, const sdrb_batch* b, bool swap, const uint32_t* out_offsets,
                           int32_t* out_ids, float* out_data, hipStream_t stream);

// range_query.hip -------------------------------------------------------------------------
// mode 0: counts of ids in [lo, hi); 1: fill those elements at out_off; 2: positions of the ids `lo` (-1: absent).
// lo / hi are HOST arrays of n_queries entries.
int launch_range_query(const sdrb_batch* b, int mode, uint64_t n_queries, const int32_t* lo, const int32_t* hi,
                       uint32_t* out, uint64_t ld_out, const uint32_t* out_off, int32_t* out_ids, float* out_data,
                       hipStream_t stream);

// transpose.hip ---------------------------------------------------------------------------
// dst_off[width + 1], dst_ids[nnz] (row numbers of src), dst_data[nnz]: the batch regrouped by id.
int launch_transpose(const sdrb_batch* src, uint32_t* dst_off, int32_t* dst_ids, float* dst_data, hipStream_t stream);

// inner.hip --------------------------------------------------------------------------------
// Sparse dot products of ArithData<float> rows for every pair (SDR::inner, SDR.hpp:1682-1692).
int launch_inner(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, float* out, uint64_t ld_out,
                 hipStream_t stream);
// lane_rows.hip: the same numbers by visiting id matches only (32-row mask tiles of A in LDS, a lane per row of B),
// and the payload-dependent size operations by the same walk
bool inner_sparse_applicable(const sdrb_batch* a, const sdrb_batch* b, bool forced);
int launch_inner_sparse(const sdrb_batch* a, const sdrb_batch* b, float* out, uint64_t ld_out, hipStream_t stream);
bool size_lane_rows_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, bool forced);
int launch_size_lane_rows(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out, uint64_t ld_out,
                          hipStream_t stream);

// unit_table.hip --------------------------------------------------------------------
// Wave-per-row kernels with an LDS position table of one row of A (declared width <= 32768).
bool unit_table_applicable(const sdrb_batch* a, const sdrb_batch* b);
bool unit_table_fill_applicable(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap);
int launch_size_unit_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                           uint32_t* out, uint64_t ld_out, hipStream_t stream);
int launch_elem_unit_table(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                           const uint32_t* out_offsets, int32_t* out_ids, float* out_data,
                           hipStream_t stream);

}  // namespace sdrb
