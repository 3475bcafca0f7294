// extern "C" boundary of the engine: batch packing into HBM, kernel selection, memory
// helpers.  No exceptions cross this layer; every entry point returns a status code.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "sdrb_internal.h"

namespace sdrb {

namespace {

thread_local std::string g_error;

// Scratch blocks are keyed by (device, stream, slot): work enqueued on different streams -- or by different
// host threads, each on its own stream -- never shares a block.  A block that has become too small is retired
// behind an event on its stream and freed once that event has completed, so growth synchronises nothing.
struct Scratch {
    void* ptr = nullptr;
    size_t bytes = 0;
};
struct Retired {
    int device;
    void* ptr;
    hipEvent_t done;
};
struct ScratchKey {
    int device;
    hipStream_t stream;
    int slot;
    bool operator<(const ScratchKey& o) const {
        if (device != o.device) return device < o.device;
        if (stream != o.stream) return stream < o.stream;
        return slot < o.slot;
    }
};
std::mutex g_ws_mutex;
std::map<ScratchKey, Scratch> g_ws;
std::vector<Retired> g_retired;

// frees the retired blocks of `device` (the current device) whose last users have finished; g_ws_mutex held
void reap_retired(int device) {
    for (size_t k = 0; k < g_retired.size();) {
        Retired& r = g_retired[k];
        if (r.device == device && hipEventQuery(r.done) == hipSuccess) {
            (void)hipEventDestroy(r.done);
            (void)hipFree(r.ptr);
            g_retired[k] = g_retired.back();
            g_retired.pop_back();
        } else {
            ++k;
        }
    }
    (void)hipGetLastError();   // hipErrorNotReady of a pending event is not an error
}

void drop_derived(sdrb_batch* b);

// the narrowed twin of a batch of 64-bit ids (a view: it owns nothing but its derived forms)
void drop_narrow(sdrb_batch* b) {
    if (b->narrow) {
        drop_derived(b->narrow);
        delete b->narrow;
        b->narrow = nullptr;
    }
    if (b->d_narrow_ids) (void)hipFree(b->d_narrow_ids);
    b->d_narrow_ids = nullptr;
}

// Data derived from the rows (bit planes, dense byte matrix) dies with the content it was built from.
void drop_tables(sdrb_batch* b);
void drop_derived(sdrb_batch* b) {
    drop_narrow(b);
    drop_tables(b);
}
void drop_tables(sdrb_batch* b) {
    if (b->d_bitplanes) (void)hipFree(b->d_bitplanes);
    if (b->d_dense8) (void)hipFree(b->d_dense8);
    b->d_bitplanes = nullptr;
    b->d_dense8 = nullptr;
    if (b->bitplanes_ready) (void)hipEventDestroy(b->bitplanes_ready);
    if (b->dense_ready) (void)hipEventDestroy(b->dense_ready);
    b->bitplanes_ready = b->dense_ready = nullptr;
}

void free_batch_storage(sdrb_batch* b) {
    if (b->owns_storage) {
        if (b->d_offsets) (void)hipFree(b->d_offsets);
        if (b->d_ids) (void)hipFree(b->d_ids);
        if (b->d_data) (void)hipFree(b->d_data);
    }
    b->owns_storage = true;
    drop_derived(b);
    b->d_offsets = nullptr;
    b->d_ids = nullptr;
    b->d_data = nullptr;
    b->cap_rows = b->cap_nnz = 0;
    b->n_rows = 0;
    b->nnz = 0;
    b->max_row_len = 0;
    b->min_id = 0;
    b->max_id = -1;
    b->min_id64 = 0;
    b->max_id64 = -1;
    settle_width(b);
}

// Makes room for n_rows rows / nnz elements, keeping the current allocation when it is large
// enough (repeated uploads of similar batches, e.g. one query batch per step, then cost no
// hipMalloc/hipFree).  Derived data of the previous content is dropped.
int reserve_storage(sdrb_batch* b, uint64_t n_rows, uint64_t nnz) {
    const bool unit = has_payload(b);
    drop_derived(b);
    {
        // set-up on this device (it is current here): look up where the static scratch of sdrb_size_op_static
        // lives, so that entry point's own first call need not (it retries and reports if this failed)
        void* unused = nullptr;
        (void)static_scratch_resolve(b->device, &unused);
    }
    if (b->owns_storage && b->d_offsets && b->cap_rows >= n_rows && b->cap_nnz >= nnz) return SDRB_OK;
    free_batch_storage(b);
    SDRB_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_offsets), (n_rows + 1) * sizeof(uint32_t)));
    if (nnz) SDRB_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_ids), nnz * (b->id_bits == 64 ? 8u : 4u)));
    if (nnz && unit) SDRB_HIP(hipMalloc(reinterpret_cast<void**>(&b->d_data), nnz * sizeof(float)));
    b->cap_rows = n_rows;
    b->cap_nnz = nnz;
    return SDRB_OK;
}

// Device-side validation of a CSR pack that is already in HBM (one thread per row).
struct PackStats {
    uint32_t max_len;
    int32_t min_id;
    int32_t max_id;
    uint32_t bad_offsets;   // 1 + first row whose offsets are not monotone / exceed nnz
    uint32_t bad_order;     // 1 + a row that is not strictly ascending
    uint32_t bad_payload;   // 1 + an element whose payload is outside [0,1] (UnitData) or not finite (ArithData)
    uint32_t nnz;           // row_offsets[n_rows]
};

__global__ __launch_bounds__(256) void k_validate_pack(const uint32_t* __restrict__ off,
                                                       const int32_t* __restrict__ ids,
                                                       const float* __restrict__ data, int data_kind,
                                                       uint64_t n_rows, PackStats* __restrict__ st) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t nnz = off[n_rows];
    if (r == 0) st->nnz = nnz;
    if (r >= n_rows) return;
    const uint32_t lo = off[r], hi = off[r + 1];
    if (hi < lo || hi > nnz || (r == 0 && lo != 0)) {
        atomicMin(&st->bad_offsets, (uint32_t)(r < 0xFFFFFFFEull ? r + 1 : 0xFFFFFFFFu));
        return;
    }
    if (hi == lo) return;
    atomicMax(&st->max_len, hi - lo);
    atomicMin(&st->min_id, ids[lo]);
    atomicMax(&st->max_id, ids[hi - 1]);
    bool order_ok = true, payload_ok = true;
    for (uint32_t k = lo; k < hi; ++k) {
        if (k > lo && ids[k] <= ids[k - 1]) order_ok = false;
        if (data_kind == SDRB_DATA_UNIT && !(data[k] >= 0.0f && data[k] <= 1.0f)) payload_ok = false;
        if (data_kind == SDRB_DATA_ARITH && !(fabsf(data[k]) <= 3.402823466e38f)) payload_ok = false;
    }
    if (!order_ok) atomicMin(&st->bad_order, (uint32_t)(r + 1));
    if (!payload_ok) atomicMin(&st->bad_payload, (uint32_t)(r + 1));
}

bool valid_op(int op) { return op >= SDRB_OP_AND && op <= SDRB_OP_RM; }

// Do the ids alone decide the result of a size op?  `this` EmptyData ignores every payload
// (EmptyData.hpp:44-80) and ors never looks at payloads (SDR.hpp:1298-1301).
bool id_only(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap) {
    const sdrb_batch* self = swap ? b : a;
    return op == SDRB_OP_OR || self->data_kind == SDRB_DATA_EMPTY;
}

int check_pair(const sdrb_batch* a, const sdrb_batch* b, int op, bool widths_may_differ = false) {
    if (!a || !b) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (!valid_op(op)) {
        set_error("unknown op %d", op);
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->device != b->device) {
        set_error("batches live on different devices (%d vs %d)", a->device, b->device);
        return SDRB_ERR_INVALID_ARG;
    }
    if (!a->d_offsets || !b->d_offsets) {
        set_error("batch has no uploaded rows");
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->id_bits != b->id_bits && !widths_may_differ) {
        set_error("one batch holds 32-bit ids and the other 64-bit ids");
        return SDRB_ERR_INVALID_ARG;
    }
    return SDRB_OK;
}

// everything except the per-pair merge kernels reads 32-bit ids
int check_ids32(const sdrb_batch* a, const sdrb_batch* b, const char* what) {
    if ((a && a->id_bits != 32) || (b && b->id_bits != 32)) {
        set_error("%s does not take batches with 64-bit ids", what);
        return SDRB_ERR_UNSUPPORTED;
    }
    return SDRB_OK;
}

// Two batches of 64-bit ids that both keep a 32-bit twin are served through the twins (every table kernel applies);
// SDRB_FLAG_FORCE_MERGE keeps the call on the 64-bit merge kernels.
bool narrow_pair(const sdrb_batch*& a, const sdrb_batch*& b, unsigned flags) {
    if (!a || !b || a->id_bits != 64 || b->id_bits != 64 || !a->narrow || !b->narrow) return false;
    if (flags & SDRB_FLAG_FORCE_MERGE) return false;
    a = a->narrow;
    b = b->narrow;
    return true;
}

// Batches whose ids differ in width.  The reference mixes them freely -- its fuzzer runs SDRElem<long, ...> rows against
// SDRElem<int, ...> rows (src/fuzz.cpp:413-423) and ids compare by value (SDRElem.hpp:56-75).  Here the 64-bit batch is
// taken through its 32-bit twin when it keeps one (every kernel of the 32-bit path then applies); when its ids do not
// fit 32 bits, the 32-bit batch is taken through a widened copy of its ids (scratch of the call's stream) and the pair
// runs on the 64-bit kernels.  Batches of equal width: narrow_pair.  After the call both pointers name batches of one
// width; `h` keeps the views alive for the call.
//
// Ids that no table can index as they are but that lie close together -- some of them negative, or 64-bit ids far from
// zero -- are moved first: what a set operation yields depends on which ids are equal and on their order, never on their
// values, so both batches are seen through copies of their ids less the smallest id of the two (32-bit ids in [0, span),
// scratch of the call's stream), with the span as their width; every kernel of the 32-bit path then applies, and an
// element fill adds the smallest id back to its result (IdHarmony::moved, ::base).  Spans beyond kMovedSpan stay on the
// labels (relabel.hip) and the merges.
constexpr unsigned kFlagIdsAsTheyAre = 1u << 29;   // internal: a caller whose kernels are merges anyway (operator==)
constexpr int64_t kMovedSpan = 32768;   // the widest table of the element kernels (elem_rank.hip)
struct IdHarmony {
    sdrb_batch wide;
    sdrb_batch at_zero[2];   // a and b with their ids moved
    bool moved = false;
    int64_t base = 0;        // what was subtracted
    hipStream_t stream = nullptr;
    IdHarmony() = default;
    IdHarmony(const IdHarmony&) = delete;
    IdHarmony& operator=(const IdHarmony&) = delete;
    ~IdHarmony() {
        if (!moved) return;
        // a form derived inside the call (bit planes of the moved ids) belongs to the views, which end here
        bool any = false;
        for (sdrb_batch& v : at_zero) any = any || v.d_bitplanes || v.d_dense8;
        if (!any) return;
        DeviceGuard guard(at_zero[0].device);   // (the caller's guard has gone by now)
        (void)hipStreamSynchronize(stream);
        for (sdrb_batch& v : at_zero) drop_tables(&v);
    }
};

// the ids' span where it is known: false for an empty batch or a 64-bit batch nobody has looked through
bool id_span(const sdrb_batch* b, int64_t* lo, int64_t* hi) {
    if (b->nnz == 0) return false;
    if (b->id_bits == 32) {
        *lo = b->min_id;
        *hi = b->max_id;
    } else {
        *lo = b->min_id64;
        *hi = b->max_id64;
    }
    return *lo <= *hi;
}

int move_ids_to_zero(const sdrb_batch*& a, const sdrb_batch*& b, unsigned flags, hipStream_t s, IdHarmony* h) {
    // (row by row, and where the caller asks for them, the merges serve: they take ids as they are)
    if ((flags & (SDRB_FLAG_FORCE_MERGE | SDRB_FLAG_FORCE_SEARCH | SDRB_FLAG_ROWWISE | kFlagLabelled | kFlagIdsAsTheyAre)) || a->device != b->device ||
        !a->d_offsets || !b->d_offsets)
        return SDRB_OK;
    const sdrb_batch* e[2] = {a->id_bits == 64 && a->narrow ? a->narrow : a, b->id_bits == 64 && b->narrow ? b->narrow : b};
    // Served by the tables as they are: those are indexed by the ids of b, and ids of a outside b's width match nothing
    // there (a width beyond every table, declared or derived, serves nobody).
    const bool b_served = e[1]->id_bits == 32 && ((e[1]->width > 0 && e[1]->width <= kMovedSpan) || e[1]->nnz == 0);
    if (b_served && e[0]->id_bits == 32) return SDRB_OK;
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (const sdrb_batch* x : e) {
        if (x->nnz == 0) continue;
        int64_t xlo, xhi;
        if (!id_span(x, &xlo, &xhi)) return SDRB_OK;
        lo = std::min(lo, xlo);
        hi = std::max(hi, xhi);
    }
    if (lo > hi || (uint64_t)hi - (uint64_t)lo >= (uint64_t)kMovedSpan) return SDRB_OK;
    DeviceGuard guard(a->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", a->device);
        return SDRB_ERR_HIP;
    }
    for (int k = 0; k < 2; ++k) {
        sdrb_batch& v = h->at_zero[k];
        v = *e[k];
        v.id_bits = 32;
        v.width = (int32_t)(hi - lo + 1);
        v.width_declared = true;
        v.owns_storage = false;
        v.narrow = nullptr;
        v.d_narrow_ids = nullptr;
        v.derived_home = nullptr;
        v.d_bitplanes = nullptr;
        v.d_dense8 = nullptr;
        v.bitplanes_ready = v.dense_ready = nullptr;
        v.min_id = 0;
        v.max_id = -1;
        int64_t xlo, xhi;
        if (id_span(e[k], &xlo, &xhi)) {
            v.min_id = (int32_t)(xlo - lo);
            v.max_id = (int32_t)(xhi - lo);
        }
        if (v.nnz) {
            int32_t* ids = static_cast<int32_t*>(workspace(v.device, s, 15 + k, (size_t)v.nnz * sizeof(int32_t)));
            if (!ids) return SDRB_ERR_HIP;
            const int rc = launch_rebase_ids(e[k]->d_ids, e[k]->id_bits, ids, v.nnz, lo, s);
            if (rc != SDRB_OK) return rc;
            v.d_ids = ids;
        }
    }
    h->moved = true;
    h->base = lo;
    h->stream = s;
    a = &h->at_zero[0];
    b = &h->at_zero[1];
    return SDRB_OK;
}

int harmonise_ids(const sdrb_batch*& a, const sdrb_batch*& b, unsigned flags, hipStream_t s, IdHarmony* h) {
    if (!a || !b) return SDRB_OK;   // the checks that follow name the null handle
    const int mrc = move_ids_to_zero(a, b, flags, s, h);
    if (mrc != SDRB_OK || h->moved) return mrc;
    if (a->id_bits == b->id_bits) {
        narrow_pair(a, b, flags);
        return SDRB_OK;
    }
    const sdrb_batch*& wide = a->id_bits == 64 ? a : b;
    const sdrb_batch*& slim = a->id_bits == 64 ? b : a;
    if (wide->narrow) {
        wide = wide->narrow;
        return SDRB_OK;
    }
    if (a->device != b->device || !slim->d_offsets) return SDRB_OK;   // check_pair reports these
    h->wide = *slim;
    sdrb_batch& v = h->wide;
    v.id_bits = 64;
    v.width = 0;
    v.width_declared = false;
    v.owns_storage = false;
    v.narrow = nullptr;
    v.d_narrow_ids = nullptr;
    v.derived_home = nullptr;
    v.d_bitplanes = nullptr;
    v.d_dense8 = nullptr;
    v.bitplanes_ready = v.dense_ready = nullptr;
    v.min_id64 = slim->min_id;
    v.max_id64 = slim->max_id;
    if (slim->nnz) {
        DeviceGuard guard(slim->device);
        if (!guard.ok) {
            set_error("hipSetDevice(%d) failed", slim->device);
            return SDRB_ERR_HIP;
        }
        int64_t* ids64 = static_cast<int64_t*>(workspace(slim->device, s, 14, (size_t)slim->nnz * sizeof(int64_t)));
        if (!ids64) return SDRB_ERR_HIP;
        const int rc = launch_widen_ids(slim->d_ids, ids64, slim->nnz, s);
        if (rc != SDRB_OK) return rc;
        v.d_ids = reinterpret_cast<int32_t*>(ids64);
    }
    slim = &h->wide;
    return SDRB_OK;
}

// Which payload pairings the set algebra takes: EmptyData and UnitData in any mix (UnitData.hpp:43-46),
// ArithData<float> with itself or with EmptyData (ArithData.hpp:37-83); ArithData defines no xor (:110-114).
int check_set_algebra_kinds(const sdrb_batch* a, const sdrb_batch* b, int op) {
    const bool arith = a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH;
    if (arith && (a->data_kind == SDRB_DATA_UNIT || b->data_kind == SDRB_DATA_UNIT)) {
        set_error("ArithData and UnitData batches do not combine");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (arith && op == SDRB_OP_XOR) {
        set_error("ArithData has no xor operations (ArithData.hpp:110-114)");
        return SDRB_ERR_UNSUPPORTED;
    }
    return SDRB_OK;
}

// A batch seen without its payloads.  Derived data built through the view is kept by the batch it views.
struct IdView {
    sdrb_batch v;
    const sdrb_batch* src;
    explicit IdView(const sdrb_batch* s) : v(*s), src(s) {
        v.data_kind = SDRB_DATA_EMPTY;
        v.d_data = nullptr;
        v.owns_storage = false;
        v.narrow = nullptr;          // the twin of a batch of 64-bit ids keeps its payload kind: not this view's twin
        v.d_narrow_ids = nullptr;
        // bit planes and byte matrix depend on the ids alone: they are those of the batch viewed
        v.derived_home = derived_owner(s);
        v.d_bitplanes = nullptr;
        v.d_dense8 = nullptr;
        v.bitplanes_ready = v.dense_ready = nullptr;
    }
};

// out(i, j) = length of the `this` row: rms of an ArithData `this`, whose matches all stay (ArithData.hpp:26-28,85-88)
__global__ __launch_bounds__(256) void k_fill_this_len(const uint32_t* __restrict__ a_off, const uint32_t* __restrict__ b_off,
                                                       uint64_t na, uint64_t nb, uint32_t swap, uint32_t rowwise,
                                                       uint32_t* __restrict__ out, uint64_t ld_out) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= nb) return;
    const uint32_t lb = b_off[j + 1] - b_off[j];
    if (rowwise) {
        out[j] = swap ? lb : a_off[j + 1] - a_off[j];
        return;
    }
    for (uint64_t i = blockIdx.y; i < na; i += gridDim.y) out[i * ld_out + j] = swap ? lb : a_off[i + 1] - a_off[i];
}

}  // namespace

std::mutex& derived_mutex() {
    static std::mutex m;
    return m;
}

int reserve_batch_storage(sdrb_batch* b, uint64_t n_rows, uint64_t nnz) { return reserve_storage(b, n_rows, nnz); }

void clear_batch_rows(sdrb_batch* b) {
    drop_derived(b);
    b->n_rows = 0;
    b->nnz = 0;
    b->max_row_len = 0;
    b->min_id = 0;
    b->max_id = -1;
    b->min_id64 = 0;
    b->max_id64 = -1;
}

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

void* workspace(int device, hipStream_t stream, int slot, size_t bytes, bool* fresh) {
    if (fresh) *fresh = false;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (!g_retired.empty()) reap_retired(device);
    Scratch& s = g_ws[ScratchKey{device, stream, slot}];
    // SDRB_POISON_WORKSPACE=1 (test runs): every block handed out is filled with 0xA5 first, so that a kernel which
    // counts on what an earlier call, or a fresh allocation, happened to leave there shows up as a wrong result
    static const bool poison = [] {
        const char* e = std::getenv("SDRB_POISON_WORKSPACE");
        return e && *e && *e != '0';
    }();
    if (s.bytes >= bytes && s.ptr) {
        if (poison) (void)hipMemsetAsync(s.ptr, 0xA5, s.bytes, stream);
        return s.ptr;
    }
    const size_t old_bytes = s.bytes;
    if (s.ptr) {
        // kernels already enqueued on this stream may still read the old block: it is freed after them
        Retired r{device, s.ptr, nullptr};
        if (hipEventCreateWithFlags(&r.done, hipEventDisableTiming) == hipSuccess &&
            hipEventRecord(r.done, stream) == hipSuccess) {
            g_retired.push_back(r);
        } else {
            (void)hipStreamSynchronize(stream);
            (void)hipFree(s.ptr);
        }
        s.ptr = nullptr;
        s.bytes = 0;
    }
    // grow by at least half so that a sequence of slowly growing requests settles quickly
    size_t want = bytes < (64u << 10) ? (64u << 10) : bytes;
    if (want < old_bytes + old_bytes / 2) want = old_bytes + old_bytes / 2;
    hipError_t e = hipMalloc(&s.ptr, want);
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu) for workspace failed: %s", want, hipGetErrorString(e));
        s.ptr = nullptr;
        return nullptr;
    }
    s.bytes = want;
    if (fresh) *fresh = true;
    if (poison) (void)hipMemsetAsync(s.ptr, 0xA5, s.bytes, stream);
    return s.ptr;
}

}  // namespace sdrb

using namespace sdrb;

extern "C" {

const char* sdrb_last_error(void) { return g_error.c_str(); }

int sdrb_device_count(int* count) {
    if (!count) return SDRB_ERR_INVALID_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *count = 0;
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return SDRB_ERR_HIP;
    }
    *count = n;
    return SDRB_OK;
}

int sdrb_batch_create(int device, int32_t width, int data_kind, sdrb_batch** out) {
    if (!out || width < 0 || device < 0 ||
        (data_kind != SDRB_DATA_EMPTY && data_kind != SDRB_DATA_UNIT && data_kind != SDRB_DATA_ARITH)) {
        set_error("sdrb_batch_create: bad argument (device=%d width=%d data_kind=%d)", device, width,
                  data_kind);
        return SDRB_ERR_INVALID_ARG;
    }
    sdrb_batch* b = new (std::nothrow) sdrb_batch();
    if (!b) {
        set_error("out of host memory");
        return SDRB_ERR_NO_MEMORY;
    }
    b->device = device;
    b->width = width;
    b->width_declared = width > 0;
    b->data_kind = data_kind;
    *out = b;
    return SDRB_OK;
}

int sdrb_batch_create64(int device, int data_kind, sdrb_batch** out) {
    if (data_kind == SDRB_DATA_ARITH) {
        set_error("sdrb_batch_create64: EmptyData or UnitData payloads only");
        return SDRB_ERR_INVALID_ARG;
    }
    const int rc = sdrb_batch_create(device, 0, data_kind, out);
    if (rc == SDRB_OK) (*out)->id_bits = 64;
    return rc;
}

int sdrb_batch_upload_device_csr(sdrb_batch* batch, uint64_t n_rows, const uint32_t* row_offsets,
                                 const int32_t* ids, const float* data, void* stream) {
    TraceRange range("sdrb_batch_upload_device_csr");
    if (!batch || !row_offsets) {
        set_error("sdrb_batch_upload_device_csr: null argument");
        return SDRB_ERR_INVALID_ARG;
    }
    if (check_ids32(batch, nullptr, "sdrb_batch_upload_device_csr") != SDRB_OK) return SDRB_ERR_UNSUPPORTED;
    if (n_rows > 0xFFFFFFFEull) {
        set_error("too many rows for 32-bit offsets");
        return SDRB_ERR_OVERFLOW;
    }
    const bool unit = has_payload(batch);
    DeviceGuard guard(batch->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", batch->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!ids || (unit && !data)) {  // only an element-free batch may come without arrays
        uint32_t total = 0;
        SDRB_HIP(hipMemcpyAsync(&total, row_offsets + n_rows, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SDRB_HIP(hipStreamSynchronize(s));
        if (total > 0) {
            set_error("ids (and payloads of a UnitData batch) must not be null");
            return SDRB_ERR_INVALID_ARG;
        }
    }
    PackStats init{0u, INT32_MAX, INT32_MIN, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
    PackStats* d_st = static_cast<PackStats*>(workspace(batch->device, s, 2, sizeof(PackStats)));
    if (!d_st) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemcpyAsync(d_st, &init, sizeof init, hipMemcpyHostToDevice, s));
    // one pass over the rows: total element count, row checks, id span (grid covers >= 1 thread)
    hipLaunchKernelGGL(k_validate_pack, dim3((uint32_t)((n_rows + 256) / 256)), dim3(256), 0, s, row_offsets,
                       ids, data, batch->data_kind, n_rows, d_st);
    SDRB_HIP(hipGetLastError());
    PackStats st;
    SDRB_HIP(hipMemcpyAsync(&st, d_st, sizeof st, hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    const uint32_t nnz = st.nnz;
    if (st.bad_offsets != 0xFFFFFFFFu) {
        set_error("row_offsets not monotone (or row_offsets[0] != 0) at row %u", st.bad_offsets - 1);
        return SDRB_ERR_INVALID_ARG;
    }
    if (st.bad_order != 0xFFFFFFFFu) {
        set_error("row %u is not strictly ascending", st.bad_order - 1);
        return SDRB_ERR_NOT_ASCENDING;
    }
    if (st.bad_payload != 0xFFFFFFFFu) {
        set_error("row %u holds a payload outside [0,1] (UnitData) or a NaN / infinity (ArithData)", st.bad_payload - 1);
        return SDRB_ERR_PAYLOAD_RANGE;
    }
    const bool any = st.max_len > 0;
    if (batch->width_declared && any && (st.min_id < 0 || st.max_id >= batch->width)) {
        set_error("id outside the declared width: ids span [%d, %d], width %d", st.min_id, st.max_id,
                  batch->width);
        return SDRB_ERR_ID_RANGE;
    }
    {
        const int rc = reserve_batch_storage(batch, n_rows, nnz);
        if (rc != SDRB_OK) return rc;
    }
    SDRB_HIP(hipMemcpyAsync(batch->d_offsets, row_offsets, (n_rows + 1) * sizeof(uint32_t),
                            hipMemcpyDeviceToDevice, s));
    if (nnz) SDRB_HIP(hipMemcpyAsync(batch->d_ids, ids, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
    if (nnz && unit)
        SDRB_HIP(hipMemcpyAsync(batch->d_data, data, (size_t)nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
    batch->n_rows = n_rows;
    batch->nnz = nnz;
    batch->max_row_len = st.max_len;
    batch->min_id = any ? st.min_id : 0;
    batch->max_id = any ? st.max_id : -1;
    settle_width(batch);
    return SDRB_OK;
}

int sdrb_batch_adopt_device_csr(sdrb_batch* batch, uint64_t n_rows, uint64_t nnz, const uint32_t* row_offsets,
                                const int32_t* ids, const float* data, uint32_t max_row_len,
                                int32_t min_id, int32_t max_id) {
    TraceRange range("sdrb_batch_adopt_device_csr");
    if (!batch || !row_offsets || (nnz && !ids) ||
        (nnz && has_payload(batch) && !data)) {
        set_error("sdrb_batch_adopt_device_csr: null argument");
        return SDRB_ERR_INVALID_ARG;
    }
    if (check_ids32(batch, nullptr, "sdrb_batch_adopt_device_csr") != SDRB_OK) return SDRB_ERR_UNSUPPORTED;
    if (nnz > 0xFFFFFFFFull || n_rows > 0xFFFFFFFEull) {
        set_error("batch too large for 32-bit offsets");
        return SDRB_ERR_OVERFLOW;
    }
    if (batch->width_declared && nnz && (min_id < 0 || max_id >= batch->width)) {
        set_error("id outside the declared width: ids span [%d, %d], width %d", min_id, max_id, batch->width);
        return SDRB_ERR_ID_RANGE;
    }
    DeviceGuard guard(batch->device);
    free_batch_storage(batch);
    batch->owns_storage = false;
    batch->d_offsets = const_cast<uint32_t*>(row_offsets);
    batch->d_ids = const_cast<int32_t*>(ids);
    batch->d_data = has_payload(batch) ? const_cast<float*>(data) : nullptr;
    batch->n_rows = n_rows;
    batch->nnz = nnz;
    batch->max_row_len = max_row_len;
    batch->min_id = nnz ? min_id : 0;
    batch->max_id = nnz ? max_id : -1;
    if (!batch->width_declared) batch->width = 0;   // the id span is the caller's word here: no width is derived from it
    return SDRB_OK;
}

int sdrb_batch_info(const sdrb_batch* batch, int* device, uint64_t* n_rows, uint64_t* nnz,
                    int32_t* width, int* data_kind, uint32_t* max_row_len) {
    if (!batch) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (device) *device = batch->device;
    if (n_rows) *n_rows = batch->n_rows;
    if (nnz) *nnz = batch->nnz;
    if (width) *width = declared_width(batch);
    if (data_kind) *data_kind = batch->data_kind;
    if (max_row_len) *max_row_len = batch->max_row_len;
    return SDRB_OK;
}

int sdrb_batch_shift(sdrb_batch* batch, int32_t amount, void* stream) {
    TraceRange range("sdrb_batch_shift");
    if (!batch) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (!batch->owns_storage) {
        set_error("sdrb_batch_shift: rows adopted from the caller's memory are not modified");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (batch->nnz == 0 || amount == 0) return SDRB_OK;
    sdrb_batch* spans = batch->id_bits == 64 ? batch->narrow : batch;   // where the id span is known
    if (!spans) {
        // 64-bit ids beyond 32 bits: their span, where the upload has left it, says whether the shift stays inside 64 bits
        if (batch->min_id64 > batch->max_id64 || (amount > 0 ? batch->max_id64 > INT64_MAX - amount : batch->min_id64 < INT64_MIN - amount)) {
            set_error("sdrb_batch_shift: these 64-bit ids cannot be shifted by %d (span unknown, or the shift leaves 64 bits)", amount);
            return SDRB_ERR_ID_RANGE;
        }
        DeviceGuard wide_guard(batch->device);
        if (!wide_guard.ok) {
            set_error("hipSetDevice(%d) failed", batch->device);
            return SDRB_ERR_HIP;
        }
        const int wrc = launch_shift_ids(batch->d_ids, 64, batch->nnz, amount, static_cast<hipStream_t>(stream));
        if (wrc != SDRB_OK) return wrc;
        drop_tables(batch);
        batch->min_id64 += amount;
        batch->max_id64 += amount;
        return SDRB_OK;   // (no 32-bit twin is made even where the shifted ids would now fit one)
    }
    const int64_t lo = (int64_t)spans->min_id + amount, hi = (int64_t)spans->max_id + amount;
    const bool fits32 = lo >= INT32_MIN && hi <= INT32_MAX;
    if (batch->id_bits == 32 && !fits32) {
        // the reference asserts the same, SDR.hpp:1586
        set_error("shift by %d takes ids out of 32 bits: they span [%d, %d]", amount, spans->min_id, spans->max_id);
        return SDRB_ERR_ID_RANGE;
    }
    if (batch->width_declared && (lo < 0 || hi >= batch->width)) {
        set_error("shift by %d takes ids out of the declared width %d: they span [%d, %d]", amount, batch->width,
                  spans->min_id, spans->max_id);
        return SDRB_ERR_ID_RANGE;
    }
    DeviceGuard guard(batch->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", batch->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = launch_shift_ids(batch->d_ids, batch->id_bits, batch->nnz, amount, s);
    if (rc != SDRB_OK) return rc;
    drop_tables(batch);   // bit planes and byte matrix were those of the old ids
    if (batch->id_bits == 64) {
        batch->min_id64 = lo;
        batch->max_id64 = hi;
        if (!fits32) {
            drop_narrow(batch);   // the shifted ids no longer fit 32 bits
            return SDRB_OK;
        }
        rc = launch_shift_ids(batch->d_narrow_ids, 32, batch->nnz, amount, s);
        if (rc != SDRB_OK) return rc;
        drop_tables(spans);
    }
    spans->min_id = (int32_t)lo;
    spans->max_id = (int32_t)hi;
    if (batch->id_bits == 64) {
        batch->min_id64 = lo;
        batch->max_id64 = hi;
    }
    if (batch->id_bits == 64) spans->width = lo >= 0 && hi < INT32_MAX ? (int32_t)hi + 1 : 0;
    else settle_width(batch);
    return SDRB_OK;
}

int sdrb_batch_table_width(const sdrb_batch* batch, int32_t* width) {
    if (!batch || !width) {
        set_error("sdrb_batch_table_width: null argument");
        return SDRB_ERR_INVALID_ARG;
    }
    *width = batch->id_bits == 32 ? batch->width : (batch->narrow ? batch->narrow->width : 0);
    return SDRB_OK;
}

int sdrb_batch_id_span(const sdrb_batch* batch, int32_t* min_id, int32_t* max_id) {
    if (!batch) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (min_id) *min_id = batch->min_id;
    if (max_id) *max_id = batch->max_id;
    return SDRB_OK;
}

int sdrb_batch_export_device_csr(const sdrb_batch* batch, uint32_t* row_offsets, int32_t* ids, float* data,
                                 void* stream) {
    TraceRange range("sdrb_batch_export_device_csr");
    if (!batch || !batch->d_offsets || !row_offsets || (batch->nnz && !ids) ||
        (batch->nnz && has_payload(batch) && !data)) {
        set_error("sdrb_batch_export_device_csr: null argument or empty handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (check_ids32(batch, nullptr, "sdrb_batch_export_device_csr") != SDRB_OK) return SDRB_ERR_UNSUPPORTED;
    DeviceGuard guard(batch->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    SDRB_HIP(hipMemcpyAsync(row_offsets, batch->d_offsets, (batch->n_rows + 1) * sizeof(uint32_t),
                            hipMemcpyDeviceToDevice, s));
    if (batch->nnz) {
        SDRB_HIP(hipMemcpyAsync(ids, batch->d_ids, batch->nnz * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
        if (has_payload(batch))
            SDRB_HIP(hipMemcpyAsync(data, batch->d_data, batch->nnz * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return SDRB_OK;
}

int sdrb_batch_device_ptrs(const sdrb_batch* batch, const uint32_t** row_offsets,
                           const int32_t** ids, const float** data) {
    if (!batch) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (row_offsets) *row_offsets = batch->d_offsets;
    if (ids) *ids = batch->d_ids;
    if (data) *data = batch->d_data;
    return SDRB_OK;
}

int sdrb_batch_destroy(sdrb_batch* batch) {
    if (!batch) return SDRB_OK;
    {
        DeviceGuard guard(batch->device);
        free_batch_storage(batch);
    }
    delete batch;
    return SDRB_OK;
}

int sdrb_size_op(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out,
                 uint64_t ld_out, void* stream) {
    TraceRange range("sdrb_size_op");
    IdHarmony harmony;
    int rc = harmonise_ids(a, b, flags, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, op);
    if (rc == SDRB_OK) rc = check_set_algebra_kinds(a, b, op);
    if (rc != SDRB_OK) return rc;
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
    if (rowwise && a->n_rows != b->n_rows) {
        set_error("row-wise evaluation needs batches of equal length (%llu vs %llu rows)",
                  (unsigned long long)a->n_rows, (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || (!rowwise && ld_out < b->n_rows)) {
        set_error("sdrb_size_op: out is null or ld_out (%llu) < rows of b (%llu)",
                  (unsigned long long)ld_out, (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) {
        // ArithData payloads are always relevant (ArithData.hpp:22-28), so sizes depend on ids alone: ands and ors
        // are those of the id sets, rms of an ArithData `this` keeps every element, an EmptyData `this` is ids only.
        const sdrb_batch* self = swap ? b : a;
        if (op == SDRB_OP_RM && self->data_kind == SDRB_DATA_ARITH) {
            DeviceGuard guard(b->device);
            if (!guard.ok) {
                set_error("hipSetDevice(%d) failed", b->device);
                return SDRB_ERR_HIP;
            }
            const uint32_t gy = rowwise ? 1u : (uint32_t)(a->n_rows < 1024 ? a->n_rows : 1024);
            hipLaunchKernelGGL(k_fill_this_len, dim3((uint32_t)((b->n_rows + 255) / 256), gy), dim3(256), 0,
                               static_cast<hipStream_t>(stream), a->d_offsets, b->d_offsets, a->n_rows, b->n_rows,
                               swap ? 1u : 0u, rowwise ? 1u : 0u, out, ld_out);
            SDRB_HIP(hipGetLastError());
            return SDRB_OK;
        }
        IdView va(a), vb(b);
        return sdrb_size_op(&va.v, &vb.v, op, flags, out, ld_out, stream);
    }
    const bool can_bitslice = a->id_bits == 32 && id_only(a, b, op, swap) && bitslice_applicable(a, b);
    if ((flags & SDRB_FLAG_FORCE_BITSLICE) && !can_bitslice) {
        set_error("bit-sliced kernel requested but not applicable to these batches/op");
        return SDRB_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // Ids no table kernel takes as they are (no declared width, a width beyond every table, 64-bit ids that do not fit
    // 32 bits): small labels first, then the table kernels on the labelled rows.
    constexpr unsigned kForced = SDRB_FLAG_FORCE_MERGE | SDRB_FLAG_FORCE_BITSLICE | SDRB_FLAG_FORCE_POPCOUNT | SDRB_FLAG_FORCE_ROWTABLE |
                                 SDRB_FLAG_FORCE_SPARSE | SDRB_FLAG_FORCE_MASKTABLE | SDRB_FLAG_FORCE_BITSLICE_LONG;
    if (!rowwise && !(flags & (kForced | kFlagLabelled))) {
        const bool direct = a->id_bits == 32 &&
                            (can_bitslice || (id_only(a, b, op, swap) && (popcount_applicable(a, b) || bitslice_long_applicable(a, b))) ||
                             size_lane_rows_applicable(a, b, op, swap, false) || mask_table_applicable(a, b) || unit_table_applicable(a, b));
        if (!direct && size_relabel_applicable(a, b, op, swap)) return launch_size_relabel(a, b, op, flags, out, ld_out, s);
    }
    if (rowwise || a->id_bits == 64) return launch_size_merge(a, b, op, swap, out, ld_out, rowwise, s);
    if ((flags & SDRB_FLAG_FORCE_MASKTABLE) && !mask_table_applicable(a, b)) {
        set_error("mask-table kernel requested but not applicable (needs a declared width whose table fits LDS)");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (flags & SDRB_FLAG_FORCE_SPARSE) {
        if (size_lane_rows_applicable(a, b, op, swap, true)) return launch_size_lane_rows(a, b, op, swap, out, ld_out, s);
        set_error("lane-per-row match kernel requested but not applicable (payload-dependent op, width <= 4096, "
                  "rows of b <= 64 ids, rows of a <= 128 ids)");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (flags & SDRB_FLAG_FORCE_BITSLICE_LONG) {
        if (a->id_bits != 32 || !id_only(a, b, op, swap) || !bitslice_long_applicable(a, b)) {
            set_error("bit-sliced kernel for long rows requested but not applicable (sizes that depend on ids alone, a declared "
                      "width whose table fits the LDS)");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_size_bitslice_long(a, b, op, swap, out, ld_out, s);
    }
    if (flags & SDRB_FLAG_FORCE_POPCOUNT) {
        if (a->id_bits != 32 || !id_only(a, b, op, swap) || !popcount_applicable(a, b)) {
            set_error("popcount kernel requested but not applicable to these batches/op");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_size_popcount(a, b, op, swap, out, ld_out, s);
    }
    if (!(flags & SDRB_FLAG_FORCE_MERGE)) {
        if (can_bitslice) return launch_size_bitslice(a, b, op, swap, out, ld_out, s);
        // longer rows, wider tables: the same formulation with the row streamed and the tile as wide as the batch
        // (dense rows of a narrow table stay with the popcount kernel, whose cost does not depend on the row length: at
        // width 2048 the two meet at rows of about 100 ids -- 64 x 1 Mi rows 0.256 against 0.267 ms; at 300 ids 1.9 against 1.1 ms)
        const bool dense_rows = popcount_applicable(a, b) && (uint64_t)b->max_row_len * 20 > (uint64_t)b->width;
        if (!(flags & (SDRB_FLAG_FORCE_ROWTABLE | SDRB_FLAG_FORCE_MASKTABLE)) && !dense_rows && id_only(a, b, op, swap) &&
            bitslice_long_applicable(a, b))
            return launch_size_bitslice_long(a, b, op, swap, out, ld_out, s);
        // rows too long for the bit-sliced kernel: the popcount kernel's cost does not depend on row length
        if (id_only(a, b, op, swap) && popcount_applicable(a, b))
            return launch_size_popcount(a, b, op, swap, out, ld_out, s);
        // payloads on short sparse rows: every match visited once by the lane that owns the row of B
        if (!(flags & (SDRB_FLAG_FORCE_ROWTABLE | SDRB_FLAG_FORCE_MASKTABLE)) && size_lane_rows_applicable(a, b, op, swap, false))
            return launch_size_lane_rows(a, b, op, swap, out, ld_out, s);
        // payloads or long rows: with several rows of A the 64-row mask table reads B once for all of them
        if (!(flags & SDRB_FLAG_FORCE_ROWTABLE) && mask_table_applicable(a, b))
            return launch_size_mask_table(a, b, op, swap, out, ld_out, s);
        if (unit_table_applicable(a, b)) return launch_size_unit_table(a, b, op, swap, out, ld_out, s);
    }
    return launch_size_merge(a, b, op, swap, out, ld_out, false, s);
}

int sdrb_size_op_narrow(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, void* out, uint64_t ld_out,
                        int out_bytes, void* stream) {
    TraceRange range("sdrb_size_op_narrow");
    IdHarmony harmony;   // 64-bit ids through their narrowed copies, ids close together moved to zero
    int rc = harmonise_ids(a, b, flags & SDRB_FLAG_SWAP, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, op);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_size_op_narrow");
    if (rc != SDRB_OK) return rc;
    if (out_bytes != 1 && out_bytes != 2) {
        set_error("sdrb_size_op_narrow: out_bytes must be 1 (uint8 counts) or 2 (uint16 counts)");
        return SDRB_ERR_INVALID_ARG;
    }
    if (flags & ~(SDRB_FLAG_SWAP | SDRB_FLAG_FORCE_BITSLICE)) {
        set_error("sdrb_size_op_narrow takes SDRB_FLAG_SWAP (and SDRB_FLAG_FORCE_BITSLICE, which it implies) only");
        return SDRB_ERR_INVALID_ARG;
    }
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) {
        // ArithData payloads are always relevant: ands and ors are those of the id sets, an EmptyData `this` is ids only
        const sdrb_batch* self = swap ? b : a;
        if (op == SDRB_OP_XOR || (op == SDRB_OP_RM && self->data_kind == SDRB_DATA_ARITH)) {
            set_error("sdrb_size_op_narrow: this op on ArithData batches is served by sdrb_size_op only");
            return SDRB_ERR_UNSUPPORTED;
        }
        IdView va(a), vb(b);
        return sdrb_size_op_narrow(&va.v, &vb.v, op, flags, out, ld_out, out_bytes, stream);
    }
    if (!id_only(a, b, op, swap) || !bitslice_applicable(a, b)) {
        set_error("sdrb_size_op_narrow serves the bit-sliced kernel only: sizes that depend on ids alone (`this` "
                  "EmptyData, or ors), b with a declared width <= 4096 and rows <= 64 ids");
        return SDRB_ERR_UNSUPPORTED;
    }
    const uint64_t bound = size_op_bound(a, b, op, swap), most = out_bytes == 1 ? 0xFFu : 0xFFFFu;
    if (bound > most) {
        set_error("sdrb_size_op_narrow: a result may reach %llu, more than %d byte(s) hold (the longest rows have %u and "
                  "%u ids)", (unsigned long long)bound, out_bytes, a->max_row_len, b->max_row_len);
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || ld_out < b->n_rows || ld_out % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 3u) != 0 ||
        ld_out * (uint64_t)out_bytes > (1ull << 27)) {
        set_error("sdrb_size_op_narrow: out must be 4-byte aligned, ld_out (%llu) a multiple of 4, >= the rows of b (%llu) "
                  "and at most 2^27 bytes a row", (unsigned long long)ld_out, (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    return launch_size_bitslice_narrow(a, b, op, swap, out, out_bytes, ld_out, static_cast<hipStream_t>(stream));
}

int sdrb_size_op_static(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out,
                        uint64_t ld_out, void* stream) {
    TraceRange range("sdrb_size_op_static");
    int rc = check_pair(a, b, op);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_size_op_static");
    if (rc != SDRB_OK) return rc;
    if (flags & ~(SDRB_FLAG_SWAP | SDRB_FLAG_FORCE_BITSLICE)) {
        set_error("sdrb_size_op_static takes SDRB_FLAG_SWAP (and SDRB_FLAG_FORCE_BITSLICE, which it implies) only");
        return SDRB_ERR_INVALID_ARG;
    }
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH || !id_only(a, b, op, swap) ||
        !bitslice_applicable(a, b)) {
        set_error("sdrb_size_op_static serves the bit-sliced kernel only: sizes that depend on ids alone (`this` "
                  "EmptyData, or ors), b with a declared width <= 4096 and rows <= 64 ids");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || ld_out < b->n_rows) {
        set_error("sdrb_size_op_static: out is null or ld_out (%llu) < rows of b (%llu)", (unsigned long long)ld_out,
                  (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    return launch_size_bitslice_static(a, b, op, swap, out, ld_out, static_cast<hipStream_t>(stream));
}

int sdrb_equal(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, uint32_t* out, uint64_t ld_out, void* stream) {
    TraceRange range("sdrb_equal");
    IdHarmony harmony;
    int rc = harmonise_ids(a, b, flags | kFlagIdsAsTheyAre, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, SDRB_OP_AND);
    if (rc == SDRB_OK) rc = check_set_algebra_kinds(a, b, SDRB_OP_AND);
    if (rc != SDRB_OK) return rc;
    if (!has_payload(a) && has_payload(b)) {
        // the reference does not compile this comparison either (EmptyData has no != for a payload, EmptyData.hpp:82-83)
        set_error("EmptyData rows compare with EmptyData rows only; put the batch with payloads first");
        return SDRB_ERR_UNSUPPORTED;
    }
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
    if (rowwise && a->n_rows != b->n_rows) {
        set_error("row-wise evaluation needs batches of equal length");
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || (!rowwise && ld_out < b->n_rows)) {
        set_error("sdrb_equal: out is null or ld_out < rows of b");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    return launch_equal(a, b, out, ld_out, rowwise, static_cast<hipStream_t>(stream));
}

static int sdrb_batch_transpose_impl(const sdrb_batch* src, sdrb_batch* dst, void* stream) {
    if (!src || !dst || src == dst || !src->d_offsets) {
        set_error("sdrb_batch_transpose: null, identical or empty handles");
        return SDRB_ERR_INVALID_ARG;
    }
    if (check_ids32(src, dst, "sdrb_batch_transpose") != SDRB_OK) return SDRB_ERR_UNSUPPORTED;
    if (!src->width_declared || src->device != dst->device || src->data_kind != dst->data_kind) {
        set_error("sdrb_batch_transpose: src needs a declared width; dst the same device and payload kind");
        return SDRB_ERR_INVALID_ARG;
    }
    if (src->n_rows > 0x7FFFFFFFull) {
        set_error("too many rows to become ids");
        return SDRB_ERR_OVERFLOW;
    }
    DeviceGuard guard(src->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint64_t width = (uint64_t)src->width;
    int rc = reserve_batch_storage(dst, width, src->nnz);
    if (rc != SDRB_OK) return rc;
    rc = launch_transpose(src, dst->d_offsets, dst->d_ids, dst->d_data, s);
    if (rc != SDRB_OK) return rc;
    // the longest column, from the column offsets (width + 1 integers)
    std::vector<uint32_t> off(width + 1);
    SDRB_HIP(hipMemcpyAsync(off.data(), dst->d_offsets, (width + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    uint32_t longest = 0;
    for (uint64_t c = 0; c < width; ++c) longest = std::max(longest, off[c + 1] - off[c]);
    dst->n_rows = width;
    dst->nnz = src->nnz;
    dst->max_row_len = longest;
    dst->width = (int32_t)src->n_rows;
    dst->width_declared = true;
    dst->min_id = 0;                                   // bounds of the row numbers (not necessarily attained)
    dst->max_id = src->nnz ? (int32_t)src->n_rows - 1 : -1;
    return SDRB_OK;
}

int sdrb_batch_transpose(const sdrb_batch* src, sdrb_batch* dst, void* stream) {
    TraceRange range("sdrb_batch_transpose");
    return sdrb::guarded("sdrb_batch_transpose", [&] { return sdrb_batch_transpose_impl(src, dst, stream); });
}

namespace {
// the exact total to the host (synchronises); a result of 2^32 elements or more cannot be addressed by the
// 32-bit offsets
int read_total(const unsigned long long* total_dev, uint64_t* total, hipStream_t s) {
    unsigned long long t64 = 0;
    SDRB_HIP(hipMemcpyAsync(&t64, total_dev, sizeof t64, hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    if (t64 > 0xFFFFFFFFull) {
        sdrb::set_error("result has %llu elements, more than 2^32-1; split the A batch", t64);
        return SDRB_ERR_OVERFLOW;
    }
    *total = t64;
    return SDRB_OK;
}

// counts[0..pairs) -> exclusive offsets[0..pairs] in place, then read_total
int finish_count(int device, uint32_t* out_offsets, uint64_t pairs, uint64_t* total, hipStream_t s) {
    unsigned long long* total_dev = static_cast<unsigned long long*>(sdrb::workspace(device, s, 10, sizeof(unsigned long long)));
    if (!total_dev) return SDRB_ERR_HIP;
    // sizes at out_offsets[0..pairs) -> exclusive offsets [0..pairs], the 64-bit total to total_dev
    const int rc = sdrb::launch_exclusive_scan_u32(device, out_offsets, pairs + 1, pairs, total_dev, s);
    if (rc != SDRB_OK) return rc;
    return read_total(total_dev, total, s);
}

int check_queries(const sdrb_batch* b, uint64_t n, const int32_t* lo, const int32_t* hi, bool ranges) {
    if (!b || !b->d_offsets) {
        sdrb::set_error("null batch handle or no uploaded rows");
        return SDRB_ERR_INVALID_ARG;
    }
    if (b->id_bits != 32) {
        sdrb::set_error("range and point queries do not take batches with 64-bit ids");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (n && (!lo || (ranges && !hi))) {
        sdrb::set_error("query arrays are null");
        return SDRB_ERR_INVALID_ARG;
    }
    if (n > 0xFFFFFFFFull) {
        sdrb::set_error("too many queries for one call");
        return SDRB_ERR_OVERFLOW;
    }
    if (ranges)
        for (uint64_t r = 0; r < n; ++r)
            if (lo[r] > hi[r]) {
                sdrb::set_error("range %llu has start %d > stop %d", (unsigned long long)r, lo[r], hi[r]);
                return SDRB_ERR_INVALID_ARG;
            }
    return SDRB_OK;
}
}  // namespace

int sdrb_range_size(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                    uint32_t* out, uint64_t ld_out, void* stream) {
    TraceRange range("sdrb_range_size");
    int rc = check_queries(b, n_ranges, lo, hi, true);
    if (rc != SDRB_OK) return rc;
    if (n_ranges == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || ld_out < b->n_rows) {
        set_error("sdrb_range_size: out is null or ld_out < rows of b");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    return launch_range_query(b, 0, n_ranges, lo, hi, out, ld_out, nullptr, nullptr, nullptr,
                              static_cast<hipStream_t>(stream));
}

int sdrb_range_elems_count(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                           uint32_t* out_offsets, uint64_t* total, void* stream) {
    TraceRange range("sdrb_range_elems_count");
    int rc = check_queries(b, n_ranges, lo, hi, true);
    if (rc != SDRB_OK) return rc;
    if (!out_offsets || !total) {
        set_error("sdrb_range_elems_count: null output");
        return SDRB_ERR_INVALID_ARG;
    }
    const uint64_t pairs = n_ranges * b->n_rows;
    if (pairs >= 0xFFFFFFFFull) {
        set_error("too many (range, row) pairs for one call; use fewer ranges per call");
        return SDRB_ERR_OVERFLOW;
    }
    DeviceGuard guard(b->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pairs) {
        rc = launch_range_query(b, 0, n_ranges, lo, hi, out_offsets, b->n_rows, nullptr, nullptr, nullptr, s);
        if (rc != SDRB_OK) return rc;
    }
    return finish_count(b->device, out_offsets, pairs, total, s);
}

int sdrb_range_elems_fill(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                          const uint32_t* out_offsets, int32_t* out_ids, float* out_data, void* stream) {
    TraceRange range("sdrb_range_elems_fill");
    int rc = check_queries(b, n_ranges, lo, hi, true);
    if (rc != SDRB_OK) return rc;
    if (n_ranges == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out_offsets || !out_ids) {
        set_error("sdrb_range_elems_fill: null output");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    return launch_range_query(b, 1, n_ranges, lo, hi, nullptr, 0, out_offsets, out_ids,
                              has_payload(b) ? out_data : nullptr, static_cast<hipStream_t>(stream));
}

int sdrb_point_query(const sdrb_batch* b, uint64_t n_ids, const int32_t* ids, int32_t* out, uint64_t ld_out,
                     void* stream) {
    TraceRange range("sdrb_point_query");
    int rc = check_queries(b, n_ids, ids, nullptr, false);
    if (rc != SDRB_OK) return rc;
    if (n_ids == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || ld_out < b->n_rows) {
        set_error("sdrb_point_query: out is null or ld_out < rows of b");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    return launch_range_query(b, 2, n_ids, ids, nullptr, reinterpret_cast<uint32_t*>(out), ld_out, nullptr, nullptr,
                              nullptr, static_cast<hipStream_t>(stream));
}

int sdrb_inner(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, float* out, uint64_t ld_out,
               void* stream) {
    TraceRange range("sdrb_inner");
    // (a dot product visits the matches in ascending id order, which ids moved to zero keep)
    IdHarmony harmony;
    int rc = harmonise_ids(a, b, flags & SDRB_FLAG_FORCE_MERGE, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, SDRB_OP_AND);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_inner");
    if (rc != SDRB_OK) return rc;
    if (a->data_kind != SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_UNIT) {
        set_error("sdrb_inner takes an ArithData batch and an ArithData or EmptyData batch");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    if (!out || ld_out < b->n_rows) {
        set_error("sdrb_inner: out is null or ld_out (%llu) < rows of b (%llu)", (unsigned long long)ld_out,
                  (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    return launch_inner(a, b, flags, out, ld_out, static_cast<hipStream_t>(stream));
}

int sdrb_allpairs_topk(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                       unsigned flags, uint32_t* out_counts, int32_t* out_rows, void* stream) {
    TraceRange range("sdrb_allpairs_topk");
    IdHarmony harmony;   // overlaps depend on which ids are equal: 64-bit ids through their narrowed copies, ids close together moved to zero
    int rc = harmonise_ids(a, b, 0, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, SDRB_OP_AND);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_allpairs_topk");
    if (rc != SDRB_OK) return rc;
    if (k == 0 || !out_counts || !out_rows) {
        set_error("sdrb_allpairs_topk: k is 0 or an output is null");
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->n_rows == 0) return SDRB_OK;
    const bool can_bitslice = allpairs_bitslice_applicable(a, b, k);
    const bool can_mfma = allpairs_mfma_applicable(a, b, k);
    const bool want_mfma = (flags & SDRB_FLAG_FORCE_MFMA) != 0, want_bitslice = (flags & SDRB_FLAG_FORCE_BITSLICE) != 0;
    if ((want_mfma && !can_mfma) || (want_bitslice && !can_bitslice) || (!can_mfma && !can_bitslice)) {
        set_error("sdrb_allpairs_topk: kernel not applicable (k <= 16 and declared widths needed; bit-sliced: "
                  "widths <= 4096, rows of a <= 64 ids, at most 16 775 168 rows of b; matrix-core: widths <= 16384, rows "
                  "shorter than 4096 ids, at most 4 192 256 rows of b; cut b into row ranges beyond that)");
        return SDRB_ERR_UNSUPPORTED;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // at SDR densities the bit-sliced kernel does a fraction of the dense product's work
    if (want_mfma || (!want_bitslice && !can_bitslice))
        return launch_allpairs_mfma(a, b, k, exclude_offset, out_counts, out_rows, s);
    return launch_allpairs_bitslice(a, b, k, exclude_offset, out_counts, out_rows, s);
}

namespace {
// checks shared by the counting entry points; *pairs = number of results
int check_count_call(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, const void* out_offsets,
                     uint64_t* pairs) {
    int rc = check_pair(a, b, op);
    if (rc == SDRB_OK) rc = check_set_algebra_kinds(a, b, op);
    if (rc != SDRB_OK) return rc;
    if (!out_offsets) {
        set_error("element op: out_offsets is null");
        return SDRB_ERR_INVALID_ARG;
    }
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
    if (rowwise && a->n_rows != b->n_rows) {
        set_error("row-wise evaluation needs batches of equal length (%llu vs %llu rows)",
                  (unsigned long long)a->n_rows, (unsigned long long)b->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    *pairs = rowwise ? b->n_rows : a->n_rows * b->n_rows;
    if (*pairs >= 0xFFFFFFFFull) {
        set_error("too many pairs for one call (%llu)", (unsigned long long)*pairs);
        return SDRB_ERR_OVERFLOW;
    }
    // the scan carries sums in 64 bits; no result can come near that (every element is output at most
    // once per row of the other batch)
    if ((unsigned __int128)a->nnz * b->n_rows + (unsigned __int128)b->nnz * a->n_rows >= ((unsigned __int128)1 << 61)) {
        set_error("result far beyond 2^32-1 elements; split the A batch");
        return SDRB_ERR_OVERFLOW;
    }
    return SDRB_OK;
}

// the most elements the result of `op` can hold, from the batches' shapes alone
uint64_t elem_op_bound(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags) {
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0, swap = (flags & SDRB_FLAG_SWAP) != 0;
    const uint64_t from_a = rowwise ? a->nnz : a->nnz * b->n_rows, from_b = rowwise ? b->nnz : b->nnz * a->n_rows;
    if (op == SDRB_OP_AND) return from_a < from_b ? from_a : from_b;
    if (op == SDRB_OP_RM) return swap ? from_b : from_a;
    return from_a + from_b;
}
}  // namespace

int sdrb_elem_op_count(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                       uint32_t* out_offsets, uint64_t* total, void* stream) {
    TraceRange range("sdrb_elem_op_count");
    IdHarmony harmony;
    uint64_t pairs = 0;
    int rc = harmonise_ids(a, b, flags, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_count_call(a, b, op, flags, out_offsets, &pairs);
    if (rc != SDRB_OK) return rc;
    if (!total) {
        set_error("sdrb_elem_op_count: null output");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pairs) {
        rc = sdrb_size_op(a, b, op, flags & (SDRB_FLAG_SWAP | SDRB_FLAG_FORCE_MERGE | SDRB_FLAG_ROWWISE | SDRB_FLAG_FORCE_ROWTABLE), out_offsets,
                          b->n_rows, stream);
        if (rc != SDRB_OK) return rc;
    }
    return finish_count(b->device, out_offsets, pairs, total, s);
}

int sdrb_elem_op_bound(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint64_t* bound) {
    int rc = check_pair(a, b, op, true);   // shapes only: the ids' widths may differ
    if (rc != SDRB_OK) return rc;
    if (!bound) {
        set_error("sdrb_elem_op_bound: null output");
        return SDRB_ERR_INVALID_ARG;
    }
    *bound = elem_op_bound(a, b, op, flags);
    return SDRB_OK;
}

namespace {
int elem_op_fill_any(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, const uint32_t* out_offsets,
                     void* out_ids_any, int want_bits, float* out_data, void* stream) {
    // The result's ids are `this`'s (SDR.hpp:151-152: ret_t defaults to this's element type); the pair itself may run
    // on ids of the other width (harmonise_ids): then the fill goes through scratch and the ids are converted.
    IdHarmony harmony;
    if (a && b && out_offsets && out_ids_any) {
        const sdrb_batch* self0 = (flags & SDRB_FLAG_SWAP) ? b : a;
        if (self0->id_bits != want_bits) {
            set_error("the result takes the ids of `this`, which are %d-bit: use %s", self0->id_bits,
                      self0->id_bits == 64 ? "sdrb_elem_op_fill64" : "sdrb_elem_op_fill");
            return SDRB_ERR_INVALID_ARG;
        }
        hipStream_t s = static_cast<hipStream_t>(stream);
        const sdrb_batch* arg0 = (flags & SDRB_FLAG_SWAP) ? a : b;
        if (want_bits == 32 && arg0->id_bits == 64 && !arg0->narrow && (op == SDRB_OP_OR || op == SDRB_OP_XOR)) {
            set_error("a union of 32-bit `this` rows with 64-bit ids that do not fit 32 bits has no 32-bit result");
            return SDRB_ERR_UNSUPPORTED;
        }
        const sdrb_batch *na = a, *nb = b;
        const int hrc = harmonise_ids(na, nb, flags, s, &harmony);
        if (hrc != SDRB_OK) return hrc;
        const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
        const uint64_t pairs = rowwise ? nb->n_rows : na->n_rows * nb->n_rows;
        if (harmony.moved) {
            // the pair ran on ids less harmony.base: the fill's ids get it back (and `this`'s width on the way)
            if (rowwise && na->n_rows != nb->n_rows) {
                set_error("row-wise evaluation needs batches of equal length");
                return SDRB_ERR_INVALID_ARG;
            }
            if (pairs == 0) return SDRB_OK;
            DeviceGuard guard(nb->device);
            if (!guard.ok) {
                set_error("hipSetDevice(%d) failed", nb->device);
                return SDRB_ERR_HIP;
            }
            if (want_bits == 32) {
                const int rc_inner = elem_op_fill_any(na, nb, op, flags, out_offsets, out_ids_any, 32, out_data, stream);
                if (rc_inner != SDRB_OK) return rc_inner;
                return launch_unbase_ids32(static_cast<int32_t*>(out_ids_any), out_offsets + pairs, (int32_t)harmony.base, s);
            }
            uint32_t total = 0;
            SDRB_HIP(hipMemcpyAsync(&total, out_offsets + pairs, sizeof total, hipMemcpyDeviceToHost, s));
            SDRB_HIP(hipStreamSynchronize(s));
            if (total == 0) return SDRB_OK;
            int32_t* tmp = static_cast<int32_t*>(workspace(nb->device, s, 11, (size_t)total * sizeof(int32_t)));
            if (!tmp) return SDRB_ERR_HIP;
            const int rc_inner = elem_op_fill_any(na, nb, op, flags, out_offsets, tmp, 32, out_data, stream);
            if (rc_inner != SDRB_OK) return rc_inner;
            return launch_unbase_ids64(tmp, static_cast<int64_t*>(out_ids_any), out_offsets + pairs, harmony.base, s);
        }
        if (na->id_bits == nb->id_bits && na->id_bits != want_bits) {
            DeviceGuard guard(nb->device);
            if (!guard.ok) {
                set_error("hipSetDevice(%d) failed", nb->device);
                return SDRB_ERR_HIP;
            }
            uint32_t total = 0;
            SDRB_HIP(hipMemcpyAsync(&total, out_offsets + pairs, sizeof total, hipMemcpyDeviceToHost, s));
            SDRB_HIP(hipStreamSynchronize(s));
            if (total == 0) return SDRB_OK;
            const size_t each = want_bits == 64 ? sizeof(int32_t) : sizeof(int64_t);
            void* tmp = workspace(nb->device, s, 11, (size_t)total * each);
            if (!tmp) return SDRB_ERR_HIP;
            const int rc_inner = elem_op_fill_any(na, nb, op, flags, out_offsets, tmp, na->id_bits, out_data, stream);
            if (rc_inner != SDRB_OK) return rc_inner;
            if (want_bits == 64) return launch_widen_ids(static_cast<const int32_t*>(tmp), static_cast<int64_t*>(out_ids_any), total, s);
            // intersections and differences of 32-bit `this` rows hold ids of `this` only: they narrow exactly
            return launch_narrow_ids(static_cast<const int64_t*>(tmp), static_cast<int32_t*>(out_ids_any), total, s);
        }
        a = na;
        b = nb;
        if (a->id_bits != want_bits) {   // (widths still differ only when a check below refuses the pair)
            set_error("these batches hold %d-bit ids", a->id_bits);
            return SDRB_ERR_INVALID_ARG;
        }
    }
    int rc = check_pair(a, b, op);
    if (rc == SDRB_OK && a->id_bits != want_bits) {
        set_error("these batches hold %d-bit ids: use %s", a->id_bits,
                  a->id_bits == 64 ? "sdrb_elem_op_fill64" : "sdrb_elem_op_fill");
        rc = SDRB_ERR_INVALID_ARG;
    }
    int32_t* out_ids = static_cast<int32_t*>(out_ids_any);
    if (rc == SDRB_OK) rc = check_set_algebra_kinds(a, b, op);
    if (rc != SDRB_OK) return rc;
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
    if (rowwise && a->n_rows != b->n_rows) {
        set_error("row-wise evaluation needs batches of equal length");
        return SDRB_ERR_INVALID_ARG;
    }
    const sdrb_batch* self = swap ? b : a;
    if (!out_offsets || !out_ids || (has_payload(self) && !out_data)) {
        set_error("sdrb_elem_op_fill: null output (results with payloads need out_data)");
        return SDRB_ERR_INVALID_ARG;
    }
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) {
        if (self->data_kind == SDRB_DATA_EMPTY) {   // results are ids only: the argument's payloads play no part
            IdView va(a), vb(b);
            return elem_op_fill_any(&va.v, &vb.v, op, flags, out_offsets, out_ids_any, want_bits, nullptr, stream);
        }
        DeviceGuard arith_guard(b->device);
        if (!arith_guard.ok) {
            set_error("hipSetDevice(%d) failed", b->device);
            return SDRB_ERR_HIP;
        }
        hipStream_t as = static_cast<hipStream_t>(stream);
        if (!(flags & SDRB_FLAG_FORCE_MERGE) && !rowwise && elem_rank_applicable(a, b))
            return launch_elem_rank(a, b, op, swap, out_offsets, out_ids, out_data, as);
        if (flags & SDRB_FLAG_FORCE_RANK) {
            set_error("the rank-table fill does not apply to these batches");
            return SDRB_ERR_UNSUPPORTED;
        }
        if (!(flags & SDRB_FLAG_FORCE_MERGE) && !rowwise && elem_search_applicable(a, b))
            return launch_elem_search(a, b, op, swap, out_offsets, out_ids_any, out_data, as);
        if (flags & SDRB_FLAG_FORCE_SEARCH) {
            set_error("the binary-search fill does not apply to these batches");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_elem_fill(a, b, op, swap, out_offsets, out_ids_any, out_data, rowwise, as);
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (flags & SDRB_FLAG_FORCE_SEARCH) {
        if (rowwise || !elem_search_applicable(a, b)) {
            set_error("the binary-search fill needs all-pairs evaluation and rows short enough to stage in LDS");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_elem_search(a, b, op, swap, out_offsets, out_ids_any, out_data, s);
    }
    if (flags & SDRB_FLAG_FORCE_RANK) {
        if (rowwise || !elem_rank_applicable(a, b)) {
            set_error("the rank-table fill needs all-pairs evaluation, 32-bit ids and a declared width <= 32768 that holds every id");
            return SDRB_ERR_UNSUPPORTED;
        }
        return launch_elem_rank(a, b, op, swap, out_offsets, out_ids, out_data, s);
    }
    if (flags & (SDRB_FLAG_FORCE_BITMAP | SDRB_FLAG_FORCE_LANEMERGE | SDRB_FLAG_FORCE_LANERANK)) {
        const bool bitmap = (flags & SDRB_FLAG_FORCE_BITMAP) != 0, lanerank = (flags & SDRB_FLAG_FORCE_LANERANK) != 0;
        if (rowwise || a->id_bits != 32 ||
            !(bitmap ? elem_bitmap_applicable(a, b, swap)
                     : lanerank ? elem_lanerank_applicable(a, b, op, swap) : elem_lanemerge_applicable(a, b, op, swap))) {
            set_error("the requested element fill kernel does not apply to these batches / this op");
            return SDRB_ERR_UNSUPPORTED;
        }
        if (bitmap) return launch_elem_bitmap(a, b, op, swap, out_offsets, out_ids, s);
        if (lanerank) return launch_elem_lanerank(a, b, op, swap, out_offsets, out_ids, s);
        return launch_elem_lanemerge(a, b, op, swap, out_offsets, out_ids, s);
    }
    if (!(flags & SDRB_FLAG_FORCE_MERGE) && !rowwise && a->id_bits == 32) {
        // intersections are small: visiting only the matches (mask table) beats enumerating whole bitsets
        if (op == SDRB_OP_AND && !(flags & SDRB_FLAG_FORCE_ROWTABLE) && mask_table_applicable(a, b))
            return launch_elem_mask_table(a, b, swap, out_offsets, out_ids, out_data, s);
        // unions of short rows, ids only: a lane per pair, coalesced runs out of an LDS stage
        if (elem_lanerank_applicable(a, b, op, swap)) return launch_elem_lanerank(a, b, op, swap, out_offsets, out_ids, s);
        if (elem_lanemerge_applicable(a, b, op, swap)) return launch_elem_lanemerge(a, b, op, swap, out_offsets, out_ids, s);
        if (elem_bitmap_applicable(a, b, swap)) return launch_elem_bitmap(a, b, op, swap, out_offsets, out_ids, s);
        if (unit_table_fill_applicable(a, b, op, swap))
            return launch_elem_unit_table(a, b, op, swap, out_offsets, out_ids, out_data, s);
        if (elem_rank_applicable(a, b)) return launch_elem_rank(a, b, op, swap, out_offsets, out_ids, out_data, s);
    }
    // ids no table indexes (any value, 32 or 64 bits): a wave per pair, ranks by binary search
    if (!(flags & SDRB_FLAG_FORCE_MERGE) && !rowwise && elem_search_applicable(a, b))
        return launch_elem_search(a, b, op, swap, out_offsets, out_ids_any, out_data, s);
    return launch_elem_fill(a, b, op, swap, out_offsets, out_ids_any, out_data, rowwise, s);
}
}  // namespace

int sdrb_elem_op_fill(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                      const uint32_t* out_offsets, int32_t* out_ids, float* out_data, void* stream) {
    TraceRange range("sdrb_elem_op_fill");
    return elem_op_fill_any(a, b, op, flags, out_offsets, out_ids, 32, out_data, stream);
}

int sdrb_elem_op_fill64(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                        const uint32_t* out_offsets, int64_t* out_ids, float* out_data, void* stream) {
    TraceRange range("sdrb_elem_op_fill64");
    return elem_op_fill_any(a, b, op, flags, out_offsets, out_ids, 64, out_data, stream);
}

int sdrb_elem_op(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out_offsets,
                 int32_t* out_ids, float* out_data, uint64_t capacity, uint64_t* total_dev, void* stream) {
    TraceRange range("sdrb_elem_op");
    uint64_t pairs = 0;
    int rc = check_count_call(a, b, op, flags, out_offsets, &pairs);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_elem_op");
    if (rc != SDRB_OK) return rc;
    const uint64_t bound = elem_op_bound(a, b, op, flags);
    if (bound > 0xFFFFFFFFull) {
        set_error("the result may hold up to %llu elements, more than 32-bit offsets address: use "
                  "sdrb_elem_op_count / sdrb_elem_op_fill (which apply the limit to the exact total) or split A",
                  (unsigned long long)bound);
        return SDRB_ERR_OVERFLOW;
    }
    if (capacity < bound) {
        set_error("sdrb_elem_op: capacity %llu is below the bound of this call (%llu elements, see sdrb_elem_op_bound)",
                  (unsigned long long)capacity, (unsigned long long)bound);
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (pairs) {
        rc = sdrb_size_op(a, b, op, flags & (SDRB_FLAG_SWAP | SDRB_FLAG_FORCE_MERGE | SDRB_FLAG_ROWWISE | SDRB_FLAG_FORCE_ROWTABLE), out_offsets,
                          b->n_rows, stream);
        if (rc != SDRB_OK) return rc;
    }
    rc = launch_exclusive_scan_u32(b->device, out_offsets, pairs + 1, pairs, reinterpret_cast<unsigned long long*>(total_dev), s);
    if (rc != SDRB_OK || pairs == 0 || bound == 0) return rc;
    return sdrb_elem_op_fill(a, b, op, flags, out_offsets, out_ids, out_data, stream);
}

int sdrb_ande_few(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, uint32_t* out_offsets, int32_t* out_ids,
                  float* out_data, uint64_t capacity, uint64_t* total_dev, void* stream) {
    TraceRange range("sdrb_ande_few");
    int rc = check_pair(a, b, SDRB_OP_AND);
    if (rc == SDRB_OK) rc = check_ids32(a, b, "sdrb_ande_few");
    if (rc == SDRB_OK) rc = check_set_algebra_kinds(a, b, SDRB_OP_AND);
    if (rc != SDRB_OK) return rc;
    if (flags & ~SDRB_FLAG_SWAP) {
        set_error("sdrb_ande_few takes SDRB_FLAG_SWAP only");
        return SDRB_ERR_INVALID_ARG;
    }
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    if (!few_ande_applicable(a, b, swap)) {
        set_error("sdrb_ande_few serves 1..32 rows of a, a declared width <= 4096 on b, rows of b <= 64 ids, EmptyData or "
                  "UnitData payloads");
        return SDRB_ERR_UNSUPPORTED;
    }
    const sdrb_batch* self = swap ? b : a;
    if (!out_offsets || !total_dev || (capacity && (!out_ids || (has_payload(self) && !out_data)))) {
        set_error("sdrb_ande_few: null output (results with payloads need out_data)");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    return launch_few_ande(a, b, swap, out_offsets, out_ids, out_data, capacity,
                           reinterpret_cast<unsigned long long*>(total_dev), static_cast<hipStream_t>(stream));
}

namespace {
int check_visit(const sdrb_batch* a, const sdrb_batch* b, int which) {
    if (!a || !b) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    if (which != SDRB_VISIT_AND && which != SDRB_VISIT_OR) {
        set_error("unknown visitor %d", which);
        return SDRB_ERR_INVALID_ARG;
    }
    return SDRB_OK;
}
}  // namespace

int sdrb_visit_count(const sdrb_batch* a, const sdrb_batch* b, int which, unsigned flags, uint32_t* out_offsets,
                     uint64_t* total, void* stream) {
    TraceRange range("sdrb_visit_count");
    IdHarmony harmony;
    int rc = check_visit(a, b, which);
    if (rc == SDRB_OK) rc = harmonise_ids(a, b, flags, static_cast<hipStream_t>(stream), &harmony);
    if (rc != SDRB_OK) return rc;
    IdView va(a), vb(b);
    // the number of callbacks is the size of the id-only intersection / union
    return sdrb_elem_op_count(&va.v, &vb.v, which == SDRB_VISIT_AND ? SDRB_OP_AND : SDRB_OP_OR,
                              flags & (SDRB_FLAG_SWAP | SDRB_FLAG_ROWWISE | SDRB_FLAG_FORCE_MERGE), out_offsets, total, stream);
}

int sdrb_visit_fill(const sdrb_batch* a, const sdrb_batch* b, int which, unsigned flags, const uint32_t* out_offsets,
                    uint32_t* this_pos, uint32_t* arg_pos, void* stream) {
    TraceRange range("sdrb_visit_fill");
    IdHarmony harmony;
    int rc = check_visit(a, b, which);
    const int op = which == SDRB_VISIT_AND ? SDRB_OP_AND : SDRB_OP_OR;
    if (rc == SDRB_OK) rc = harmonise_ids(a, b, flags, static_cast<hipStream_t>(stream), &harmony);
    if (rc == SDRB_OK) rc = check_pair(a, b, op);
    if (rc != SDRB_OK) return rc;
    if (a->n_rows == 0 || b->n_rows == 0) return SDRB_OK;
    const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
    const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
    if (rowwise && a->n_rows != b->n_rows) {
        set_error("row-wise evaluation needs batches of equal length");
        return SDRB_ERR_INVALID_ARG;
    }
    if (!out_offsets || !this_pos || !arg_pos) {
        set_error("sdrb_visit_fill: null output");
        return SDRB_ERR_INVALID_ARG;
    }
    DeviceGuard guard(b->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", b->device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    IdView va(a), vb(b);
    if (!(flags & (SDRB_FLAG_FORCE_MERGE | SDRB_FLAG_FORCE_SEARCH)) && !rowwise && a->id_bits == 32 && elem_rank_applicable(&va.v, &vb.v))
        return launch_elem_rank(&va.v, &vb.v, op, swap, out_offsets, reinterpret_cast<int32_t*>(this_pos),
                                reinterpret_cast<float*>(arg_pos), s, true);
    if (!(flags & SDRB_FLAG_FORCE_MERGE) && !rowwise && elem_search_applicable(&va.v, &vb.v))
        return launch_elem_search(&va.v, &vb.v, op, swap, out_offsets, this_pos, reinterpret_cast<float*>(arg_pos), s, true);
    if (flags & SDRB_FLAG_FORCE_SEARCH) {
        set_error("the binary-search kernel does not apply to these batches");
        return SDRB_ERR_UNSUPPORTED;
    }
    return launch_elem_visit(&va.v, &vb.v, op, swap, out_offsets, this_pos, arg_pos, rowwise, s);
}

int sdrb_malloc(int device, size_t bytes, void** ptr) {
    if (!ptr) return SDRB_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SDRB_ERR_HIP;
    }
    SDRB_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    return SDRB_OK;
}

int sdrb_free(int device, void* ptr) {
    if (!ptr) return SDRB_OK;
    DeviceGuard guard(device);
    SDRB_HIP(hipFree(ptr));
    return SDRB_OK;
}

int sdrb_memcpy_h2d(int device, void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    DeviceGuard guard(device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bytes) SDRB_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    SDRB_HIP(hipStreamSynchronize(s));
    return SDRB_OK;
}

int sdrb_memcpy_d2h(int device, void* dst_host, const void* src_dev, size_t bytes, void* stream) {
    DeviceGuard guard(device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (bytes) SDRB_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    return SDRB_OK;
}

int sdrb_stream_release(int device, void* stream) {
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return SDRB_ERR_HIP;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    SDRB_HIP(hipStreamSynchronize(s));   // nothing enqueued on the stream reads its scratch any more
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    reap_retired(device);
    for (auto it = g_ws.begin(); it != g_ws.end();) {
        if (it->first.device == device && it->first.stream == s) {
            if (it->second.ptr) (void)hipFree(it->second.ptr);
            it = g_ws.erase(it);
        } else {
            ++it;
        }
    }
    return SDRB_OK;
}

int sdrb_stream_sync(int device, void* stream) {
    DeviceGuard guard(device);
    SDRB_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return SDRB_OK;
}

static int sdrb_synth_csr_impl(uint64_t seed, uint64_t first_row, uint64_t n_rows, int32_t width,
                   int32_t active, uint64_t* row_offsets, int32_t* ids, float* data) {
    if (width <= 0 || active < 0 || active > width || !row_offsets || (!ids && n_rows && active)) {
        set_error("sdrb_synth_csr: bad argument (width=%d active=%d)", width, active);
        return SDRB_ERR_INVALID_ARG;
    }
    constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;
    std::vector<int32_t> perm(width);
    for (int32_t w = 0; w < width; ++w) perm[w] = w;
    std::vector<int32_t> swapped(active);
    for (uint64_t k = 0; k < n_rows; ++k) {
        const uint64_t r = first_row + k;
        uint64_t state = seed ^ (r * kGolden);
        auto next = [&state]() {
            state += kGolden;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            return z ^ (z >> 31);
        };
        for (int32_t i = 0; i < active; ++i) {
            const int32_t jx = i + (int32_t)(next() % (uint64_t)(width - i));
            swapped[i] = jx;
            std::swap(perm[i], perm[jx]);
        }
        int32_t* row = ids + k * (uint64_t)active;
        for (int32_t i = 0; i < active; ++i) row[i] = perm[i];
        // undo the swaps so perm is the identity again for the next row
        for (int32_t i = active - 1; i >= 0; --i) std::swap(perm[i], perm[swapped[i]]);
        std::sort(row, row + active);
        if (data) {
            float* drow = data + k * (uint64_t)active;
            for (int32_t i = 0; i < active; ++i) drow[i] = (float)(next() >> 40) * 0x1p-24f;
        }
        row_offsets[k] = k * (uint64_t)active;
    }
    row_offsets[n_rows] = n_rows * (uint64_t)active;
    return SDRB_OK;
}

int sdrb_synth_csr(uint64_t seed, uint64_t first_row, uint64_t n_rows, int32_t width, int32_t active, uint64_t* row_offsets, int32_t* ids, float* data) {
    return sdrb::guarded("sdrb_synth_csr", [&] { return sdrb_synth_csr_impl(seed, first_row, n_rows, width, active, row_offsets, ids, data); });
}

}  // extern "C"
