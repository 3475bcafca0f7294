// Ingest: host CSR -> HBM.
//
// A large batch crosses the bus first and is checked where it lands: a team of host threads copies pieces of the
// three arrays into a ring of page-locked staging buffers (the offsets narrowed to 32 bits on the way) while the
// calling thread enqueues the transfer of every piece that is ready, in order, over a few copy streams of the
// library's own (one stream pays a fixed cost per transfer that several streams overlap), and two kernels then
// check the rows in HBM (offsets monotone, ids strictly ascending inside the declared
// width, payloads in range).  A refused batch leaves the handle without rows.  Small batches (a query batch per
// step) are checked on the host by the calling thread before anything is copied, and sdrb_csr_check runs the
// same checks on the host alone, with a team of threads when the arrays are large.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr size_t kSlotBytes = 2u << 20;        // one staging buffer; a transfer is at most this long
constexpr size_t kFirstPiece = 1u << 20;       // the first pieces of an upload are shorter: the bus starts sooner
constexpr size_t kRingSlots = 32;              // staging buffers in the ring (64 MiB of page-locked memory)
constexpr int kCopyStreams = 4;
constexpr uint64_t kSmallBytes = 2u << 20;     // batches below this are served by the calling thread alone
constexpr int kMaxThreads = 64;
constexpr int kMaxDevices = 64;

// The ring of page-locked staging buffers, allocated on first use and kept until the process ends, each with an
// event per device that marks the end of the transfer that last read it; and the copy streams of each device.
// One large upload at a time uses them (g_stage_mutex is held for its duration): the bus is shared anyway.
struct StageBuffer {
    void* host = nullptr;
    hipEvent_t done[kMaxDevices] = {nullptr};
};
std::mutex g_stage_mutex;
std::vector<StageBuffer*> g_stage;
hipStream_t g_copy_stream[kMaxDevices][kCopyStreams] = {{nullptr}};
hipEvent_t g_order_event[kMaxDevices][kCopyStreams + 1] = {{nullptr}};

// the ring, its events and the copy streams for `device` (the current device); g_stage_mutex held
int ensure_stage(int device) {
    if (device < 0 || device >= kMaxDevices) {
        set_error("device %d is beyond the staging pool's %d devices", device, kMaxDevices);
        return SDRB_ERR_INVALID_ARG;
    }
    if (g_stage.empty()) {
        // one allocation for the whole ring: page-locking memory costs per call as well as per byte
        void* ring = nullptr;
        const hipError_t e = hipHostMalloc(&ring, kRingSlots * kSlotBytes, hipHostMallocPortable);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu) for the staging ring failed: %s", kRingSlots * kSlotBytes, hipGetErrorString(e));
            return SDRB_ERR_HIP;
        }
        for (size_t k = 0; k < kRingSlots; ++k) {
            StageBuffer* sb = new StageBuffer;
            sb->host = static_cast<char*>(ring) + k * kSlotBytes;
            g_stage.push_back(sb);
        }
    }
    for (size_t k = 0; k < kRingSlots; ++k)
        if (!g_stage[k]->done[device]) SDRB_HIP(hipEventCreateWithFlags(&g_stage[k]->done[device], hipEventDisableTiming));
    for (int k = 0; k < kCopyStreams; ++k)
        if (!g_copy_stream[device][k]) SDRB_HIP(hipStreamCreateWithFlags(&g_copy_stream[device][k], hipStreamNonBlocking));
    for (int k = 0; k <= kCopyStreams; ++k)
        if (!g_order_event[device][k]) SDRB_HIP(hipEventCreateWithFlags(&g_order_event[device][k], hipEventDisableTiming));
    return SDRB_OK;
}

// all threads of a team wait here until the last one has arrived
class Barrier {
 public:
    explicit Barrier(int n) : n_(n) {}
    void wait() {
        std::unique_lock<std::mutex> lock(m_);
        const uint64_t gen = gen_;
        if (++arrived_ == n_) {
            arrived_ = 0;
            ++gen_;
            cv_.notify_all();
        } else {
            cv_.wait(lock, [&] { return gen_ != gen; });
        }
    }

 private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, arrived_ = 0;
    uint64_t gen_ = 0;
};

// Helper threads of one call.  They are created behind a gate: if one cannot be created, those that exist are told
// to leave and are joined before the exception travels on (a thread that is destroyed unjoined ends the process).
class Team {
 public:
    template <typename F>
    void start(int n, F body) {   // body(index), index in [1, n]
        try {
            for (int t = 1; t <= n; ++t)
                threads_.emplace_back([this, body, t] {
                    int g;
                    while ((g = gate_.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
                    if (g > 0) body(t);
                });
        } catch (...) {
            gate_.store(-1, std::memory_order_release);
            join();
            throw;
        }
        gate_.store(1, std::memory_order_release);
    }
    void join() {
        for (std::thread& th : threads_)
            if (th.joinable()) th.join();
    }
    ~Team() {
        if (gate_.load() == 0) gate_.store(-1);
        join();
    }

 private:
    std::vector<std::thread> threads_;
    std::atomic<int> gate_{0};
};

// What a worker found wrong, by the order in which the checks are reported: the first row with broken offsets
// or order, then (by the caller, from the id span) ids outside the width, then the first payload out of range.
struct Findings {
    uint64_t bad_offset_row = UINT64_MAX;
    uint64_t bad_order_row = UINT64_MAX, bad_order_at = 0;   // row, element index inside the batch
    uint64_t bad_payload_at = UINT64_MAX;
    uint32_t max_len = 0;
    int64_t min_id = INT64_MAX, max_id = INT64_MIN;
    int hip_rc = SDRB_OK;
    char hip_msg[256] = {0};
    void merge(const Findings& o) {
        if (o.bad_offset_row < bad_offset_row) bad_offset_row = o.bad_offset_row;
        if (o.bad_order_row < bad_order_row) {
            bad_order_row = o.bad_order_row;
            bad_order_at = o.bad_order_at;
        }
        if (o.bad_payload_at < bad_payload_at) bad_payload_at = o.bad_payload_at;
        if (o.max_len > max_len) max_len = o.max_len;
        if (o.min_id < min_id) min_id = o.min_id;
        if (o.max_id > max_id) max_id = o.max_id;
        if (hip_rc == SDRB_OK && o.hip_rc != SDRB_OK) {
            hip_rc = o.hip_rc;
            std::memcpy(hip_msg, o.hip_msg, sizeof hip_msg);
        }
    }
};

// offsets of rows [r0, r1): monotone, inside the element arrays
void check_offsets(const uint64_t* off, uint64_t r0, uint64_t r1, uint64_t nnz, Findings* f) {
    uint64_t longest = 0, broken = 0;
    for (uint64_t r = r0; r < r1; ++r) {
        const uint64_t lo = off[r], hi = off[r + 1];
        broken += (hi < lo) | (hi > nnz);
        const uint64_t len = hi - lo;
        longest = len > longest ? len : longest;
    }
    if (broken) {
        for (uint64_t r = r0; r < r1; ++r)
            if (off[r + 1] < off[r] || off[r + 1] > nnz) {
                f->bad_offset_row = r;
                return;
            }
    }
    if (longest > f->max_len) f->max_len = (uint32_t)(longest > 0xFFFFFFFFull ? 0xFFFFFFFFull : longest);
}

inline bool payload_ok(float v, int kind) {
    return kind == SDRB_DATA_UNIT ? (v >= 0.0f && v <= 1.0f) : (std::fabs(v) <= 3.402823466e38f);
}

// The flat loops of the element checks, compiled a second time for AVX2 hosts (chosen at load time).
#if defined(__HIP_DEVICE_COMPILE__)
#define SDRB_HOST_SIMD
#else
#define SDRB_HOST_SIMD __attribute__((target_clones("avx2", "default")))
#endif

// descents at positions [k0, k1): ids[k] <= ids[k-1]  (k0 >= 1)
SDRB_HOST_SIMD uint64_t count_descents(const int32_t* ids, uint64_t k0, uint64_t k1) {
    uint64_t descents = 0;
    for (uint64_t k = k0; k < k1; ++k) descents += ids[k] <= ids[k - 1];
    return descents;
}
SDRB_HOST_SIMD uint64_t count_descents(const int64_t* ids, uint64_t k0, uint64_t k1) {
    uint64_t descents = 0;
    for (uint64_t k = k0; k < k1; ++k) descents += ids[k] <= ids[k - 1];
    return descents;
}
SDRB_HOST_SIMD uint64_t count_outside_unit(const float* data, uint64_t e0, uint64_t e1) {
    uint64_t bad = 0;
    for (uint64_t k = e0; k < e1; ++k) bad += !(data[k] >= 0.0f && data[k] <= 1.0f);
    return bad;
}
SDRB_HOST_SIMD uint64_t count_not_finite(const float* data, uint64_t e0, uint64_t e1) {
    uint64_t bad = 0;
    for (uint64_t k = e0; k < e1; ++k) bad += !(std::fabs(data[k]) <= 3.402823466e38f);
    return bad;
}

// Elements of rows [r0, r1) (offsets already checked): order inside every row, id span, payload range.
// The order test runs over the flat element range -- loops without branches, piece by piece (flat_descents,
// flat_bad_payloads) -- and the descents it finds are compared with those that are legitimate because they sit
// at the start of a row (finish_rows).
struct ChunkCheck {
    uint64_t descents = 0, bad_payloads = 0;
};

template <typename Id>
void check_piece(const Id* ids, const float* data, int data_kind, uint64_t e0, uint64_t p0, uint64_t p1, ChunkCheck* c) {
    c->descents += count_descents(ids, p0 > e0 ? p0 : e0 + 1, p1);
    if (data_kind == SDRB_DATA_UNIT) c->bad_payloads += count_outside_unit(data, p0, p1);
    else if (data_kind == SDRB_DATA_ARITH) c->bad_payloads += count_not_finite(data, p0, p1);
}

// true when the chunk holds a finding (recorded in *f if it is the worker's first of its kind)
template <typename Id>
bool finish_rows(const uint64_t* off, const Id* ids, const float* data, int data_kind, uint64_t r0, uint64_t r1,
                 const ChunkCheck& c, Findings* f) {
    const uint64_t e0 = off[r0], e1 = off[r1];
    uint64_t at_row_starts = 0;
    int64_t lowest = f->min_id, highest = f->max_id;
    for (uint64_t r = r0; r < r1; ++r) {
        const uint64_t lo = off[r], hi = off[r + 1];
        if (hi == lo) continue;
        if (lo > e0) at_row_starts += ids[lo] <= ids[lo - 1];
        const int64_t first = ids[lo], last = ids[hi - 1];   // rows ascend: their ends are their extremes
        lowest = first < lowest ? first : lowest;
        highest = last > highest ? last : highest;
    }
    f->min_id = lowest;
    f->max_id = highest;
    bool finding = false;
    if (c.descents != at_row_starts) {
        finding = true;
        for (uint64_t r = r0; r < r1 && f->bad_order_row == UINT64_MAX; ++r)
            for (uint64_t k = off[r] + 1; k < off[r + 1]; ++k)
                if (ids[k] <= ids[k - 1]) {
                    f->bad_order_row = r;
                    f->bad_order_at = k;
                    break;
                }
    }
    if (c.bad_payloads) {
        finding = true;
        for (uint64_t k = e0; k < e1 && f->bad_payload_at == UINT64_MAX; ++k)
            if (!payload_ok(data[k], data_kind)) f->bad_payload_at = k;
    }
    return finding;
}

template <typename Id>
void check_elements(const uint64_t* off, const Id* ids, const float* data, int data_kind, uint64_t r0, uint64_t r1,
                    Findings* f) {
    const uint64_t e0 = off[r0], e1 = off[r1];
    if (e1 == e0) return;
    ChunkCheck c;
    check_piece<Id>(ids, data, data_kind, e0, e0, e1, &c);
    (void)finish_rows<Id>(off, ids, data, data_kind, r0, r1, c, f);
}

int team_size(uint64_t bytes) {
    int n = 0;
    if (const char* env = std::getenv("SDRB_HOST_THREADS")) n = std::atoi(env);
    if (n <= 0) {
        n = (int)std::thread::hardware_concurrency();
        if (n <= 0) n = 1;
        if (n > 16) n = 16;   // a job's share of a GPU host, not the whole machine
    }
    if (n > kMaxThreads) n = kMaxThreads;
    // a thread should have a few staging buffers' worth of work
    const uint64_t useful = bytes / (2 * kSlotBytes) + 1;
    if ((uint64_t)n > useful) n = (int)useful;
    return n;
}

// first row whose start offset is >= e (offsets monotone)
uint64_t row_at(const uint64_t* off, uint64_t n_rows, uint64_t e) {
    uint64_t lo = 0, hi = n_rows;
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (off[mid] < e) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// ---- the host-side checks by a team of threads (sdrb_csr_check, the file writer) ----

struct Job {
    int data_kind;
    uint64_t n_rows, nnz;
    const uint64_t* off;
    const void* ids;
    const float* data;
    int threads;
    uint64_t chunk_elems, n_elem_chunks;
    std::atomic<uint64_t> next{0};
    std::atomic<bool> failed{false};
};

template <typename Id>
void run_checker(Job* job, Findings* found, Barrier* barrier, int tid) {
    // phase 1: the offsets, a contiguous share of the rows per thread
    {
        const uint64_t per = (job->n_rows + job->threads - 1) / job->threads;
        const uint64_t r0 = std::min(job->n_rows, per * tid), r1 = std::min(job->n_rows, r0 + per);
        check_offsets(job->off, r0, r1, job->nnz, found);
        if (found->bad_offset_row != UINT64_MAX) job->failed.store(true);
    }
    barrier->wait();
    if (job->failed.load()) return;   // offsets are not to be trusted: no element is touched
    // phase 2: chunks of rows, handed out in ascending order; after a finding the rest is still checked, so that
    // the FIRST finding of each kind is the one reported
    const Id* ids = static_cast<const Id*>(job->ids);
    for (;;) {
        const uint64_t c = job->next.fetch_add(1);
        if (c >= job->n_elem_chunks) break;
        // rows whose first element lies in [c, c+1) * chunk_elems
        const uint64_t r0 = row_at(job->off, job->n_rows, c * job->chunk_elems);
        const uint64_t r1 = row_at(job->off, job->n_rows, (c + 1) * job->chunk_elems);
        if (r1 > r0) check_elements<Id>(job->off, ids, job->data, job->data_kind, r0, r1, found);
    }
}

// ---- the copy team of an upload ----

// a piece of one of the three arrays: `bytes` of device memory at dst, from src (offsets: narrowed on the way)
struct Piece {
    char* dst;
    const char* src;
    size_t bytes;
    bool narrow;   // src holds uint64 offsets, dst takes uint32
};

// Piece i travels through ring slot i % kRingSlots.  Workers fill slots (piece i may be written once the transfer of
// piece i - kRingSlots has finished: `transferred` counts the leading pieces whose transfers are complete) and mark
// them staged; the calling thread submits staged pieces in order and advances `transferred` by polling the events.
struct CopyJob {
    std::vector<Piece> pieces;
    std::vector<std::atomic<unsigned char>> staged;
    std::atomic<size_t> next{0};
    std::atomic<size_t> transferred{0};
    std::atomic<bool> failed{false};
    std::atomic<uint64_t> offset_high_bits{0};   // OR of the upper halves of the offsets narrowed (must stay 0)
    explicit CopyJob(std::vector<Piece>&& p) : pieces(std::move(p)), staged(pieces.size()) {
        for (auto& f : staged) f.store(0, std::memory_order_relaxed);
    }
};

inline void spin_pause() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
}

// cuts [dst, dst + bytes) into pieces: `small` pieces of kFirstPiece first, then pieces of kSlotBytes
void add_pieces(std::vector<Piece>* out, void* dst, const void* src, size_t bytes, bool narrow, size_t* small) {
    char* d = static_cast<char*>(dst);
    const char* s = static_cast<const char*>(src);
    while (bytes) {
        size_t n = *small ? kFirstPiece : kSlotBytes;
        if (*small) --*small;
        if (n > bytes) n = bytes;
        out->push_back(Piece{d, s, n, narrow});
        d += n;
        s += narrow ? 2 * n : n;
        bytes -= n;
    }
}

void run_stager(CopyJob* job) {
    for (;;) {
        const size_t i = job->next.fetch_add(1);
        if (i >= job->pieces.size()) break;
        while (i >= job->transferred.load(std::memory_order_acquire) + kRingSlots) {
            if (job->failed.load()) return;
            spin_pause();
        }
        if (job->failed.load()) return;
        const Piece& p = job->pieces[i];
        char* host = static_cast<char*>(g_stage[i % kRingSlots]->host);
        if (p.narrow) {
            const uint64_t* wide = reinterpret_cast<const uint64_t*>(p.src);
            uint32_t* out = reinterpret_cast<uint32_t*>(host);
            const size_t n = p.bytes / sizeof(uint32_t);
            uint64_t high = 0;
            for (size_t r = 0; r < n; ++r) {
                out[r] = (uint32_t)wide[r];
                high |= wide[r];
            }
            if (high >> 32) job->offset_high_bits.fetch_or(high >> 32);
        } else {
            std::memcpy(host, p.src, p.bytes);
        }
        job->staged[i].store(1, std::memory_order_release);
    }
}

// the calling thread: every staged piece, in order, onto the copy streams; returns the first HIP error
hipError_t run_submitter(CopyJob* job, int device) {
    const size_t n = job->pieces.size();
    size_t done = 0;
    auto poll = [&](size_t submitted) {
        while (done < submitted) {
            const hipError_t q = hipEventQuery(g_stage[done % kRingSlots]->done[device]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return q;
            ++done;
        }
        job->transferred.store(done, std::memory_order_release);
        return hipSuccess;
    };
    for (size_t i = 0; i < n; ++i) {
        while (!job->staged[i].load(std::memory_order_acquire)) {
            const hipError_t q = poll(i);
            if (q != hipSuccess) return q;
            spin_pause();
        }
        const Piece& p = job->pieces[i];
        hipStream_t cs = g_copy_stream[device][i % kCopyStreams];
        hipError_t e = hipMemcpyAsync(p.dst, g_stage[i % kRingSlots]->host, p.bytes, hipMemcpyHostToDevice, cs);
        if (e == hipSuccess) e = hipEventRecord(g_stage[i % kRingSlots]->done[device], cs);
        if (e != hipSuccess) return e;
        const hipError_t q = poll(i + 1);
        if (q != hipSuccess) return q;
    }
    return hipSuccess;
}

// ---- the checks of an uploaded batch, in HBM ----

struct DeviceFindings {
    unsigned long long descents;        // positions k with ids[k] <= ids[k-1], over the flat element array
    unsigned long long at_row_starts;   // those of them that sit at the start of a row (legitimate)
    unsigned long long bad_payloads;
    unsigned long long bad_offset_row;  // lowest row whose offsets are not monotone / leave the arrays (or ~0)
    unsigned long long min_id, max_id;  // biased by 2^63 so that unsigned order is signed order
    unsigned long long bad_order_row, bad_payload_row;   // filled by k_locate_findings only
    unsigned int max_len;
};

constexpr unsigned long long kBias = 0x8000000000000000ull;

__device__ inline unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ inline unsigned long long wave_min(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ inline unsigned long long wave_max(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(v, d, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ inline bool payload_bad(float v, int kind) {
    return kind == SDRB_DATA_UNIT ? !(v >= 0.0f && v <= 1.0f) : !(fabsf(v) <= 3.402823466e38f);
}

// Sums / extremes of a block: the waves' values meet in LDS and thread 0 returns the block's (others: undefined).
// All threads of the block must call; `part` is a caller-provided LDS array of 4 entries per call site.
__device__ inline unsigned long long block_sum(unsigned long long v, unsigned long long* part) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    return part[0] + part[1] + part[2] + part[3];
}
__device__ inline unsigned long long block_min(unsigned long long v, unsigned long long* part) {
    v = wave_min(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned long long a = part[0] < part[1] ? part[0] : part[1], b = part[2] < part[3] ? part[2] : part[3];
    return a < b ? a : b;
}
__device__ inline unsigned long long block_max(unsigned long long v, unsigned long long* part) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    const unsigned long long a = part[0] > part[1] ? part[0] : part[1], b = part[2] > part[3] ? part[2] : part[3];
    return a > b ? a : b;
}

// Every element once, coalesced, four to a thread (one 16-byte load of ids, one of payloads, and the element before
// the four): descents against the element before, id span, payload range.  A block folds its findings in LDS and
// touches the shared record once (all blocks update the same few words).
template <typename Id>
__global__ __launch_bounds__(256) void k_check_flat(const Id* __restrict__ ids, const float* __restrict__ data, int data_kind,
                                                    uint64_t nnz, DeviceFindings* __restrict__ f) {
    __shared__ unsigned long long part[4][4];
    unsigned long long descents = 0, bad = 0, lo = ~0ull, hi = 0ull;
    const uint64_t quads = nnz / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * 256) {
        const uint64_t k = q * 4;
        Id x[4];
        if constexpr (sizeof(Id) == 4) {
            const int4 v = *reinterpret_cast<const int4*>(ids + k);   // hipMalloc'ed arrays: 16-byte aligned at multiples of four
            x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
        } else {
            const longlong2 v0 = *reinterpret_cast<const longlong2*>(ids + k), v1 = *reinterpret_cast<const longlong2*>(ids + k + 2);
            x[0] = v0.x; x[1] = v0.y; x[2] = v1.x; x[3] = v1.y;
        }
        Id before = k > 0 ? ids[k - 1] : x[0];
        if (k > 0) descents += x[0] <= before;
#pragma unroll
        for (int u = 1; u < 4; ++u) descents += x[u] <= x[u - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned long long w = (unsigned long long)(long long)x[u] ^ kBias;
            lo = w < lo ? w : lo;
            hi = w > hi ? w : hi;
        }
        if (data_kind != SDRB_DATA_EMPTY) {
            const float4 d = *reinterpret_cast<const float4*>(data + k);
            bad += payload_bad(d.x, data_kind) + payload_bad(d.y, data_kind) + payload_bad(d.z, data_kind) + payload_bad(d.w, data_kind);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t)(nnz - quads * 4)) {   // the last one to three elements
        const uint64_t k = quads * 4 + threadIdx.x;
        const Id x = ids[k];
        if (k > 0) descents += x <= ids[k - 1];
        const unsigned long long w = (unsigned long long)(long long)x ^ kBias;
        lo = w < lo ? w : lo;
        hi = w > hi ? w : hi;
        if (data_kind != SDRB_DATA_EMPTY) bad += payload_bad(data[k], data_kind);
    }
    descents = block_sum(descents, part[0]);
    bad = block_sum(bad, part[1]);
    lo = block_min(lo, part[2]);
    hi = block_max(hi, part[3]);
    if (threadIdx.x == 0) {   // one record a block, folded by k_fold_findings: no two blocks write the same word
        DeviceFindings* mine = f + 1 + blockIdx.x;
        mine->descents = descents;
        mine->bad_payloads = bad;
        mine->min_id = lo;
        mine->max_id = hi;
    }
}

// every row once: offsets monotone and inside the arrays, longest row, descents that sit at a row start
template <typename Id>
__global__ __launch_bounds__(256) void k_check_rows(const uint32_t* __restrict__ off, const Id* __restrict__ ids, uint64_t n_rows,
                                                    uint64_t nnz, DeviceFindings* __restrict__ f) {
    __shared__ unsigned long long part[3][4];
    unsigned long long starts = 0, bad_row = ~0ull, len = 0;
    for (uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x; r < n_rows; r += (uint64_t)gridDim.x * 256) {
        const uint64_t lo = off[r], hi = off[r + 1];
        if (hi < lo || hi > nnz) {
            bad_row = r < bad_row ? r : bad_row;
        } else if (hi > lo) {
            len = hi - lo > len ? hi - lo : len;
            if (lo > 0) starts += ids[lo] <= ids[lo - 1];
        }
    }
    starts = block_sum(starts, part[0]);
    bad_row = block_min(bad_row, part[1]);
    len = block_max(len, part[2]);
    if (threadIdx.x == 0) {
        DeviceFindings* mine = f + 1 + blockIdx.x;
        mine->at_row_starts = starts;
        mine->bad_offset_row = bad_row;
        mine->max_len = (unsigned int)len;
    }
}

// records that change nothing when folded: sums 0, minima ~0, maxima 0
__global__ __launch_bounds__(256) void k_neutral_findings(DeviceFindings* __restrict__ f, uint32_t n) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n) return;
    f[b].bad_offset_row = f[b].min_id = f[b].bad_order_row = f[b].bad_payload_row = ~0ull;
}

// folds the blocks' records f[1 .. n] into f[0] (one block)
__global__ __launch_bounds__(256) void k_fold_findings(DeviceFindings* __restrict__ f, uint32_t n) {
    __shared__ unsigned long long part[7][4];
    unsigned long long descents = 0, starts = 0, bad = 0, bad_row = ~0ull, lo = ~0ull, hi = 0ull, len = 0;
    for (uint32_t b = threadIdx.x; b < n; b += 256) {
        const DeviceFindings& r = f[1 + b];
        descents += r.descents;
        starts += r.at_row_starts;
        bad += r.bad_payloads;
        bad_row = r.bad_offset_row < bad_row ? r.bad_offset_row : bad_row;
        lo = r.min_id < lo ? r.min_id : lo;
        hi = r.max_id > hi ? r.max_id : hi;
        len = r.max_len > len ? r.max_len : len;
    }
    descents = block_sum(descents, part[0]);
    starts = block_sum(starts, part[1]);
    bad = block_sum(bad, part[2]);
    bad_row = block_min(bad_row, part[3]);
    lo = block_min(lo, part[4]);
    hi = block_max(hi, part[5]);
    len = block_max(len, part[6]);
    if (threadIdx.x == 0) {
        f->descents = descents;
        f->at_row_starts = starts;
        f->bad_payloads = bad;
        f->bad_offset_row = bad_row;
        f->min_id = lo;
        f->max_id = hi;
        f->max_len = (unsigned int)len;
        f->bad_order_row = f->bad_payload_row = ~0ull;
    }
}

// after a finding: the lowest row that is not strictly ascending, the lowest row with a payload out of range
template <typename Id>
__global__ __launch_bounds__(256) void k_locate_findings(const uint32_t* __restrict__ off, const Id* __restrict__ ids,
                                                         const float* __restrict__ data, int data_kind, uint64_t n_rows,
                                                         DeviceFindings* __restrict__ f) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    bool order_ok = true, payload_ok = true;
    for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
        if (k > off[r] && ids[k] <= ids[k - 1]) order_ok = false;
        if (data_kind != SDRB_DATA_EMPTY && payload_bad(data[k], data_kind)) payload_ok = false;
    }
    if (!order_ok) atomicMin(&f->bad_order_row, (unsigned long long)r);
    if (!payload_ok) atomicMin(&f->bad_payload_row, (unsigned long long)r);
}

// Checks the batch's arrays in HBM (n_rows rows, nnz elements as the host offsets state) and turns what the kernels
// found into Findings, looking the details of a finding up in the host arrays.  Synchronises the stream.
template <typename Id>
int check_on_device(const sdrb_batch* batch, uint64_t n_rows, uint64_t nnz, const uint64_t* off, const Id* ids,
                    const float* data, hipStream_t s, Findings* out) {
    // f[0]: the batch's record; f[1 ..]: one record per block of the two kernels (the same blocks' slots serve both:
    // they write different fields), neutral where a kernel has fewer blocks
    const uint64_t flat_blocks = nnz ? std::min<uint64_t>(4096, (nnz / 4 + 256 * 4 - 1) / (256 * 4) + 1) : 0;
    const uint64_t row_blocks = n_rows ? std::min<uint64_t>(4096, (n_rows + 256 * 4 - 1) / (256 * 4)) : 0;
    const uint32_t slots = (uint32_t)std::max<uint64_t>(std::max(flat_blocks, row_blocks), 1);
    DeviceFindings* d_f = static_cast<DeviceFindings*>(workspace(batch->device, s, 2, (size_t)(slots + 1) * sizeof(DeviceFindings)));
    if (!d_f) return SDRB_ERR_HIP;
    DeviceFindings f{};
    const Id* d_ids = reinterpret_cast<const Id*>(batch->d_ids);
    const int kind = batch->data_kind;
    // neutral records: zero sums, ~0 for the minima (0xFF bytes), then the maxima cleared by the kernels that own them
    SDRB_HIP(hipMemsetAsync(d_f, 0, (size_t)(slots + 1) * sizeof(DeviceFindings), s));
    hipLaunchKernelGGL(k_neutral_findings, dim3((slots + 1 + 255) / 256), dim3(256), 0, s, d_f, slots + 1);
    SDRB_HIP(hipGetLastError());
    if (nnz) {
        hipLaunchKernelGGL((k_check_flat<Id>), dim3((uint32_t)flat_blocks), dim3(256), 0, s, d_ids, batch->d_data, kind, nnz, d_f);
        SDRB_HIP(hipGetLastError());
    }
    if (n_rows) {
        hipLaunchKernelGGL((k_check_rows<Id>), dim3((uint32_t)row_blocks), dim3(256), 0, s, batch->d_offsets, d_ids, n_rows, nnz, d_f);
        SDRB_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(k_fold_findings, dim3(1), dim3(256), 0, s, d_f, slots);
    SDRB_HIP(hipGetLastError());
    SDRB_HIP(hipMemcpyAsync(&f, d_f, sizeof f, hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    out->max_len = f.max_len;
    if (f.min_id <= f.max_id) {
        out->min_id = (int64_t)(f.min_id ^ kBias);
        out->max_id = (int64_t)(f.max_id ^ kBias);
    }
    if (f.bad_offset_row != ~0ull) {
        out->bad_offset_row = f.bad_offset_row;
        return SDRB_OK;
    }
    if (f.descents == f.at_row_starts && f.bad_payloads == 0) return SDRB_OK;
    hipLaunchKernelGGL((k_locate_findings<Id>), dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, s, batch->d_offsets, d_ids,
                       batch->d_data, kind, n_rows, d_f);
    SDRB_HIP(hipGetLastError());
    SDRB_HIP(hipMemcpyAsync(&f, d_f, sizeof f, hipMemcpyDeviceToHost, s));
    SDRB_HIP(hipStreamSynchronize(s));
    if (f.bad_order_row != ~0ull) {
        const uint64_t r = f.bad_order_row;
        for (uint64_t k = off[r] + 1; k < off[r + 1]; ++k)
            if (ids[k] <= ids[k - 1]) {
                out->bad_order_row = r;
                out->bad_order_at = k;
                break;
            }
    }
    if (f.bad_payload_row != ~0ull) {
        const uint64_t r = f.bad_payload_row;
        for (uint64_t k = off[r]; k < off[r + 1]; ++k)
            if (!payload_ok(data[k], kind)) {
                out->bad_payload_at = k;
                break;
            }
    }
    return SDRB_OK;
}

// The calling thread alone: check everything, then copy (nothing on the device is touched by a refused batch).
template <typename Id>
void check_serial(uint64_t n_rows, uint64_t nnz, const uint64_t* off, const Id* ids, const float* data, int data_kind,
                  Findings* f) {
    check_offsets(off, 0, n_rows, nnz, f);
    if (f->bad_offset_row != UINT64_MAX) return;
    check_elements<Id>(off, ids, data, data_kind, 0, n_rows, f);
}

// turns the findings into the status code and message of the upload; 0 when the rows are fine
template <typename Id>
int verdict(const Findings& f, int32_t width, int data_kind, const uint64_t* off, const Id* ids, const float* data) {
    if (f.bad_offset_row != UINT64_MAX) {
        set_error("row_offsets not monotone at row %llu", (unsigned long long)f.bad_offset_row);
        return SDRB_ERR_INVALID_ARG;
    }
    if (f.bad_order_row != UINT64_MAX) {
        const uint64_t k = f.bad_order_at;
        set_error("row %llu is not strictly ascending at element %llu (%lld after %lld)",
                  (unsigned long long)f.bad_order_row, (unsigned long long)(k - off[f.bad_order_row]), (long long)ids[k],
                  (long long)ids[k - 1]);
        return SDRB_ERR_NOT_ASCENDING;
    }
    if (width > 0 && f.min_id <= f.max_id && (f.min_id < 0 || f.max_id >= width)) {
        set_error("id outside the declared width: ids span [%lld, %lld], width %d", (long long)f.min_id,
                  (long long)f.max_id, width);
        return SDRB_ERR_ID_RANGE;
    }
    if (f.bad_payload_at != UINT64_MAX) {
        const uint64_t k = f.bad_payload_at;
        if (data_kind == SDRB_DATA_ARITH) set_error("ArithData payload at element %llu is not finite", (unsigned long long)k);
        else set_error("UnitData payload %g at element %llu is outside [0,1]", (double)data[k], (unsigned long long)k);
        return SDRB_ERR_PAYLOAD_RANGE;
    }
    return SDRB_OK;
}

// ids[k] (64 bits) -> out[k] (32 bits), for the narrowed twin of a batch whose ids all fit
__global__ __launch_bounds__(256) void k_narrow_ids(const int64_t* __restrict__ ids, int32_t* __restrict__ out, uint64_t n) {
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < n) out[k] = (int32_t)ids[k];
}

// the host-side checks of large arrays by a team of threads; the findings of all of them end up in *all
template <typename Id>
void check_by_team(int data_kind, uint64_t n_rows, const uint64_t* off, const Id* ids, const float* data, uint64_t bytes,
                   Findings* all) {
    Job job;
    job.data_kind = data_kind;
    job.n_rows = n_rows;
    job.nnz = off[n_rows];
    job.off = off;
    job.ids = ids;
    job.data = data;
    job.threads = team_size(bytes);
    job.chunk_elems = kSlotBytes / sizeof(Id);
    job.n_elem_chunks = (job.nnz + job.chunk_elems - 1) / job.chunk_elems;
    std::vector<Findings> found(job.threads);
    Barrier barrier(job.threads);
    Team team;
    Job* jp = &job;
    Findings* fp = found.data();
    Barrier* bp = &barrier;
    team.start(job.threads - 1, [jp, fp, bp](int t) { run_checker<Id>(jp, fp + t, bp, t); });
    run_checker<Id>(&job, &found[0], &barrier, 0);
    team.join();
    for (const Findings& f : found) all->merge(f);
}

// Copies the three arrays into the batch's storage (reserved by the caller, its device current) through the ring of
// staging buffers.  The transfers run on the library's copy streams, after what `s` held before the call; on return
// `s` waits for them (they are enqueued, not waited for by the host).
bool page_locked(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();   // ordinary host memory is an error to this call on some runtimes: not one of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

template <typename Id>
int copy_by_team(sdrb_batch* batch, uint64_t n_rows, const uint64_t* off, const Id* ids, const float* data, uint64_t bytes,
                 hipStream_t s) {
    const uint64_t nnz = off[n_rows];
    const int device = batch->device;
    // eight threads stage faster than the bus carries (measured: 16 gain nothing and cost their start-up)
    const int threads = std::min(team_size(bytes), 9);
    std::vector<Piece> pieces;
    size_t small = (size_t)threads;
    // An array the caller already holds in page-locked memory needs no staging: it crosses the bus in one copy of its
    // own (the offsets are always staged: they are narrowed on the way).  The upload returns after its checks have run
    // on the device, which is after these copies.
    const bool ids_direct = nnz != 0 && page_locked(ids), data_direct = has_payload(batch) && nnz != 0 && page_locked(data);
    if (!ids_direct) add_pieces(&pieces, batch->d_ids, ids, nnz * sizeof(Id), false, &small);
    if (has_payload(batch) && !data_direct) add_pieces(&pieces, batch->d_data, data, nnz * sizeof(float), false, &small);
    add_pieces(&pieces, batch->d_offsets, off, (n_rows + 1) * sizeof(uint32_t), true, &small);
    CopyJob job(std::move(pieces));
    std::lock_guard<std::mutex> stage_lock(g_stage_mutex);
    const int rc = ensure_stage(device);
    if (rc != SDRB_OK) return rc;
    // the copy streams start after the work already enqueued on s (it may still read the storage written here)
    SDRB_HIP(hipEventRecord(g_order_event[device][kCopyStreams], s));
    for (int k = 0; k < kCopyStreams; ++k) SDRB_HIP(hipStreamWaitEvent(g_copy_stream[device][k], g_order_event[device][kCopyStreams], 0));
    if (ids_direct) SDRB_HIP(hipMemcpyAsync(batch->d_ids, ids, nnz * sizeof(Id), hipMemcpyHostToDevice, g_copy_stream[device][kCopyStreams - 1]));
    if (data_direct) SDRB_HIP(hipMemcpyAsync(batch->d_data, data, nnz * sizeof(float), hipMemcpyHostToDevice, g_copy_stream[device][kCopyStreams - 2]));
    Team team;
    CopyJob* jp = &job;
    team.start(threads > 1 ? threads - 1 : 1, [jp](int) { run_stager(jp); });
    const hipError_t e = run_submitter(&job, device);
    if (e != hipSuccess) job.failed.store(true);
    team.join();
    // s continues after the transfers
    hipError_t tail = hipSuccess;
    for (int k = 0; k < kCopyStreams && tail == hipSuccess; ++k) {
        tail = hipEventRecord(g_order_event[device][k], g_copy_stream[device][k]);
        if (tail == hipSuccess) tail = hipStreamWaitEvent(s, g_order_event[device][k], 0);
    }
    if (e != hipSuccess || tail != hipSuccess) {
        for (int k = 0; k < kCopyStreams; ++k) (void)hipStreamSynchronize(g_copy_stream[device][k]);
        set_error("upload: a transfer failed: %s", hipGetErrorString(e != hipSuccess ? e : tail));
        return SDRB_ERR_HIP;
    }
    if (job.offset_high_bits.load()) {
        // an offset beyond 2^32 cannot be told from a small one once narrowed: the host names the row
        for (int k = 0; k < kCopyStreams; ++k) (void)hipStreamSynchronize(g_copy_stream[device][k]);
        Findings f;
        check_offsets(off, 0, n_rows, nnz, &f);
        set_error("row_offsets not monotone at row %llu", (unsigned long long)f.bad_offset_row);
        return SDRB_ERR_INVALID_ARG;
    }
    return SDRB_OK;
}

// the checks of an upload without a device: status code and message of the first finding
template <typename Id>
int check_rows(uint64_t n_rows, const uint64_t* off, const Id* ids, const float* data, int32_t width, int data_kind,
               Findings* all) {
    const uint64_t nnz = off[n_rows];
    const uint64_t bytes = nnz * (sizeof(Id) + (data_kind != SDRB_DATA_EMPTY ? 4 : 0)) + (n_rows + 1) * 4;
    if (bytes < kSmallBytes) {
        check_serial<Id>(n_rows, nnz, off, ids, data, data_kind, all);
    } else {
        check_by_team<Id>(data_kind, n_rows, off, ids, data, bytes, all);
    }
    return verdict<Id>(*all, width, data_kind, off, ids, data);
}

template <typename Id>
int upload_rows(sdrb_batch* batch, uint64_t n_rows, const uint64_t* off, const Id* ids, const float* data,
                hipStream_t s, Findings* out) {
    const uint64_t nnz = off[n_rows];
    const bool payload = has_payload(batch);
    const uint64_t bytes = nnz * (sizeof(Id) + (payload ? 4 : 0)) + (n_rows + 1) * 4;
    Findings all;
    if (bytes < kSmallBytes) {
        // check first, then copy: nothing on the device is touched by a refused batch
        int rc = check_rows<Id>(n_rows, off, ids, data, declared_width(batch), batch->data_kind, &all);
        if (rc != SDRB_OK) return rc;
        DeviceGuard guard(batch->device);
        if (!guard.ok) {
            set_error("hipSetDevice(%d) failed", batch->device);
            return SDRB_ERR_HIP;
        }
        rc = reserve_batch_storage(batch, n_rows, nnz);
        if (rc != SDRB_OK) return rc;
        std::vector<uint32_t> off32(n_rows + 1);
        for (uint64_t r = 0; r <= n_rows; ++r) off32[r] = (uint32_t)off[r];
        SDRB_HIP(hipMemcpyAsync(batch->d_offsets, off32.data(), (n_rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        if (nnz) SDRB_HIP(hipMemcpyAsync(batch->d_ids, ids, nnz * sizeof(Id), hipMemcpyHostToDevice, s));
        if (nnz && payload) SDRB_HIP(hipMemcpyAsync(batch->d_data, data, nnz * sizeof(float), hipMemcpyHostToDevice, s));
        SDRB_HIP(hipStreamSynchronize(s));
        *out = all;
        return SDRB_OK;
    }

    DeviceGuard guard(batch->device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", batch->device);
        return SDRB_ERR_HIP;
    }
    int rc = reserve_batch_storage(batch, n_rows, nnz);
    if (rc != SDRB_OK) return rc;
    rc = copy_by_team<Id>(batch, n_rows, off, ids, data, bytes, s);
    if (rc == SDRB_OK) rc = check_on_device<Id>(batch, n_rows, nnz, off, ids, data, s, &all);
    else (void)hipStreamSynchronize(s);
    if (rc == SDRB_OK) rc = verdict<Id>(all, declared_width(batch), batch->data_kind, off, ids, data);
    if (rc != SDRB_OK) {
        clear_batch_rows(batch);   // parts of the refused rows may have reached the device: the batch holds none
        return rc;
    }
    *out = all;
    return SDRB_OK;
}

}  // namespace

int check_host_csr(uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids, const float* data, int32_t width,
                   int data_kind) {
    Findings f;
    return check_rows<int32_t>(n_rows, row_offsets, ids, data, width, data_kind, &f);
}

}  // namespace sdrb

using namespace sdrb;

static int upload_csr_impl(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids,
                           const float* data, void* stream) {
    if (!batch || !row_offsets) {
        set_error("sdrb_batch_upload_csr: null argument");
        return SDRB_ERR_INVALID_ARG;
    }
    if (batch->id_bits != 32) {
        set_error("this batch holds 64-bit ids: use sdrb_batch_upload_csr64");
        return SDRB_ERR_INVALID_ARG;
    }
    if (row_offsets[0] != 0) {
        set_error("row_offsets[0] must be 0");
        return SDRB_ERR_INVALID_ARG;
    }
    const uint64_t nnz = row_offsets[n_rows];
    if (nnz > 0xFFFFFFFFull || n_rows > 0xFFFFFFFEull) {
        set_error("batch too large for 32-bit offsets (%llu rows, %llu elements)", (unsigned long long)n_rows,
                  (unsigned long long)nnz);
        return SDRB_ERR_OVERFLOW;
    }
    if (nnz > 0 && !ids) {
        set_error("ids is null");
        return SDRB_ERR_INVALID_ARG;
    }
    if (has_payload(batch) && nnz > 0 && !data) {
        set_error("a batch with payloads needs a payload array");
        return SDRB_ERR_INVALID_ARG;
    }
    Findings f;
    const int rc = upload_rows<int32_t>(batch, n_rows, row_offsets, ids, data, static_cast<hipStream_t>(stream), &f);
    if (rc != SDRB_OK) return rc;
    const bool any = f.min_id <= f.max_id;
    batch->n_rows = n_rows;
    batch->nnz = nnz;
    batch->max_row_len = f.max_len;
    batch->min_id = any ? (int32_t)f.min_id : 0;
    batch->max_id = any ? (int32_t)f.max_id : -1;
    settle_width(batch);
    return SDRB_OK;
}

static int upload_csr64_impl(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets, const int64_t* ids,
                             const float* data, void* stream) {
    if (!batch || !row_offsets || batch->id_bits != 64) {
        set_error("sdrb_batch_upload_csr64: null argument or not a 64-bit batch");
        return SDRB_ERR_INVALID_ARG;
    }
    if (row_offsets[0] != 0) {
        set_error("row_offsets[0] must be 0");
        return SDRB_ERR_INVALID_ARG;
    }
    const uint64_t nnz = row_offsets[n_rows];
    if (nnz > 0xFFFFFFFFull || n_rows > 0xFFFFFFFEull) {
        set_error("batch too large for 32-bit offsets");
        return SDRB_ERR_OVERFLOW;
    }
    const bool payload = has_payload(batch);
    if (nnz && (!ids || (payload && !data))) {
        set_error("ids (and payloads of a UnitData batch) must not be null");
        return SDRB_ERR_INVALID_ARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    Findings f;
    const int rc = upload_rows<int64_t>(batch, n_rows, row_offsets, ids, data, s, &f);
    if (rc != SDRB_OK) return rc;
    batch->n_rows = n_rows;
    batch->nnz = nnz;
    batch->max_row_len = f.max_len;
    batch->min_id = 0;
    batch->max_id = -1;
    // ids that all fit 32 bits (the reference's SDRElem<long> rows usually do): keep them narrowed as well
    const bool any = f.min_id <= f.max_id;
    batch->min_id64 = any ? (int64_t)f.min_id : 0;
    batch->max_id64 = any ? (int64_t)f.max_id : -1;
    if (!any || (f.min_id >= INT32_MIN && f.max_id < INT32_MAX)) {
        DeviceGuard guard(batch->device);
        if (nnz) {
            SDRB_HIP(hipMalloc(reinterpret_cast<void**>(&batch->d_narrow_ids), nnz * sizeof(int32_t)));
            hipLaunchKernelGGL(k_narrow_ids, dim3((uint32_t)((nnz + 255) / 256)), dim3(256), 0, s,
                               reinterpret_cast<const int64_t*>(batch->d_ids), batch->d_narrow_ids, nnz);
            SDRB_HIP(hipGetLastError());
            SDRB_HIP(hipStreamSynchronize(s));
        }
        sdrb_batch* twin = new sdrb_batch;
        twin->device = batch->device;
        twin->data_kind = batch->data_kind;
        twin->id_bits = 32;
        twin->width = (any && f.min_id >= 0) ? (int32_t)(f.max_id + 1) : 0;
        twin->n_rows = n_rows;
        twin->nnz = nnz;
        twin->max_row_len = f.max_len;
        twin->min_id = any ? (int32_t)f.min_id : 0;
        twin->max_id = any ? (int32_t)f.max_id : -1;
        twin->d_offsets = batch->d_offsets;
        twin->d_ids = batch->d_narrow_ids;
        twin->d_data = batch->d_data;
        twin->owns_storage = false;
        batch->narrow = twin;
    }
    return SDRB_OK;
}

extern "C" {

int sdrb_csr_check(uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids, const float* data, int32_t width,
                   int data_kind) {
    return sdrb::guarded("sdrb_csr_check", [&] {
        if (!row_offsets || width < 0 ||
            (data_kind != SDRB_DATA_EMPTY && data_kind != SDRB_DATA_UNIT && data_kind != SDRB_DATA_ARITH)) {
            set_error("sdrb_csr_check: bad argument");
            return SDRB_ERR_INVALID_ARG;
        }
        if (row_offsets[0] != 0) {
            set_error("row_offsets[0] must be 0");
            return SDRB_ERR_INVALID_ARG;
        }
        const uint64_t nnz = row_offsets[n_rows];
        if (nnz && (!ids || (data_kind != SDRB_DATA_EMPTY && !data))) {
            set_error("ids (and payloads of a batch that has them) must not be null");
            return SDRB_ERR_INVALID_ARG;
        }
        return check_host_csr(n_rows, row_offsets, ids, data, width, data_kind);
    });
}

int sdrb_batch_upload_csr(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids,
                          const float* data, void* stream) {
    TraceRange range("sdrb_batch_upload_csr");
    return sdrb::guarded("sdrb_batch_upload_csr", [&] { return upload_csr_impl(batch, n_rows, row_offsets, ids, data, stream); });
}

int sdrb_batch_upload_csr64(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets, const int64_t* ids,
                            const float* data, void* stream) {
    TraceRange range("sdrb_batch_upload_csr64");
    return sdrb::guarded("sdrb_batch_upload_csr64", [&] { return upload_csr64_impl(batch, n_rows, row_offsets, ids, data, stream); });
}

}  // extern "C"
