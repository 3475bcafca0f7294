// SDRBatch.hpp -- header-only C++17 host side of the MI355X batched SDR engine.
//
// Sits next to the reference's SparseDistributedRepresentation/SDR.hpp: it packs any range of
// SDR objects (anything with cbegin()/cend() over elements that have id() and data(), i.e. the
// reference's SDR<SDRElem<Id,Data>, Container> for every container it supports, including
// IDContiguousContainer) into a device-resident CSR batch and evaluates the reference's
// binary operations for every pair (this[i], db[j]) on the GPU through the C ABI of
// sdr_amd.h.  Operation names and meaning are the reference's (SDR.hpp:151-318):
//
//     std::vector<SDR<>> queries = ..., rows = ...;
//     auto q  = sdr_amd::SDRBatch::pack(queries.begin(), queries.end(), /*device*/0, /*width*/2048);
//     auto db = sdr_amd::SDRBatch::pack(rows.begin(), rows.end(), 0, 2048);
//     std::vector<uint32_t> counts = q.ands(db).to_host();   // counts[i*db.size()+j] == queries[i].ands(rows[j])
//     auto r = q.ore(db);                                    // CSR of queries[i].ore(rows[j])
//     SDR<> first = sdr_amd::Elements::unpack<SDR<>>(r.to_host(), db.size(), 0, 0);   // == queries[0].ore(rows[0])
//
// Element types: ids of 32 or 64 bits (SDRElem<int,...>, SDRElem<long,...>); payloads EmptyData,
// UnitData or ArithData<float>.  Besides the binary operations: inner (sparse dot products), the
// row-wise / in-place forms andi ori xori rmi, range and point queries, transpose, all-pairs top-k,
// and database files (save / load_rows / SDRBatch::from_file).
//
// Link with -lsdr_amd (no HIP headers needed here).  Errors become sdr_amd::Error.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../sdr_amd.h"

namespace sdr_amd {

class Error : public std::runtime_error {
 public:
    Error(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
    int code() const { return code_; }

 private:
    int code_;
};

inline void check(int rc) {
    if (rc != SDRB_OK) throw Error(rc, sdrb_last_error());
}

namespace detail {

// A payload without value() is EmptyData (EmptyData.hpp:18-92).  One with a float value() is
// UnitData (UnitData.hpp:11-140) or ArithData<float> (ArithData.hpp:11-107); the two are told
// apart by what defines them here: a zero payload is relevant for ArithData (ArithData.hpp:22-24)
// and not for UnitData (UnitData.hpp:27-29).
template <typename D, typename = void>
struct has_value : std::false_type {};
template <typename D>
struct has_value<D, std::void_t<decltype(std::declval<const D&>().value())>> : std::true_type {};

template <typename D>
int kind_of() {
    if constexpr (!has_value<D>::value) {
        return SDRB_DATA_EMPTY;
    } else {
        static_assert(std::is_same_v<std::decay_t<decltype(std::declval<const D&>().value())>, float>,
                      "payload values must be float (UnitData or ArithData<float>)");
        return D(0.0f).relevant() ? SDRB_DATA_ARITH : SDRB_DATA_UNIT;
    }
}

template <typename SDR_t>
using elem_of = typename SDR_t::value_type;
template <typename SDR_t>
using data_of = typename elem_of<SDR_t>::data_type;
template <typename SDR_t>
using id_of = typename elem_of<SDR_t>::id_type;
template <typename SDR_t>
constexpr bool wide_ids = sizeof(id_of<SDR_t>) > 4;

// number of elements of an SDR: its size() where it has one (SDR.hpp:343-349), else the length of its range
template <typename S, typename = void>
struct has_size : std::false_type {};
template <typename S>
struct has_size<S, std::void_t<decltype(std::declval<const S&>().size())>> : std::true_type {};
template <typename S>
size_t row_len(const S& s) {
    if constexpr (has_size<S>::value) return static_cast<size_t>(s.size());
    else return static_cast<size_t>(std::distance(s.cbegin(), s.cend()));
}

// Rows whose elements lie in memory one after the other as bare ids: SDRElem<int, EmptyData> is four bytes, its id
// (SDRElem.hpp:78-79, EmptyData.hpp:89-92), so a std::vector of them -- the reference's default container,
// SDR.hpp:26 -- is a packed int array (the layout README.md:185-202 lets a caller rely on) and a row is copied
// with one memcpy instead of element by element.
template <typename SDR_t>
constexpr bool bare_id_rows() {
    using Elem = elem_of<SDR_t>;
    using It = decltype(std::declval<const SDR_t&>().cbegin());
    return !has_value<data_of<SDR_t>>::value && sizeof(Elem) == sizeof(id_of<SDR_t>) && std::is_standard_layout_v<Elem> &&
           (std::is_pointer_v<It> || std::is_same_v<It, typename std::vector<Elem>::const_iterator>);
}

// host threads for packing: SDRB_HOST_THREADS, else the machine's, at most 16
inline int host_threads(size_t rows) {
    int n = 0;
    if (const char* env = std::getenv("SDRB_HOST_THREADS")) n = std::atoi(env);
    if (n <= 0) {
        n = static_cast<int>(std::thread::hardware_concurrency());
        if (n <= 0) n = 1;
        if (n > 16) n = 16;
    }
    const size_t useful = rows / 4096 + 1;   // a thread should have a few thousand rows
    return static_cast<size_t>(n) > useful ? static_cast<int>(useful) : n;
}

// runs body(t) for t in [0, threads) on that many threads (the caller's among them)
template <typename F>
void on_threads(int threads, F&& body) {
    std::vector<std::thread> team;
    team.reserve(threads > 0 ? threads - 1 : 0);
    for (int t = 1; t < threads; ++t) team.emplace_back([&body, t] { body(t); });
    body(0);
    for (std::thread& th : team) th.join();
}

struct DeviceBuffer {
    int device = 0;
    void* ptr = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(int dev, size_t n) : device(dev), bytes(n) { check(sdrb_malloc(dev, n, &ptr)); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : device(o.device), ptr(o.ptr), bytes(o.bytes) { o.ptr = nullptr; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        if (this != &o) {
            release();
            device = o.device; ptr = o.ptr; bytes = o.bytes; o.ptr = nullptr;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) sdrb_free(device, ptr);
        ptr = nullptr;
    }
    template <typename T>
    std::vector<T> to_host(size_t count, void* stream = nullptr) const {
        std::vector<T> h(count);
        check(sdrb_memcpy_d2h(device, h.data(), ptr, count * sizeof(T), stream));
        return h;
    }
};

}  // namespace detail

// Streams the SDRs of [first, last) into a CSR database file, `chunk_rows` at a time (the reference
// has no serialisation, SDR.hpp:1670-1680 prints text only).  Rows are read by iteration, so any
// container works; payloads with a float value() are stored.
template <typename Iter>
void save(Iter first, Iter last, const std::string& path, int32_t width = 0, size_t chunk_rows = 65536) {
    using SDR_t = std::decay_t<decltype(*first)>;
    constexpr bool unit = detail::has_value<detail::data_of<SDR_t>>::value;
    static_assert(!detail::wide_ids<SDR_t>, "database files hold 32-bit ids");
    sdrb_file_writer* w = nullptr;
    check(sdrb_file_writer_open(path.c_str(), width, detail::kind_of<detail::data_of<SDR_t>>(), &w));
    std::vector<uint64_t> off;
    std::vector<int32_t> ids;
    std::vector<float> data;
    auto flush = [&]() {
        const int rc = sdrb_file_writer_append(w, off.size() - 1, off.data(), ids.data(), unit ? data.data() : nullptr);
        if (rc != SDRB_OK) {
            const std::string msg = sdrb_last_error();
            sdrb_file_writer_abort(w);
            throw Error(rc, msg);
        }
    };
    while (first != last) {
        off.assign(1, 0);
        ids.clear();
        data.clear();
        for (size_t n = 0; n < chunk_rows && first != last; ++n, ++first) {
            for (auto it = first->cbegin(); it != first->cend(); ++it) {
                ids.push_back(static_cast<int32_t>(it->id()));
                if constexpr (unit) data.push_back(static_cast<float>(it->data().value()));
            }
            off.push_back(ids.size());
        }
        flush();
    }
    check(sdrb_file_writer_close(w));
}

// Rows [row_lo, row_hi) of a CSR database file as SDR objects of the caller's type (row_hi = 0: to the end).
template <typename SDR_t>
std::vector<SDR_t> load_rows(const std::string& path, uint64_t row_lo = 0, uint64_t row_hi = 0) {
    using Elem = detail::elem_of<SDR_t>;
    using Data = detail::data_of<SDR_t>;
    uint64_t n_rows = 0, span = 0;
    int kind = SDRB_DATA_EMPTY;
    check(sdrb_file_info(path.c_str(), &n_rows, nullptr, nullptr, &kind));
    if (row_hi == 0) row_hi = n_rows;
    check(sdrb_file_row_span(path.c_str(), row_lo, row_hi, &span));
    std::vector<uint64_t> off(row_hi - row_lo + 1);
    std::vector<int32_t> ids(span ? span : 1);
    std::vector<float> data(kind != SDRB_DATA_EMPTY ? (span ? span : 1) : 0);
    check(sdrb_file_read_rows(path.c_str(), row_lo, row_hi, off.data(), ids.data(), data.empty() ? nullptr : data.data()));
    std::vector<SDR_t> rows(row_hi - row_lo);
    for (size_t r = 0; r < rows.size(); ++r)
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
            if constexpr (detail::has_value<Data>::value) {
                rows[r].push_back(data.empty() ? Elem(ids[k]) : Elem(ids[k], Data(data[k])));
            } else {
                rows[r].push_back(Elem(ids[k]));
            }
        }
    return rows;
}

// T table[na][nb] in device memory
template <typename T>
class Table {
 public:
    Table(int device, size_t na, size_t nb) : na_(na), nb_(nb), buf_(device, (na * nb ? na * nb : 1) * sizeof(T)) {}
    size_t rows() const { return na_; }
    size_t cols() const { return nb_; }
    T* device_ptr() const { return static_cast<T*>(buf_.ptr); }
    std::vector<T> to_host(void* stream = nullptr) const { return buf_.to_host<T>(na_ * nb_, stream); }

 private:
    size_t na_, nb_;
    detail::DeviceBuffer buf_;
};
// Sizes of results as uint8_t or uint16_t (sdrb_size_op_narrow): rows of the table are `ld` counts apart, ld the next
// multiple of four of the number of columns.
template <typename T>
class NarrowCounts {
 public:
    NarrowCounts(int device, size_t na, size_t nb)
        : na_(na), nb_(nb), ld_((nb + 3) / 4 * 4), buf_(device, (na * ld_ ? na * ld_ : 4) * sizeof(T)) {}
    size_t rows() const { return na_; }
    size_t cols() const { return nb_; }
    size_t ld() const { return ld_; }
    T* device_ptr() const { return static_cast<T*>(buf_.ptr); }
    // counts[i * cols() + j], the padding dropped
    std::vector<T> to_host(void* stream = nullptr) const {
        const std::vector<T> padded = buf_.to_host<T>(na_ * ld_, stream);
        std::vector<T> h(na_ * nb_);
        for (size_t i = 0; i < na_; ++i) std::copy(padded.begin() + i * ld_, padded.begin() + i * ld_ + nb_, h.begin() + i * nb_);
        return h;
    }

 private:
    size_t na_, nb_, ld_;
    detail::DeviceBuffer buf_;
};

using Counts = Table<uint32_t>;     // sizes of results
using Values = Table<float>;        // dot products
using Positions = Table<int32_t>;   // positions inside rows, -1 = absent

// Per row of the left batch: the k rows of the right batch with the largest id overlap
// (overlap descending, then row ascending; unused slots hold count 0, row -1)
class TopK {
 public:
    TopK(int device, size_t na, size_t k)
        : na_(na), k_(k), counts_(device, (na * k ? na * k : 1) * sizeof(uint32_t)), rows_(device, (na * k ? na * k : 1) * sizeof(int32_t)) {}
    size_t rows() const { return na_; }
    size_t k() const { return k_; }
    uint32_t* counts_ptr() const { return static_cast<uint32_t*>(counts_.ptr); }
    int32_t* rows_ptr() const { return static_cast<int32_t*>(rows_.ptr); }
    std::vector<uint32_t> counts_to_host(void* stream = nullptr) const { return counts_.to_host<uint32_t>(na_ * k_, stream); }
    std::vector<int32_t> rows_to_host(void* stream = nullptr) const { return rows_.to_host<int32_t>(na_ * k_, stream); }

 private:
    size_t na_, k_;
    detail::DeviceBuffer counts_, rows_;
};

// What the reference's visitors are called with (andv / orv), for all pairs: pair (i, j) owns the records
// offsets[i*nb+j] .. offsets[i*nb+j+1], each a (position in this, position in arg); SDRB_POS_NONE marks the
// absent side of an orv single.
class Visits {
 public:
    size_t na = 0, nb = 0;
    uint64_t total = 0;
    detail::DeviceBuffer offsets, this_pos, arg_pos;

    struct Host {
        std::vector<uint32_t> offsets, this_pos, arg_pos;
    };
    // The reference's five-callback orv (SDR.hpp:284-285, 1099-1150) stops calling the single-side visitors once one
    // side is exhausted and hands the rest of the other side over in ONE call: arg_only_position(arg position) when
    // `this` runs out (it is asked first, so also when both end together), else this_only_position(this position).
    // For pair `p` of an orv result, given the lengths of the two rows: the records [offsets[p], head_end) are the
    // visitor calls made before that, then comes the one position call.
    struct OrvTail {
        uint32_t head_end;      // index into this_pos / arg_pos: the records from here on are the handed-over tail
        bool arg_only;          // true: arg_only_position(position), false: this_only_position(position)
        uint32_t position;      // where the tail starts in its row (may be the row's end)
    };
    static OrvTail orv_tail(const Host& h, size_t p, uint32_t this_len, uint32_t arg_len) {
        uint32_t k = h.offsets[p], seen_this = 0, seen_arg = 0;
        const uint32_t end = h.offsets[p + 1];
        // the merge asks `this` first, then `arg`; after a call it asks the side(s) it has just advanced
        if (this_len == 0) return {k, true, 0};
        if (arg_len == 0) return {k, false, 0};
        for (; k < end; ++k) {
            const bool has_this = h.this_pos[k] != SDRB_POS_NONE, has_arg = h.arg_pos[k] != SDRB_POS_NONE;
            seen_this += has_this;
            seen_arg += has_arg;
            if (has_this && seen_this == this_len) return {k + 1, true, seen_arg};
            if (has_arg && seen_arg == arg_len) return {k + 1, false, seen_this};
        }
        return {end, true, seen_arg};   // not reached for rows of the stated lengths
    }
    Host to_host(void* stream = nullptr) const {
        Host h;
        h.offsets = offsets.to_host<uint32_t>(na * nb + 1, stream);
        h.this_pos = this_pos.to_host<uint32_t>(total, stream);
        h.arg_pos = arg_pos.to_host<uint32_t>(total, stream);
        return h;
    }
};

// CSR result of an element op over all pairs
class Elements {
 public:
    size_t na = 0, nb = 0;
    uint64_t total = 0;
    bool has_data = false;
    int id_bits = 32;                         // ids are int32, or int64 for batches of 64-bit ids
    detail::DeviceBuffer offsets, ids, data;  // row-wise results: na = 1, one result per column

    struct Host {
        std::vector<uint32_t> offsets;
        std::vector<int64_t> ids;             // widened on the host so that one unpack serves both id types
        std::vector<float> data;
    };
    Host to_host(void* stream = nullptr) const {
        Host h;
        h.offsets = offsets.to_host<uint32_t>(na * nb + 1, stream);
        if (id_bits == 64) {
            h.ids = ids.to_host<int64_t>(total, stream);
        } else {
            const std::vector<int32_t> narrow = ids.to_host<int32_t>(total, stream);
            h.ids.assign(narrow.begin(), narrow.end());
        }
        if (has_data) h.data = data.to_host<float>(total, stream);
        return h;
    }
    // Rebuilds result (i, j) as an SDR object of the caller's type.
    template <typename SDR_t>
    static SDR_t unpack(const Host& h, size_t nb, size_t i, size_t j) {
        using Elem = detail::elem_of<SDR_t>;
        using Data = detail::data_of<SDR_t>;
        using Id = detail::id_of<SDR_t>;
        SDR_t r;
        for (uint32_t k = h.offsets[i * nb + j]; k < h.offsets[i * nb + j + 1]; ++k) {
            if constexpr (detail::has_value<Data>::value) {
                r.push_back(h.data.empty() ? Elem(static_cast<Id>(h.ids[k])) : Elem(static_cast<Id>(h.ids[k]), Data(h.data[k])));
            } else {
                r.push_back(Elem(static_cast<Id>(h.ids[k])));
            }
        }
        return r;
    }
};

// An element-op result too long for one CSR with 32-bit offsets (more than 2^32-1 elements): rows
// starts[c] .. starts[c+1] of A are held by chunks[c], an Elements over those rows of A and all rows of B.
class ChunkedElements {
 public:
    size_t na = 0, nb = 0;
    std::vector<size_t> starts;
    std::vector<Elements> chunks;
    uint64_t total() const {
        uint64_t t = 0;
        for (const Elements& e : chunks) t += e.total;
        return t;
    }
    // which chunk holds row i of A, and which of the chunk's rows it is
    std::pair<size_t, size_t> locate(size_t i) const {
        size_t c = 0;
        while (c + 1 < chunks.size() && starts[c + 1] <= i) ++c;
        return {c, i - starts[c]};
    }
};

class SDRBatch {
 public:
    SDRBatch(int device, int32_t width, int data_kind, int id_bits = 32)
        : device_(device), kind_(data_kind), id_bits_(id_bits) {
        if (id_bits == 64) check(sdrb_batch_create64(device, data_kind, &h_));
        else check(sdrb_batch_create(device, width, data_kind, &h_));
    }
    SDRBatch(const SDRBatch&) = delete;
    SDRBatch& operator=(const SDRBatch&) = delete;
    SDRBatch(SDRBatch&& o) noexcept
        : h_(o.h_), device_(o.device_), n_rows_(o.n_rows_), kind_(o.kind_), id_bits_(o.id_bits_),
          view_data_(std::move(o.view_data_)) { o.h_ = nullptr; }
    ~SDRBatch() { sdrb_batch_destroy(h_); }

    // Host CSR of a range of SDRs: what pack() uploads.  The arrays are not value-initialised, so that the
    // threads that fill them are the first to touch their pages.
    template <typename Id>
    struct HostRows {
        size_t n_rows = 0;
        uint64_t nnz = 0;
        std::unique_ptr<uint64_t[]> off;
        std::unique_ptr<Id[]> ids;
        std::unique_ptr<float[]> data;   // null without payloads
    };

    // The SDRs of [first, last) as host CSR, read by iterating each one (the portable way to read any container,
    // IDContiguousContainer.hpp:263-303 included).  With random-access iterators over the SDRs the work is done
    // in two passes -- lengths, then elements -- by a team of host threads, each filling the rows of a contiguous
    // share; a std::vector of bare 4-byte ids is copied row by row with memcpy.  Other iterators are walked once.
    template <typename Iter>
    static auto pack_host(Iter first, Iter last) {
        using SDR_t = std::decay_t<decltype(*first)>;
        using Data = detail::data_of<SDR_t>;
        constexpr bool payload = detail::has_value<Data>::value;
        using Id = std::conditional_t<detail::wide_ids<SDR_t>, int64_t, int32_t>;
        HostRows<Id> h;
        constexpr bool random_access =
            std::is_base_of_v<std::random_access_iterator_tag, typename std::iterator_traits<Iter>::iterator_category>;
        if constexpr (random_access) {
            const size_t n = static_cast<size_t>(last - first);
            h.n_rows = n;
            h.off.reset(new uint64_t[n + 1]);
            const int threads = detail::host_threads(n);
            const size_t share = (n + threads - 1) / threads;
            detail::on_threads(threads, [&](int t) {
                const size_t r0 = std::min(n, share * t), r1 = std::min(n, r0 + share);
                for (size_t r = r0; r < r1; ++r) h.off[r + 1] = detail::row_len(first[r]);
            });
            h.off[0] = 0;
            for (size_t r = 0; r < n; ++r) h.off[r + 1] += h.off[r];
            h.nnz = h.off[n];
            h.ids.reset(new Id[h.nnz ? h.nnz : 1]);
            if constexpr (payload) h.data.reset(new float[h.nnz ? h.nnz : 1]);
            detail::on_threads(threads, [&](int t) {
                const size_t r0 = std::min(n, share * t), r1 = std::min(n, r0 + share);
                for (size_t r = r0; r < r1; ++r) {
                    const SDR_t& row = first[r];
                    Id* out = h.ids.get() + h.off[r];
                    if constexpr (detail::bare_id_rows<SDR_t>() && sizeof(Id) == sizeof(detail::id_of<SDR_t>)) {
                        const size_t len = static_cast<size_t>(h.off[r + 1] - h.off[r]);
                        if (len) std::memcpy(out, static_cast<const void*>(&*row.cbegin()), len * sizeof(Id));
                    } else {
                        float* pay = payload ? h.data.get() + h.off[r] : nullptr;
                        for (auto it = row.cbegin(); it != row.cend(); ++it) {
                            *out++ = static_cast<Id>(it->id());
                            if constexpr (payload) *pay++ = static_cast<float>(it->data().value());
                        }
                    }
                }
            });
        } else {
            std::vector<uint64_t> off{0};
            std::vector<Id> ids;
            std::vector<float> data;
            for (; first != last; ++first) {
                for (auto it = first->cbegin(); it != first->cend(); ++it) {
                    ids.push_back(static_cast<Id>(it->id()));
                    if constexpr (payload) data.push_back(static_cast<float>(it->data().value()));
                }
                off.push_back(ids.size());
            }
            h.n_rows = off.size() - 1;
            h.nnz = ids.size();
            h.off.reset(new uint64_t[off.size()]);
            std::memcpy(h.off.get(), off.data(), off.size() * sizeof(uint64_t));
            h.ids.reset(new Id[ids.size() ? ids.size() : 1]);
            if (!ids.empty()) std::memcpy(h.ids.get(), ids.data(), ids.size() * sizeof(Id));
            if constexpr (payload) {
                h.data.reset(new float[data.size() ? data.size() : 1]);
                if (!data.empty()) std::memcpy(h.data.get(), data.data(), data.size() * sizeof(float));
            }
        }
        return h;
    }

    // Packs the SDRs of [first, last) into a device batch (pack_host, then one upload that checks the rows
    // and moves them through page-locked staging).  width = 0: arbitrary ids.
    // Elements with ids wider than 32 bits (SDRElem<long, ...>) make a batch of 64-bit ids.
    template <typename Iter>
    static SDRBatch pack(Iter first, Iter last, int device = 0, int32_t width = 0, void* stream = nullptr) {
        using SDR_t = std::decay_t<decltype(*first)>;
        using Data = detail::data_of<SDR_t>;
        constexpr bool wide = detail::wide_ids<SDR_t>;
        const auto h = pack_host(first, last);
        SDRBatch b(device, width, detail::kind_of<Data>(), wide ? 64 : 32);
        if constexpr (wide) check(sdrb_batch_upload_csr64(b.h_, h.n_rows, h.off.get(), h.ids.get(), h.data.get(), stream));
        else check(sdrb_batch_upload_csr(b.h_, h.n_rows, h.off.get(), h.ids.get(), h.data.get(), stream));
        b.n_rows_ = h.n_rows;
        return b;
    }

    // Rows [row_lo, row_hi) of a CSR database file (see save()) as a device batch; row_hi = 0: to the end.
    static SDRBatch from_file(const std::string& path, int device = 0, uint64_t row_lo = 0, uint64_t row_hi = 0,
                              void* stream = nullptr) {
        uint64_t n_rows = 0;
        int32_t width = 0;
        int kind = SDRB_DATA_EMPTY;
        check(sdrb_file_info(path.c_str(), &n_rows, nullptr, &width, &kind));
        if (row_hi == 0) row_hi = n_rows;
        SDRBatch b(device, width, kind);
        check(sdrb_batch_upload_file(b.h_, path.c_str(), row_lo, row_hi, stream));
        b.n_rows_ = row_hi - row_lo;
        return b;
    }

    void upload(const std::vector<uint64_t>& off, const std::vector<int32_t>& ids,
                const std::vector<float>& data, void* stream = nullptr) {
        check(sdrb_batch_upload_csr(h_, off.size() - 1, off.data(), ids.data(),
                                    data.empty() ? nullptr : data.data(), stream));
        n_rows_ = off.size() - 1;
    }

    size_t size() const { return n_rows_; }
    int device() const { return device_; }
    int data_kind() const { return kind_; }
    int id_bits() const { return id_bits_; }
    const sdrb_batch* handle() const { return h_; }

    // out(i, j) = (*this)[i].ands(db[j]) ... (SDR.hpp:175-176,229-230,257-258,317-318)
    Counts ands(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_AND, db, flags, stream); }
    Counts ors(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_OR, db, flags, stream); }
    Counts xors(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_XOR, db, flags, stream); }
    Counts rms(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_RM, db, flags, stream); }

    // The same sizes as uint8_t or uint16_t counts (T): for results that depend on ids alone (`this` EmptyData, or ors)
    // on rows of at most 64 ids over a declared width of at most 4096; refused -- never truncated -- when the longest
    // rows allow a value that T cannot hold.  op: SDRB_OP_AND / OR / XOR / RM.
    template <typename T>
    NarrowCounts<T> size_op_narrow(int op, const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const {
        static_assert(std::is_same_v<T, uint8_t> || std::is_same_v<T, uint16_t>, "narrow counts are uint8_t or uint16_t");
        NarrowCounts<T> c(db.device_, n_rows_, db.n_rows_);
        check(sdrb_size_op_narrow(h_, db.h_, op, flags, c.device_ptr(), c.ld(), (int)sizeof(T), stream));
        return c;
    }

    // out(i, j) = (*this)[i].inner(db[j]) on ArithData<float> rows (SDR.hpp:1682-1692), bit for bit.
    // These are also the entries of diff_mul (SDR.hpp:1886-1923; *this = matrix rows, db = the columns
    // of the other matrix) and of row_major_mul_vec (SDR.hpp:1729-1750; *this = the vector, db = matrix rows).
    Values inner(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const {
        Values v(db.device_, n_rows_, db.n_rows_);
        check(sdrb_inner(h_, db.h_, flags, v.device_ptr(), db.n_rows_, stream));
        return v;
    }

    // For every row i: the k rows j of db with the largest |(*this)[i] ∩ db[j]| -- the value ands returns
    // on id-only elements (SDR.hpp:1083-1095); pair (i, i + exclude_offset) is skipped when exclude_offset >= 0.
    // The reference has no top-k: ties go to the lower row.
    TopK allpairs_topk(const SDRBatch& db, uint32_t k, int64_t exclude_offset = -1, unsigned flags = 0,
                       void* stream = nullptr) const {
        TopK t(db.device_, n_rows_, k);
        check(sdrb_allpairs_topk(h_, db.h_, k, exclude_offset, flags, t.counts_ptr(), t.rows_ptr(), stream));
        return t;
    }

    // (*this)[i].ande(db[j]) ... as CSR (SDR.hpp:151-152,209-212,237-240,292-293)
    Elements ande(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return elem_op(SDRB_OP_AND, db, flags, stream); }
    Elements ore(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return elem_op(SDRB_OP_OR, db, flags, stream); }
    Elements xore(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return elem_op(SDRB_OP_XOR, db, flags, stream); }
    Elements rme(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return elem_op(SDRB_OP_RM, db, flags, stream); }

    // The same rows under another payload type, sharing this batch's offsets and ids on the device (the view must
    // not outlive *this, nor an upload into it).  ids_only: payloads dropped.  as_unit (EmptyData batches): every
    // element carries UnitData(EmptyData) = 1 (UnitData.hpp:13-14, SDRElem.hpp:29-30).  Batches of 32-bit ids.
    SDRBatch ids_only() const { return view(SDRB_DATA_EMPTY, nullptr); }
    SDRBatch as_unit(void* stream = nullptr) const {
        if (kind_ != SDRB_DATA_EMPTY) throw Error(SDRB_ERR_INVALID_ARG, "as_unit converts EmptyData batches");
        return view(SDRB_DATA_UNIT, stream);
    }

    // Element ops with the reference's result-type argument, a.ande<ret_t, c_ret_t>(b) (SDR.hpp:151-152, 209-212,
    // 237-240, 292-293).  ret = SDRB_DATA_EMPTY: a match's data is converted to EmptyData before its relevance is
    // asked, so payloads play no part (EmptyData.hpp:27-37: intersections keep every common id, xore and rme drop
    // every common id).  ret = SDRB_DATA_UNIT on an EmptyData `this`: a match gets EmptyData::op<UnitData>() =
    // UnitData() = 1 and is kept (EmptyData.hpp:44-45, 53-54, 64-65), elements of `this` alone become
    // UnitData(EmptyData) = 1, elements of the argument alone keep their payload -- for ore and for xore alike;
    // rme has no such form (EmptyData::rme returns EmptyData whatever is asked for, EmptyData.hpp:73-74).
    // ret = this batch's own kind: the plain op.
    Elements elem_op_as(int ret, int op, const SDRBatch& db, void* stream = nullptr) const {
        if (ret == kind_) return elem_op(op, db, 0, stream);
        if (ret == SDRB_DATA_EMPTY) {
            const SDRBatch mine = ids_only(), theirs = db.ids_only();
            Elements e = mine.elem_op(op, theirs, 0, stream);
            check(sdrb_stream_sync(device_, stream));   // the views end here
            return e;
        }
        if (ret != SDRB_DATA_UNIT || kind_ != SDRB_DATA_EMPTY)
            throw Error(SDRB_ERR_UNSUPPORTED, "result types: EmptyData from any batch, UnitData from an EmptyData or UnitData batch");
        if (op == SDRB_OP_RM) throw Error(SDRB_ERR_UNSUPPORTED, "rme of an EmptyData batch yields EmptyData whatever result type is asked for");
        if (op == SDRB_OP_AND) {
            const SDRBatch theirs = db.ids_only();
            Elements e = elem_op(op, theirs, 0, stream);
            check(sdrb_stream_sync(device_, stream));
            const std::vector<float> ones(e.total ? e.total : 1, 1.0f);
            e.data = detail::DeviceBuffer(db.device_, ones.size() * sizeof(float));
            check(sdrb_memcpy_h2d(db.device_, e.data.ptr, ones.data(), ones.size() * sizeof(float), stream));
            e.has_data = true;
            return e;
        }
        const SDRBatch mine = as_unit(stream);
        Elements e = mine.elem_op(SDRB_OP_OR, db, 0, stream);
        check(sdrb_stream_sync(device_, stream));
        return e;
    }

    // rmi with the reference's rmi_combiner (SDR.hpp:304-310, 1447-1513), row by row: for every id that row j of
    // *this shares with row j of arg, the element's payload becomes combine(this_value, arg_value) -- arg_value is
    // a null pointer when arg is EmptyData -- and the element is removed when the new payload is not rm_relevant
    // (UnitData: below 0.1, UnitData.hpp:22-28; ArithData: never, ArithData.hpp:22-28); other elements stay.  The
    // combiner runs on the host over the matches andv reports (ascending ids within a row).  Batches of 32-bit ids.
    template <typename Combine>
    SDRBatch& rmi(const SDRBatch& arg, Combine combine, void* stream = nullptr) {
        if (kind_ == SDRB_DATA_EMPTY) throw Error(SDRB_ERR_INVALID_ARG, "a combiner needs payloads on this batch");
        if (id_bits_ != 32 || arg.id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "rmi with a combiner serves batches of 32-bit ids");
        const Visits::Host hits = visit(SDRB_VISIT_AND, arg, SDRB_FLAG_ROWWISE, stream).to_host(stream);
        Elements::Host mine = to_host(stream);
        Elements::Host theirs;
        if (arg.kind_ != SDRB_DATA_EMPTY) theirs = arg.to_host(stream);
        std::vector<char> keep(mine.ids.size(), 1);
        for (size_t j = 0; j < n_rows_; ++j)
            for (uint32_t k = hits.offsets[j]; k < hits.offsets[j + 1]; ++k) {
                const size_t at = mine.offsets[j] + hits.this_pos[k];
                const float* arg_value = theirs.data.empty() ? nullptr : &theirs.data[theirs.offsets[j] + hits.arg_pos[k]];
                const float v = combine(mine.data[at], arg_value);
                mine.data[at] = v;
                if (kind_ == SDRB_DATA_UNIT && !(v >= 0.1f)) keep[at] = 0;   // binary32 compare: the reference's double compare gives the same verdict
            }
        std::vector<uint64_t> off{0};
        std::vector<int32_t> ids;
        std::vector<float> data;
        for (size_t j = 0; j < n_rows_; ++j) {
            for (uint32_t k = mine.offsets[j]; k < mine.offsets[j + 1]; ++k)
                if (keep[k]) {
                    ids.push_back((int32_t)mine.ids[k]);
                    data.push_back(mine.data[k]);
                }
            off.push_back(ids.size());
        }
        check(sdrb_batch_upload_csr(h_, n_rows_, off.data(), ids.data(), data.empty() ? nullptr : data.data(), stream));
        return *this;
    }

    // The same four results when they may exceed 2^32-1 elements (e.g. 64 rows of 328 ids in union with 128 Ki
    // such rows): the rows of *this are cut into consecutive ranges whose results fit, one CSR each.  op is
    // SDRB_OP_AND / OR / XOR / RM; `limit` (elements per CSR) exists for tests.  Batches of 32-bit ids.
    ChunkedElements elem_op_chunked(int op, const SDRBatch& db, unsigned flags = 0, void* stream = nullptr,
                                    uint64_t limit = 0xFFFFFFFFull) const {
        if (id_bits_ != 32 || db.id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "chunked results serve batches of 32-bit ids");
        if (flags & SDRB_FLAG_ROWWISE) throw Error(SDRB_ERR_INVALID_ARG, "row-wise results never need chunks");
        ChunkedElements out;
        out.na = n_rows_;
        out.nb = db.n_rows_;
        const uint32_t* d_off = nullptr;
        const int32_t* d_ids = nullptr;
        const float* d_data = nullptr;
        check(sdrb_batch_device_ptrs(h_, &d_off, &d_ids, &d_data));
        std::vector<uint32_t> off(n_rows_ + 1);
        check(sdrb_memcpy_d2h(device_, off.data(), d_off, off.size() * sizeof(uint32_t), stream));
        uint64_t b_nnz = 0;
        check(sdrb_batch_info(db.h_, nullptr, nullptr, &b_nnz, nullptr, nullptr, nullptr));
        // the most elements row i of *this can contribute (the bound sdrb_elem_op_bound uses, row by row)
        const bool swap = (flags & SDRB_FLAG_SWAP) != 0;
        auto row_bound = [&](size_t i) -> uint64_t {
            const uint64_t from_a = (uint64_t)(off[i + 1] - off[i]) * db.n_rows_;
            if (op == SDRB_OP_AND) return from_a < b_nnz ? from_a : b_nnz;
            if (op == SDRB_OP_RM) return swap ? b_nnz : from_a;
            return from_a + b_nnz;
        };
        out.starts.push_back(0);
        uint64_t acc = 0;
        for (size_t i = 0; i < n_rows_; ++i) {
            const uint64_t r = row_bound(i);
            if (r > limit) throw Error(SDRB_ERR_OVERFLOW, "one row against all rows of the other batch exceeds the limit: split the other batch");
            if (acc + r > limit) {
                out.starts.push_back(i);
                acc = 0;
            }
            acc += r;
        }
        out.starts.push_back(n_rows_);
        uint32_t max_len = 0;
        int32_t width = 0, lo_id = 0, hi_id = -1;
        check(sdrb_batch_info(h_, nullptr, nullptr, nullptr, &width, nullptr, &max_len));
        check(sdrb_batch_id_span(h_, &lo_id, &hi_id));
        for (size_t c = 0; c + 1 < out.starts.size(); ++c) {
            const size_t i0 = out.starts[c], i1 = out.starts[c + 1];
            std::vector<uint32_t> sub(off.begin() + i0, off.begin() + i1 + 1);
            for (uint32_t& o : sub) o -= off[i0];
            detail::DeviceBuffer sub_off(device_, sub.size() * sizeof(uint32_t));
            check(sdrb_memcpy_h2d(device_, sub_off.ptr, sub.data(), sub.size() * sizeof(uint32_t), stream));
            SDRBatch part(device_, width, kind_);
            check(sdrb_batch_adopt_device_csr(part.h_, i1 - i0, off[i1] - off[i0], static_cast<const uint32_t*>(sub_off.ptr),
                                              d_ids + off[i0], d_data ? d_data + off[i0] : nullptr, max_len, lo_id, hi_id));
            part.n_rows_ = i1 - i0;
            out.chunks.push_back(part.elem_op(op, db, flags, stream));
            check(sdrb_stream_sync(device_, stream));   // the view's offsets are released at the end of this turn
        }
        return out;
    }

    // Row-wise forms for two batches of equal length: result j = (*this)[j].op(arg[j]).
    // op is SDRB_OP_AND / OR / XOR / RM; sizes come as a 1 x n table, elements as a CSR with one result per row.
    Counts rowwise_size(int op, const SDRBatch& arg, unsigned flags = 0, void* stream = nullptr) const {
        Counts c(arg.device_, 1, arg.n_rows_);
        check(sdrb_size_op(h_, arg.h_, op, flags | SDRB_FLAG_ROWWISE, c.device_ptr(), arg.n_rows_, stream));
        return c;
    }
    Elements rowwise_elem(int op, const SDRBatch& arg, unsigned flags = 0, void* stream = nullptr) const {
        return elem_op(op, arg, flags | SDRB_FLAG_ROWWISE, stream);
    }

    // In-place forms: every row j becomes row.andi(arg[j]) ... (SDR.hpp:1027-1081,1239-1292,1366-1422,
    // 1436-1513).  `arg` may be *this (a.op(a) is legal in the reference, src/test.cpp:241-272).
    SDRBatch& andi(const SDRBatch& arg, void* stream = nullptr) { return inplace(SDRB_OP_AND, arg, stream); }
    SDRBatch& ori(const SDRBatch& arg, void* stream = nullptr) { return inplace(SDRB_OP_OR, arg, stream); }
    SDRBatch& xori(const SDRBatch& arg, void* stream = nullptr) { return inplace(SDRB_OP_XOR, arg, stream); }
    SDRBatch& rmi(const SDRBatch& arg, void* stream = nullptr) { return inplace(SDRB_OP_RM, arg, stream); }

    // out(i, j) = ((*this)[i] == db[j]) as 1 / 0 -- SDR::operator== (SDR.hpp:470-483); SDRB_FLAG_ROWWISE: out(0, j) for row j of both
    Counts equals(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const {
        Counts c(db.device_, (flags & SDRB_FLAG_ROWWISE) ? 1 : n_rows_, db.n_rows_);
        check(sdrb_equal(h_, db.h_, flags, c.device_ptr(), db.n_rows_, stream));
        return c;
    }

    // Batched visitors (the reference's extension API, README.md:133-155): the calls that
    // (*this)[i].andv(db[j], visitor) (SDR.hpp:186-194) and (*this)[i].orv(db[j], this_v, arg_v, both_v)
    // (SDR.hpp:268-269) make, as arrays of element positions -- the raw material for custom operations.
    Visits andv(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return visit(SDRB_VISIT_AND, db, flags, stream); }
    Visits orv(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return visit(SDRB_VISIT_OR, db, flags, stream); }

    // The reference's operator sugar (SDR.hpp:431-460), batch against batch: & * -> ande, && -> ands,
    // | + -> ore, || -> ors, ^ -> xore, - -> rme for every pair; &= *= |= += ^= -= -> the in-place forms row by row.
    Elements operator&(const SDRBatch& o) const { return ande(o); }
    Elements operator*(const SDRBatch& o) const { return ande(o); }
    Counts operator&&(const SDRBatch& o) const { return ands(o); }
    Elements operator|(const SDRBatch& o) const { return ore(o); }
    Elements operator+(const SDRBatch& o) const { return ore(o); }
    Counts operator||(const SDRBatch& o) const { return ors(o); }
    Elements operator^(const SDRBatch& o) const { return xore(o); }
    Elements operator-(const SDRBatch& o) const { return rme(o); }
    SDRBatch& operator&=(const SDRBatch& o) { return andi(o); }
    SDRBatch& operator*=(const SDRBatch& o) { return andi(o); }
    SDRBatch& operator|=(const SDRBatch& o) { return ori(o); }
    SDRBatch& operator+=(const SDRBatch& o) { return ori(o); }
    SDRBatch& operator^=(const SDRBatch& o) { return xori(o); }
    SDRBatch& operator-=(const SDRBatch& o) { return rmi(o); }
    // SDR::shift (SDR.hpp:1579-1590) for every row: ids += amount, in place; << >> give a shifted copy, <<= >>= shift in
    // place (SDR.hpp:461-468).  Refused (Error, SDRB_ERR_ID_RANGE) when an id would leave 32 bits or the declared width.
    SDRBatch& shift(int amount, void* stream = nullptr) {
        check(sdrb_batch_shift(h_, amount, stream));
        return *this;
    }
    SDRBatch& operator<<=(int amount) { return shift(amount); }
    SDRBatch& operator>>=(int amount) { return shift(-amount); }
    SDRBatch operator<<(int amount) const { SDRBatch c = copy(); c.shift(amount); return c; }
    SDRBatch operator>>(int amount) const { SDRBatch c = copy(); c.shift(-amount); return c; }
    // a batch of its own with the same rows (device-to-device; batches of 32-bit ids)
    SDRBatch copy(void* stream = nullptr) const {
        if (id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "copy serves batches of 32-bit ids");
        uint64_t nnz = 0;
        int32_t width = 0;
        check(sdrb_batch_info(h_, nullptr, nullptr, &nnz, &width, nullptr, nullptr));
        detail::DeviceBuffer off(device_, (n_rows_ + 1) * sizeof(uint32_t)), ids(device_, (nnz ? nnz : 1) * sizeof(int32_t));
        detail::DeviceBuffer data;
        if (kind_ != SDRB_DATA_EMPTY) data = detail::DeviceBuffer(device_, (nnz ? nnz : 1) * sizeof(float));
        check(sdrb_batch_export_device_csr(h_, static_cast<uint32_t*>(off.ptr), static_cast<int32_t*>(ids.ptr),
                                           static_cast<float*>(data.ptr), stream));
        SDRBatch c(device_, width, kind_);
        check(sdrb_batch_upload_device_csr(c.h_, n_rows_, static_cast<const uint32_t*>(off.ptr), static_cast<const int32_t*>(ids.ptr),
                                           static_cast<const float*>(data.ptr), stream));
        check(sdrb_stream_sync(device_, stream));   // the buffers are released on return
        c.n_rows_ = n_rows_;
        return c;
    }

    // Range queries: out(r, j) = (*this)[j].ands(lo[r], hi[r]), the ids of row j in [lo, hi) (SDR.hpp:931-956),
    // and the elements themselves, (*this)[j].ande(lo[r], hi[r]) (SDR.hpp:881-929), as a CSR over (r, j).
    Counts range_size(const std::vector<int32_t>& lo, const std::vector<int32_t>& hi, void* stream = nullptr) const {
        if (lo.size() != hi.size()) throw Error(SDRB_ERR_INVALID_ARG, "lo and hi differ in length");
        Counts c(device_, lo.size(), n_rows_);
        check(sdrb_range_size(h_, lo.size(), lo.data(), hi.data(), c.device_ptr(), n_rows_, stream));
        return c;
    }
    Elements range_elems(const std::vector<int32_t>& lo, const std::vector<int32_t>& hi, void* stream = nullptr) const {
        if (lo.size() != hi.size()) throw Error(SDRB_ERR_INVALID_ARG, "lo and hi differ in length");
        Elements e;
        e.na = lo.size();
        e.nb = n_rows_;
        e.has_data = kind_ != SDRB_DATA_EMPTY;
        e.offsets = detail::DeviceBuffer(device_, (e.na * e.nb + 1) * sizeof(uint32_t));
        check(sdrb_range_elems_count(h_, lo.size(), lo.data(), hi.data(), static_cast<uint32_t*>(e.offsets.ptr), &e.total, stream));
        e.ids = detail::DeviceBuffer(device_, (e.total ? e.total : 1) * sizeof(int32_t));
        if (e.has_data) e.data = detail::DeviceBuffer(device_, (e.total ? e.total : 1) * sizeof(float));
        check(sdrb_range_elems_fill(h_, lo.size(), lo.data(), hi.data(), static_cast<const uint32_t*>(e.offsets.ptr),
                                    static_cast<int32_t*>(e.ids.ptr), static_cast<float*>(e.data.ptr), stream));
        return e;
    }
    // Point queries: out(q, j) = position of ids[q] in row j or -1, the batched ande(id) (SDR.hpp:860-879).
    Positions point_query(const std::vector<int32_t>& ids, void* stream = nullptr) const {
        Positions p(device_, ids.size(), n_rows_);
        check(sdrb_point_query(h_, ids.size(), ids.data(), p.device_ptr(), n_rows_, stream));
        return p;
    }

    // The batch regrouped by id: row c of the result lists the rows of *this that contain id c, ascending,
    // with their payloads -- SDR::transpose (SDR.hpp:1807-1826) for a matrix stored as the rows of a batch.
    SDRBatch transpose(void* stream = nullptr) const {
        SDRBatch t(device_, 0, kind_);
        check(sdrb_batch_transpose(h_, t.h_, stream));
        uint64_t rows = 0;
        check(sdrb_batch_info(t.h_, nullptr, &rows, nullptr, nullptr, nullptr, nullptr));
        t.n_rows_ = rows;
        return t;
    }

    // The packed rows back on the host (batches of 32-bit ids), e.g. to rebuild SDR objects after an in-place op.
    Elements::Host to_host(void* stream = nullptr) const {
        if (id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "to_host serves batches of 32-bit ids");
        const uint32_t* off = nullptr;
        const int32_t* ids = nullptr;
        const float* data = nullptr;
        uint64_t nnz = 0;
        check(sdrb_batch_device_ptrs(h_, &off, &ids, &data));
        check(sdrb_batch_info(h_, nullptr, nullptr, &nnz, nullptr, nullptr, nullptr));
        Elements::Host h;
        h.offsets.resize(n_rows_ + 1);
        check(sdrb_memcpy_d2h(device_, h.offsets.data(), off, h.offsets.size() * sizeof(uint32_t), stream));
        std::vector<int32_t> narrow(nnz);
        if (nnz) check(sdrb_memcpy_d2h(device_, narrow.data(), ids, nnz * sizeof(int32_t), stream));
        h.ids.assign(narrow.begin(), narrow.end());
        if (kind_ != SDRB_DATA_EMPTY && nnz) {
            h.data.resize(nnz);
            check(sdrb_memcpy_d2h(device_, h.data.data(), data, nnz * sizeof(float), stream));
        }
        return h;
    }

 private:
    SDRBatch view(int kind, void* stream) const {
        if (id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "views exist for batches of 32-bit ids");
        const uint32_t* d_off = nullptr;
        const int32_t* d_ids = nullptr;
        const float* d_data = nullptr;
        uint64_t nnz = 0;
        uint32_t max_len = 0;
        int32_t width = 0, lo_id = 0, hi_id = -1;
        check(sdrb_batch_device_ptrs(h_, &d_off, &d_ids, &d_data));
        check(sdrb_batch_info(h_, nullptr, nullptr, &nnz, &width, nullptr, &max_len));
        check(sdrb_batch_id_span(h_, &lo_id, &hi_id));
        SDRBatch v(device_, width, kind);
        const float* payloads = nullptr;
        if (kind != SDRB_DATA_EMPTY) {
            const std::vector<float> ones(nnz ? nnz : 1, 1.0f);
            v.view_data_ = detail::DeviceBuffer(device_, ones.size() * sizeof(float));
            check(sdrb_memcpy_h2d(device_, v.view_data_.ptr, ones.data(), ones.size() * sizeof(float), stream));
            payloads = static_cast<const float*>(v.view_data_.ptr);
        }
        check(sdrb_batch_adopt_device_csr(v.h_, n_rows_, nnz, d_off, d_ids, payloads, max_len, lo_id, hi_id));
        v.n_rows_ = n_rows_;
        return v;
    }
    Counts size_op(int op, const SDRBatch& db, unsigned flags, void* stream) const {
        Counts c(db.device_, n_rows_, db.n_rows_);
        check(sdrb_size_op(h_, db.h_, op, flags, c.device_ptr(), db.n_rows_, stream));
        return c;
    }
    Elements elem_op(int op, const SDRBatch& db, unsigned flags, void* stream) const {
        const bool rowwise = (flags & SDRB_FLAG_ROWWISE) != 0;
        const SDRBatch& self = (flags & SDRB_FLAG_SWAP) ? db : *this;
        Elements e;
        e.na = rowwise ? 1 : n_rows_;
        e.nb = db.n_rows_;
        e.has_data = self.kind_ != SDRB_DATA_EMPTY;
        e.id_bits = self.id_bits_;   // a result holds ids of `this`'s width (the two batches may differ)
        e.offsets = detail::DeviceBuffer(db.device_, (e.na * e.nb + 1) * sizeof(uint32_t));
        if (!rowwise && id_bits_ == 32 && db.id_bits_ == 32 && elem_op_one_enqueue(op, db, flags, stream, &e)) return e;
        check(sdrb_elem_op_count(h_, db.h_, op, flags, static_cast<uint32_t*>(e.offsets.ptr), &e.total, stream));
        e.ids = detail::DeviceBuffer(db.device_, (e.total ? e.total : 1) * (e.id_bits / 8));
        if (e.has_data) e.data = detail::DeviceBuffer(db.device_, (e.total ? e.total : 1) * sizeof(float));
        if (e.id_bits == 64) {
            check(sdrb_elem_op_fill64(h_, db.h_, op, flags, static_cast<const uint32_t*>(e.offsets.ptr),
                                      static_cast<int64_t*>(e.ids.ptr), static_cast<float*>(e.data.ptr), stream));
        } else {
            check(sdrb_elem_op_fill(h_, db.h_, op, flags, static_cast<const uint32_t*>(e.offsets.ptr),
                                    static_cast<int32_t*>(e.ids.ptr), static_cast<float*>(e.data.ptr), stream));
        }
        return e;
    }
    // Count, scan and fill enqueued back to back, the element arrays sized from the shapes (sdrb_elem_op), or, for
    // ande of a few rows against a sparse database, from the expected number of matches (sdrb_ande_few; a result
    // that outgrows the estimate is computed again with room).  The host waits once, for the element total.
    // false: the shapes do not allow it (the caller falls back to the two-call protocol).
    bool elem_op_one_enqueue(int op, const SDRBatch& db, unsigned flags, void* stream, Elements* e) const {
        uint64_t bound = 0;
        if (sdrb_elem_op_bound(h_, db.h_, op, flags, &bound) != SDRB_OK || bound > 0xFFFFFFFFull) return false;
        uint64_t a_nnz = 0, b_nnz = 0;
        int32_t b_width = 0;
        uint32_t b_longest = 0;
        check(sdrb_batch_info(h_, nullptr, nullptr, &a_nnz, nullptr, nullptr, nullptr));
        check(sdrb_batch_info(db.h_, nullptr, nullptr, &b_nnz, nullptr, nullptr, &b_longest));
        check(sdrb_batch_table_width(db.h_, &b_width));   // the declared width, or the one derived from the ids
        const bool few = op == SDRB_OP_AND && n_rows_ >= 1 && n_rows_ <= 32 && b_width > 0 && b_width <= 4096 &&
                         b_longest <= 64 && (flags & ~SDRB_FLAG_SWAP) == 0;
        // ande of dense rows has a bound far above its result: only the estimate-sized call pays there
        if (op == SDRB_OP_AND && !few && bound > (64ull << 20)) return false;
        detail::DeviceBuffer total_dev(db.device_, sizeof(uint64_t));
        uint64_t capacity = bound ? bound : 1;
        if (few) {
            const double expected = (double)a_nnz * (double)b_nnz / (double)b_width;
            const uint64_t estimate = (uint64_t)(expected * 1.3 + 8.0 * std::sqrt(expected)) + 4096;
            if (estimate < capacity) capacity = estimate;
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            e->ids = detail::DeviceBuffer(db.device_, capacity * sizeof(int32_t));
            if (e->has_data) e->data = detail::DeviceBuffer(db.device_, capacity * sizeof(float));
            int rc = SDRB_ERR_UNSUPPORTED;
            if (few)
                rc = sdrb_ande_few(h_, db.h_, flags, static_cast<uint32_t*>(e->offsets.ptr), static_cast<int32_t*>(e->ids.ptr),
                                   static_cast<float*>(e->data.ptr), capacity, static_cast<uint64_t*>(total_dev.ptr), stream);
            if (rc == SDRB_ERR_UNSUPPORTED) {
                if (capacity < bound) {   // the few-rows kernel declined: the general one needs the full bound
                    capacity = bound;
                    e->ids = detail::DeviceBuffer(db.device_, capacity * sizeof(int32_t));
                    if (e->has_data) e->data = detail::DeviceBuffer(db.device_, capacity * sizeof(float));
                }
                rc = sdrb_elem_op(h_, db.h_, op, flags, static_cast<uint32_t*>(e->offsets.ptr), static_cast<int32_t*>(e->ids.ptr),
                                  static_cast<float*>(e->data.ptr), capacity, static_cast<uint64_t*>(total_dev.ptr), stream);
                if (rc == SDRB_ERR_UNSUPPORTED) return false;
            }
            check(rc);
            e->total = total_dev.to_host<uint64_t>(1, stream)[0];
            if (e->total <= capacity) return true;
            capacity = e->total;   // the estimate was too small: once more, with room
        }
        throw std::runtime_error("sdr_amd: element total changed between two runs of one operation");
    }
    Visits visit(int which, const SDRBatch& db, unsigned flags, void* stream) const {
        Visits v;
        v.na = (flags & SDRB_FLAG_ROWWISE) ? 1 : n_rows_;
        v.nb = db.n_rows_;
        v.offsets = detail::DeviceBuffer(db.device_, (v.na * v.nb + 1) * sizeof(uint32_t));
        check(sdrb_visit_count(h_, db.h_, which, flags, static_cast<uint32_t*>(v.offsets.ptr), &v.total, stream));
        v.this_pos = detail::DeviceBuffer(db.device_, (v.total ? v.total : 1) * sizeof(uint32_t));
        v.arg_pos = detail::DeviceBuffer(db.device_, (v.total ? v.total : 1) * sizeof(uint32_t));
        check(sdrb_visit_fill(h_, db.h_, which, flags, static_cast<const uint32_t*>(v.offsets.ptr),
                              static_cast<uint32_t*>(v.this_pos.ptr), static_cast<uint32_t*>(v.arg_pos.ptr), stream));
        return v;
    }
    SDRBatch& inplace(int op, const SDRBatch& arg, void* stream) {
        if (id_bits_ != 32) throw Error(SDRB_ERR_UNSUPPORTED, "in-place operations serve batches of 32-bit ids");
        Elements e = rowwise_elem(op, arg, 0, stream);
        check(sdrb_batch_upload_device_csr(h_, n_rows_, static_cast<const uint32_t*>(e.offsets.ptr),
                                           static_cast<const int32_t*>(e.ids.ptr),
                                           e.has_data ? static_cast<const float*>(e.data.ptr) : nullptr, stream));
        check(sdrb_stream_sync(device_, stream));   // the result buffers are released on return
        return *this;
    }

    sdrb_batch* h_ = nullptr;
    int device_ = 0;
    size_t n_rows_ = 0;
    int kind_ = SDRB_DATA_EMPTY;
    int id_bits_ = 32;
    detail::DeviceBuffer view_data_;   // a view's own payloads (as_unit)
};

}  // namespace sdr_amd
