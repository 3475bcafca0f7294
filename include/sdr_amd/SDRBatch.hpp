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
//     SDR<> first = r.unpack<SDR<>>(0, 0);                   // == queries[0].ore(rows[0])
//
// Link with -lsdr_amd (no HIP headers needed here).  Errors become sdr_amd::Error.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
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

// A payload with a float value() is treated as UnitData (UnitData.hpp:19); anything else as
// EmptyData (EmptyData.hpp:18-92).
template <typename D, typename = void>
struct has_value : std::false_type {};
template <typename D>
struct has_value<D, std::void_t<decltype(std::declval<const D&>().value())>> : std::true_type {};

template <typename SDR_t>
using elem_of = typename SDR_t::value_type;
template <typename SDR_t>
using data_of = typename elem_of<SDR_t>::data_type;

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
    sdrb_file_writer* w = nullptr;
    check(sdrb_file_writer_open(path.c_str(), width, unit ? SDRB_DATA_UNIT : SDRB_DATA_EMPTY, &w));
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
    std::vector<float> data(kind == SDRB_DATA_UNIT ? (span ? span : 1) : 0);
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

// uint32 counts[na][nb] in device memory
class Counts {
 public:
    Counts(int device, size_t na, size_t nb) : na_(na), nb_(nb), buf_(device, (na * nb ? na * nb : 1) * sizeof(uint32_t)) {}
    size_t rows() const { return na_; }
    size_t cols() const { return nb_; }
    uint32_t* device_ptr() const { return static_cast<uint32_t*>(buf_.ptr); }
    std::vector<uint32_t> to_host(void* stream = nullptr) const { return buf_.to_host<uint32_t>(na_ * nb_, stream); }

 private:
    size_t na_, nb_;
    detail::DeviceBuffer buf_;
};

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

// CSR result of an element op over all pairs
class Elements {
 public:
    size_t na = 0, nb = 0;
    uint64_t total = 0;
    bool has_data = false;
    detail::DeviceBuffer offsets, ids, data;

    struct Host {
        std::vector<uint32_t> offsets;
        std::vector<int32_t> ids;
        std::vector<float> data;
    };
    Host to_host(void* stream = nullptr) const {
        Host h;
        h.offsets = offsets.to_host<uint32_t>(na * nb + 1, stream);
        h.ids = ids.to_host<int32_t>(total, stream);
        if (has_data) h.data = data.to_host<float>(total, stream);
        return h;
    }
    // Rebuilds result (i, j) as an SDR object of the caller's type.
    template <typename SDR_t>
    static SDR_t unpack(const Host& h, size_t nb, size_t i, size_t j) {
        using Elem = detail::elem_of<SDR_t>;
        using Data = detail::data_of<SDR_t>;
        SDR_t r;
        for (uint32_t k = h.offsets[i * nb + j]; k < h.offsets[i * nb + j + 1]; ++k) {
            if constexpr (detail::has_value<Data>::value) r.push_back(Elem(h.ids[k], Data(h.data[k])));
            else r.push_back(Elem(h.ids[k]));
        }
        return r;
    }
};

class SDRBatch {
 public:
    SDRBatch(int device, int32_t width, int data_kind) : device_(device), unit_(data_kind == SDRB_DATA_UNIT) {
        check(sdrb_batch_create(device, width, data_kind, &h_));
    }
    SDRBatch(const SDRBatch&) = delete;
    SDRBatch& operator=(const SDRBatch&) = delete;
    SDRBatch(SDRBatch&& o) noexcept : h_(o.h_), device_(o.device_), n_rows_(o.n_rows_), unit_(o.unit_) { o.h_ = nullptr; }
    ~SDRBatch() { sdrb_batch_destroy(h_); }

    // Packs the SDRs of [first, last) by iterating each one (the portable way to read any
    // container, IDContiguousContainer.hpp:263-303 included).  width = 0: arbitrary ids.
    template <typename Iter>
    static SDRBatch pack(Iter first, Iter last, int device = 0, int32_t width = 0, void* stream = nullptr) {
        using SDR_t = std::decay_t<decltype(*first)>;
        using Data = detail::data_of<SDR_t>;
        constexpr bool unit = detail::has_value<Data>::value;
        std::vector<uint64_t> off{0};
        std::vector<int32_t> ids;
        std::vector<float> data;
        for (; first != last; ++first) {
            for (auto it = first->cbegin(); it != first->cend(); ++it) {
                ids.push_back(static_cast<int32_t>(it->id()));
                if constexpr (unit) data.push_back(static_cast<float>(it->data().value()));
            }
            off.push_back(ids.size());
        }
        SDRBatch b(device, width, unit ? SDRB_DATA_UNIT : SDRB_DATA_EMPTY);
        b.upload(off, ids, data, stream);
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
    const sdrb_batch* handle() const { return h_; }

    // out(i, j) = (*this)[i].ands(db[j]) ... (SDR.hpp:175-176,229-230,257-258,317-318)
    Counts ands(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_AND, db, flags, stream); }
    Counts ors(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_OR, db, flags, stream); }
    Counts xors(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_XOR, db, flags, stream); }
    Counts rms(const SDRBatch& db, unsigned flags = 0, void* stream = nullptr) const { return size_op(SDRB_OP_RM, db, flags, stream); }

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

 private:
    Counts size_op(int op, const SDRBatch& db, unsigned flags, void* stream) const {
        Counts c(db.device_, n_rows_, db.n_rows_);
        check(sdrb_size_op(h_, db.h_, op, flags, c.device_ptr(), db.n_rows_, stream));
        return c;
    }
    Elements elem_op(int op, const SDRBatch& db, unsigned flags, void* stream) const {
        Elements e;
        e.na = n_rows_;
        e.nb = db.n_rows_;
        e.has_data = (flags & SDRB_FLAG_SWAP) ? db.unit_ : unit_;
        e.offsets = detail::DeviceBuffer(db.device_, (e.na * e.nb + 1) * sizeof(uint32_t));
        check(sdrb_elem_op_count(h_, db.h_, op, flags, static_cast<uint32_t*>(e.offsets.ptr), &e.total, stream));
        e.ids = detail::DeviceBuffer(db.device_, (e.total ? e.total : 1) * sizeof(int32_t));
        if (e.has_data) e.data = detail::DeviceBuffer(db.device_, (e.total ? e.total : 1) * sizeof(float));
        check(sdrb_elem_op_fill(h_, db.h_, op, flags, static_cast<const uint32_t*>(e.offsets.ptr),
                                static_cast<int32_t*>(e.ids.ptr), static_cast<float*>(e.data.ptr), stream));
        return e;
    }

    sdrb_batch* h_ = nullptr;
    int device_ = 0;
    size_t n_rows_ = 0;
    bool unit_ = false;
};

}  // namespace sdr_amd
