// CSR database files: a flat, mmap-able container for a packed batch, with a streaming writer.
//
// The reference has no serialisation (SDR.hpp:1670-1680 only prints text), so this format is
// this engine's own.  It holds exactly what a batch holds -- the SoA layout of
// IDContiguousContainer.hpp:246-250, ids ascending per row (SDR.hpp:679,1621), UnitData payloads
// in [0,1] (UnitData.hpp:15-17) -- so a row range can be uploaded straight from the mapping.
//
//   byte 0   char     magic[8] = "SDRBCSR1"
//        8   uint32   version (1)           12  uint32  flags (bit 0: float payloads present;
//                                                                  bit 1: they are ArithData, else UnitData)
//       16   uint32   id_bits (32)          20  int32   width (0 = undeclared)
//       24   uint64   n_rows                32  uint64  nnz
//       40   uint64   ids_pos               48  uint64  data_pos (0 = none)      56  uint64  offsets_pos
//       64   ... sections, each 64-byte aligned: int32 ids[nnz], float data[nnz], uint64 offsets[n_rows+1]
// (little endian; ids come first so that a writer can stream rows without knowing their number).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <string>
#include <vector>

#include "sdrb_internal.h"

namespace {

using sdrb::set_error;

constexpr char kMagic[8] = {'S', 'D', 'R', 'B', 'C', 'S', 'R', '1'};
constexpr uint64_t kAlign = 64;

struct Header {
    char magic[8];
    uint32_t version, flags, id_bits;
    int32_t width;
    uint64_t n_rows, nnz, ids_pos, data_pos, offsets_pos;
};
static_assert(sizeof(Header) == 64, "header is 64 bytes");

uint64_t align_up(uint64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

bool write_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        const ssize_t w = ::write(fd, c, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        c += w;
        n -= (size_t)w;
    }
    return true;
}

bool pad_to(int fd, uint64_t& pos, uint64_t target) {
    static const char zeros[kAlign] = {0};
    while (pos < target) {
        const size_t n = (size_t)(target - pos < kAlign ? target - pos : kAlign);
        if (!write_all(fd, zeros, n)) return false;
        pos += n;
    }
    return true;
}

// A read-only mapping of a validated file.
struct Mapping {
    void* base = MAP_FAILED;
    size_t bytes = 0;
    const Header* h = nullptr;
    const int32_t* ids = nullptr;
    const float* data = nullptr;
    const uint64_t* offsets = nullptr;
    ~Mapping() {
        if (base != MAP_FAILED) ::munmap(base, bytes);
    }
};

int map_file(const char* path, Mapping* m) {
    if (!path) {
        set_error("null path");
        return SDRB_ERR_INVALID_ARG;
    }
    const int fd = ::open(path, O_RDONLY);
    if (fd < 0) {
        set_error("cannot open %s: %s", path, std::strerror(errno));
        return SDRB_ERR_IO;
    }
    struct stat st;
    if (::fstat(fd, &st) != 0 || (uint64_t)st.st_size < sizeof(Header)) {
        ::close(fd);
        set_error("%s is too short to be a CSR database file", path);
        return SDRB_ERR_FORMAT;
    }
    m->bytes = (size_t)st.st_size;
    m->base = ::mmap(nullptr, m->bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    ::close(fd);
    if (m->base == MAP_FAILED) {
        set_error("mmap of %s failed: %s", path, std::strerror(errno));
        return SDRB_ERR_IO;
    }
    const Header* h = static_cast<const Header*>(m->base);
    if (std::memcmp(h->magic, kMagic, 8) != 0 || h->version != 1 || h->id_bits != 32) {
        set_error("%s: not a version-1 CSR database file with 32-bit ids", path);
        return SDRB_ERR_FORMAT;
    }
    const bool has_data = (h->flags & 1u) != 0;
    const uint64_t size = m->bytes;
    auto section_ok = [&](uint64_t pos, uint64_t count, uint64_t elem) {
        return pos % kAlign == 0 && pos >= sizeof(Header) && pos <= size && count <= (size - pos) / elem;
    };
    // n_rows + 1 offsets must fit: compared as n_rows < capacity, so that n_rows = 2^64-1 cannot wrap to zero
    const bool rows_ok = h->offsets_pos <= size && h->n_rows < (size - h->offsets_pos) / 8;
    if (h->nnz > 0xFFFFFFFFFFull || !rows_ok || !section_ok(h->ids_pos, h->nnz, 4) ||
        !section_ok(h->offsets_pos, h->n_rows + 1, 8) || (has_data && !section_ok(h->data_pos, h->nnz, 4)) ||
        (!has_data && h->data_pos != 0)) {
        set_error("%s: section table is inconsistent with the file size", path);
        return SDRB_ERR_FORMAT;
    }
    // the sections may come in any order but never share bytes
    struct Span { uint64_t lo, hi; };
    Span spans[3] = {{h->ids_pos, h->ids_pos + h->nnz * 4}, {h->offsets_pos, h->offsets_pos + (h->n_rows + 1) * 8},
                     {has_data ? h->data_pos : 0, has_data ? h->data_pos + h->nnz * 4 : 0}};
    for (int x = 0; x < 3; ++x)
        for (int y = x + 1; y < 3; ++y)
            if (spans[x].lo < spans[y].hi && spans[y].lo < spans[x].hi) {
                set_error("%s: sections overlap", path);
                return SDRB_ERR_FORMAT;
            }
    const char* c = static_cast<const char*>(m->base);
    m->h = h;
    m->ids = reinterpret_cast<const int32_t*>(c + h->ids_pos);
    m->data = has_data ? reinterpret_cast<const float*>(c + h->data_pos) : nullptr;
    m->offsets = reinterpret_cast<const uint64_t*>(c + h->offsets_pos);
    if (m->offsets[0] != 0 || m->offsets[h->n_rows] != h->nnz) {
        set_error("%s: row offsets do not span the element arrays", path);
        return SDRB_ERR_FORMAT;
    }
    return SDRB_OK;
}

int check_range(const Mapping& m, uint64_t row_lo, uint64_t row_hi) {
    if (row_lo > row_hi || row_hi > m.h->n_rows) {
        set_error("row range [%llu, %llu) outside the file's %llu rows", (unsigned long long)row_lo,
                  (unsigned long long)row_hi, (unsigned long long)m.h->n_rows);
        return SDRB_ERR_INVALID_ARG;
    }
    for (uint64_t r = row_lo; r < row_hi; ++r) {
        if (m.offsets[r + 1] < m.offsets[r] || m.offsets[r + 1] > m.h->nnz) {
            set_error("row offsets not monotone at row %llu", (unsigned long long)r);
            return SDRB_ERR_FORMAT;
        }
    }
    return SDRB_OK;
}

}  // namespace

struct sdrb_file_writer {
    int fd = -1;
    int data_fd = -1;            // payloads are streamed to a sidecar and appended on close
    std::string path, data_path;
    int32_t width = 0;
    bool unit = false;           // a float payload per element (UnitData or ArithData)
    bool arith = false;          // ... of ArithData: any finite value instead of [0,1]
    uint64_t pos = 0;            // write position in the main file
    uint64_t nnz = 0;
    bool failed = false;         // a write stopped part-way: the file no longer matches `pos`, only abort is left
    std::vector<uint64_t> offsets{0};
};

extern "C" {

static int sdrb_file_writer_open_impl(const char* path, int32_t width, int data_kind, sdrb_file_writer** out) {
    if (!path || !out || width < 0 || (data_kind != SDRB_DATA_EMPTY && data_kind != SDRB_DATA_UNIT && data_kind != SDRB_DATA_ARITH)) {
        set_error("sdrb_file_writer_open: bad argument");
        return SDRB_ERR_INVALID_ARG;
    }
    sdrb_file_writer* w = new sdrb_file_writer;   // a failed allocation is reported by the entry point's guard
    w->path = path;
    w->width = width;
    w->unit = data_kind != SDRB_DATA_EMPTY;
    w->arith = data_kind == SDRB_DATA_ARITH;
    w->fd = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (w->fd < 0) {
        set_error("cannot create %s: %s", path, std::strerror(errno));
        delete w;
        return SDRB_ERR_IO;
    }
    if (w->unit) {
        w->data_path = w->path + ".payload.tmp";
        w->data_fd = ::open(w->data_path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
        if (w->data_fd < 0) {
            set_error("cannot create %s: %s", w->data_path.c_str(), std::strerror(errno));
            ::close(w->fd);
            delete w;
            return SDRB_ERR_IO;
        }
    }
    Header h{};   // placeholder; rewritten on close
    if (!write_all(w->fd, &h, sizeof h)) {
        set_error("write to %s failed: %s", path, std::strerror(errno));
        sdrb_file_writer_abort(w);
        return SDRB_ERR_IO;
    }
    w->pos = sizeof h;
    *out = w;
    return SDRB_OK;
}

int sdrb_file_writer_open(const char* path, int32_t width, int data_kind, sdrb_file_writer** out) {
    return sdrb::guarded("sdrb_file_writer_open", [&] { return sdrb_file_writer_open_impl(path, width, data_kind, out); });
}

static int sdrb_file_writer_append_impl(sdrb_file_writer* w, uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids,
                            const float* data) {
    if (!w || !row_offsets || row_offsets[0] != 0) {
        set_error("sdrb_file_writer_append: null argument or row_offsets[0] != 0");
        return SDRB_ERR_INVALID_ARG;
    }
    if (w->failed) {
        set_error("sdrb_file_writer_append: an earlier write to %s failed; abort the writer", w->path.c_str());
        return SDRB_ERR_IO;
    }
    const uint64_t nnz = row_offsets[n_rows];
    if ((nnz && !ids) || (nnz && w->unit && !data)) {
        set_error("sdrb_file_writer_append: element arrays missing");
        return SDRB_ERR_INVALID_ARG;
    }
    // the same preconditions an upload enforces (checked by the same code), so that every file holds a valid batch
    {
        const int rc = sdrb::check_host_csr(n_rows, row_offsets, ids, data, w->width,
                                            !w->unit ? SDRB_DATA_EMPTY : w->arith ? SDRB_DATA_ARITH : SDRB_DATA_UNIT);
        if (rc != SDRB_OK) return rc;
    }
    if (!write_all(w->fd, ids, nnz * sizeof(int32_t)) ||
        (w->unit && !write_all(w->data_fd, data, nnz * sizeof(float)))) {
        set_error("write to %s failed: %s", w->path.c_str(), std::strerror(errno));
        w->failed = true;
        return SDRB_ERR_IO;
    }
    w->pos += nnz * sizeof(int32_t);
    for (uint64_t r = 1; r <= n_rows; ++r) w->offsets.push_back(w->nnz + row_offsets[r]);
    w->nnz += nnz;
    return SDRB_OK;
}

int sdrb_file_writer_append(sdrb_file_writer* w, uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids, const float* data) {
    return sdrb::guarded("sdrb_file_writer_append", [&] { return sdrb_file_writer_append_impl(w, n_rows, row_offsets, ids, data); });
}

static int sdrb_file_writer_close_impl(sdrb_file_writer* w) {
    if (!w) return SDRB_OK;
    if (w->failed) {
        const std::string broken = w->path;
        sdrb_file_writer_abort(w);
        set_error("%s: an earlier write failed, the file was removed", broken.c_str());
        return SDRB_ERR_IO;
    }
    Header h{};
    std::memcpy(h.magic, kMagic, 8);
    h.version = 1;
    h.flags = (w->unit ? 1u : 0u) | (w->arith ? 2u : 0u);
    h.id_bits = 32;
    h.width = w->width;
    h.n_rows = w->offsets.size() - 1;
    h.nnz = w->nnz;
    h.ids_pos = sizeof(Header);
    bool ok = pad_to(w->fd, w->pos, align_up(w->pos));
    if (ok && w->unit) {
        h.data_pos = w->pos;
        std::vector<char> buf(1 << 20);
        ok = ::lseek(w->data_fd, 0, SEEK_SET) == 0;
        uint64_t left = w->nnz * sizeof(float);
        while (ok && left) {
            const size_t want = (size_t)(left < buf.size() ? left : buf.size());
            const ssize_t got = ::read(w->data_fd, buf.data(), want);
            if (got <= 0) {
                ok = false;
                break;
            }
            ok = write_all(w->fd, buf.data(), (size_t)got);
            w->pos += (uint64_t)got;
            left -= (uint64_t)got;
        }
        ok = ok && pad_to(w->fd, w->pos, align_up(w->pos));
    }
    if (ok) {
        h.offsets_pos = w->pos;
        ok = write_all(w->fd, w->offsets.data(), w->offsets.size() * sizeof(uint64_t));
    }
    ok = ok && ::lseek(w->fd, 0, SEEK_SET) == 0 && write_all(w->fd, &h, sizeof h);
    const int saved = errno;
    const std::string path = w->path;
    if (ok) {
        ::close(w->fd);
        if (w->data_fd >= 0) {
            ::close(w->data_fd);
            ::unlink(w->data_path.c_str());
        }
        delete w;
        return SDRB_OK;
    }
    sdrb_file_writer_abort(w);
    set_error("finishing %s failed: %s", path.c_str(), std::strerror(saved));
    return SDRB_ERR_IO;
}

int sdrb_file_writer_close(sdrb_file_writer* w) {
    return sdrb::guarded("sdrb_file_writer_close", [&] { return sdrb_file_writer_close_impl(w); });
}

int sdrb_file_writer_abort(sdrb_file_writer* w) {
    if (!w) return SDRB_OK;
    if (w->fd >= 0) ::close(w->fd);
    if (w->data_fd >= 0) {
        ::close(w->data_fd);
        ::unlink(w->data_path.c_str());
    }
    ::unlink(w->path.c_str());
    delete w;
    return SDRB_OK;
}

int sdrb_file_info(const char* path, uint64_t* n_rows, uint64_t* nnz, int32_t* width, int* data_kind) {
    Mapping m;
    const int rc = map_file(path, &m);
    if (rc != SDRB_OK) return rc;
    if (n_rows) *n_rows = m.h->n_rows;
    if (nnz) *nnz = m.h->nnz;
    if (width) *width = m.h->width;
    if (data_kind) *data_kind = !(m.h->flags & 1u) ? SDRB_DATA_EMPTY : (m.h->flags & 2u) ? SDRB_DATA_ARITH : SDRB_DATA_UNIT;
    return SDRB_OK;
}

int sdrb_file_row_span(const char* path, uint64_t row_lo, uint64_t row_hi, uint64_t* nnz) {
    Mapping m;
    int rc = map_file(path, &m);
    if (rc != SDRB_OK) return rc;
    rc = check_range(m, row_lo, row_hi);
    if (rc != SDRB_OK) return rc;
    if (nnz) *nnz = m.offsets[row_hi] - m.offsets[row_lo];
    return SDRB_OK;
}

int sdrb_file_read_rows(const char* path, uint64_t row_lo, uint64_t row_hi, uint64_t* row_offsets, int32_t* ids,
                        float* data) {
    Mapping m;
    int rc = map_file(path, &m);
    if (rc != SDRB_OK) return rc;
    rc = check_range(m, row_lo, row_hi);
    if (rc != SDRB_OK) return rc;
    if (!row_offsets) {
        set_error("sdrb_file_read_rows: row_offsets is null");
        return SDRB_ERR_INVALID_ARG;
    }
    const uint64_t e0 = m.offsets[row_lo], e1 = m.offsets[row_hi];
    for (uint64_t r = row_lo; r <= row_hi; ++r) row_offsets[r - row_lo] = m.offsets[r] - e0;
    if (e1 > e0) {
        if (!ids) {
            set_error("sdrb_file_read_rows: ids is null");
            return SDRB_ERR_INVALID_ARG;
        }
        std::memcpy(ids, m.ids + e0, (e1 - e0) * sizeof(int32_t));
        if (m.data && data) std::memcpy(data, m.data + e0, (e1 - e0) * sizeof(float));
    }
    return SDRB_OK;
}

static int sdrb_batch_upload_file_impl(sdrb_batch* batch, const char* path, uint64_t row_lo, uint64_t row_hi, void* stream) {
    if (!batch) {
        set_error("null batch handle");
        return SDRB_ERR_INVALID_ARG;
    }
    Mapping m;
    int rc = map_file(path, &m);
    if (rc != SDRB_OK) return rc;
    rc = check_range(m, row_lo, row_hi);
    if (rc != SDRB_OK) return rc;
    const bool file_data = (m.h->flags & 1u) != 0, file_arith = (m.h->flags & 2u) != 0;
    if (sdrb::has_payload(batch) && !file_data) {
        set_error("%s holds no payloads but the batch has them", path);
        return SDRB_ERR_INVALID_ARG;
    }
    if (batch->data_kind == SDRB_DATA_UNIT && file_arith) {
        set_error("%s holds ArithData payloads but the batch is UnitData", path);
        return SDRB_ERR_INVALID_ARG;
    }
    // rows are uploaded straight from the mapping; only the rebased offsets are materialised
    const uint64_t e0 = m.offsets[row_lo];
    std::vector<uint64_t> off(row_hi - row_lo + 1);
    for (uint64_t r = row_lo; r <= row_hi; ++r) off[r - row_lo] = m.offsets[r] - e0;
    return sdrb_batch_upload_csr(batch, row_hi - row_lo, off.data(), m.ids + e0,
                                 sdrb::has_payload(batch) ? m.data + e0 : nullptr, stream);
}

int sdrb_batch_upload_file(sdrb_batch* batch, const char* path, uint64_t row_lo, uint64_t row_hi, void* stream) {
    return sdrb::guarded("sdrb_batch_upload_file", [&] { return sdrb_batch_upload_file_impl(batch, path, row_lo, row_hi, stream); });
}

}  // extern "C"
