// Host-side checks of the C ABI that need no GPU, meant to be run from a build with
// -fsanitize=address,undefined on the host code (make -C sdr_amd/csrc asan): the CSR database
// file writer and reader on good, refused and damaged inputs, the synthetic row generator into
// exactly sized buffers, and argument validation of the batch entry points.  Every allocation
// is sized to the documented extent, so a write or read past it is reported by the sanitizer.
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sdr_amd.h"

namespace {

int g_checks = 0;

#define CHECK(cond)                                                                        \
    do {                                                                                   \
        ++g_checks;                                                                        \
        if (!(cond)) {                                                                     \
            std::fprintf(stderr, "%s:%d: CHECK(%s) failed; last error: %s\n", __FILE__,    \
                         __LINE__, #cond, sdrb_last_error());                              \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

struct Rows {
    std::vector<uint64_t> offsets;
    std::vector<int32_t> ids;
    std::vector<float> data;
};

Rows synth(uint64_t seed, uint64_t first, uint64_t n, int32_t width, int32_t active, bool payload) {
    Rows r;
    r.offsets.resize(n + 1);
    r.ids.resize(n * (uint64_t)active);
    if (payload) r.data.resize(n * (uint64_t)active);
    CHECK(sdrb_synth_csr(seed, first, n, width, active, r.offsets.data(), r.ids.data(),
                         payload ? r.data.data() : nullptr) == SDRB_OK);
    return r;
}

std::vector<unsigned char> slurp(const std::string& path) {
    std::vector<unsigned char> bytes;
    FILE* f = std::fopen(path.c_str(), "rb");
    CHECK(f != nullptr);
    unsigned char buf[4096];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return bytes;
}

void spill(const std::string& path, const std::vector<unsigned char>& bytes, size_t n) {
    FILE* f = std::fopen(path.c_str(), "wb");
    CHECK(f != nullptr);
    CHECK(std::fwrite(bytes.data(), 1, n, f) == n);
    std::fclose(f);
}

void generator_fills_exactly_its_buffers() {
    for (int32_t active : {0, 1, 40, 64}) {
        Rows r = synth(7, 3, 17, 64, active, true);
        CHECK(r.offsets.front() == 0 && r.offsets.back() == 17u * (uint64_t)active);
        for (uint64_t row = 0; row < 17; ++row)
            for (uint64_t k = r.offsets[row] + 1; k < r.offsets[row + 1]; ++k) CHECK(r.ids[k - 1] < r.ids[k]);
        for (float v : r.data) CHECK(v >= 0.0f && v < 1.0f);
    }
    // rows depend on the global row index alone: a window equals the same rows of a longer run
    Rows whole = synth(1, 0, 32, 2048, 40, false), window = synth(1, 8, 8, 2048, 40, false);
    CHECK(std::memcmp(whole.ids.data() + 8 * 40, window.ids.data(), 8 * 40 * sizeof(int32_t)) == 0);
    uint64_t off[2];
    int32_t id[1];
    CHECK(sdrb_synth_csr(1, 0, 1, 8, 9, off, id, nullptr) != SDRB_OK);   // more active ids than the width holds
    CHECK(sdrb_synth_csr(1, 0, 1, 8, 1, nullptr, id, nullptr) != SDRB_OK);
}

void files_round_trip(const std::string& dir, bool payload) {
    const std::string path = dir + (payload ? "/unit.sdrb" : "/empty.sdrb");
    const int32_t width = 2048, active = 40;
    const uint64_t n = 1000;
    Rows all = synth(5, 0, n, width, active, payload);
    sdrb_file_writer* w = nullptr;
    CHECK(sdrb_file_writer_open(path.c_str(), width, payload ? SDRB_DATA_UNIT : SDRB_DATA_EMPTY, &w) == SDRB_OK);
    // ragged chunks, an empty one among them
    const uint64_t cuts[] = {0, 1, 1, 250, 999, 1000};
    for (size_t c = 0; c + 1 < sizeof cuts / sizeof cuts[0]; ++c) {
        const uint64_t lo = cuts[c], hi = cuts[c + 1];
        std::vector<uint64_t> rel(hi - lo + 1);
        for (uint64_t r = lo; r <= hi; ++r) rel[r - lo] = all.offsets[r] - all.offsets[lo];
        CHECK(sdrb_file_writer_append(w, hi - lo, rel.data(), all.ids.data() + all.offsets[lo],
                                      payload ? all.data.data() + all.offsets[lo] : nullptr) == SDRB_OK);
    }
    CHECK(sdrb_file_writer_close(w) == SDRB_OK);

    uint64_t rows = 0, nnz = 0;
    int32_t got_width = 0;
    int kind = -1;
    CHECK(sdrb_file_info(path.c_str(), &rows, &nnz, &got_width, &kind) == SDRB_OK);
    CHECK(rows == n && nnz == n * active && got_width == width);
    CHECK(kind == (payload ? SDRB_DATA_UNIT : SDRB_DATA_EMPTY));
    CHECK(sdrb_file_info(path.c_str(), nullptr, nullptr, nullptr, nullptr) == SDRB_OK);

    for (auto span : {std::pair<uint64_t, uint64_t>{0, n}, {0, 0}, {n, n}, {123, 124}, {500, 1000}}) {
        uint64_t count = ~0ull;
        CHECK(sdrb_file_row_span(path.c_str(), span.first, span.second, &count) == SDRB_OK);
        CHECK(count == all.offsets[span.second] - all.offsets[span.first]);
        std::vector<uint64_t> off(span.second - span.first + 1);
        std::vector<int32_t> ids(count);
        std::vector<float> data(payload ? count : 0);
        CHECK(sdrb_file_read_rows(path.c_str(), span.first, span.second, off.data(), ids.data(),
                                  payload ? data.data() : nullptr) == SDRB_OK);
        for (uint64_t r = span.first; r <= span.second; ++r)
            CHECK(off[r - span.first] == all.offsets[r] - all.offsets[span.first]);
        CHECK(count == 0 || std::memcmp(ids.data(), all.ids.data() + all.offsets[span.first], count * 4) == 0);
        if (payload && count)
            CHECK(std::memcmp(data.data(), all.data.data() + all.offsets[span.first], count * 4) == 0);
    }
    uint64_t count;
    CHECK(sdrb_file_row_span(path.c_str(), 5, 4, &count) != SDRB_OK);
    CHECK(sdrb_file_row_span(path.c_str(), 0, n + 1, &count) != SDRB_OK);
    uint64_t off1[2];
    CHECK(sdrb_file_read_rows(path.c_str(), 0, 1, off1, nullptr, nullptr) != SDRB_OK);   // ids missing

    // damaged copies: every prefix class and a flipped byte in each header field must be refused or read safely
    const std::vector<unsigned char> bytes = slurp(path);
    const std::string bad = dir + "/damaged.sdrb";
    for (size_t keep : {(size_t)0, (size_t)7, (size_t)63, (size_t)64, bytes.size() / 2, bytes.size() - 1}) {
        spill(bad, bytes, keep);
        CHECK(sdrb_file_info(bad.c_str(), &rows, &nnz, &got_width, &kind) != SDRB_OK);
    }
    for (size_t pos = 0; pos < 64; ++pos) {
        for (unsigned char flip : {(unsigned char)0x01, (unsigned char)0x80, (unsigned char)0xFF}) {
            std::vector<unsigned char> copy = bytes;
            copy[pos] ^= flip;
            spill(bad, copy, copy.size());
            uint64_t r2 = 0, e2 = 0;
            if (sdrb_file_info(bad.c_str(), &r2, &e2, nullptr, nullptr) != SDRB_OK) continue;
            // a header the reader accepts must still be readable inside the mapping
            const uint64_t take = r2 < 8 ? r2 : 8;
            uint64_t span_nnz = 0;
            if (sdrb_file_row_span(bad.c_str(), 0, take, &span_nnz) != SDRB_OK) continue;
            std::vector<uint64_t> off(take + 1);
            std::vector<int32_t> ids(span_nnz);
            std::vector<float> data(span_nnz);
            (void)sdrb_file_read_rows(bad.c_str(), 0, take, off.data(), ids.data(), payload ? data.data() : nullptr);
        }
    }
    // the row-offset table damaged: offsets that run backwards or past the element arrays
    {
        std::vector<unsigned char> copy = bytes;
        uint64_t offsets_pos;
        std::memcpy(&offsets_pos, copy.data() + 56, 8);
        const uint64_t huge = ~0ull >> 1;
        std::memcpy(copy.data() + offsets_pos + 8 * 10, &huge, 8);
        spill(bad, copy, copy.size());
        uint64_t span_nnz = 0;
        CHECK(sdrb_file_row_span(bad.c_str(), 0, 20, &span_nnz) != SDRB_OK);
        std::vector<uint64_t> off(21);
        std::vector<int32_t> ids(20 * active);
        std::vector<float> data(20 * active);
        CHECK(sdrb_file_read_rows(bad.c_str(), 0, 20, off.data(), ids.data(), payload ? data.data() : nullptr) != SDRB_OK);
    }
    ::unlink(bad.c_str());
    ::unlink(path.c_str());
}

void writer_refuses_bad_rows(const std::string& dir) {
    const std::string path = dir + "/refused.sdrb";
    struct stat st;
    sdrb_file_writer* w = nullptr;
    CHECK(sdrb_file_writer_open(nullptr, 16, SDRB_DATA_EMPTY, &w) != SDRB_OK);
    CHECK(sdrb_file_writer_open(path.c_str(), 16, SDRB_DATA_EMPTY, nullptr) != SDRB_OK);
    CHECK(sdrb_file_writer_open((dir + "/no/such/dir/x.sdrb").c_str(), 16, SDRB_DATA_EMPTY, &w) == SDRB_ERR_IO);
    CHECK(sdrb_file_writer_open(path.c_str(), 16, SDRB_DATA_UNIT, &w) == SDRB_OK);

    const uint64_t off[] = {0, 3};
    const float ok_data[] = {0.0f, 0.5f, 1.0f};
    const int32_t unsorted[] = {1, 1, 2}, outside[] = {1, 2, 16}, negative[] = {-1, 2, 3}, fine[] = {1, 2, 15};
    const float too_big[] = {0.0f, 0.5f, 1.5f}, nan_data[] = {0.0f, 0.5f, __builtin_nanf("")};
    const uint64_t backwards[] = {0, 3, 2}, not_zero[] = {1, 3};
    CHECK(sdrb_file_writer_append(w, 1, off, unsorted, ok_data) == SDRB_ERR_NOT_ASCENDING);
    CHECK(sdrb_file_writer_append(w, 1, off, outside, ok_data) == SDRB_ERR_ID_RANGE);
    CHECK(sdrb_file_writer_append(w, 1, off, negative, ok_data) == SDRB_ERR_ID_RANGE);
    CHECK(sdrb_file_writer_append(w, 1, off, fine, too_big) == SDRB_ERR_PAYLOAD_RANGE);
    CHECK(sdrb_file_writer_append(w, 1, off, fine, nan_data) == SDRB_ERR_PAYLOAD_RANGE);
    CHECK(sdrb_file_writer_append(w, 1, off, fine, nullptr) != SDRB_OK);
    CHECK(sdrb_file_writer_append(w, 2, backwards, fine, ok_data) != SDRB_OK);
    CHECK(sdrb_file_writer_append(w, 1, not_zero, fine, ok_data) != SDRB_OK);
    CHECK(sdrb_file_writer_append(nullptr, 1, off, fine, ok_data) != SDRB_OK);
    // a refused chunk leaves nothing behind: the file holds the accepted rows only
    CHECK(sdrb_file_writer_append(w, 1, off, fine, ok_data) == SDRB_OK);
    CHECK(sdrb_file_writer_close(w) == SDRB_OK);
    uint64_t rows = 0, nnz = 0;
    CHECK(sdrb_file_info(path.c_str(), &rows, &nnz, nullptr, nullptr) == SDRB_OK);
    CHECK(rows == 1 && nnz == 3);
    ::unlink(path.c_str());

    CHECK(sdrb_file_writer_open(path.c_str(), 16, SDRB_DATA_EMPTY, &w) == SDRB_OK);
    CHECK(sdrb_file_writer_append(w, 1, off, fine, nullptr) == SDRB_OK);
    CHECK(sdrb_file_writer_abort(w) == SDRB_OK);
    CHECK(::stat(path.c_str(), &st) != 0);   // abort removes the file
    CHECK(sdrb_file_writer_close(nullptr) == SDRB_OK && sdrb_file_writer_abort(nullptr) == SDRB_OK);   // like free(NULL)
    CHECK(sdrb_file_info((dir + "/absent.sdrb").c_str(), &rows, &nnz, nullptr, nullptr) == SDRB_ERR_IO);
    CHECK(sdrb_file_info(nullptr, &rows, &nnz, nullptr, nullptr) != SDRB_OK);
}

// Without a GPU every device call fails; with one, the null and shape checks come first.  Either way the
// entry points must return a status and leave a message, never touch the null handles.
void batch_entry_points_check_their_arguments() {
    uint32_t out[4];
    float outf[4];
    uint64_t total = 0;
    CHECK(sdrb_size_op(nullptr, nullptr, SDRB_OP_AND, 0, out, 4, nullptr) != SDRB_OK);
    CHECK(std::strlen(sdrb_last_error()) > 0);
    CHECK(sdrb_size_op_static(nullptr, nullptr, SDRB_OP_AND, 0, out, 4, nullptr) != SDRB_OK);
    CHECK(sdrb_equal(nullptr, nullptr, 0, out, 4, nullptr) != SDRB_OK);
    CHECK(sdrb_inner(nullptr, nullptr, 0, outf, 4, nullptr) != SDRB_OK);
    CHECK(sdrb_elem_op_count(nullptr, nullptr, SDRB_OP_OR, 0, out, &total, nullptr) != SDRB_OK);
    CHECK(sdrb_elem_op_bound(nullptr, nullptr, SDRB_OP_OR, 0, &total) != SDRB_OK);
    CHECK(sdrb_batch_destroy(nullptr) == SDRB_OK);   // like free(NULL)
    CHECK(sdrb_batch_create(0, 16, SDRB_DATA_EMPTY, nullptr) != SDRB_OK);
    CHECK(sdrb_batch_create(0, -5, SDRB_DATA_EMPTY, nullptr) != SDRB_OK);
    sdrb_batch* b = nullptr;
    CHECK(sdrb_batch_create(0, 16, 99, &b) != SDRB_OK);
    CHECK(sdrb_batch_upload_file(nullptr, "/nonexistent", 0, 0, nullptr) != SDRB_OK);
    int devices = -1;
    CHECK(sdrb_device_count(&devices) == SDRB_OK ? devices >= 0 : true);
}

}   // namespace

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    generator_fills_exactly_its_buffers();
    files_round_trip(dir, false);
    files_round_trip(dir, true);
    writer_refuses_bad_rows(dir);
    batch_entry_points_check_their_arguments();
    std::printf("host checks passed: %d\n", g_checks);
    return 0;
}
