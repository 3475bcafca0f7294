// Ingest timing through the C++ front end: N std::vector-backed SDRs -> host CSR (SDRBatch::pack_host) -> HBM
// (sdrb_batch_upload_csr), next to a plain page-locked copy of the same bytes.  Prints one JSON line.
//
// Two builds of this file:
//   sdr_amd/bench_ingest       rows are a stand-in with the memory layout of the reference's default SDR
//                              (a std::vector of 4-byte {id} or 8-byte {id, float} elements) -- what bench.py runs
//   oracle/_ref/bench_ingest   -DSDR_REFERENCE: rows are the reference's own SDR<SDRElem<int, EmptyData>> and
//                              SDR<SDRElem<int, UnitData>> objects (built only where the reference checkout exists)
//
//   bench_ingest [rows=1048576] [width=2048] [active=40] [repeats=5]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#ifdef SDR_REFERENCE
#include "SparseDistributedRepresentation/SDR.hpp"
#include "SparseDistributedRepresentation/DataTypes/UnitData.hpp"
using namespace sparse_distributed_representation;
using IdRow = SDR<SDRElem<int, EmptyData>>;
using UnitRow = SDR<SDRElem<int, UnitData>>;
static const char* kRows = "reference SDR objects";
#else
// rows with the layout and the accessors the packer needs (cbegin/cend over elements with id() and data())
struct NoData {};
struct FloatData {
    float v;
    float value() const { return v; }
    bool relevant() const { return v >= 0.1f; }   // tells UnitData from ArithData for the packer
    explicit FloatData(float x = 1.0f) : v(x) {}
};
template <typename D>
struct Elem {
    using id_type = int;
    using data_type = D;
    int id_;
    D data_;
    int id() const { return id_; }
    const D& data() const { return data_; }
};
template <>
struct Elem<NoData> {
    using id_type = int;
    using data_type = NoData;
    int id_;
    int id() const { return id_; }
    NoData data() const { return NoData(); }
};
template <typename D>
struct Row {
    using value_type = Elem<D>;
    std::vector<Elem<D>> v;
    auto cbegin() const { return v.cbegin(); }
    auto cend() const { return v.cend(); }
    size_t size() const { return v.size(); }
};
using IdRow = Row<NoData>;
using UnitRow = Row<FloatData>;
static const char* kRows = "stand-in rows with the reference's std::vector layout";
#endif

#include "sdr_amd/SDRBatch.hpp"

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename RowT>
static std::vector<RowT> make_rows(uint64_t seed, size_t n, int width, int active, bool payload) {
    std::vector<uint64_t> off(n + 1);
    std::vector<int32_t> ids(n * (size_t)active);
    std::vector<float> data(payload ? ids.size() : 0);
    sdr_amd::check(sdrb_synth_csr(seed, 0, n, width, active, off.data(), ids.data(), payload ? data.data() : nullptr));
    std::vector<RowT> rows(n);
    for (size_t r = 0; r < n; ++r) {
#ifdef SDR_REFERENCE
        using E = typename RowT::value_type;
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
            if constexpr (std::is_same_v<typename E::data_type, UnitData>) rows[r].push_back(E(ids[k], UnitData(data[k])));
            else rows[r].push_back(E(ids[k]));
        }
#else
        rows[r].v.reserve(active);
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
            typename RowT::value_type e;
            e.id_ = ids[k];
            if constexpr (sizeof(e) > 4) e.data_ = FloatData(data[k]);
            rows[r].v.push_back(e);
        }
#endif
    }
    return rows;
}

struct Timing {
    double first_upload_s = 0;   // the process's first large upload: the staging ring and the copy streams are made here
    double pack_s = 1e30, upload_s = 1e30, pinned_s = 1e30, pageable_s = 1e30;
    uint64_t bytes = 0;
};

template <typename RowT>
static Timing measure(const std::vector<RowT>& rows, int width, int repeats) {
    Timing t;
    void* pinned = nullptr;
    void* dev = nullptr;
    using RowData = typename RowT::value_type::data_type;
    // one batch, uploaded into again and again (its device storage is kept between uploads)
    sdr_amd::SDRBatch b(0, width, sdr_amd::detail::kind_of<RowData>());
    for (int rep = 0; rep < repeats + 1; ++rep) {   // the first round warms up (staging buffers, first-touch pages)
        const double t0 = now();
        auto h = sdr_amd::SDRBatch::pack_host(rows.begin(), rows.end());
        const double t1 = now();
        sdr_amd::check(sdrb_batch_upload_csr(const_cast<sdrb_batch*>(b.handle()), h.n_rows, h.off.get(), h.ids.get(), h.data.get(), nullptr));
        const double t2 = now();
        const uint64_t bytes = h.nnz * (h.data ? 8 : 4) + (h.n_rows + 1) * 4;   // what reaches the device
        if (!pinned) {
            t.bytes = bytes;
            if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc(&dev, bytes) != hipSuccess) {
                std::fprintf(stderr, "cannot allocate the comparison buffers\n");
                std::exit(1);
            }
            std::memset(pinned, 1, bytes);
        }
        // the same bytes from page-locked memory in one copy, and from pageable memory in one copy
        const double t3 = now();
        (void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice);
        const double t4 = now();
        (void)hipMemcpy(dev, h.ids.get(), h.nnz * 4, hipMemcpyHostToDevice);
        const double t5 = now();
        if (rep == 0) {
            t.first_upload_s = t2 - t1;
            continue;
        }
        t.pack_s = std::min(t.pack_s, t1 - t0);
        t.upload_s = std::min(t.upload_s, t2 - t1);
        t.pinned_s = std::min(t.pinned_s, t4 - t3);
        t.pageable_s = std::min(t.pageable_s, (t5 - t4) * (double)bytes / (double)(h.nnz * 4));
    }
    (void)hipHostFree(pinned);
    (void)hipFree(dev);
    return t;
}

static void print(const char* name, const Timing& t, size_t rows, bool last) {
    std::printf("\"%s\": {\"bytes\": %llu, \"pack_ms\": %.3f, \"pack_rows_per_s\": %.4g, \"pack_GBps\": %.2f, "
                "\"upload_ms\": %.3f, \"first_upload_ms\": %.3f, \"upload_GBps\": %.2f, \"pinned_h2d_ms\": %.3f, \"pinned_h2d_GBps\": %.2f, "
                "\"pageable_h2d_GBps\": %.2f, \"upload_vs_pinned\": %.3f, \"rows_per_s_end_to_end\": %.4g}%s",
                name, (unsigned long long)t.bytes, t.pack_s * 1e3, rows / t.pack_s, t.bytes / t.pack_s / 1e9,
                t.upload_s * 1e3, t.first_upload_s * 1e3, t.bytes / t.upload_s / 1e9, t.pinned_s * 1e3, t.bytes / t.pinned_s / 1e9,
                t.bytes / t.pageable_s / 1e9, t.pinned_s / t.upload_s, rows / (t.pack_s + t.upload_s), last ? "" : ", ");
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : (1u << 20);
    const int width = argc > 2 ? std::atoi(argv[2]) : 2048;
    const int active = argc > 3 ? std::atoi(argv[3]) : 40;
    const int repeats = argc > 4 ? std::atoi(argv[4]) : 5;
    try {
        Timing ids, unit;
        {
            const auto rows = make_rows<IdRow>(1, n, width, active, false);
            ids = measure(rows, width, repeats);
        }
        {
            const auto rows = make_rows<UnitRow>(1, n, width, active, true);
            unit = measure(rows, width, repeats);
        }
        std::printf("{\"rows\": %zu, \"width\": %d, \"active\": %d, \"row_type\": \"%s\", \"host_threads\": %d, ", n, width, active,
                    kRows, sdr_amd::detail::host_threads(n));
        print("empty", ids, n, false);
        print("unit", unit, n, true);
        std::printf("}\n");
    } catch (const std::exception& e) {
        std::fprintf(stderr, "bench_ingest: %s\n", e.what());
        return 1;
    }
    return 0;
}
