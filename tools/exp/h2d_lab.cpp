// Lab: what a host-to-device copy of N bytes costs when it is cut into pieces that travel from a ring of page-locked
// buffers -- by piece size, number of streams, and with or without an event per piece.  Not product code.
//   g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include h2d_lab.cpp -L/opt/rocm/lib -lamdhip64 -o _bin/h2d_lab
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime_api.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); std::exit(1); } } while (0)

int main() {
    const size_t total = 172u << 20;
    void* dev;
    CK(hipMalloc(&dev, total));
    void* big;
    CK(hipHostMalloc(&big, total, hipHostMallocPortable));
    std::memset(big, 1, total);
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        CK(hipMemcpy(dev, big, total, hipMemcpyHostToDevice));
        std::printf("one copy from page-locked memory: %.3f ms\n", (now() - t0) * 1e3);
    }
    hipStream_t streams[8];
    for (auto& s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int kSlots = 32;
    for (size_t piece : {256u << 10, 1u << 20, 2u << 20, 4u << 20, 8u << 20, 16u << 20}) {
        std::vector<hipEvent_t> ev(kSlots);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (int ns : {1, 2, 4}) {
            for (int with_events : {0, 1}) {
                double best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    const double t0 = now();
                    size_t i = 0;
                    for (size_t at = 0; at < total; at += piece, ++i) {
                        const size_t n = total - at < piece ? total - at : piece;
                        hipStream_t s = streams[i % ns];
                        // source: the big pinned buffer (a ring would hold the same bytes)
                        CK(hipMemcpyAsync((char*)dev + at, (char*)big + at, n, hipMemcpyHostToDevice, s));
                        if (with_events) CK(hipEventRecord(ev[i % kSlots], s));
                    }
                    for (int k = 0; k < ns; ++k) CK(hipStreamSynchronize(streams[k]));
                    best = std::min(best, now() - t0);
                }
                std::printf("piece %5zu KiB, %d stream(s), events %d: %.3f ms = %.1f GB/s\n", piece >> 10, ns, with_events, best * 1e3,
                            total / best / 1e9);
            }
        }
        for (auto& e : ev) CK(hipEventDestroy(e));
    }
    // the legacy default stream, as a caller who passes NULL gets it
    for (size_t piece : {1u << 20, 4u << 20}) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            const double t0 = now();
            for (size_t at = 0; at < total; at += piece) {
                const size_t n = total - at < piece ? total - at : piece;
                CK(hipMemcpyAsync((char*)dev + at, (char*)big + at, n, hipMemcpyHostToDevice, nullptr));
            }
            CK(hipStreamSynchronize(nullptr));
            best = std::min(best, now() - t0);
        }
        std::printf("piece %5zu KiB, NULL stream: %.3f ms = %.1f GB/s\n", piece >> 10, best * 1e3, total / best / 1e9);
    }
    return 0;
}
