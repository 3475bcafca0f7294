// Perf lab (not shipped): which store pattern streams a [Q][N] uint32 matrix fastest on MI355X?
// Build: hipcc --offload-arch=gfx950 -O3 -o lab_store lab_store.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

constexpr int Q = 4096;
constexpr size_t N = 1 << 20;

// (a) linear 16B/lane grid-stride fill
__global__ void k_linear16(uint4* out, size_t n16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) out[i] = make_uint4(1, 2, 3, 4);
}
// (b) product pattern: block covers ROWS = blockDim.x*RPL columns, loops over all q; lane stores RPL dwords
template <int RPL, bool NT>
__global__ void k_pattern(uint32_t* out, size_t ld, int q0, int q1) {
    size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * RPL;
    for (int q = q0 + blockIdx.y; q < q1; q += gridDim.y) {
        uint32_t* p = out + (size_t)q * ld + j;
        if constexpr (RPL == 1) { if (NT) __builtin_nontemporal_store((uint32_t)q, p); else *p = q; }
        else if constexpr (RPL == 2) { uint2 v = make_uint2(q, q); if (NT) { __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p+1);} else *(uint2*)p = v; }
        else { typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 v = {(uint32_t)q,(uint32_t)q,(uint32_t)q,(uint32_t)q}; if (NT) __builtin_nontemporal_store(v, (v4*)p); else *(v4*)p = v; }
    }
}
// (c) q-major tiles like the product: for tile of 64 q: 64 stores back-to-back
template <int RPL>
__global__ void k_tiles(uint32_t* out, size_t ld) {
    size_t j = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * RPL;
    for (int t = 0; t < Q / 64; ++t) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            uint32_t* p = out + (size_t)(t * 64 + q) * ld + j;
            if constexpr (RPL == 1) *p = q + t;
            else { typedef uint32_t v4 __attribute__((ext_vector_type(4))); v4 v = {(uint32_t)q,(uint32_t)t,(uint32_t)q,(uint32_t)q}; *(v4*)p = v; }
        }
    }
}

// (c') the same tiles with 16 bytes per lane after a 4x4 exchange inside each quad of lanes: lane 4g+r stores
// columns 4g..4g+3 of row q0+r, so one instruction covers four rows x 256 B and a tile takes 16 instructions
__global__ void k_tiles_quad(uint32_t* out, size_t ld) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const size_t j = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~3u);
    const int r = threadIdx.x & 3;
    for (int t = 0; t < Q / 64; ++t) {
#pragma unroll
        for (int q4 = 0; q4 < 16; ++q4) {
            uint32_t* p = out + (size_t)(t * 64 + q4 * 4 + r) * ld + j;
            v4 v = {(uint32_t)q4, (uint32_t)t, (uint32_t)r, (uint32_t)q4};
            *(v4*)p = v;
        }
    }
}
// (c'') rows of a quad 8 apart (lane 4g+r stores row 8r+b): what the byte-sliced exchange of the count kernel yields
__global__ void k_tiles_quad8(uint32_t* out, size_t ld) {
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    const size_t j = (size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~3u);
    const int r = threadIdx.x & 3;
    for (int t = 0; t < Q / 64; ++t) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                uint32_t* p = out + (size_t)(t * 64 + 32 * h + 8 * r + b) * ld + j;
                v4 v = {(uint32_t)b, (uint32_t)t, (uint32_t)r, (uint32_t)h};
                *(v4*)p = v;
            }
    }
}


// (c3) the dword tiles with the block index remapped so that each XCD (blocks are dealt to the eight XCDs
// round robin) owns runs of C consecutive 1-KiB column segments; C = 0: one contiguous eighth of the columns
template <int NT>
__global__ void k_tiles_xcd(uint32_t* out, size_t ld, int C) {
    const uint32_t bid = blockIdx.x, nblk = gridDim.x, xcd = bid & 7, idx = bid >> 3;
    const uint32_t jb = C == 0 ? xcd * (nblk / 8) + idx : (idx / C) * 8 * C + xcd * C + idx % C;
    size_t j = (size_t)jb * blockDim.x + threadIdx.x;
    for (int t = 0; t < Q / 64; ++t) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            uint32_t* p = out + (size_t)(t * 64 + q) * ld + j;
            if (NT) __builtin_nontemporal_store((uint32_t)(q + t), p); else *p = q + t;
        }
    }
}
// (c4) tiles walked by a block that covers S consecutive 1-KiB segments per row (S stores of one row back to back)
template <int S>
__global__ void k_tiles_wide(uint32_t* out, size_t ld) {
    size_t j = (size_t)blockIdx.x * blockDim.x * S + threadIdx.x;
    for (int t = 0; t < Q / 64; ++t) {
#pragma unroll 8
        for (int q = 0; q < 64; ++q) {
            uint32_t* p = out + (size_t)(t * 64 + q) * ld + j;
#pragma unroll
            for (int s = 0; s < S; ++s) p[s * 256] = q + t + s;
        }
    }
}

// (d) each block owns one contiguous chunk and sweeps it front to back
template <int W>
__global__ void k_chunked(uint32_t* out, size_t chunk_elems) {
    typedef uint32_t vN __attribute__((ext_vector_type(W)));
    vN* p = (vN*)(out + (size_t)blockIdx.x * chunk_elems);
    size_t n = chunk_elems / W;
    vN v; for (int i = 0; i < W; ++i) v[i] = i;
    for (size_t i = threadIdx.x; i < n; i += blockDim.x) p[i] = v;
}
template <>
__global__ void k_chunked<1>(uint32_t* out, size_t chunk_elems) {
    uint32_t* p = out + (size_t)blockIdx.x * chunk_elems;
    for (size_t i = threadIdx.x; i < chunk_elems; i += blockDim.x) p[i] = i;
}

// (n) narrow count matrices: [Q][N] of EB-byte counts (EB = 1 or 2), a block = 256 columns, tiles of 64 rows.
// QUAD: what the count kernel does -- lane 4g+r stores the four columns 4g..4g+3 of row 8r+b (4 EB bytes), so a wave
//       instruction writes 64 EB bytes to each of four rows;
// else: one row per wave instruction, the wave's 64 lanes covering the block's 256 columns of it (4 EB bytes a lane):
//       what a stage in LDS would make of the same tile -- 256 EB contiguous bytes per instruction.
template <int EB, bool QUAD>
__global__ void k_tiles_narrow(unsigned char* out, size_t ld_bytes) {
    typedef uint32_t v2 __attribute__((ext_vector_type(2)));
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* base = out + (size_t)blockIdx.x * 256 * EB;
    for (int t = 0; t < Q / 64; ++t) {
        unsigned char* tile = base + (size_t)t * 64 * ld_bytes;
        if constexpr (QUAD) {
            const uint32_t r = tid & 3, g4 = tid & ~3u;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    unsigned char* p = tile + (size_t)(32 * h + 8 * r + b) * ld_bytes + (size_t)g4 * EB;
                    if constexpr (EB == 1) *reinterpret_cast<uint32_t*>(p) = (uint32_t)(t + b);
                    else *reinterpret_cast<v2*>(p) = v2{(uint32_t)t, (uint32_t)b};
                }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {   // wave w writes rows w, w + 4, ... of the tile
                unsigned char* p = tile + (size_t)(4 * k + wave) * ld_bytes + (size_t)lane * 4 * EB;
                if constexpr (EB == 1) *reinterpret_cast<uint32_t*>(p) = (uint32_t)(t + k);
                else *reinterpret_cast<v2*>(p) = v2{(uint32_t)t, (uint32_t)k};
            }
        }
    }
}

template <typename F>
float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    uint32_t* out; size_t bytes = (size_t)Q * N * 4;
    CK(hipMalloc(&out, bytes));
    if (argc > 1 && argv[1][0] == 'x') {   // block-to-column mappings and store hints for the dword tiles
        auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
        rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(out, 1, bytes, 0)); }));
        rep("tiles RPL1 block=256", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }));
        for (int C : {0, 1, 2, 4, 16, 64}) {
            char nm[64]; snprintf(nm, 64, "tiles xcd-remap C=%d", C);
            rep(nm, timeit([&] { k_tiles_xcd<0><<<N / 256, 256>>>(out, N, C); }));
        }
        rep("tiles NT", timeit([&] { k_tiles_xcd<1><<<N / 256, 256>>>(out, N, 1); }));
        rep("tiles NT xcd-remap C=0", timeit([&] { k_tiles_xcd<1><<<N / 256, 256>>>(out, N, 0); }));
        rep("tiles NT xcd-remap C=4", timeit([&] { k_tiles_xcd<1><<<N / 256, 256>>>(out, N, 4); }));
        rep("tiles wide S=2", timeit([&] { k_tiles_wide<2><<<N / 512, 256>>>(out, N); }));
        rep("tiles wide S=4", timeit([&] { k_tiles_wide<4><<<N / 1024, 256>>>(out, N); }));
        rep("tiles wide S=8", timeit([&] { k_tiles_wide<8><<<N / 2048, 256>>>(out, N); }));
        rep("tiles RPL1 block=512", timeit([&] { k_tiles<1><<<N / 512, 512>>>(out, N); }));
        rep("tiles RPL1 block=1024", timeit([&] { k_tiles<1><<<N / 1024, 1024>>>(out, N); }));
        for (int g : {1024, 4096, 16384}) { char nm[64]; size_t chunk = (size_t)Q * N / g;
            snprintf(nm, 64, "chunked x4 grid=%d block=256", g); rep(nm, timeit([&] { k_chunked<4><<<g, 256>>>(out, chunk); })); }
        rep("linear16 grid=8192", timeit([&] { k_linear16<<<8192, 256>>>((uint4*)out, bytes / 16); }));
        rep("tiles RPL1 block=256", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }));
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'n') {   // narrow count matrices: the kernel's quad pattern against whole-row pieces
        auto rep = [&](const char* name, float ms, size_t b) { printf("%-52s %8.3f ms  %7.1f GB/s\n", name, ms, b / ms / 1e6); fflush(stdout); };
        unsigned char* o8 = reinterpret_cast<unsigned char*>(out);
        rep("uint32 tiles RPL1 block=256 (256-B pieces)", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }), bytes);
        rep("uint16 quad pattern (4 rows x 128 B an instruction)", timeit([&] { k_tiles_narrow<2, true><<<N / 256, 256>>>(o8, N * 2); }), bytes / 2);
        rep("uint16 whole rows (512 B of one row an instruction)", timeit([&] { k_tiles_narrow<2, false><<<N / 256, 256>>>(o8, N * 2); }), bytes / 2);
        rep("uint8 quad pattern (4 rows x 64 B an instruction)", timeit([&] { k_tiles_narrow<1, true><<<N / 256, 256>>>(o8, N); }), bytes / 4);
        rep("uint8 whole rows (256 B of one row an instruction)", timeit([&] { k_tiles_narrow<1, false><<<N / 256, 256>>>(o8, N); }), bytes / 4);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'q') {   // only the quad-exchange shapes against the dword tiles
        auto rep = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
        rep("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(out, 1, bytes, 0)); }));
        rep("tiles RPL1 block=256", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }));
        rep("tiles quad (rows q0+r) block=256", timeit([&] { k_tiles_quad<<<N / 256, 256>>>(out, N); }));
        rep("tiles quad8 (rows 8r+b) block=256", timeit([&] { k_tiles_quad8<<<N / 256, 256>>>(out, N); }));
        rep("tiles RPL4 block=256", timeit([&] { k_tiles<4><<<N / 1024, 256>>>(out, N); }));
        rep("tiles RPL1 block=256", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }));
        return 0;
    }
    if (argc > 1) {  // calibration mode for PMC runs: one known-size pattern, three launches
        for (int i = 0; i < 3; ++i) k_tiles<1><<<N / 256, 256>>>(out, N);
        CK(hipDeviceSynchronize());
        printf("calibration: k_tiles<1> wrote %zu bytes per launch\n", bytes);
        return 0;
    }
    auto report = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    report("hipMemsetAsync", timeit([&] { CK(hipMemsetAsync(out, 1, bytes, 0)); }));
    {   // leading-dimension skew: is the 4 MiB power-of-two row stride a DRAM channel/bank problem?
        uint32_t* out2; CK(hipMalloc(&out2, (size_t)Q * (N + 4096) * 4));
        for (size_t pad : {0, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096}) {
            char nm[64]; snprintf(nm, 64, "tiles RPL1 block=256 ld=N+%zu", pad);
            report(nm, timeit([&] { k_tiles<1><<<N / 256, 256>>>(out2, N + pad); }));
        }
        for (size_t pad : {0, 64, 1024}) {
            char nm[64]; snprintf(nm, 64, "pattern RPL4 block=256 ld=N+%zu", pad);
            report(nm, timeit([&] { k_pattern<4, false><<<dim3(N / 1024, 1), 256>>>(out2, N + pad, 0, Q); }));
        }
        CK(hipFree(out2));
    }
    for (int g : {256, 512, 1024, 2048, 4096, 16384, 65536}) for (int bs : {256, 1024}) {
        char nm[64]; size_t chunk = (size_t)Q * N / g;
        snprintf(nm, 64, "chunked x4 grid=%d block=%d", g, bs); report(nm, timeit([&] { k_chunked<4><<<g, bs>>>(out, chunk); }));
        snprintf(nm, 64, "chunked x1 grid=%d block=%d", g, bs); report(nm, timeit([&] { k_chunked<1><<<g, bs>>>(out, chunk); }));
    }
    for (int g : {2048, 8192, 65536})
        { char nm[64]; snprintf(nm, 64, "linear16 grid=%d", g); report(nm, timeit([&] { k_linear16<<<g, 256>>>((uint4*)out, bytes / 16); })); }
    for (int bs : {256, 512, 1024}) {
        char nm[96];
        snprintf(nm, 96, "pattern RPL1 block=%d (all q in-block)", bs);
        report(nm, timeit([&] { k_pattern<1, false><<<dim3(N / bs, 1), bs>>>(out, N, 0, Q); }));
        snprintf(nm, 96, "pattern RPL1 NT block=%d", bs);
        report(nm, timeit([&] { k_pattern<1, true><<<dim3(N / bs, 1), bs>>>(out, N, 0, Q); }));
        snprintf(nm, 96, "pattern RPL4 block=%d", bs);
        report(nm, timeit([&] { k_pattern<4, false><<<dim3(N / bs / 4, 1), bs>>>(out, N, 0, Q); }));
        snprintf(nm, 96, "pattern RPL4 NT block=%d", bs);
        report(nm, timeit([&] { k_pattern<4, true><<<dim3(N / bs / 4, 1), bs>>>(out, N, 0, Q); }));
        snprintf(nm, 96, "pattern RPL2 block=%d", bs);
        report(nm, timeit([&] { k_pattern<2, false><<<dim3(N / bs / 2, 1), bs>>>(out, N, 0, Q); }));
    }
    report("tiles RPL1 block=256", timeit([&] { k_tiles<1><<<N / 256, 256>>>(out, N); }));
    report("tiles RPL4 block=256", timeit([&] { k_tiles<4><<<N / 1024, 256>>>(out, N); }));
    report("tiles RPL1 block=1024", timeit([&] { k_tiles<1><<<N / 1024, 1024>>>(out, N); }));
    // y-split: each block handles a slice of q (more, shorter blocks)
    report("pattern RPL1 block=256 ysplit=8", timeit([&] { k_pattern<1, false><<<dim3(N / 256, 8), 256>>>(out, N, 0, Q); }));
    report("pattern RPL4 block=256 ysplit=8", timeit([&] { k_pattern<4, false><<<dim3(N / 1024, 8), 256>>>(out, N, 0, Q); }));
    return 0;
}
