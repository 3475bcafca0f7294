// Element operations ande/ore/xore/rme over every pair of two batches:
// count (a size op) -> exclusive scan -> fill.  This file holds the scan and the general fill.
//
// Fill: one lane owns pair (i, j) and walks row i of A against row j of B with the
// three-way merge of the reference's union visitor orv (SDR.hpp:1099-1150), emitting
//   ande: kept matches only                                  (SDR.hpp:1015-1021)
//   ore : this-only, converted arg-only, every match          (SDR.hpp:1207-1217)
//   xore: this-only, converted arg-only, kept matches         (SDR.hpp:1332-1344)
//   rme : this-only and kept matches                          (SDR.hpp:1540-1560)
// in ascending id order at the offset the scan assigned to the pair.  An arg-only element
// is converted to the result's element type (SDRElem.hpp:29-30): an EmptyData argument
// becomes UnitData 1.0 (UnitData.hpp:13-14), a UnitData argument loses its payload in an
// EmptyData result (EmptyData.hpp:25-27).
#include <type_traits>

#include "sdrb_device.h"

#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;

// VISIT (ids only, ande / ore): instead of elements, the positions a visitor of andv / orv would be called
// with -- out_ids (as uint32) gets the position in `this`, out_data (as uint32) the position in `arg`,
// 0xFFFFFFFF for the absent side of a single.
template <typename ID, int OP, bool SWAP, bool A_UNIT, bool B_UNIT, bool VISIT = false, bool ARITH = false>
__global__ __launch_bounds__(kBlock) void k_elem_fill(
    const uint32_t* __restrict__ a_off, const ID* __restrict__ a_ids,
    const float* __restrict__ a_data, uint64_t na, const uint32_t* __restrict__ b_off,
    const ID* __restrict__ b_ids, const float* __restrict__ b_data, uint64_t nb,
    const uint32_t* __restrict__ out_off, ID* __restrict__ out_ids,
    float* __restrict__ out_data, uint32_t rowwise) {
    constexpr bool THIS_UNIT = SWAP ? B_UNIT : A_UNIT;
    constexpr bool ARG_UNIT = SWAP ? A_UNIT : B_UNIT;
    constexpr bool KEEP_THIS_ONLY = OP != SDRB_OP_AND;
    constexpr bool KEEP_ARG_ONLY = OP == SDRB_OP_OR || OP == SDRB_OP_XOR;
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nb) return;
    const uint32_t b_beg = b_off[j];
    const uint32_t lb = b_off[j + 1] - b_beg;
    const ID* bi = b_ids + b_beg;
    const float* bd = B_UNIT ? b_data + b_beg : nullptr;

    // all pairs, or row-wise: row j of A only (result j)
    for (uint64_t i = rowwise ? j : blockIdx.y; i < na; i += rowwise ? na : gridDim.y) {
        const uint32_t a_beg = a_off[i];
        const uint32_t la = a_off[i + 1] - a_beg;
        const ID* ai = a_ids + a_beg;
        const float* ad = A_UNIT ? a_data + a_beg : nullptr;
        uint32_t w = out_off[rowwise ? j : i * nb + j];
        constexpr uint32_t NONE = 0xFFFFFFFFu;
        // a visitor's call: positions in A_i and B_j (NONE = absent), stored as (this, arg)
        auto emit_pos = [&](uint32_t pa, uint32_t pb) {
            reinterpret_cast<uint32_t*>(out_ids)[w] = SWAP ? pb : pa;
            reinterpret_cast<uint32_t*>(out_data)[w] = SWAP ? pa : pb;
            ++w;
        };
        auto emit = [&](ID id, float d) {
            out_ids[w] = id;
            if constexpr (THIS_UNIT) out_data[w] = d;
            ++w;
        };
        // an element present only in A / only in B
        auto only_a = [&](uint32_t pa) {
            if constexpr (VISIT) {   // orv's this/arg visitor; andv has none
                if constexpr (OP == SDRB_OP_OR) emit_pos(pa, NONE);
                return;
            }
            const float va = A_UNIT ? ad[pa] : 0.0f;
            if constexpr (!SWAP) {
                if constexpr (KEEP_THIS_ONLY) emit(ai[pa], va);
            } else {
                if constexpr (KEEP_ARG_ONLY) emit(ai[pa], converted_arg<ARG_UNIT, ARITH>(va));
            }
        };
        auto only_b = [&](uint32_t pb) {
            if constexpr (VISIT) {
                if constexpr (OP == SDRB_OP_OR) emit_pos(NONE, pb);
                return;
            }
            const float vb = B_UNIT ? bd[pb] : 0.0f;
            if constexpr (SWAP) {
                if constexpr (KEEP_THIS_ONLY) emit(bi[pb], vb);
            } else {
                if constexpr (KEEP_ARG_ONLY) emit(bi[pb], converted_arg<ARG_UNIT, ARITH>(vb));
            }
        };
        uint32_t pa = 0, pb = 0;
        while (pa < la && pb < lb) {
            const ID x = ai[pa];
            const ID y = bi[pb];
            if (x == y) {
                if constexpr (VISIT) {
                    emit_pos(pa, pb);
                } else {
                    const float va = A_UNIT ? ad[pa] : 0.0f;
                    const float vb = B_UNIT ? bd[pb] : 0.0f;
                    const float d = SWAP ? match_data<OP, ARG_UNIT, ARITH>(vb, va) : match_data<OP, ARG_UNIT, ARITH>(va, vb);
                    if (match_kept<OP, THIS_UNIT, ARITH>(d)) emit(x, d);
                }
                ++pa;
                ++pb;
            } else if (x < y) {
                only_a(pa++);
            } else {
                only_b(pb++);
            }
        }
        if constexpr (SWAP ? KEEP_ARG_ONLY : KEEP_THIS_ONLY) {
            while (pa < la) only_a(pa++);
        }
        if constexpr (SWAP ? KEEP_THIS_ONLY : KEEP_ARG_ONLY) {
            while (pb < lb) only_b(pb++);
        }
    }
}

struct FillArgs {
    const sdrb_batch* a;
    const sdrb_batch* b;
    const uint32_t* out_off;
    void* out_ids;
    float* out_data;
    bool rowwise;
    hipStream_t stream;
};

// ArithData<float> payloads (32-bit ids; `this` carries the payloads -- an EmptyData `this` is served as ids only)
template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
void launch_arith(const FillArgs& g) {
    const uint32_t gx = (uint32_t)((g.b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = g.rowwise ? 1u : (uint32_t)(g.a->n_rows < 65535 ? g.a->n_rows : 65535);
    hipLaunchKernelGGL((k_elem_fill<int32_t, OP, SWAP, A_UNIT, B_UNIT, false, true>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                       g.a->d_offsets, g.a->d_ids, g.a->d_data, g.a->n_rows, g.b->d_offsets, g.b->d_ids,
                       g.b->d_data, g.b->n_rows, g.out_off, static_cast<int32_t*>(g.out_ids), g.out_data,
                       g.rowwise ? 1u : 0u);
}

template <int OP>
void launch_arith_kinds(const FillArgs& g, bool swap) {
    const bool ap = g.a->data_kind != SDRB_DATA_EMPTY, bp = g.b->data_kind != SDRB_DATA_EMPTY;
    if (ap && bp) {
        if (swap) launch_arith<OP, true, true, true>(g);
        else launch_arith<OP, false, true, true>(g);
    } else if (ap) {
        launch_arith<OP, false, true, false>(g);   // this = A (payloads), arg = B (ids only)
    } else {
        launch_arith<OP, true, false, true>(g);    // this = B (payloads), arg = A (ids only)
    }
}

template <int OP, bool SWAP, bool A_UNIT, bool B_UNIT>
void launch(const FillArgs& g) {
    const uint32_t gx = (uint32_t)((g.b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = g.rowwise ? 1u : (uint32_t)(g.a->n_rows < 65535 ? g.a->n_rows : 65535);
    if (g.a->id_bits == 64)
        hipLaunchKernelGGL((k_elem_fill<long long, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                           g.a->d_offsets, reinterpret_cast<const long long*>(g.a->d_ids), g.a->d_data, g.a->n_rows,
                           g.b->d_offsets, reinterpret_cast<const long long*>(g.b->d_ids), g.b->d_data, g.b->n_rows,
                           g.out_off, static_cast<long long*>(g.out_ids), g.out_data, g.rowwise ? 1u : 0u);
    else
        hipLaunchKernelGGL((k_elem_fill<int32_t, OP, SWAP, A_UNIT, B_UNIT>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                           g.a->d_offsets, g.a->d_ids, g.a->d_data, g.a->n_rows, g.b->d_offsets, g.b->d_ids,
                           g.b->d_data, g.b->n_rows, g.out_off, static_cast<int32_t*>(g.out_ids), g.out_data,
                           g.rowwise ? 1u : 0u);
}

template <int OP, bool SWAP>
void launch_kinds(const FillArgs& g) {
    const bool au = g.a->data_kind == SDRB_DATA_UNIT, bu = g.b->data_kind == SDRB_DATA_UNIT;
    if (au && bu) launch<OP, SWAP, true, true>(g);
    else if (au) launch<OP, SWAP, true, false>(g);
    else if (bu) launch<OP, SWAP, false, true>(g);
    else launch<OP, SWAP, false, false>(g);
}

template <int OP>
void launch_swap(const FillArgs& g, bool swap) {
    if (swap) launch_kinds<OP, true>(g);
    else launch_kinds<OP, false>(g);
}

}  // namespace

namespace {

// ---- exclusive scan of the offsets: tile sums -> scan of the sums -> local scans --------------------------
// Three launches with no waiting between workgroups: (1) every tile of 4096 entries is summed exactly, in 64
// bits; (2) one workgroup scans the tile sums (a few thousand values) and writes the grand total; (3) every
// tile scans its entries locally, starting from its tile's prefix.  The entries are read twice -- the second
// time out of the Infinity Cache for arrays that fit it -- and written once.  (A single-pass variant with
// chained tiles and decoupled look-back was measured slower here: 70 us against 40 us on 16.7 M entries; the
// polling of a few status lines by hundreds of resident workgroups is what held it back.)
// Offsets are kept modulo 2^32; the 64-bit total tells the caller whether they are valid.
constexpr int kScanSlabs = 4;                        // a thread holds 4 consecutive entries of each slab
constexpr int kScanSlab = kBlock * 4;                // 1024 entries
constexpr int kScanTile = kScanSlabs * kScanSlab;    // 4096 entries
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

template <bool ALIGNED>
__device__ __forceinline__ void load_scan_tile(const uint32_t* __restrict__ data, uint64_t base, uint64_t n_valid,
                                               uint32_t tid, u32x4 (&v)[kScanSlabs]) {
#pragma unroll
    for (int k = 0; k < kScanSlabs; ++k) {
        const uint64_t idx = base + (uint64_t)k * kScanSlab + tid * 4u;
        if (ALIGNED && idx + 4 <= n_valid) {
            v[k] = *reinterpret_cast<const u32x4*>(data + idx);
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[k][c] = idx + c < n_valid ? data[idx + c] : 0u;
        }
    }
}

// (1) sums[tile] = exact sum of the tile's entries
template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void k_scan_tile_sums(const uint32_t* __restrict__ data, uint64_t n_valid,
                                                           unsigned long long* __restrict__ sums) {
    __shared__ unsigned long long s_wave_total[kBlock / 64];
    const uint32_t tid = threadIdx.x;
    u32x4 v[kScanSlabs];
    load_scan_tile<ALIGNED>(data, (uint64_t)blockIdx.x * kScanTile, n_valid, tid, v);
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanSlabs; ++k) mine += (unsigned long long)v[k][0] + v[k][1] + v[k][2] + v[k][3];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d, 64);
    if ((tid & 63) == 0) s_wave_total[tid >> 6] = mine;
    __syncthreads();
    if (tid == 0) {
        unsigned long long total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) total += s_wave_total[w];
        sums[blockIdx.x] = total;
    }
}

// (2) one workgroup: sums[] -> exclusive prefixes in place, grand total to *total_out
constexpr int kSumsBlock = 1024;
__global__ __launch_bounds__(kSumsBlock) void k_scan_of_sums(unsigned long long* __restrict__ sums, uint32_t n_tiles,
                                                             unsigned long long* __restrict__ total_out) {
    __shared__ unsigned long long s_wave[kSumsBlock / 64];
    __shared__ unsigned long long s_carry;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    // chunks of 4096 tile sums (four per thread), chained through s_carry
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += kSumsBlock * 4) {
        unsigned long long x[4], mine = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = c0 + tid * 4 + c;
            x[c] = t < n_tiles ? sums[t] : 0ull;
            mine += x[c];
        }
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d, 64);
            if (lane >= (uint32_t)d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned long long run = s_carry + incl - mine;
        unsigned long long chunk_total = 0;
#pragma unroll
        for (int w = 0; w < kSumsBlock / 64; ++w) {
            if ((uint32_t)w < wave) run += s_wave[w];
            chunk_total += s_wave[w];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t t = c0 + tid * 4 + c;
            if (t < n_tiles) sums[t] = run;
            run += x[c];
        }
        __syncthreads();
        if (tid == 0) s_carry += chunk_total;
        __syncthreads();
    }
    if (tid == 0 && total_out) *total_out = s_carry;
}

// (3) data[] -> exclusive prefix sums in place, tile by tile, each from its tile's prefix
template <bool ALIGNED>
__global__ __launch_bounds__(kBlock) void k_scan_tiles(uint32_t* __restrict__ data, uint64_t n, uint64_t n_valid,
                                                       const unsigned long long* __restrict__ prefixes) {
    __shared__ uint32_t s_wave_sum[kScanSlabs][kBlock / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile;
    u32x4 v[kScanSlabs];
    load_scan_tile<ALIGNED>(data, base, n_valid, tid, v);
    const uint32_t prefix = (uint32_t)prefixes[blockIdx.x];
    uint32_t incl[kScanSlabs];      // inclusive scan of the threads' slab sums over the wave (mod 2^32)
#pragma unroll
    for (int k = 0; k < kScanSlabs; ++k) {
        incl[k] = wave_inclusive_scan_u32(v[k][0] + v[k][1] + v[k][2] + v[k][3]);
        if (lane == 63) s_wave_sum[k][wave] = incl[k];
    }
    __syncthreads();
    uint32_t slab_base = prefix;
#pragma unroll
    for (int k = 0; k < kScanSlabs; ++k) {
        uint32_t before = slab_base;   // everything ahead of this wave's part of slab k
        uint32_t slab_total = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            const uint32_t ws = s_wave_sum[k][w];
            if ((uint32_t)w < wave) before += ws;
            slab_total += ws;
        }
        const uint32_t own = v[k][0] + v[k][1] + v[k][2] + v[k][3];
        uint32_t run = before + incl[k] - own;
        u32x4 o;
        o[0] = run;
        run += v[k][0];
        o[1] = run;
        run += v[k][1];
        o[2] = run;
        run += v[k][2];
        o[3] = run;
        const uint64_t idx = base + (uint64_t)k * kScanSlab + tid * 4u;
        if (ALIGNED && idx + 4 <= n) {
            *reinterpret_cast<u32x4*>(data + idx) = o;
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (idx + c < n) data[idx + c] = o[c];
        }
        slab_base += slab_total;
    }
}

}  // namespace

int launch_exclusive_scan_u32(int device, uint32_t* data, uint64_t n, uint64_t n_valid,
                              unsigned long long* total_dev, hipStream_t stream) {
    if (n == 0) {
        if (total_dev) SDRB_HIP(hipMemsetAsync(total_dev, 0, sizeof(unsigned long long), stream));
        return SDRB_OK;
    }
    const uint64_t tiles = (n + kScanTile - 1) / kScanTile;
    if (tiles > 0x7FFFFFFFull) {
        set_error("scan of %llu entries is too long", (unsigned long long)n);
        return SDRB_ERR_OVERFLOW;
    }
    unsigned long long* sums =
        static_cast<unsigned long long*>(workspace(device, stream, 1, (size_t)tiles * sizeof(unsigned long long)));
    if (!sums) return SDRB_ERR_HIP;
    const bool aligned = (reinterpret_cast<uintptr_t>(data) & 15u) == 0;
    if (aligned) hipLaunchKernelGGL(k_scan_tile_sums<true>, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, data, n_valid, sums);
    else hipLaunchKernelGGL(k_scan_tile_sums<false>, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, data, n_valid, sums);
    hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(kSumsBlock), 0, stream, sums, (uint32_t)tiles, total_dev);
    if (aligned) hipLaunchKernelGGL(k_scan_tiles<true>, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, data, n, n_valid, sums);
    else hipLaunchKernelGGL(k_scan_tiles<false>, dim3((uint32_t)tiles), dim3(kBlock), 0, stream, data, n, n_valid, sums);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

namespace {
template <typename ID, int OP, bool SWAP>
void launch_visit(const FillArgs& g, uint32_t gx, uint32_t gy) {
    hipLaunchKernelGGL((k_elem_fill<ID, OP, SWAP, false, false, true>), dim3(gx, gy), dim3(kBlock), 0, g.stream,
                       g.a->d_offsets, reinterpret_cast<const ID*>(g.a->d_ids), nullptr, g.a->n_rows, g.b->d_offsets,
                       reinterpret_cast<const ID*>(g.b->d_ids), nullptr, g.b->n_rows, g.out_off,
                       static_cast<ID*>(g.out_ids), g.out_data, g.rowwise ? 1u : 0u);
}
}  // namespace

// op: SDRB_OP_AND (andv) or SDRB_OP_OR (orv); this_pos / arg_pos: uint32 positions, any id width
int launch_elem_visit(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, const uint32_t* out_offsets,
                      uint32_t* this_pos, uint32_t* arg_pos, bool rowwise, hipStream_t stream) {
    // the kernel indexes its id output by the id type: for 64-bit ids positions are written through a
    // 32-bit view, which emit_pos does itself, so the output pointers are passed untyped
    FillArgs g{a, b, out_offsets, this_pos, reinterpret_cast<float*>(arg_pos), rowwise, stream};
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    const uint32_t gy = rowwise ? 1u : (uint32_t)(a->n_rows < 65535 ? a->n_rows : 65535);
    const bool wide = a->id_bits == 64, is_and = op == SDRB_OP_AND;
    if (wide) {
        if (is_and) { if (swap) launch_visit<long long, SDRB_OP_AND, true>(g, gx, gy); else launch_visit<long long, SDRB_OP_AND, false>(g, gx, gy); }
        else { if (swap) launch_visit<long long, SDRB_OP_OR, true>(g, gx, gy); else launch_visit<long long, SDRB_OP_OR, false>(g, gx, gy); }
    } else {
        if (is_and) { if (swap) launch_visit<int32_t, SDRB_OP_AND, true>(g, gx, gy); else launch_visit<int32_t, SDRB_OP_AND, false>(g, gx, gy); }
        else { if (swap) launch_visit<int32_t, SDRB_OP_OR, true>(g, gx, gy); else launch_visit<int32_t, SDRB_OP_OR, false>(g, gx, gy); }
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

int launch_elem_fill(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap,
                     const uint32_t* out_offsets, void* out_ids, float* out_data, bool rowwise,
                     hipStream_t stream) {
    FillArgs g{a, b, out_offsets, out_ids, out_data, rowwise, stream};
    if (a->data_kind == SDRB_DATA_ARITH || b->data_kind == SDRB_DATA_ARITH) {   // caller: `this` is the ArithData side
        switch (op) {
        case SDRB_OP_AND: launch_arith_kinds<SDRB_OP_AND>(g, swap); break;
        case SDRB_OP_OR: launch_arith_kinds<SDRB_OP_OR>(g, swap); break;
        case SDRB_OP_RM: launch_arith_kinds<SDRB_OP_RM>(g, swap); break;
        default: set_error("ArithData has no xor (ArithData.hpp:110-114)"); return SDRB_ERR_UNSUPPORTED;
        }
        SDRB_HIP(hipGetLastError());
        return SDRB_OK;
    }
    switch (op) {
    case SDRB_OP_AND: launch_swap<SDRB_OP_AND>(g, swap); break;
    case SDRB_OP_OR: launch_swap<SDRB_OP_OR>(g, swap); break;
    case SDRB_OP_XOR: launch_swap<SDRB_OP_XOR>(g, swap); break;
    case SDRB_OP_RM: launch_swap<SDRB_OP_RM>(g, swap); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

__global__ __launch_bounds__(256) void k_widen_ids(const int32_t* __restrict__ in, int64_t* __restrict__ out, uint64_t n) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) out[k] = (int64_t)in[k];
}

__global__ __launch_bounds__(256) void k_narrow_result_ids(const int64_t* __restrict__ in, int32_t* __restrict__ out, uint64_t n) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) out[k] = (int32_t)in[k];
}

// ids known to fit 32 bits (elements of a 32-bit `this` that ran through the 64-bit kernels)
int launch_narrow_ids(const int64_t* in, int32_t* out, uint64_t n, hipStream_t stream) {
    if (n == 0) return SDRB_OK;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_narrow_result_ids, dim3((uint32_t)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, in, out, n);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <typename Id>
__global__ __launch_bounds__(256) void k_shift_ids(Id* __restrict__ ids, uint64_t n, Id amount) {
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) ids[k] += amount;
}

// ids[k] += amount (SDR::shift, SDR.hpp:1579-1590); the caller has checked that nothing overflows
int launch_shift_ids(void* ids, int id_bits, uint64_t n, int64_t amount, hipStream_t stream) {
    if (n == 0) return SDRB_OK;
    const uint64_t blocks = (n + 255) / 256;
    const dim3 grid((uint32_t)(blocks < 65536 ? blocks : 65536));
    if (id_bits == 64) hipLaunchKernelGGL((k_shift_ids<long long>), grid, dim3(256), 0, stream, static_cast<long long*>(ids), n, (long long)amount);
    else hipLaunchKernelGGL((k_shift_ids<int32_t>), grid, dim3(256), 0, stream, static_cast<int32_t*>(ids), n, (int32_t)amount);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

template <typename Id>
__global__ __launch_bounds__(256) void k_rebase_ids(const Id* __restrict__ in, int32_t* __restrict__ out, uint64_t n, Id base) {
    // (unsigned arithmetic: the difference fits 31 bits, the operands' own distance from zero need not fit the id type's
    // signed range once the other batch's smallest id is subtracted)
    using U = std::make_unsigned_t<Id>;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256)
        out[k] = (int32_t)(uint32_t)((U)in[k] - (U)base);
}

int launch_rebase_ids(const void* in, int id_bits, int32_t* out, uint64_t n, int64_t base, hipStream_t stream) {
    if (n == 0) return SDRB_OK;
    const uint64_t blocks = (n + 255) / 256;
    const dim3 grid((uint32_t)(blocks < 65536 ? blocks : 65536));
    if (id_bits == 64) hipLaunchKernelGGL((k_rebase_ids<long long>), grid, dim3(256), 0, stream, static_cast<const long long*>(in), out, n, (long long)base);
    else hipLaunchKernelGGL((k_rebase_ids<int32_t>), grid, dim3(256), 0, stream, static_cast<const int32_t*>(in), out, n, (int32_t)base);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

// the element count of a result lives on the device (its last offset): the grid is fixed and strides over it
__global__ __launch_bounds__(256) void k_unbase_ids32(int32_t* __restrict__ ids, const uint32_t* __restrict__ n_dev, int32_t base) {
    const uint64_t n = *n_dev;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256)
        ids[k] = (int32_t)((uint32_t)ids[k] + (uint32_t)base);
}
__global__ __launch_bounds__(256) void k_unbase_ids64(const int32_t* __restrict__ in, int64_t* __restrict__ out,
                                                      const uint32_t* __restrict__ n_dev, long long base) {
    const uint64_t n = *n_dev;
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) out[k] = (long long)in[k] + base;
}

int launch_unbase_ids32(int32_t* ids, const uint32_t* n_dev, int32_t base, hipStream_t stream) {
    hipLaunchKernelGGL(k_unbase_ids32, dim3(2048), dim3(256), 0, stream, ids, n_dev, base);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}
int launch_unbase_ids64(const int32_t* in, int64_t* out, const uint32_t* n_dev, int64_t base, hipStream_t stream) {
    hipLaunchKernelGGL(k_unbase_ids64, dim3(2048), dim3(256), 0, stream, in, out, n_dev, (long long)base);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

int launch_widen_ids(const int32_t* in, int64_t* out, uint64_t n, hipStream_t stream) {
    if (n == 0) return SDRB_OK;
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_widen_ids, dim3((uint32_t)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, in, out, n);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
