// Transpose of a batch seen as a sparse matrix (rows = the batch's rows, columns = ids):
// the result has one row per id c of the declared width, listing the rows of the source that
// contain c in ascending order, each with the payload it had there.
//
// The reference's transpose (SDR.hpp:1807-1826) walks a row-major matrix in column order with a
// priority queue keyed by the column id (MatrixUtils.hpp:13-75) and appends (row id, payload) to
// the column's bucket (MatrixUtils.hpp:150-190).  On the device the same regrouping is a STABLE
// sort of the elements by id: the packed CSR already lists elements by ascending row, so within
// an id the sorted order is ascending row -- the order the reference's own known answers show
// (src/test.cpp:517-571).  The sort is a least-significant-digit radix sort over the id bits, four bits a pass,
// stable by construction: a wave owns a contiguous tile of the elements; pass by pass it counts its tile's digits
// (k_digit_counts), the counts of all (digit, tile) pairs are scanned in digit-major order, and the wave walks its
// tile again in order, ranking the 64 elements of a turn with one ballot per digit value and sending each element to
// its digit's next free place (k_digit_scatter).  The other kernels label elements with their rows' positions, count
// the columns and gather.
#include <utility>

#include "sdrb_internal.h"

namespace sdrb {

namespace {

// element k -> the row that holds it, and its id as an unsigned sort key; one wave per row
__global__ __launch_bounds__(256) void k_label_elements(const uint32_t* __restrict__ off,
                                                        const int32_t* __restrict__ ids, uint32_t n_rows,
                                                        uint32_t* __restrict__ keys, uint32_t* __restrict__ elem,
                                                        uint32_t* __restrict__ col_count) {
    const uint32_t row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const uint32_t beg = off[row], end = off[row + 1];
    for (uint32_t k = beg + lane; k < end; k += 64) {
        const uint32_t c = (uint32_t)ids[k];
        keys[k] = c;
        elem[k] = k;
        atomicAdd(&col_count[c], 1u);
    }
}

// first element index of every row, scattered so that a max-scan would label elements; cheaper here:
// each sorted element looks its row up by binary search over the (L2-resident) offsets
__global__ __launch_bounds__(256) void k_gather_transposed(const uint32_t* __restrict__ src_off, uint32_t n_rows,
                                                           const float* __restrict__ src_data,
                                                           const uint32_t* __restrict__ sorted_elem, uint64_t nnz,
                                                           int32_t* __restrict__ dst_ids, float* __restrict__ dst_data) {
    const uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= nnz) return;
    const uint32_t k = sorted_elem[p];
    uint32_t lo = 0, hi = n_rows;            // last row whose offset is <= k
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (src_off[mid] <= k) lo = mid;
        else hi = mid;
    }
    dst_ids[p] = (int32_t)lo;
    if (dst_data) dst_data[p] = src_data[k];
}

constexpr int kDigitBits = 4, kDigits = 1 << kDigitBits;
constexpr uint32_t kSortTile = 4096;   // elements a wave owns

// counts[d * n_tiles + tile] = elements of the tile whose digit (key >> shift) & 15 is d
__global__ __launch_bounds__(64) void k_digit_counts(const uint32_t* __restrict__ keys, uint64_t n, uint32_t shift,
                                                     uint32_t n_tiles, uint32_t* __restrict__ counts) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)tile * kSortTile, t1 = t0 + kSortTile < n ? t0 + kSortTile : n;
    uint32_t mine[kDigits];
#pragma unroll
    for (int d = 0; d < kDigits; ++d) mine[d] = 0;
    for (uint64_t k = t0 + lane; k < t1; k += 64) {
        const uint32_t digit = (keys[k] >> shift) & (kDigits - 1);
#pragma unroll
        for (int d = 0; d < kDigits; ++d) mine[d] += digit == (uint32_t)d ? 1u : 0u;
    }
#pragma unroll
    for (int d = 0; d < kDigits; ++d) {
        uint32_t v = mine[d];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += (uint32_t)__shfl_xor((int)v, s, 64);
        if (lane == 0) counts[(uint64_t)d * n_tiles + tile] = v;
    }
}

// starts = the exclusive scan of counts: where the tile's elements of each digit go.  Elements keep their order.
__global__ __launch_bounds__(64) void k_digit_scatter(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                      uint64_t n, uint32_t shift, uint32_t n_tiles,
                                                      const uint32_t* __restrict__ starts, uint32_t* __restrict__ keys_out,
                                                      uint32_t* __restrict__ vals_out) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)tile * kSortTile, t1 = t0 + kSortTile < n ? t0 + kSortTile : n;
    uint32_t next[kDigits];   // uniform over the wave: the next free place of each digit
#pragma unroll
    for (int d = 0; d < kDigits; ++d) next[d] = starts[(uint64_t)d * n_tiles + tile];
    const unsigned long long below = (1ull << lane) - 1ull;
    for (uint64_t k0 = t0; k0 < t1; k0 += 64) {
        const uint64_t k = k0 + lane;
        const bool live = k < t1;
        const uint32_t key = live ? keys[k] : 0u, val = live ? vals[k] : 0u;
        const uint32_t digit = live ? (key >> shift) & (kDigits - 1) : 0xFFFFFFFFu;
        uint32_t place = 0;
#pragma unroll
        for (int d = 0; d < kDigits; ++d) {
            const unsigned long long same = __builtin_amdgcn_ballot_w64(digit == (uint32_t)d);
            if (digit == (uint32_t)d) place = next[d] + (uint32_t)__builtin_popcountll(same & below);
            next[d] += (uint32_t)__builtin_popcountll(same);
        }
        if (live) {
            keys_out[place] = key;
            vals_out[place] = val;
        }
    }
}

}  // namespace

int launch_transpose(const sdrb_batch* src, uint32_t* dst_off, int32_t* dst_ids, float* dst_data, hipStream_t stream) {
    const uint32_t width = (uint32_t)src->width, n_rows = (uint32_t)src->n_rows;
    const uint64_t nnz = src->nnz;
    SDRB_HIP(hipMemsetAsync(dst_off, 0, ((size_t)width + 1) * sizeof(uint32_t), stream));
    if (nnz == 0) return SDRB_OK;
    int bits = 1;
    while ((1ull << bits) < width) ++bits;
    const uint32_t n_tiles = (uint32_t)((nnz + kSortTile - 1) / kSortTile);
    const size_t arr = (nnz * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t counts_bytes = ((size_t)kDigits * n_tiles + 1) * sizeof(uint32_t);
    char* ws = static_cast<char*>(workspace(src->device, stream, 8, 4 * arr + counts_bytes));
    if (!ws) return SDRB_ERR_HIP;
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(ws + arr);
    uint32_t* elem_in = reinterpret_cast<uint32_t*>(ws + 2 * arr);
    uint32_t* elem_out = reinterpret_cast<uint32_t*>(ws + 3 * arr);
    uint32_t* counts = reinterpret_cast<uint32_t*>(ws + 4 * arr);
    hipLaunchKernelGGL(k_label_elements, dim3((n_rows + 3) / 4), dim3(256), 0, stream, src->d_offsets, src->d_ids,
                       n_rows, keys_in, elem_in, dst_off);
    SDRB_HIP(hipGetLastError());
    // column counts -> column offsets (the extra last entry is zero, so the scan's last value is nnz)
    int rc = launch_exclusive_scan_u32(src->device, dst_off, (uint64_t)width + 1, (uint64_t)width, nullptr, stream);
    if (rc != SDRB_OK) return rc;
    for (int shift = 0; shift < bits; shift += kDigitBits) {
        hipLaunchKernelGGL(k_digit_counts, dim3(n_tiles), dim3(64), 0, stream, keys_in, nnz, (uint32_t)shift, n_tiles, counts);
        SDRB_HIP(hipGetLastError());
        rc = launch_exclusive_scan_u32(src->device, counts, (uint64_t)kDigits * n_tiles + 1, (uint64_t)kDigits * n_tiles, nullptr, stream);
        if (rc != SDRB_OK) return rc;
        hipLaunchKernelGGL(k_digit_scatter, dim3(n_tiles), dim3(64), 0, stream, keys_in, elem_in, nnz, (uint32_t)shift, n_tiles, counts,
                           keys_out, elem_out);
        SDRB_HIP(hipGetLastError());
        std::swap(keys_in, keys_out);
        std::swap(elem_in, elem_out);
    }
    elem_out = elem_in;   // the last pass's output
    hipLaunchKernelGGL(k_gather_transposed, dim3((uint32_t)((nnz + 255) / 256)), dim3(256), 0, stream, src->d_offsets,
                       n_rows, src->d_data, elem_out, nnz, dst_ids, has_payload(src) ? dst_data : nullptr);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
