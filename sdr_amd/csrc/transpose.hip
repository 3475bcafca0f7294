// Transpose of a batch seen as a sparse matrix (rows = the batch's rows, columns = ids):
// the result has one row per id c of the declared width, listing the rows of the source that
// contain c in ascending order, each with the payload it had there.
//
// The reference's transpose (SDR.hpp:1807-1826) walks a row-major matrix in column order with a
// priority queue keyed by the column id (MatrixUtils.hpp:13-75) and appends (row id, payload) to
// the column's bucket (MatrixUtils.hpp:150-190).  On the device the same regrouping is a STABLE
// sort of the elements by id: the packed CSR already lists elements by ascending row, so within
// an id the sorted order is ascending row -- the order the reference's own known answers show
// (src/test.cpp:517-571).  The sort is rocPRIM's LSD radix sort over the id bits (stable by
// construction); the kernels here label elements with their rows, count the columns and gather.
#include <rocprim/device/device_radix_sort.hpp>

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

}  // namespace

int launch_transpose(const sdrb_batch* src, uint32_t* dst_off, int32_t* dst_ids, float* dst_data, hipStream_t stream) {
    const uint32_t width = (uint32_t)src->width, n_rows = (uint32_t)src->n_rows;
    const uint64_t nnz = src->nnz;
    SDRB_HIP(hipMemsetAsync(dst_off, 0, ((size_t)width + 1) * sizeof(uint32_t), stream));
    if (nnz == 0) return SDRB_OK;
    int bits = 1;
    while ((1ull << bits) < width) ++bits;
    size_t temp_bytes = 0;
    SDRB_HIP(rocprim::radix_sort_pairs(nullptr, temp_bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                       (uint32_t*)nullptr, nnz, 0, bits, stream));
    const size_t arr = (nnz * sizeof(uint32_t) + 255) / 256 * 256;
    char* ws = static_cast<char*>(workspace(src->device, stream, 8, 4 * arr + temp_bytes));
    if (!ws) return SDRB_ERR_HIP;
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(ws);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(ws + arr);
    uint32_t* elem_in = reinterpret_cast<uint32_t*>(ws + 2 * arr);
    uint32_t* elem_out = reinterpret_cast<uint32_t*>(ws + 3 * arr);
    void* temp = ws + 4 * arr;
    hipLaunchKernelGGL(k_label_elements, dim3((n_rows + 3) / 4), dim3(256), 0, stream, src->d_offsets, src->d_ids,
                       n_rows, keys_in, elem_in, dst_off);
    SDRB_HIP(hipGetLastError());
    // column counts -> column offsets (the extra last entry is zero, so the scan's last value is nnz)
    int rc = launch_exclusive_scan_u32(src->device, dst_off, (uint64_t)width + 1, (uint64_t)width, nullptr, stream);
    if (rc != SDRB_OK) return rc;
    SDRB_HIP(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, elem_in, elem_out, nnz, 0, bits, stream));
    hipLaunchKernelGGL(k_gather_transposed, dim3((uint32_t)((nnz + 255) / 256)), dim3(256), 0, stream, src->d_offsets,
                       n_rows, src->d_data, elem_out, nnz, dst_ids, has_payload(src) ? dst_data : nullptr);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
