// Range and point queries for every row of a batch.
//
//   ands(start, stop)   SDR.hpp:931-956   number of ids in [start, stop)
//   ande(start, stop)   SDR.hpp:881-929   the elements with ids in [start, stop)
//   ande(id)            SDR.hpp:860-879   pointer to the payload of `id`, or NULL
//
// The reference finds both ends with std::lower_bound on the sorted row (SDR.hpp:890,900,945-946)
// and so does a lane here, for its row and every query of the call: out(r, j) indexed [r][j] with j
// fastest, so lanes write coalesced rows.  A point query returns the element's position in its row
// (-1 when absent), which is where its payload lives: data[row_offsets[j] + position].
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;

// first position in ids[0, n) whose id is >= key
__device__ __forceinline__ uint32_t lower_bound(const int32_t* __restrict__ ids, uint32_t n, int32_t key) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ids[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// MODE 0: counts;  MODE 1: fill the slices at out_off;  MODE 2: point queries (q_lo holds the ids)
template <int MODE>
__global__ __launch_bounds__(kBlock) void k_range_query(
    const uint32_t* __restrict__ b_off, const int32_t* __restrict__ b_ids, const float* __restrict__ b_data,
    uint64_t nb, uint32_t n_queries, const int32_t* __restrict__ q_lo, const int32_t* __restrict__ q_hi,
    uint32_t* __restrict__ out, uint64_t ld_out, const uint32_t* __restrict__ out_off,
    int32_t* __restrict__ out_ids, float* __restrict__ out_data) {
    const uint64_t j = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= nb) return;
    const uint32_t beg = b_off[j], len = b_off[j + 1] - beg;
    const int32_t* ids = b_ids + beg;
    for (uint32_t r = blockIdx.y; r < n_queries; r += gridDim.y) {
        const int32_t lo = q_lo[r];
        const uint32_t first = lower_bound(ids, len, lo);
        if constexpr (MODE == 2) {
            const bool found = first < len && ids[first] == lo;
            reinterpret_cast<int32_t*>(out)[(uint64_t)r * ld_out + j] = found ? (int32_t)first : -1;
        } else {
            // the second search starts where the first one ended (SDR.hpp:900,946)
            const uint32_t last = first + lower_bound(ids + first, len - first, q_hi[r]);
            if constexpr (MODE == 0) {
                out[(uint64_t)r * ld_out + j] = last - first;
            } else {
                uint32_t w = out_off[(uint64_t)r * nb + j];
                for (uint32_t k = first; k < last; ++k, ++w) {
                    out_ids[w] = ids[k];
                    if (out_data) out_data[w] = b_data[beg + k];
                }
            }
        }
    }
}

// host query arrays -> device scratch (slot 7), ordered on the stream
int stage_queries(int device, uint64_t n, const int32_t* lo, const int32_t* hi, const int32_t** d_lo,
                  const int32_t** d_hi, hipStream_t stream) {
    int32_t* ws = static_cast<int32_t*>(workspace(device, stream, 7, 2 * n * sizeof(int32_t)));
    if (!ws) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemcpyAsync(ws, lo, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    if (hi) SDRB_HIP(hipMemcpyAsync(ws + n, hi, n * sizeof(int32_t), hipMemcpyHostToDevice, stream));
    // the host arrays may be pageable: make sure they have been read before the caller reuses them
    SDRB_HIP(hipStreamSynchronize(stream));
    *d_lo = ws;
    *d_hi = hi ? ws + n : nullptr;
    return SDRB_OK;
}

dim3 grid_for(const sdrb_batch* b, uint64_t n_queries) {
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 2048 ? 1u : (2048u + gx - 1) / gx;
    if (gy > n_queries) gy = (uint32_t)n_queries;
    if (gy > 65535u) gy = 65535u;
    return dim3(gx, gy ? gy : 1u);
}

}  // namespace

int launch_range_query(const sdrb_batch* b, int mode, uint64_t n_queries, const int32_t* lo, const int32_t* hi,
                       uint32_t* out, uint64_t ld_out, const uint32_t* out_off, int32_t* out_ids, float* out_data,
                       hipStream_t stream) {
    if (n_queries == 0 || b->n_rows == 0) return SDRB_OK;
    const int32_t *d_lo = nullptr, *d_hi = nullptr;
    const int rc = stage_queries(b->device, n_queries, lo, hi, &d_lo, &d_hi, stream);
    if (rc != SDRB_OK) return rc;
    const dim3 grid = grid_for(b, n_queries);
    if (mode == 0)
        hipLaunchKernelGGL(k_range_query<0>, grid, dim3(kBlock), 0, stream, b->d_offsets, b->d_ids, b->d_data, b->n_rows,
                           (uint32_t)n_queries, d_lo, d_hi, out, ld_out, out_off, out_ids, out_data);
    else if (mode == 1)
        hipLaunchKernelGGL(k_range_query<1>, grid, dim3(kBlock), 0, stream, b->d_offsets, b->d_ids, b->d_data, b->n_rows,
                           (uint32_t)n_queries, d_lo, d_hi, out, ld_out, out_off, out_ids, out_data);
    else
        hipLaunchKernelGGL(k_range_query<2>, grid, dim3(kBlock), 0, stream, b->d_offsets, b->d_ids, b->d_data, b->n_rows,
                           (uint32_t)n_queries, d_lo, d_hi, out, ld_out, out_off, out_ids, out_data);
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
