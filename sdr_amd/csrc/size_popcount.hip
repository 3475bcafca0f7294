// Dense-bitset popcount kernel for the id-only size operations.
//
// For EmptyData the size ops are functions of |this|, |arg| and the number of common ids
// (SDR.hpp:1094, 1303, 1433, 1576).  When both sides are stored as W-bit sets the common-id
// count is popcount(a & b), the form that suits densely populated rows (its cost depends on
// W, not on how many ids a row holds), e.g. the struct-of-arrays rows of
// IDContiguousContainer.hpp:246-250 kept as bit words.
//
// Layout: B is converted once per upload into bit PLANES, plane w holding word w of every
// row (bt[w][j], j fastest), so that a lane owning row j reads its W/32 words with
// coalesced loads and keeps them in registers.  Rows of A are bit words in row-major order
// and reach the lanes as scalar operands.  Per pair: W/32 x (v_and + v_bcnt).  Output is
// written exactly like the bit-sliced kernel does (one 256-byte store per wave per row of A).
#include "sdrb_internal.h"

namespace sdrb {

namespace {

constexpr int kBlock = 256;

__global__ __launch_bounds__(256) void k_scatter_bitplanes(const uint32_t* __restrict__ b_off,
                                                            const int32_t* __restrict__ b_ids,
                                                            uint32_t nb, uint32_t* __restrict__ planes) {
    // one wave per row, one lane per element
    const uint32_t j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= nb) return;
    const uint32_t beg = b_off[j], len = b_off[j + 1] - beg;
    for (uint32_t k = threadIdx.x & 63; k < len; k += 64) {
        const uint32_t id = (uint32_t)b_ids[beg + k];
        atomicOr(&planes[(uint64_t)(id >> 5) * nb + j], 1u << (id & 31));
    }
}

__global__ __launch_bounds__(64) void k_build_row_bits(const uint32_t* __restrict__ a_off,
                                                       const int32_t* __restrict__ a_ids, uint32_t nw,
                                                       uint32_t width, uint32_t* __restrict__ bits,
                                                       uint32_t* __restrict__ a_len) {
    const uint64_t i = blockIdx.x;
    const uint32_t beg = a_off[i], len = a_off[i + 1] - beg;
    if (threadIdx.x == 0) a_len[i] = len;
    for (uint32_t k = threadIdx.x; k < len; k += 64) {
        const uint32_t id = (uint32_t)a_ids[beg + k];
        if (id < width) atomicOr(&bits[i * nw + (id >> 5)], 1u << (id & 31));   // others never match
    }
}

// MODE: 0 -> m;  1 -> L - m;  2 -> L - 2m   (L = use_la*|A_i| + use_lb*|B_j|)
template <int NW, int MODE>
__global__ __launch_bounds__(kBlock) void k_size_popcount(
    const uint32_t* __restrict__ planes, const uint32_t* __restrict__ b_off, uint32_t nb,
    const uint32_t* __restrict__ a_bits, const uint32_t* __restrict__ a_len, uint32_t na,
    uint32_t rows_a_per_block, uint32_t use_la, uint32_t use_lb, uint32_t* __restrict__ out,
    uint64_t ld_out) {
    const uint32_t j0 = blockIdx.x * kBlock;
    const uint32_t j = j0 + threadIdx.x;
    const uint32_t jc = j < nb ? j : nb - 1;   // clamped: out-of-range lanes compute and drop
    uint32_t r[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) r[w] = planes[(uint64_t)w * nb + jc];
    uint32_t lbterm = 0;
    if (MODE != 0 && use_lb) lbterm = b_off[jc + 1] - b_off[jc];

    const uint32_t i0 = blockIdx.y * rows_a_per_block;
    const uint32_t i1 = i0 + rows_a_per_block < na ? i0 + rows_a_per_block : na;
    const uint32_t rows_here = nb - j0 < (uint32_t)kBlock ? nb - j0 : (uint32_t)kBlock;
    char* row_ptr = reinterpret_cast<char*>(out) + ((uint64_t)i0 * ld_out + j0) * 4;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t* q = a_bits + (uint64_t)i * NW;   // uniform -> scalar loads
        uint32_t m = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) m += (uint32_t)__builtin_popcount(r[w] & q[w]);
        uint32_t v = m;
        if constexpr (MODE == 1) v = ((use_la ? a_len[i] : 0u) + lbterm) - m;
        if constexpr (MODE == 2) v = ((use_la ? a_len[i] : 0u) + lbterm) - 2 * m;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(row_ptr, 0, (int)(rows_here * 4), 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, (int)(threadIdx.x * 4), 0, 0);
        row_ptr += ld_out * 4;
    }
}

struct Args {
    const uint32_t* planes;
    const sdrb_batch* a;
    const sdrb_batch* b;
    const uint32_t* a_bits;
    const uint32_t* a_len;
    uint32_t per, use_la, use_lb;
    dim3 grid;
    uint32_t* out;
    uint64_t ld_out;
    hipStream_t stream;
};

template <int NW, int MODE>
void launch(const Args& g) {
    hipLaunchKernelGGL((k_size_popcount<NW, MODE>), g.grid, dim3(kBlock), 0, g.stream, g.planes,
                       g.b->d_offsets, (uint32_t)g.b->n_rows, g.a_bits, g.a_len, (uint32_t)g.a->n_rows,
                       g.per, g.use_la, g.use_lb, g.out, g.ld_out);
}

template <int MODE>
void launch_w(const Args& g, int nw) {
    if (nw == 16) launch<16, MODE>(g);
    else if (nw == 32) launch<32, MODE>(g);
    else launch<64, MODE>(g);
}

int words_for(int32_t width) { return width <= 512 ? 16 : width <= 1024 ? 32 : 64; }

}  // namespace

bool popcount_applicable(const sdrb_batch* a, const sdrb_batch* b) {
    (void)a;
    return b->width > 0 && b->width <= 2048 && b->n_rows > 0;
}

int ensure_bitplanes(const sdrb_batch* b, hipStream_t stream) {
    sdrb_batch* mb = derived_owner(b);   // derived data cached in the handle
    std::lock_guard<std::mutex> lock(derived_mutex());
    if (mb->d_bitplanes) return derived_wait(mb->bitplanes_ready, mb->bitplanes_stream, stream);
    const int nw = words_for(b->width);
    const size_t bytes = (size_t)nw * b->n_rows * sizeof(uint32_t);
    uint32_t* planes = nullptr;
    SDRB_HIP(hipMalloc(reinterpret_cast<void**>(&planes), bytes));
    int rc = SDRB_OK;
    if (hipMemsetAsync(planes, 0, bytes, stream) != hipSuccess) rc = SDRB_ERR_HIP;
    if (rc == SDRB_OK) {
        hipLaunchKernelGGL(k_scatter_bitplanes, dim3((uint32_t)((b->n_rows + 3) / 4)), dim3(256), 0, stream,
                           b->d_offsets, b->d_ids, (uint32_t)b->n_rows, planes);
        if (hipGetLastError() != hipSuccess) rc = SDRB_ERR_HIP;
    }
    if (rc == SDRB_OK) rc = derived_built(&mb->bitplanes_ready, &mb->bitplanes_stream, stream);
    if (rc != SDRB_OK) {
        (void)hipStreamSynchronize(stream);
        (void)hipFree(planes);
        set_error("building the bit planes of a batch failed");
        return rc;
    }
    mb->d_bitplanes = planes;   // published last: whoever sees the pointer also sees the event of its build
    return SDRB_OK;
}

int launch_size_popcount(const sdrb_batch* a, const sdrb_batch* b, int op, bool swap, uint32_t* out,
                         uint64_t ld_out, hipStream_t stream) {
    if (!popcount_applicable(a, b)) {
        set_error("popcount kernel needs a declared width <= 2048 on the streamed batch");
        return SDRB_ERR_UNSUPPORTED;
    }
    if (a->n_rows == 0) return SDRB_OK;
    int rc = ensure_bitplanes(b, stream);
    if (rc != SDRB_OK) return rc;
    const int nw = words_for(b->width);
    const size_t bits_bytes = (size_t)a->n_rows * nw * sizeof(uint32_t);
    const size_t len_bytes = (size_t)a->n_rows * sizeof(uint32_t);
    char* ws = static_cast<char*>(workspace(b->device, stream, 0, bits_bytes + len_bytes));
    if (!ws) return SDRB_ERR_HIP;
    SDRB_HIP(hipMemsetAsync(ws, 0, bits_bytes + len_bytes, stream));
    uint32_t* a_bits = reinterpret_cast<uint32_t*>(ws);
    uint32_t* a_len = reinterpret_cast<uint32_t*>(ws + bits_bytes);
    hipLaunchKernelGGL(k_build_row_bits, dim3((uint32_t)a->n_rows), dim3(64), 0, stream, a->d_offsets,
                       a->d_ids, (uint32_t)nw, (uint32_t)b->width, a_bits, a_len);
    SDRB_HIP(hipGetLastError());

    Args g;
    g.planes = derived_owner(b)->d_bitplanes;
    g.a = a;
    g.b = b;
    g.a_bits = a_bits;
    g.a_len = a_len;
    const uint32_t gx = (uint32_t)((b->n_rows + kBlock - 1) / kBlock);
    uint32_t gy = gx >= 1024 ? 1u : (1024u + gx - 1) / gx;
    if (gy > a->n_rows) gy = (uint32_t)a->n_rows;
    g.per = (uint32_t)((a->n_rows + gy - 1) / gy);
    gy = (uint32_t)((a->n_rows + g.per - 1) / g.per);
    g.grid = dim3(gx, gy);
    g.out = out;
    g.ld_out = ld_out;
    g.stream = stream;
    switch (op) {
    case SDRB_OP_AND: g.use_la = g.use_lb = 0; launch_w<0>(g, nw); break;
    case SDRB_OP_OR: g.use_la = g.use_lb = 1; launch_w<1>(g, nw); break;
    case SDRB_OP_XOR: g.use_la = g.use_lb = 1; launch_w<2>(g, nw); break;
    case SDRB_OP_RM: g.use_la = swap ? 0 : 1; g.use_lb = swap ? 1 : 0; launch_w<1>(g, nw); break;
    default: set_error("unknown op %d", op); return SDRB_ERR_INVALID_ARG;
    }
    SDRB_HIP(hipGetLastError());
    return SDRB_OK;
}

}  // namespace sdrb
