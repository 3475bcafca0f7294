/*
 * sdr_amd.h -- C ABI of the MI355X (gfx950) batched SDR set-algebra engine.
 *
 * An SDR (sparse distributed representation) is an ascending, duplicate-free sequence of
 * (int id, payload) elements.  The reference library (jagprog5/SDR, header-only C++17)
 * evaluates one pair of SDRs per call on one CPU thread; it has no FFI of its own, so each
 * entry point below cites the reference member function(s) whose per-pair result it
 * reproduces, bit for bit, for every pair of a batch.  Paths are relative to the
 * reference's include/SparseDistributedRepresentation/.
 *
 * Conventions
 *  - A "batch" is a device-resident CSR pack of many SDRs of one element type:
 *    uint32 row_offsets[n_rows+1], int32 ids[nnz], optional float data[nnz].
 *  - Binary operations take two batches A and B and evaluate every pair:
 *        result(i, j) = A_i.op(B_j)            (A_i is `this`, B_j is `arg`)
 *    or, with SDRB_FLAG_SWAP, B_j.op(A_i).  Results are indexed [i][j], j fastest.
 *    B is the long, streamed side (the database); A is the short side (the queries).
 *  - All functions return SDRB_OK (0) or a negative error code and never throw;
 *    sdrb_last_error() gives a thread-local message for the last failure.
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is
 *    enqueued asynchronously unless a function says it synchronises.
 *  - Device pointers passed in (out buffers, offsets) must live on the batch's device.
 *  - Handles are thread-compatible: one thread at a time may upload into a handle; any number of threads may
 *    read it (pass it to the operations below) at once.
 *  - Scratch memory is kept per (device, stream): calls on different streams, from one thread or several, run
 *    concurrently and never share a table (sdrb_size_op_static is the exception, see there).  Forms of a batch
 *    that the library derives and caches in the handle (bit planes, byte matrix) are built on the stream of the
 *    first call that needs them; calls on other streams wait for that build through an event.
 */
#ifndef SDR_AMD_H
#define SDR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRB_OK 0
#define SDRB_ERR_INVALID_ARG (-1)
#define SDRB_ERR_NOT_ASCENDING (-2) /* a row's ids are not strictly ascending */
#define SDRB_ERR_ID_RANGE (-3)      /* an id lies outside [0, width) of a batch with a declared width */
#define SDRB_ERR_HIP (-4)           /* a HIP runtime call failed; see sdrb_last_error() */
#define SDRB_ERR_UNSUPPORTED (-5)
#define SDRB_ERR_OVERFLOW (-6)      /* more than 2^32-1 elements in one batch or one result */
#define SDRB_ERR_PAYLOAD_RANGE (-7) /* a UnitData payload outside [0,1] (UnitData.hpp:15-17) */
#define SDRB_ERR_IO (-8)            /* a file could not be opened, read or written */
#define SDRB_ERR_FORMAT (-9)        /* not a CSR database file, or its tables are inconsistent */
#define SDRB_ERR_NO_MEMORY (-10)    /* a host allocation failed */

/* set operations */
#define SDRB_OP_AND 0
#define SDRB_OP_OR 1
#define SDRB_OP_XOR 2
#define SDRB_OP_RM 3

/* payload kinds of a batch's elements */
#define SDRB_DATA_EMPTY 0 /* DataTypes/EmptyData.hpp:18-92: zero-size payload */
#define SDRB_DATA_UNIT 1  /* DataTypes/UnitData.hpp:11-140: float in [0,1] */
#define SDRB_DATA_ARITH 2 /* DataTypes/ArithData.hpp:11-107 with arith_t = float; payloads must be finite here
                             (the reference allows any float).  Besides sdrb_inner such batches take the set
                             algebra with ArithData or EmptyData partners: and = product, or = sum, rm =
                             difference, every payload relevant (ArithData.hpp:22-83); there is no xor
                             (ArithData.hpp:110-114) and no mixing with UnitData */

/* flags of the binary operations */
#define SDRB_FLAG_SWAP 1u           /* result(i,j) = B_j.op(A_i) instead of A_i.op(B_j) */
#define SDRB_FLAG_FORCE_MERGE 2u    /* always use the general per-pair merge kernels */
#define SDRB_FLAG_FORCE_BITSLICE 4u /* size ops: fail with UNSUPPORTED unless the bit-sliced kernel applies */
#define SDRB_FLAG_FORCE_POPCOUNT 8u /* size ops: fail with UNSUPPORTED unless the dense-bitset popcount kernel applies */
#define SDRB_FLAG_ROWWISE 32u       /* size/element ops: result j = A_j.op(B_j) for batches of equal length instead of
                                       every pair; outputs have one entry per row (out[j], offsets[n+1]).  This is the
                                       batched form of the in-place ops andi/ori/xori/rmi (SDR.hpp:1027-1081,1239-1292,
                                       1366-1513), whose results equal those of the element ops */
#define SDRB_FLAG_FORCE_ROWTABLE 64u /* size ops: use the one-row-of-A position-table kernel where the 64-row mask-table
                                       kernel would be chosen (tests and experiments) */
#define SDRB_FLAG_FORCE_RANK 128u    /* element fill: fail with UNSUPPORTED unless the wave-per-pair rank-table kernel applies */
#define SDRB_FLAG_FORCE_BITMAP 256u   /* element fill: fail with UNSUPPORTED unless the dense-bitset fill applies */
#define SDRB_FLAG_FORCE_LANEMERGE 512u /* element fill: fail with UNSUPPORTED unless the lane-per-pair merge fill (unions of short rows, ids only) applies */
#define SDRB_FLAG_FORCE_LANERANK 1024u /* element fill: fail with UNSUPPORTED unless the lane-per-pair rank fill (unions of short rows, ids only, width <= 4096) applies */
#define SDRB_FLAG_FORCE_SPARSE 2048u /* sdrb_inner, size ops: fail with UNSUPPORTED unless the lane-per-row match kernel applies
                                        (sdrb_inner also takes FORCE_ROWTABLE: the dense value-table kernel, and FORCE_MERGE:
                                        the per-pair merge) */
#define SDRB_FLAG_FORCE_MASKTABLE 4096u /* size ops: fail with UNSUPPORTED unless the 64-row mask-table kernel applies */
#define SDRB_FLAG_FORCE_BITSLICE_LONG 8192u /* size ops: fail with UNSUPPORTED unless the bit-sliced kernel for long rows / wide tables applies */
#define SDRB_FLAG_FORCE_SEARCH 16384u /* element fill, visitors: fail with UNSUPPORTED unless the wave-per-pair binary-search kernel applies (any ids, as they are) */
#define SDRB_FLAG_FORCE_MFMA 16u    /* all-pairs top-k: fail with UNSUPPORTED unless the matrix-core kernel applies */

typedef struct sdrb_batch sdrb_batch;
typedef struct sdrb_file_writer sdrb_file_writer;

/* Message of the calling thread's most recent failure ("" if none). */
const char* sdrb_last_error(void);

/* Number of visible HIP devices. */
int sdrb_device_count(int* count);

/*
 * Creates an empty batch on `device`.
 * width > 0 declares that every id lies in [0, width) (checked at upload); width == 0
 * means arbitrary signed ids (SDRElem.hpp:15,73-75).  The table-driven kernels index
 * tables by id.  A batch without a declared width whose uploaded ids are all non-negative
 * is served by them as if largest id + 1 had been declared (sdrb_batch_info still reports
 * 0, and the transpose still asks for a declared width); with a negative id, or ids beyond
 * every table, size operations go through small per-call labels and the table kernels,
 * element operations through the per-pair merge kernels.
 * Replaces: the container side of SDR<SDRElem<int,Data>, Container> (SDR.hpp:26-27,573);
 * the struct-of-arrays layout mirrors IDContiguousContainer.hpp:246-250.
 */
int sdrb_batch_create(int device, int32_t width, int data_kind, sdrb_batch** out);

/*
 * Packs n_rows SDRs given as host CSR (row_offsets has n_rows+1 entries, ids/data have
 * row_offsets[n_rows] entries; data is NULL for SDRB_DATA_EMPTY) into HBM, replacing any
 * previous content.  Rows must be strictly ascending (SDR.hpp:679,1621 assert the same),
 * ids must respect the declared width, UnitData payloads must lie in [0,1].  Payloads
 * below the relevance threshold are kept as they are, exactly like the reference's
 * iterator constructor and push_back (SDR.hpp:83,612-637,1614-1628).
 * Synchronises: the host buffers may be reused on return.
 * Large batches are checked and staged by a team of host threads (at most 16, or SDRB_HOST_THREADS) through
 * page-locked buffers while earlier pieces are already crossing the bus; such a batch, when it is refused, leaves
 * the handle without rows (small batches are checked before anything is copied and leave the handle as it was).
 * ids / data that already lie in page-locked host memory are copied as they are, without staging.
 */
int sdrb_batch_upload_csr(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets,
                          const int32_t* ids, const float* data, void* stream);

/*
 * The checks of sdrb_batch_upload_csr on their own, without a device: offsets monotone, rows strictly ascending
 * (SDR.hpp:679,1621), ids in [0, width) when width > 0, UnitData payloads in [0,1] (UnitData.hpp:15-17), ArithData
 * payloads finite.  Same status codes and messages as the upload; large arrays are checked by a team of host threads.
 */
int sdrb_csr_check(uint64_t n_rows, const uint64_t* row_offsets, const int32_t* ids, const float* data, int32_t width,
                   int data_kind);

/*
 * Same as sdrb_batch_upload_csr for a CSR pack that already lives in this device's memory
 * (uint32 row_offsets[n_rows+1]), e.g. a query batch received over RCCL.  The same checks
 * run in a device kernel (one stream synchronisation to read its verdict); the arrays are then
 * copied on `stream`, so the caller keeps ownership but must leave them unchanged until the
 * stream has passed this call.  Later calls on the same stream see the new content.
 */
int sdrb_batch_upload_device_csr(sdrb_batch* batch, uint64_t n_rows, const uint32_t* row_offsets,
                                 const int32_t* ids, const float* data, void* stream);

/*
 * Batches with 64-bit ids (the reference's SDRElem<long, ...>, fuzzed in src/fuzz.cpp:413-423).
 * Such a batch has no declared width and EmptyData or UnitData payloads; it is taken by sdrb_size_op,
 * sdrb_equal, sdrb_elem_op_count / _bound, sdrb_elem_op_fill64 and the visitors (the per-pair merge kernels serve
 * them).  When every id of a batch fits 32 bits the handle also keeps the ids narrowed; two such batches are served
 * by the kernels of the 32-bit path (ids of results widened back), and sdrb_allpairs_topk takes them as well.
 * The two batches of a call may differ in id width, as the fuzzed pairs do (`long` ids against `int` ids): the
 * 64-bit batch is taken through its narrowed ids when it keeps them, else the 32-bit batch through a widened copy
 * of its ids made for the call.  A result holds ids of `this`'s width (SDR.hpp:151-152): sdrb_elem_op_fill64 when
 * `this` is the 64-bit batch, sdrb_elem_op_fill otherwise (refused for ore / xore against 64-bit ids that do not
 * fit 32 bits, which such a result could not hold).  Ids compare by value; the reference, handed a `long` id beyond
 * 32 bits in a search over `int` elements, truncates it (SDRElem.hpp:73-75 takes the element's own id type), so
 * that case has no reference result to match.  Everything else refuses 64-bit batches.
 *
 * Ids no table indexes as they are -- some negative, 64-bit ids far from zero -- but within 32 768 of each other over
 * BOTH batches of a call are moved for the call: the kernels see copies of the ids less the smallest one (a pass over
 * each batch, scratch of the stream), with every table kernel open to them, and an element fill adds it back.  Results
 * depend on which ids are equal and on their order, never on their values (SDRElem.hpp:56-75).  Ids further apart are
 * given small labels for the size ops (batches without a usable width above); their elements are filled by a wave per
 * pair with binary searches over the rows, whatever the ids are.
 */
int sdrb_batch_create64(int device, int data_kind, sdrb_batch** out);
int sdrb_batch_upload_csr64(sdrb_batch* batch, uint64_t n_rows, const uint64_t* row_offsets,
                            const int64_t* ids, const float* data, void* stream);

/*
 * Makes the batch a VIEW of a CSR pack in this device's memory: nothing is copied, checked or
 * synchronised.  The caller vouches that the arrays hold a batch that has already been
 * validated (e.g. the arrays of another rank's uploaded batch, received over RCCL together with
 * that batch's statistics) and keeps them alive and unchanged while the batch is in use.
 * max_row_len, min_id and max_id are the values sdrb_batch_info / sdrb_batch_id_span report
 * for the validated original.
 */
int sdrb_batch_adopt_device_csr(sdrb_batch* batch, uint64_t n_rows, uint64_t nnz, const uint32_t* row_offsets,
                                const int32_t* ids, const float* data, uint32_t max_row_len,
                                int32_t min_id, int32_t max_id);

/* Shape of a batch; any out pointer may be NULL. */
int sdrb_batch_info(const sdrb_batch* batch, int* device, uint64_t* n_rows, uint64_t* nnz,
                    int32_t* width, int* data_kind, uint32_t* max_row_len);

/* Smallest and largest id over all rows (min_id > max_id for a batch without elements). */
int sdrb_batch_id_span(const sdrb_batch* batch, int32_t* min_id, int32_t* max_id);

/*
 * Adds `amount` to every id of the batch, in place, on `stream`: SDR::shift (SDR.hpp:1579-1590) and the operators
 * << >> <<= >>= built on it (SDR.hpp:461-468), for all rows at once.  The reference asserts that no id overflows
 * (SDR.hpp:1586); here such a shift is refused with SDRB_ERR_ID_RANGE and nothing is changed, and so is one that
 * takes an id out of a declared width.  Batches of 64-bit ids are shifted within 64 bits (ids that leave 32 bits lose
 * their narrowed copy; ids that come to fit 32 bits get none before the next upload); rows adopted from caller memory
 * are not modified (SDRB_ERR_UNSUPPORTED).
 */
int sdrb_batch_shift(sdrb_batch* batch, int32_t amount, void* stream);

/*
 * The width the table kernels index this batch by: the declared width; for a batch created without one, largest id + 1
 * once rows without a negative id have been uploaded; 0 when there is none (negative ids, 64-bit ids, adopted rows).
 * A caller that sizes buffers by the expected number of matches (sdrb_ande_few) divides by this.
 * No counterpart in the reference, whose ids need no width (SDRElem.hpp:15).
 */
int sdrb_batch_table_width(const sdrb_batch* batch, int32_t* width);

/*
 * Copies the packed CSR into caller buffers on the same device (device-to-device, ordered on
 * `stream`): row_offsets[n_rows+1], ids[nnz], data[nnz] (UnitData batches; may be NULL otherwise).
 * This is the sending side of an exchange whose receiving side is sdrb_batch_adopt_device_csr.
 */
int sdrb_batch_export_device_csr(const sdrb_batch* batch, uint32_t* row_offsets, int32_t* ids, float* data,
                                 void* stream);

/* Device pointers of the packed CSR (read-only views, valid until the next upload/destroy). */
int sdrb_batch_device_ptrs(const sdrb_batch* batch, const uint32_t** row_offsets,
                           const int32_t** ids, const float** data);

int sdrb_batch_destroy(sdrb_batch* batch);

/*
 * Size operations for every pair: out[i*ld_out + j] (uint32, device memory, ld_out >= B rows)
 *   SDRB_OP_AND -> ands  (SDR.hpp:1083-1095)     SDRB_OP_OR -> ors  (SDR.hpp:1294-1304)
 *   SDRB_OP_XOR -> xors  (SDR.hpp:1424-1434)     SDRB_OP_RM -> rms  (SDR.hpp:1568-1577)
 * including the payload predicates of EmptyData.hpp:50-51,70-71,79-80 and
 * UnitData.hpp:48-51,94-97,117-120 (single IEEE binary32 operations, threshold >= 0.1).
 * The reference returns size_t; counts here are uint32 (a row has < 2^32 elements).
 * Kernel choice: when the result depends on ids only (`this` side is EmptyData, or the
 * op is ors) and B has a declared width that fits the LDS table, a bit-sliced
 * table kernel is used; otherwise a per-pair merge kernel (the visitor merge
 * andv, SDR.hpp:958-997, restated per lane).
 */
int sdrb_size_op(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                 uint32_t* out, uint64_t ld_out, void* stream);

/*
 * The same size operations with NARROW counts: out[i*ld_out + j] as uint8 (out_bytes 1) or uint16 (out_bytes 2).
 * The reference returns size_t (SDR.hpp:31,1085); every value here is bounded by the longest rows -- min(|A_i|, |B_j|)
 * for ands, |this| for rms, |A_i| + |B_j| for ors and xors -- and at 4096 x 1 Mi pairs the count matrix IS the
 * memory traffic, so a consumer that thresholds or ranks counts of at most 40 reads and writes a quarter of the bytes.
 * Served by the bit-sliced kernel only: sizes that depend on ids alone (`this` EmptyData, or SDRB_OP_OR), 32-bit ids,
 * b with a declared width <= 4096 and rows of at most 64 ids; SDRB_ERR_UNSUPPORTED otherwise, and also when the
 * bound above does not fit out_bytes (nothing is ever truncated).  out must be 4-byte aligned, ld_out a multiple of
 * 4 and >= the rows of b; the padding columns of a row up to the next multiple of 4 may be written.
 * Flags: SDRB_FLAG_SWAP.
 */
int sdrb_size_op_narrow(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, void* out, uint64_t ld_out,
                        int out_bytes, void* stream);

/*
 * sdrb_size_op for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same results, same output
 * layout and the same kernels as sdrb_size_op's bit-sliced path, but the bit table of A lives in a static
 * 2 MiB + 32 KiB array of device memory inside the library instead of a workspace that grows on demand: the
 * call makes no hipMalloc / hipFree.  Each GPU the library runs on holds its own copy of the array until the
 * process ends; its address is looked up once per device (when a batch is first uploaded there, else on the
 * first call).
 * Limits: sizes that depend on ids alone (`this` EmptyData, or SDRB_OP_OR), 32-bit ids, b with a declared
 * width <= 4096 and rows of at most 64 ids, and rows of a: at most 8192 for widths <= 2048, 4096 for widths
 * <= 4096 (SDRB_ERR_UNSUPPORTED beyond that, before any work is enqueued).  Flags: SDRB_FLAG_SWAP
 * (SDRB_FLAG_FORCE_BITSLICE is accepted: this entry point has no other kernel).
 * Every call on a GPU shares that GPU's array: calls that overlap in time on one GPU (two streams, two
 * threads) race; issue them on one stream or order them with events.
 */
int sdrb_size_op_static(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                        uint32_t* out, uint64_t ld_out, void* stream);

/*
 * Equality for every pair: out[i*ld_out + j] = (A_i == B_j) ? 1 : 0  -- SDR::operator== (SDR.hpp:470-483): the same
 * number of elements, the same ids position by position and, where BOTH batches carry payloads, payloads that
 * compare equal as floats (UnitData.hpp:122-124, ArithData.hpp:90-92; a payload always equals EmptyData,
 * UnitData.hpp:130-132).  SDRB_FLAG_ROWWISE: out[j] = (A_j == B_j).  Ids of 32 or 64 bits; UnitData and ArithData
 * batches do not combine, and an EmptyData A takes an EmptyData B only (the reference compiles no other: EmptyData
 * has no != for a payload, EmptyData.hpp:82-83).
 */
int sdrb_equal(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, uint32_t* out, uint64_t ld_out, void* stream);

/*
 * Sparse dot products for every pair: out[i*ld_out + j] (float, device memory, ld_out >= B rows)
 *   = A_i.inner(B_j)  (SDR.hpp:1682-1692): over the id matches in ascending id order,
 *     acc += a_value * b_value, a rounded product followed by a rounded sum (ArithData.hpp:37-43,
 *     59-63), starting from 0; bit-exact against the reference.
 * `a` must be an SDRB_DATA_ARITH batch; `b` is SDRB_DATA_ARITH, or SDRB_DATA_EMPTY (its matches add
 * a_value unchanged, ArithData.hpp:39).  The same numbers are entry (i, j) of diff_mul when `a`
 * holds the rows of a row-major matrix and `b` the columns of a column-major one (SDR.hpp:1886-1923),
 * and entry j of row_major_mul_vec(vector) when `a` holds the single vector and `b` the matrix rows
 * (SDR.hpp:1729-1750; ArithData is always relevant, so no entry is dropped).
 */
int sdrb_inner(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, float* out, uint64_t ld_out,
               void* stream);

/*
 * Transpose of a batch seen as a sparse matrix (rows x ids): `dst` gets one row per id c in
 * [0, width of src), listing the rows of `src` that contain c in ascending order, each with the
 * payload it had there; dst's declared width becomes the row count of src.  This is SDR::transpose
 * (SDR.hpp:1807-1826, MatrixUtils.hpp:13-190) for a matrix stored as the rows of a batch -- CSR to
 * CSC -- e.g. to build the inverted index of a database.  `src` needs a declared width, `dst` the
 * same payload kind and device; `dst` may not be `src`.
 */
int sdrb_batch_transpose(const sdrb_batch* src, sdrb_batch* dst, void* stream);

/*
 * Range and point queries for every row of a batch; `lo`, `hi`, `ids` are HOST arrays, outputs are
 * device memory indexed [query][row] (ld_out >= rows of b).
 *   sdrb_range_size:        out[r*ld_out + j] = B_j.ands(lo[r], hi[r])  (SDR.hpp:931-956): ids in [lo, hi)
 *   sdrb_range_elems_count/_fill: B_j.ande(lo[r], hi[r])  (SDR.hpp:881-929) as CSR over the pairs (r, j),
 *                            the same two-call protocol as the element ops; out_data may be NULL
 *   sdrb_point_query:       out[q*ld_out + j] = position of ids[q] in row j, or -1 -- the batched
 *                            ande(id) (SDR.hpp:860-879): present or not, and where its payload is
 *                            (data[row_offsets[j] + position])
 * A range with lo > hi violates the reference's precondition (SDR.hpp:884) and is refused.
 */
int sdrb_range_size(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                    uint32_t* out, uint64_t ld_out, void* stream);
int sdrb_range_elems_count(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                           uint32_t* out_offsets, uint64_t* total, void* stream);
int sdrb_range_elems_fill(const sdrb_batch* b, uint64_t n_ranges, const int32_t* lo, const int32_t* hi,
                          const uint32_t* out_offsets, int32_t* out_ids, float* out_data, void* stream);
int sdrb_point_query(const sdrb_batch* b, uint64_t n_ids, const int32_t* ids, int32_t* out, uint64_t ld_out,
                     void* stream);

/*
 * All-pairs overlap with a fused per-row top-k.  For every row i of `a`, the k rows j of `b`
 * with the largest id overlap |a_i and b_j| -- the value `ands` returns for EmptyData elements
 * (SDR.hpp:1083-1095; payloads of UnitData batches are ignored here) -- ordered by overlap
 * descending, then j ascending.  The reference has no top-k: each count is pinned by `ands`,
 * the selection and its tie rule are this engine's own.  The na x nb count matrix is never
 * materialised.  When exclude_offset >= 0 the pair (i, j = i + exclude_offset) is skipped (the
 * self match when `a` is rows [exclude_offset, exclude_offset + na) of `b`).
 * out_counts / out_rows: device memory, [a rows][k]; unused slots hold count 0, row -1.
 * Kernels (k <= 16, declared widths): a bit-sliced table kernel (widths <= 4096, rows of `a`
 * <= 64 ids; the default when it applies, SDRB_FLAG_FORCE_BITSLICE insists on it) and a dense
 * int8 0/1 product on the matrix cores, v_mfma_i32_32x32x32_i8 (widths <= 16384, rows shorter
 * than 4096 ids on one side; SDRB_FLAG_FORCE_MFMA selects it).  Both give identical results.
 */
int sdrb_allpairs_topk(const sdrb_batch* a, const sdrb_batch* b, uint32_t k, int64_t exclude_offset,
                       unsigned flags, uint32_t* out_counts, int32_t* out_rows, void* stream);

/*
 * Element operations for every pair, as a two-call CSR protocol:
 *   SDRB_OP_AND -> ande (SDR.hpp:999-1025)       SDRB_OP_OR -> ore  (SDR.hpp:1183-1222)
 *   SDRB_OP_XOR -> xore (SDR.hpp:1307-1349)      SDRB_OP_RM -> rme  (SDR.hpp:1515-1566)
 * The in-place forms andi/ori/xori/rmi (SDR.hpp:1027-1081,1239-1292,1366-1513) produce the
 * same elements; a caller replaces row i of its own storage with result (i, j).
 * Result elements have the element type of the `this` side (the reference's default
 * ret_t): ids always, payloads only when `this` is UnitData.
 *
 * 1) sdrb_elem_op_count fills out_offsets[na*nb + 1] (uint32, device) with the exclusive
 *    prefix sum of the result lengths in [i][j] order and returns the total in
 *    *total (host).  Synchronises the stream.  A call may hold fewer than 2^32-1 pairs and at most
 *    2^32-1 result elements (SDRB_ERR_OVERFLOW otherwise: split A); where the worst case of the
 *    shapes exceeds that, the lengths are summed exactly before the limit is applied.
 * 2) the caller allocates out_ids[total] (and out_data[total] when `this` is UnitData);
 * 3) sdrb_elem_op_fill writes result (i, j) at out_offsets[i*nb + j], ascending by id.
 */
int sdrb_elem_op_count(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                       uint32_t* out_offsets, uint64_t* total, void* stream);
int sdrb_elem_op_fill64(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                        const uint32_t* out_offsets, int64_t* out_ids, float* out_data, void* stream);
int sdrb_elem_op_fill(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags,
                      const uint32_t* out_offsets, int32_t* out_ids, float* out_data,
                      void* stream);

/*
 * The same element operations in ONE call that never waits for the GPU: count, scan and fill are enqueued back to
 * back on `stream`.  The caller sizes out_ids (and out_data) from the shapes alone: `capacity` elements, at least
 * sdrb_elem_op_bound() -- |A_i| + |B_j| per pair for ore / xore, |this| for rme, min(|A_i|, |B_j|) for ande --
 * which for unions of sparse rows is within a few percent of the exact total.  out_offsets[na*nb + 1] receives
 * the exclusive offsets as above; the exact number of elements is out_offsets[pairs] and, as 64 bits, *total_dev
 * (device memory, may be NULL), readable once the stream has passed the call.  A call whose bound exceeds 2^32-1
 * elements is refused (SDRB_ERR_OVERFLOW): use the two-call protocol, which applies the limit to the exact total.
 * 32-bit ids.
 */
int sdrb_elem_op_bound(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint64_t* bound);
int sdrb_elem_op(const sdrb_batch* a, const sdrb_batch* b, int op, unsigned flags, uint32_t* out_offsets,
                 int32_t* out_ids, float* out_data, uint64_t capacity, uint64_t* total_dev, void* stream);

/*
 * ande (SDR.hpp:999-1025) of a FEW rows of `a` -- 1 to 32: one query against a database is the reference's own use
 * (README.md:63-90) -- with every row of `b`, in one call that never waits for the GPU and never passes over the
 * per-pair offsets between counting and filling.  At SDR densities such a result is sparse (0.78 elements a pair at
 * 2048/40), so the caller sizes out_ids (and out_data when `this` carries payloads) by an ESTIMATE: `capacity`
 * elements.  out_offsets[na*nb + 1] always receives the complete exclusive offsets and *total_dev (device memory,
 * required) the exact element total; elements at positions >= capacity are not written -- a caller who reads a total
 * above its capacity calls again with room.  Needs a declared width <= 4096 on `b`, rows of `b` of at most 64 ids,
 * EmptyData or UnitData payloads, 32-bit ids.  Flags: SDRB_FLAG_SWAP.
 */
int sdrb_ande_few(const sdrb_batch* a, const sdrb_batch* b, unsigned flags, uint32_t* out_offsets, int32_t* out_ids,
                  float* out_data, uint64_t capacity, uint64_t* total_dev, void* stream);

/*
 * Batched visitors: what the reference's extension API hands to its callbacks (README.md:133-155), for
 * every pair, as arrays -- the raw material for operations the library does not have.
 *   SDRB_VISIT_AND -> andv (SDR.hpp:186-194, 958-997): one record per id match, ascending by id:
 *                     (position in `this`, position in `arg`)
 *   SDRB_VISIT_OR  -> orv (SDR.hpp:268-285, 1099-1166): one record per id of the union, ascending: a match,
 *                     or a single whose absent side is SDRB_POS_NONE (this_visitor / arg_visitor / both_visitor)
 * Positions index the rows' elements (ids[row_offsets[r] + position], data[...] likewise); payloads play no part,
 * so batches of any payload kind are taken.  Same two-call protocol, orientation and flags (SDRB_FLAG_SWAP,
 * SDRB_FLAG_ROWWISE, SDRB_FLAG_FORCE_MERGE) as the element ops; ids of 32 or 64 bits.
 */
#define SDRB_VISIT_AND 0
#define SDRB_VISIT_OR 1
#define SDRB_POS_NONE 0xFFFFFFFFu
int sdrb_visit_count(const sdrb_batch* a, const sdrb_batch* b, int which, unsigned flags, uint32_t* out_offsets,
                     uint64_t* total, void* stream);
int sdrb_visit_fill(const sdrb_batch* a, const sdrb_batch* b, int which, unsigned flags, const uint32_t* out_offsets,
                    uint32_t* this_pos, uint32_t* arg_pos, void* stream);

/*
 * Device-memory helpers for callers that do not link a HIP runtime themselves (the C++
 * SDRBatch front end, ctypes/cgo/JNI bindings).  sdrb_memcpy_* synchronise.
 */
int sdrb_malloc(int device, size_t bytes, void** ptr);
int sdrb_free(int device, void* ptr);
int sdrb_memcpy_h2d(int device, void* dst_dev, const void* src_host, size_t bytes, void* stream);
int sdrb_memcpy_d2h(int device, void* dst_host, const void* src_dev, size_t bytes, void* stream);
int sdrb_stream_sync(int device, void* stream);

/*
 * Scratch memory is kept per (device, stream) and grows on demand; it is not freed while the stream may be used
 * again.  An application that creates and destroys streams calls this before it destroys one: the stream is
 * synchronised and every scratch block kept for it is freed (a later call on a stream with the same handle value
 * starts with none).  No counterpart in the reference.
 */
int sdrb_stream_release(int device, void* stream);

/*
 * CSR database files (host side; no GPU needed except for sdrb_batch_upload_file).  The reference
 * has no serialisation (SDR.hpp:1670-1680 prints text only); the container is this engine's own:
 * a 64-byte header {magic "SDRBCSR1", version, payload flag, id bits, width, rows, elements,
 * section positions} followed by 64-byte aligned sections int32 ids[], float data[] (UnitData
 * only) and uint64 row_offsets[rows+1], little endian, mmap-able.
 *
 * Writer: rows are appended in chunks of any size (row_offsets relative to the chunk, first 0)
 * and checked like an upload (ascending ids SDR.hpp:679,1621; declared width; payloads in [0,1]
 * UnitData.hpp:15-17); close() finishes the file, abort() removes it.  A writer is freed by
 * close or abort.
 */
int sdrb_file_writer_open(const char* path, int32_t width, int data_kind, sdrb_file_writer** out);
int sdrb_file_writer_append(sdrb_file_writer* writer, uint64_t n_rows, const uint64_t* row_offsets,
                            const int32_t* ids, const float* data);
int sdrb_file_writer_close(sdrb_file_writer* writer);
int sdrb_file_writer_abort(sdrb_file_writer* writer);

/* Shape of a file; any out pointer may be NULL. */
int sdrb_file_info(const char* path, uint64_t* n_rows, uint64_t* nnz, int32_t* width, int* data_kind);
/* Elements held by rows [row_lo, row_hi). */
int sdrb_file_row_span(const char* path, uint64_t row_lo, uint64_t row_hi, uint64_t* nnz);
/* Rows [row_lo, row_hi) into caller arrays: row_offsets[row_hi-row_lo+1] rebased to 0, ids, data (may be NULL). */
int sdrb_file_read_rows(const char* path, uint64_t row_lo, uint64_t row_hi, uint64_t* row_offsets,
                        int32_t* ids, float* data);
/* Uploads rows [row_lo, row_hi) of a file into a batch straight from the mapping (a shard of a database file). */
int sdrb_batch_upload_file(sdrb_batch* batch, const char* path, uint64_t row_lo, uint64_t row_hi, void* stream);

/*
 * Deterministic synthetic rows for the bench/verify harness (no reference counterpart;
 * the reference's fuzzer draws from std::mt19937, src/fuzz.cpp:181-219).
 * Row r (global index first_row + k) is the sorted first `active` entries of a partial
 * Fisher-Yates shuffle of 0..width-1 driven by splitmix64 seeded with
 * seed ^ (r * 0x9E3779B97F4A7C15); payloads, when `data` is non-NULL, are
 * (x >> 40) * 2^-24 drawn after the ids, in id order, and are not filtered for relevance.
 * Host-only: writes row_offsets[n_rows+1], ids[n_rows*active], data[n_rows*active].
 */
int sdrb_synth_csr(uint64_t seed, uint64_t first_row, uint64_t n_rows, int32_t width,
                   int32_t active, uint64_t* row_offsets, int32_t* ids, float* data);

#ifdef __cplusplus
}
#endif
#endif /* SDR_AMD_H */
