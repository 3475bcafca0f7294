/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED ENGINE.
 *
 * CPU restatement, in plain C, of the single-pair SDR set algebra of jagprog5/SDR
 * (sorted, duplicate-free int ids with an optional per-id payload).  Only the test
 * suite, the smoke check and the CPU-baseline leg of the benchmark may load this
 * library; the HIP engine never calls it.
 *
 * Every function names the reference file:line it restates (paths relative to
 * /root/reference/include/SparseDistributedRepresentation/).
 *
 * Parity status: PINNED.  The restatement is checked bit-for-bit against the real
 * reference compiled from its own headers (oracle/_ref, see Makefile) on the
 * exhaustive 8-id fuzz grid of src/fuzz.cpp:236-252 and on seeded random rows, and
 * against the known answers of README.md:65-70,113-128 and src/test.cpp:172-272.
 */
#ifndef SDR_ORACLE_H
#define SDR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* payload kinds */
enum { OSDR_EMPTY = 0, /* DataTypes/EmptyData.hpp:18-92  */
       OSDR_UNIT = 1,  /* DataTypes/UnitData.hpp:11-140  */
       OSDR_ARITH = 2  /* DataTypes/ArithData.hpp:11-107 with arith_t = float; no xor ops */ };

/* set operations */
enum { OSDR_AND = 0, OSDR_OR = 1, OSDR_XOR = 2, OSDR_RM = 3 };

/* One SDR: ascending duplicate-free ids, payload array (NULL for OSDR_EMPTY). */
typedef struct {
    const int32_t* ids;
    const float* data;
    size_t n;
    int kind;
} osdr_view;

typedef void (*osdr_and_visitor)(void* ctx, size_t this_pos, size_t arg_pos);
typedef void (*osdr_one_visitor)(void* ctx, size_t pos);

/* SDR.hpp:958-997 -- leap-frog lower_bound intersection. */
void osdr_andv(const int32_t* t, size_t tn, const int32_t* a, size_t an,
               osdr_and_visitor visitor, void* ctx);

/* SDR.hpp:1099-1166 -- three-way union merge (3-callback form, default tails). */
void osdr_orv(const int32_t* t, size_t tn, const int32_t* a, size_t an,
              osdr_one_visitor this_visitor, osdr_one_visitor arg_visitor,
              osdr_and_visitor both_visitor, void* ctx);

/* SDR.hpp:1083-1095 (ands), 1294-1304 (ors), 1424-1434 (xors), 1568-1577 (rms). */
size_t osdr_size_op(int op, const osdr_view* t, const osdr_view* a);

/* SDR.hpp:999-1025 (ande), 1183-1222 (ore), 1307-1349 (xore), 1515-1566 (rme).
 * Result element type = this's element type (the default ret_t).  out_ids/out_data
 * must hold t->n + a->n entries (out_data may be NULL when t->kind is OSDR_EMPTY).
 * Returns the result length. */
size_t osdr_elem_op(int op, const osdr_view* t, const osdr_view* a,
                    int32_t* out_ids, float* out_data);

/* SDR.hpp:1027-1043 (andi, vector path), 1239-1244 (ori), 1366-1371 (xori),
 * 1447-1472 (rmi, vector path).  Operates on this's own arrays, which must have
 * capacity t_n + a->n.  Returns the new length.  Aliasing (a == this) is legal
 * (src/test.cpp:241-272). */
size_t osdr_inplace_op(int op, int32_t* t_ids, float* t_data, size_t t_n, int t_kind,
                       const osdr_view* a);

/* Host loops over CSR batches: out[i*a_n + j] = this_i.op(arg_j). */
/* Range and point queries on every row of a batch.
 *   counts[r * n + j]   = row_j.ands(lo[r], hi[r])   (SDR.hpp:931-956): two lower_bounds, the second
 *                         starting at the first one's result
 *   position[q * n + j] = index of qid[q] in row j or -1: what ande(id) finds (SDR.hpp:860-879:
 *                         lower_bound, then the id must be equal)
 * The slices of ande(lo, hi) (SDR.hpp:881-929) are rows[j][first .. first + count). */
void osdr_batch_range_size(const uint64_t* off, const int32_t* ids, uint64_t n, uint64_t n_ranges,
                           const int32_t* lo, const int32_t* hi, uint32_t* counts, uint32_t* firsts);
void osdr_batch_point(const uint64_t* off, const int32_t* ids, uint64_t n, uint64_t n_ids,
                      const int32_t* qid, int32_t* position);

/* SDR::inner on ArithData<float> elements (SDR.hpp:1682-1692): for every id match, in ascending
 * id order, acc += this_value * arg_value -- a rounded product (ArithData.hpp:37-43, value_ *=)
 * followed by a rounded sum (ArithData.hpp:59-63, value_ +=), starting from ArithData() = 0
 * (ArithData.hpp:16).  An EmptyData argument leaves the value unchanged (ArithData.hpp:39), so
 * its matches add this_value.  a_data == NULL means the argument rows carry no payload.
 * out[i * a_n + j] = this_i.inner(arg_j); also diff_mul's entry (i, j) (SDR.hpp:1886-1923) and
 * row_major_mul_vec's entry for row j when this is the vector (SDR.hpp:1729-1750). */
void osdr_batch_inner(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, uint64_t a_n,
                      float* out);

void osdr_batch_size_op(int op,
                        const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                        int t_kind, uint64_t t_n,
                        const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                        int a_kind, uint64_t a_n,
                        uint32_t* out);

/* out_off has t_n*a_n+1 entries (always written).  With out_ids == NULL only the
 * offsets are produced.  Returns the total number of result elements. */
uint64_t osdr_batch_elem_op(int op,
                            const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                            int t_kind, uint64_t t_n,
                            const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                            int a_kind, uint64_t a_n,
                            uint64_t* out_off, int32_t* out_ids, float* out_data);

/* out[i*a_n + j] = (this_i == arg_j), SDR::operator== (SDR.hpp:470-483): 1 or 0. */
void osdr_batch_equal(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, int t_kind, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, int a_kind, uint64_t a_n,
                      uint32_t* out);

/* The callbacks of andv (which = 0, SDR.hpp:958-997) or orv (which = 1, SDR.hpp:1099-1166) of every pair, in
 * call order, as (this position, arg position) records; the absent side of an orv single is 0xFFFFFFFF.
 * out_off has t_n*a_n+1 entries; this_pos / arg_pos may be NULL (count only).  Returns the total. */
uint64_t osdr_batch_visit(int which, const uint64_t* t_off, const int32_t* t_ids, uint64_t t_n,
                          const uint64_t* a_off, const int32_t* a_ids, uint64_t a_n,
                          uint64_t* out_off, uint32_t* this_pos, uint32_t* arg_pos);

/* Times `repeats` passes of the batched size op on one thread and returns the best
 * wall time in seconds; *checksum receives the sum of all results of one pass. */
double osdr_time_size_op(int op,
                         const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                         int t_kind, uint64_t t_n,
                         const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                         int a_kind, uint64_t a_n,
                         int repeats, uint64_t* checksum);

#ifdef __cplusplus
}
#endif
#endif
