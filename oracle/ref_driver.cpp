/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE SHIPPED ENGINE.
 *
 * extern "C" driver over the REAL reference headers.  It is compiled only where
 * /root/reference exists (see oracle/Makefile); the headers are included from where
 * they lie and nothing of them is copied here.  The resulting oracle/_ref/libsdr_ref.so
 * is what pins the C restatement in sdr_oracle.c, generates the golden fixtures, and
 * serves as the "reference" CPU baseline of the benchmark.
 *
 * Same batch signatures as sdr_oracle.h with an rsdr_ prefix.  Rows become
 * SDR<SDRElem<int,EmptyData>> / SDR<SDRElem<int,UnitData>> on std::vector, the
 * reference's default container (SDR.hpp:26), built with push_back (SDR.hpp:1614-1628:
 * no relevance filtering, ascending order asserted).
 */
#include "SparseDistributedRepresentation/SDR.hpp"
#include "SparseDistributedRepresentation/DataTypes/ArithData.hpp"
#include "SparseDistributedRepresentation/DataTypes/UnitData.hpp"

#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

using namespace sparse_distributed_representation;

namespace {

using EmptyElem = SDRElem<int, EmptyData>;
using UnitElem = SDRElem<int, UnitData>;
using ArithElem = SDRElem<int, ArithData<float>>;
using EmptySDR = SDR<EmptyElem>;
using UnitSDR = SDR<UnitElem>;
using ArithSDR = SDR<ArithElem>;

enum { K_EMPTY = 0, K_UNIT = 1, K_ARITH = 2 };
enum { OP_AND = 0, OP_OR = 1, OP_XOR = 2, OP_RM = 3 };

template <typename S>
std::vector<S> build(const uint64_t* off, const int32_t* ids, const float* data, uint64_t n) {
    std::vector<S> rows(n);
    for (uint64_t r = 0; r < n; ++r) {
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
            if constexpr (std::is_same_v<S, UnitSDR>) {
                rows[r].push_back(UnitElem(ids[k], UnitData(data[k])));
            } else if constexpr (std::is_same_v<S, ArithSDR>) {
                rows[r].push_back(ArithElem(ids[k], ArithData<float>(data[k])));
            } else {
                rows[r].push_back(EmptyElem(ids[k]));
            }
        }
    }
    return rows;
}

// ArithData defines no xor (ArithData.hpp:110-114): those calls do not compile in the reference, so they are
// kept out of the instantiations that involve it (callers never ask for them).
template <typename A, typename B>
constexpr bool has_xor = !std::is_same_v<A, ArithSDR> && !std::is_same_v<B, ArithSDR>;

template <typename A, typename B>
size_t size_op(int op, const A& a, const B& b) {
    switch (op) {
    case OP_AND: return a.ands(b);
    case OP_OR: return a.ors(b);
    case OP_XOR:
        if constexpr (has_xor<A, B>) return a.xors(b);
        else return 0;
    default: return a.rms(b);
    }
}

template <typename A, typename B>
A elem_op(int op, const A& a, const B& b) {
    switch (op) {
    case OP_AND: return a.ande(b);
    case OP_OR: return a.ore(b);
    case OP_XOR:
        if constexpr (has_xor<A, B>) return a.xore(b);
        else return A();
    default: return a.rme(b);
    }
}

template <typename A, typename B>
void inplace_op(int op, A& a, const B& b) {
    switch (op) {
    case OP_AND: a.andi(b); break;
    case OP_OR: a.ori(b); break;
    case OP_XOR:
        if constexpr (has_xor<A, B>) a.xori(b);
        break;
    default: a.rmi(b); break;
    }
}

template <typename S>
float payload(const typename S::value_type& e) {
    if constexpr (std::is_same_v<S, EmptySDR>) return 0.0f;
    else return e.data().value();
}

template <typename A, typename B>
void batch_size(int op, const std::vector<A>& t, const std::vector<B>& a, uint32_t* out) {
    for (size_t i = 0; i < t.size(); ++i)
        for (size_t j = 0; j < a.size(); ++j)
            out[i * a.size() + j] = (uint32_t)size_op(op, t[i], a[j]);
}

template <typename A, typename B>
uint64_t batch_elem(int op, bool inplace, const std::vector<A>& t, const std::vector<B>& a,
                    uint64_t* out_off, int32_t* out_ids, float* out_data) {
    uint64_t total = 0;
    for (size_t i = 0; i < t.size(); ++i)
        for (size_t j = 0; j < a.size(); ++j) {
            out_off[i * a.size() + j] = total;
            A r;
            if (inplace) {
                r = t[i];
                inplace_op(op, r, a[j]);
            } else {
                r = elem_op(op, t[i], a[j]);
            }
            if (out_ids) {
                uint64_t k = total;
                for (auto it = r.cbegin(); it != r.cend(); ++it, ++k) {
                    out_ids[k] = it->id();
                    if (out_data) out_data[k] = payload<A>(*it);
                }
            }
            total += r.size();
        }
    out_off[t.size() * a.size()] = total;
    return total;
}

template <typename A, typename B>
double time_size(int op, const std::vector<A>& t, const std::vector<B>& a, int repeats,
                 int threads, uint64_t* checksum) {
    double best = 1e300;
    uint64_t sum_out = 0;
    if (threads < 1) threads = 1;
    for (int r = 0; r < repeats; ++r) {
        std::vector<uint64_t> sums(threads, 0);
        auto work = [&](int w) {
            // arg rows statically partitioned over the workers
            size_t lo = a.size() * w / threads, hi = a.size() * (w + 1) / threads;
            uint64_t s = 0;
            for (size_t i = 0; i < t.size(); ++i)
                for (size_t j = lo; j < hi; ++j) s += size_op(op, t[i], a[j]);
            sums[w] = s;
        };
        auto t0 = std::chrono::steady_clock::now();
        if (threads == 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (int w = 0; w < threads; ++w) pool.emplace_back(work, w);
            for (auto& th : pool) th.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        double s = std::chrono::duration<double>(t1 - t0).count();
        if (s < best) best = s;
        sum_out = 0;
        for (auto v : sums) sum_out += v;
    }
    if (checksum) *checksum = sum_out;
    return best;
}

}  // namespace

#define DISPATCH(tk, ak, CALL)                                                        \
    do {                                                                              \
        if ((tk) == K_EMPTY && (ak) == K_EMPTY) { CALL(EmptySDR, EmptySDR); }         \
        else if ((tk) == K_EMPTY && (ak) == K_UNIT) { CALL(EmptySDR, UnitSDR); }      \
        else if ((tk) == K_UNIT && (ak) == K_EMPTY) { CALL(UnitSDR, EmptySDR); }      \
        else if ((tk) == K_ARITH && (ak) == K_ARITH) { CALL(ArithSDR, ArithSDR); }    \
        else if ((tk) == K_ARITH && (ak) == K_EMPTY) { CALL(ArithSDR, EmptySDR); }    \
        else if ((tk) == K_EMPTY && (ak) == K_ARITH) { CALL(EmptySDR, ArithSDR); }    \
        else { CALL(UnitSDR, UnitSDR); }                                              \
    } while (0)

extern "C" {

void rsdr_batch_size_op(int op,
                        const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                        int t_kind, uint64_t t_n,
                        const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                        int a_kind, uint64_t a_n, uint32_t* out) {
#define CALL(A, B)                                          \
    auto t = build<A>(t_off, t_ids, t_data, t_n);           \
    auto a = build<B>(a_off, a_ids, a_data, a_n);           \
    batch_size(op, t, a, out)
    DISPATCH(t_kind, a_kind, CALL);
#undef CALL
}

uint64_t rsdr_batch_elem_op(int op, int inplace,
                            const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                            int t_kind, uint64_t t_n,
                            const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                            int a_kind, uint64_t a_n,
                            uint64_t* out_off, int32_t* out_ids, float* out_data) {
    uint64_t total = 0;
#define CALL(A, B)                                          \
    auto t = build<A>(t_off, t_ids, t_data, t_n);           \
    auto a = build<B>(a_off, a_ids, a_data, a_n);           \
    total = batch_elem(op, inplace != 0, t, a, out_off, out_ids, t_kind != K_EMPTY ? out_data : nullptr)
    DISPATCH(t_kind, a_kind, CALL);
#undef CALL
    return total;
}

/* a.op(a) with the SAME object on both sides (src/test.cpp:241-272). Returns new length. */
/* SDR::transpose (SDR.hpp:1807-1826) of the matrix whose rows are the given batch rows (row id = row
 * number, ArithData<float> payloads; empty rows are left out, MatrixUtils.hpp:59-61).  Writes the
 * result column by column, in the order the reference produced it: col_off[width + 1], rows[], vals[]. */
void rsdr_transpose(const uint64_t* off, const int32_t* ids, const float* data, uint64_t n, int32_t width,
                    uint64_t* col_off, int32_t* rows, float* vals) {
    using Elem = SDRElem<int, ArithData<float>>;
    using Row = SDRElem<int, SDR<Elem>>;
    SDR<Row> m;
    for (uint64_t r = 0; r < n; ++r) {
        if (off[r] == off[r + 1]) continue;
        SDR<Elem> row;
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) row.push_back(Elem(ids[k], ArithData<float>(data[k])));
        m.push_back(Row((int)r, std::move(row)));
    }
    auto t = m.transpose();
    uint64_t w = 0;
    int32_t next_col = 0;
    for (auto it = t.cbegin(); it != t.cend(); ++it) {
        for (; next_col <= it->id(); ++next_col) col_off[next_col] = w;
        for (auto e = it->data().cbegin(); e != it->data().cend(); ++e, ++w) {
            rows[w] = e->id();
            vals[w] = e->data().value();
        }
    }
    for (; next_col <= width; ++next_col) col_off[next_col] = w;
}

/* The size and element ops on rows with `long` ids, SDR<SDRElem<long, Data>> (the id type the reference's
 * fuzzer also covers, src/fuzz.cpp:413-423). */
}  // extern "C"  (templates need C++ linkage)
namespace {
template <typename Data>
std::vector<SDR<SDRElem<long, Data>>> build_long(const uint64_t* off, const int64_t* ids, const float* data, uint64_t n) {
    std::vector<SDR<SDRElem<long, Data>>> rows(n);
    for (uint64_t r = 0; r < n; ++r)
        for (uint64_t k = off[r]; k < off[r + 1]; ++k) {
            if constexpr (std::is_same_v<Data, UnitData>) rows[r].push_back(SDRElem<long, Data>((long)ids[k], UnitData(data[k])));
            else rows[r].push_back(SDRElem<long, Data>((long)ids[k]));
        }
    return rows;
}

template <typename A, typename B>
uint64_t run_long(int op, bool elems, const std::vector<A>& t, const std::vector<B>& a, uint32_t* sizes,
                  uint64_t* out_off, int64_t* out_ids, float* out_data) {
    uint64_t total = 0;
    for (size_t i = 0; i < t.size(); ++i)
        for (size_t j = 0; j < a.size(); ++j) {
            if (!elems) {
                sizes[i * a.size() + j] = (uint32_t)size_op(op, t[i], a[j]);
                continue;
            }
            out_off[i * a.size() + j] = total;
            A r = elem_op(op, t[i], a[j]);
            if (out_ids) {
                uint64_t k = total;
                for (auto it = r.cbegin(); it != r.cend(); ++it, ++k) {
                    out_ids[k] = (int64_t)it->id();
                    if constexpr (std::is_same_v<typename A::value_type::data_type, UnitData>) {
                        if (out_data) out_data[k] = it->data().value();
                    }
                }
            }
            total += r.size();
        }
    if (elems) out_off[t.size() * a.size()] = total;
    return total;
}
}  // namespace
extern "C" {

uint64_t rsdr_batch_op64(int op, int elems,
                         const uint64_t* t_off, const int64_t* t_ids, const float* t_data, int t_kind, uint64_t t_n,
                         const uint64_t* a_off, const int64_t* a_ids, const float* a_data, int a_kind, uint64_t a_n,
                         uint32_t* sizes, uint64_t* out_off, int64_t* out_ids, float* out_data) {
    if (t_kind == K_UNIT && a_kind == K_UNIT)
        return run_long(op, elems != 0, build_long<UnitData>(t_off, t_ids, t_data, t_n),
                        build_long<UnitData>(a_off, a_ids, a_data, a_n), sizes, out_off, out_ids, out_data);
    if (t_kind == K_UNIT)
        return run_long(op, elems != 0, build_long<UnitData>(t_off, t_ids, t_data, t_n),
                        build_long<EmptyData>(a_off, a_ids, a_data, a_n), sizes, out_off, out_ids, out_data);
    if (a_kind == K_UNIT)
        return run_long(op, elems != 0, build_long<EmptyData>(t_off, t_ids, t_data, t_n),
                        build_long<UnitData>(a_off, a_ids, a_data, a_n), sizes, out_off, out_ids, out_data);
    return run_long(op, elems != 0, build_long<EmptyData>(t_off, t_ids, t_data, t_n),
                    build_long<EmptyData>(a_off, a_ids, a_data, a_n), sizes, out_off, out_ids, out_data);
}

/* Range and point queries by the reference's own members: ands(lo, hi) (SDR.hpp:931-956), the ids of
 * ande(lo, hi) (SDR.hpp:881-929; digest = sum of (k+1)*id over the slice, enough to pin content and order
 * next to the count) and ande(id) (SDR.hpp:860-879; found flag and, for UnitData, the payload found). */
void rsdr_batch_range(const uint64_t* off, const int32_t* ids, const float* data, int kind, uint64_t n,
                      uint64_t n_ranges, const int32_t* lo, const int32_t* hi, uint32_t* counts, int64_t* digests) {
    auto run = [&](auto rows) {
        for (uint64_t r = 0; r < n_ranges; ++r)
            for (uint64_t j = 0; j < n; ++j) {
                counts[r * n + j] = (uint32_t)rows[j].ands(lo[r], hi[r]);
                auto slice = rows[j].ande(lo[r], hi[r]);
                int64_t d = 0, k = 0;
                for (auto it = slice.cbegin(); it != slice.cend(); ++it) d += (++k) * (int64_t)it->id();
                digests[r * n + j] = d + (int64_t)slice.size() * 1000003;
            }
    };
    if (kind == K_UNIT) run(build<UnitSDR>(off, ids, data, n));
    else run(build<EmptySDR>(off, ids, data, n));
}

void rsdr_batch_point(const uint64_t* off, const int32_t* ids, const float* data, int kind, uint64_t n,
                      uint64_t n_ids, const int32_t* qid, uint8_t* found, float* value) {
    if (kind == K_UNIT) {
        auto rows = build<UnitSDR>(off, ids, data, n);
        for (uint64_t q = 0; q < n_ids; ++q)
            for (uint64_t j = 0; j < n; ++j) {
                const UnitData* p = rows[j].ande(qid[q]);
                found[q * n + j] = p != nullptr;
                value[q * n + j] = p ? p->value() : -1.0f;
            }
    } else {
        auto rows = build<EmptySDR>(off, ids, data, n);
        for (uint64_t q = 0; q < n_ids; ++q)
            for (uint64_t j = 0; j < n; ++j) {
                found[q * n + j] = rows[j].ande(qid[q]) != nullptr;
                value[q * n + j] = -1.0f;
            }
    }
}

/* this_i == arg_j with the reference's operator== (SDR.hpp:470-483).  The reference compiles it for an EmptyData
 * `this` only against EmptyData (EmptyData has no != for payloads), so those are the pairings served. */
void rsdr_batch_equal(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, int t_kind, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, int a_kind, uint64_t a_n,
                      uint32_t* out) {
#define CALL(A, B)                                          \
    do {                                                    \
        auto t = build<A>(t_off, t_ids, t_data, t_n);       \
        auto a = build<B>(a_off, a_ids, a_data, a_n);       \
        for (size_t i = 0; i < t.size(); ++i)               \
            for (size_t j = 0; j < a.size(); ++j) out[i * a.size() + j] = (t[i] == a[j]) ? 1u : 0u; \
    } while (0)
    if (t_kind == K_EMPTY && a_kind == K_EMPTY) CALL(EmptySDR, EmptySDR);
    else if (t_kind == K_UNIT && a_kind == K_UNIT) CALL(UnitSDR, UnitSDR);
    else if (t_kind == K_UNIT && a_kind == K_EMPTY) CALL(UnitSDR, EmptySDR);
    else if (t_kind == K_ARITH && a_kind == K_ARITH) CALL(ArithSDR, ArithSDR);
    else if (t_kind == K_ARITH && a_kind == K_EMPTY) CALL(ArithSDR, EmptySDR);
#undef CALL
}

/* What the reference's visitors are called with: andv (which = 0, SDR.hpp:186-194) and orv (which = 1,
 * SDR.hpp:268-269) on SDR<SDRElem<int, EmptyData>> rows in std::vector containers, every callback recorded
 * as (this position, arg position); 0xFFFFFFFF marks the absent side of an orv single. */
uint64_t rsdr_batch_visit(int which, const uint64_t* t_off, const int32_t* t_ids, uint64_t t_n,
                          const uint64_t* a_off, const int32_t* a_ids, uint64_t a_n,
                          uint64_t* out_off, uint32_t* this_pos, uint32_t* arg_pos) {
    std::vector<EmptySDR> t = build<EmptySDR>(t_off, t_ids, nullptr, t_n);
    std::vector<EmptySDR> a = build<EmptySDR>(a_off, a_ids, nullptr, a_n);
    uint64_t n = 0;
    for (uint64_t i = 0; i < t_n; ++i)
        for (uint64_t j = 0; j < a_n; ++j) {
            out_off[i * a_n + j] = n;
            EmptySDR& ti = t[i];
            EmptySDR& aj = a[j];
            auto rec = [&](uint32_t tp, uint32_t ap) {
                if (this_pos) {
                    this_pos[n] = tp;
                    arg_pos[n] = ap;
                }
                ++n;
            };
            auto both = [&](EmptySDR::iterator tp, EmptySDR::iterator ap) {
                rec((uint32_t)(tp - ti.begin()), (uint32_t)(ap - aj.begin()));
            };
            if (which == 0) {
                ti.andv(aj, both);
            } else {
                auto only_this = [&](EmptySDR::iterator tp) { rec((uint32_t)(tp - ti.begin()), 0xFFFFFFFFu); };
                auto only_arg = [&](EmptySDR::iterator ap) { rec(0xFFFFFFFFu, (uint32_t)(ap - aj.begin())); };
                ti.orv(aj, only_this, only_arg, both);
            }
        }
    out_off[t_n * a_n] = n;
    return n;
}

/* this_i.inner(arg_j) on SDR<SDRElem<int, ArithData<float>>> rows (SDR.hpp:1682-1692); a_data == NULL:
 * the argument rows are SDR<SDRElem<int, EmptyData>>.  Also checks that diff_mul on the same rows,
 * taken as a row-major and a column-major matrix, produces the same entries (SDR.hpp:1886-1923). */
void rsdr_batch_inner(const uint64_t* t_off, const int32_t* t_ids, const float* t_data, uint64_t t_n,
                      const uint64_t* a_off, const int32_t* a_ids, const float* a_data, uint64_t a_n,
                      float* out) {
    using ArithElem = SDRElem<int, ArithData<float>>;
    using ArithSDR = SDR<ArithElem>;
    std::vector<ArithSDR> t(t_n);
    for (uint64_t r = 0; r < t_n; ++r)
        for (uint64_t k = t_off[r]; k < t_off[r + 1]; ++k) t[r].push_back(ArithElem(t_ids[k], ArithData<float>(t_data[k])));
    if (a_data) {
        std::vector<ArithSDR> a(a_n);
        for (uint64_t r = 0; r < a_n; ++r)
            for (uint64_t k = a_off[r]; k < a_off[r + 1]; ++k) a[r].push_back(ArithElem(a_ids[k], ArithData<float>(a_data[k])));
        for (uint64_t i = 0; i < t_n; ++i)
            for (uint64_t j = 0; j < a_n; ++j) out[i * a_n + j] = t[i].inner<ArithData<float>>(a[j]).value();
    } else {
        std::vector<EmptySDR> a = build<EmptySDR>(a_off, a_ids, nullptr, a_n);
        for (uint64_t i = 0; i < t_n; ++i)
            for (uint64_t j = 0; j < a_n; ++j) out[i * a_n + j] = t[i].inner<ArithData<float>>(a[j]).value();
    }
}

uint64_t rsdr_inplace_self(int op, int32_t* ids, float* data, uint64_t n, int kind) {
    uint64_t off[2] = {0, n};
    uint64_t len = 0;
    if (kind == K_UNIT) {
        auto v = build<UnitSDR>(off, ids, data, 1);
        inplace_op(op, v[0], v[0]);
        for (auto it = v[0].cbegin(); it != v[0].cend(); ++it, ++len) {
            ids[len] = it->id();
            data[len] = it->data().value();
        }
    } else {
        auto v = build<EmptySDR>(off, ids, data, 1);
        inplace_op(op, v[0], v[0]);
        for (auto it = v[0].cbegin(); it != v[0].cend(); ++it, ++len) ids[len] = it->id();
    }
    return len;
}

double rsdr_time_size_op(int op,
                         const uint64_t* t_off, const int32_t* t_ids, const float* t_data,
                         int t_kind, uint64_t t_n,
                         const uint64_t* a_off, const int32_t* a_ids, const float* a_data,
                         int a_kind, uint64_t a_n,
                         int repeats, int threads, uint64_t* checksum) {
    double best = 0;
#define CALL(A, B)                                          \
    auto t = build<A>(t_off, t_ids, t_data, t_n);           \
    auto a = build<B>(a_off, a_ids, a_data, a_n);           \
    best = time_size(op, t, a, repeats, threads, checksum)
    DISPATCH(t_kind, a_kind, CALL);
#undef CALL
    return best;
}

}  // extern "C"
