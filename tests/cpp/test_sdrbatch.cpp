// Drop-in check of include/sdr_amd/SDRBatch.hpp against the REAL reference: SDR objects are built
// with the reference's own classes and containers, packed, evaluated on the GPU, and every pair
// is compared with the reference's per-pair call using the reference's operator== (ids and
// payloads, SDR.hpp:470-483).  Built only where /root/reference exists (oracle/Makefile ->
// oracle/_ref/test_sdrbatch); the GPU test-suite runs the binary when it is present.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <forward_list>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "SparseDistributedRepresentation/SDR.hpp"
#include "SparseDistributedRepresentation/DataTypes/ArithData.hpp"
#include "SparseDistributedRepresentation/DataTypes/UnitData.hpp"
#include "SparseDistributedRepresentation/IDContiguousContainer.hpp"
#include "sdr_amd/SDRBatch.hpp"

using namespace sparse_distributed_representation;

static long g_checks = 0;
#define REQUIRE(x) do { ++g_checks; if (!(x)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

// long ids are spread beyond 32 bits unless `small_long` keeps them as they are (ids that 32 bits would hold too)
static bool g_small_long = false;
// 1: ids close together but far from zero (long ids near 2^40, int ids from -1700 on): the engine moves such ids to zero
//    for a call; 2: int ids far apart and partly negative (times 1 100 003, less 7): no table, a wave per pair searches
static int g_id_mode = 0;

template <typename SDR_t>
std::vector<SDR_t> make_rows(std::mt19937& rng, size_t n, int width, int max_len) {
    using Elem = typename SDR_t::value_type;
    using Data = typename Elem::data_type;
    std::vector<SDR_t> rows(n);
    using Id = typename Elem::id_type;
    std::uniform_real_distribution<float> val(0.0f, 1.0f);
    std::normal_distribution<float> gauss(0.0f, 3.0f);
    for (auto& r : rows) {
        int len = (int)(rng() % (unsigned)(max_len + 1));
        std::set<int> drawn;
        while ((int)drawn.size() < len) drawn.insert((int)(rng() % (unsigned)width));
        std::set<Id> ids;
        for (int small : drawn) {
            Id id = small;
            if constexpr (sizeof(Id) > 4) {
                // ids that need 64 bits, of either sign; every third id stays small, so that rows of `int` ids can share it
                if (g_id_mode == 1) id = (Id)small + ((Id)1 << 40);
                else if (!g_small_long && small % 3 == 1) id = (Id)small * 0x100000003L + 0x4000000000L;
                else if (!g_small_long && small % 3 == 2) id = -(Id)small * 0x100000003L - 0x4000000000L;
            } else {
                if (g_id_mode == 1) id = (Id)(small - 1700);
                else if (g_id_mode == 2) id = (Id)(small * 1100003 - 7 * 1100003);
            }
            ids.insert(id);
        }
        std::vector<Elem> elems;
        for (Id id : ids) {
            if constexpr (std::is_same_v<Data, UnitData>) elems.push_back(Elem(id, UnitData(val(rng))));
            else if constexpr (std::is_same_v<Data, ArithData<float>>) elems.push_back(Elem(id, ArithData<float>(gauss(rng))));
            else elems.push_back(Elem(id));
        }
        if constexpr (SDR_t::uses_flist_like) {
            r = SDR_t(elems.begin(), elems.end());   // iterator ctor: no relevance filtering (SDR.hpp:83)
        } else {
            for (auto& e : elems) r.push_back(e);    // no filtering either (SDR.hpp:1614-1628)
        }
    }
    return rows;
}

template <typename A, typename B>
int run(const char* name, std::mt19937& rng, int width, int32_t declared_width) {
    auto qs = make_rows<A>(rng, 37, width, 40);
    auto ds = make_rows<B>(rng, 301, width, 40);
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, declared_width);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, declared_width);
    auto c_and = q.ands(d).to_host(), c_or = q.ors(d).to_host(), c_xor = q.xors(d).to_host(), c_rm = q.rms(d).to_host();
    auto e_and = q.ande(d), e_or = q.ore(d), e_xor = q.xore(d), e_rm = q.rme(d);
    auto h_and = e_and.to_host(), h_or = e_or.to_host(), h_xor = e_xor.to_host(), h_rm = e_rm.to_host();
    // narrow counts: where the sizes depend on ids alone and the batch declares a width they equal the reference's too
    constexpr bool ids_decide = !sdr_amd::detail::has_value<typename A::value_type::data_type>::value;
    if (ids_decide && declared_width > 0 && sizeof(typename A::value_type::id_type) <= 4) {
        auto n_and = q.template size_op_narrow<uint8_t>(SDRB_OP_AND, d).to_host();
        auto n_xor = q.template size_op_narrow<uint16_t>(SDRB_OP_XOR, d).to_host();
        for (size_t i = 0; i < qs.size(); ++i)
            for (size_t j = 0; j < ds.size(); ++j) {
                REQUIRE(n_and[i * ds.size() + j] == qs[i].ands(ds[j]));
                REQUIRE(n_xor[i * ds.size() + j] == qs[i].xors(ds[j]));
            }
    }
    // the default result type of the reference is this's element type in a std::vector
    using R = SDR<typename A::value_type, std::vector<typename A::value_type>>;
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            const size_t k = i * ds.size() + j;
            REQUIRE(c_and[k] == qs[i].ands(ds[j]));
            REQUIRE(c_or[k] == qs[i].ors(ds[j]));
            REQUIRE(c_xor[k] == qs[i].xors(ds[j]));
            REQUIRE(c_rm[k] == qs[i].rms(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_and, ds.size(), i, j) == qs[i].ande(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_or, ds.size(), i, j) == qs[i].ore(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_xor, ds.size(), i, j) == qs[i].xore(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_rm, ds.size(), i, j) == qs[i].rme(ds[j]));
        }
    // results in chunks over the rows of `this` (what a result of more than 2^32-1 elements comes back as; the limit
    // is lowered so that these small batches need several chunks): every pair equals the reference's call
    if constexpr (sizeof(typename A::value_type::id_type) <= 4 && sizeof(typename B::value_type::id_type) <= 4) {
        const int ops[4] = {SDRB_OP_AND, SDRB_OP_OR, SDRB_OP_XOR, SDRB_OP_RM};
        for (int op : ops) {
            auto parts = q.elem_op_chunked(op, d, 0, nullptr, /*limit*/ 60000);
            REQUIRE(parts.chunks.size() > 1);
            REQUIRE(parts.starts.front() == 0 && parts.starts.back() == qs.size());
            std::vector<sdr_amd::Elements::Host> hosts;
            for (auto& c : parts.chunks) hosts.push_back(c.to_host());
            for (size_t i = 0; i < qs.size(); ++i) {
                const auto where = parts.locate(i);
                for (size_t j = 0; j < ds.size(); j += 7) {
                    const R got = sdr_amd::Elements::unpack<R>(hosts[where.first], ds.size(), where.second, j);
                    if (op == SDRB_OP_AND) REQUIRE(got == qs[i].ande(ds[j]));
                    else if (op == SDRB_OP_OR) REQUIRE(got == qs[i].ore(ds[j]));
                    else if (op == SDRB_OP_XOR) REQUIRE(got == qs[i].xore(ds[j]));
                    else REQUIRE(got == qs[i].rme(ds[j]));
                }
            }
        }
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// ande of a few rows of `this` against a database (the one-enqueue path sized by an estimate): every pair equals the
// reference's ande; with `copies` the database repeats the queries' rows, so the result outgrows the estimate and
// is computed a second time with room.
template <typename A>
int run_few(const char* name, std::mt19937& rng, int width, int len, bool copies) {
    auto qs = make_rows<A>(rng, 5, width, len);
    auto ds = make_rows<A>(rng, 301, width, len);
    if (copies)
        for (size_t j = 0; j < ds.size(); ++j) ds[j] = qs[j % qs.size()];
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, width);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, width);
    using R = SDR<typename A::value_type, std::vector<typename A::value_type>>;
    auto e = q.ande(d);
    auto h = e.to_host();
    auto e_sw = d.ande(q, SDRB_FLAG_SWAP);   // this = the argument batch: result (j, i) = qs[i].ande(ds[j])
    auto h_sw = e_sw.to_host();
    uint64_t total = 0;
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            const R want = qs[i].ande(ds[j]);
            total += want.size();
            REQUIRE(sdr_amd::Elements::unpack<R>(h, ds.size(), i, j) == want);
            REQUIRE(sdr_amd::Elements::unpack<R>(h_sw, qs.size(), j, i) == want);
        }
    REQUIRE(e.total == total && e_sw.total == total);
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// Result types other than this's own, a.ande<ret_t, c_ret_t>(b) of the reference (SDR.hpp:151-152, 209-212,
// 237-240, 292-293), pair by pair against the reference's own template instantiations.
template <typename A, typename B, typename RetElem, bool kWithRme>
int run_retarget(const char* name, std::mt19937& rng, int ret_kind) {
    auto qs = make_rows<A>(rng, 23, 200, 40);
    auto ds = make_rows<B>(rng, 150, 200, 40);
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, 200);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 200);
    using R = SDR<RetElem, std::vector<RetElem>>;
    auto h_and = q.elem_op_as(ret_kind, SDRB_OP_AND, d).to_host();
    auto h_or = q.elem_op_as(ret_kind, SDRB_OP_OR, d).to_host();
    auto h_xor = q.elem_op_as(ret_kind, SDRB_OP_XOR, d).to_host();
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            REQUIRE(sdr_amd::Elements::unpack<R>(h_and, ds.size(), i, j) == (qs[i].template ande<RetElem, std::vector<RetElem>>(ds[j])));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_or, ds.size(), i, j) == (qs[i].template ore<RetElem, std::vector<RetElem>>(ds[j])));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_xor, ds.size(), i, j) == (qs[i].template xore<RetElem, std::vector<RetElem>>(ds[j])));
        }
    // an EmptyData `this` has no rme into UnitData: EmptyData::rme returns EmptyData, the reference does not compile it
    if constexpr (kWithRme) {
        auto h_rm = q.elem_op_as(ret_kind, SDRB_OP_RM, d).to_host();
        for (size_t i = 0; i < qs.size(); ++i)
            for (size_t j = 0; j < ds.size(); ++j)
                REQUIRE(sdr_amd::Elements::unpack<R>(h_rm, ds.size(), i, j) == (qs[i].template rme<RetElem, std::vector<RetElem>>(ds[j])));
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// rmi with a caller's combiner (SDR.hpp:304-310, 1447-1513): the same lambda body handed to the reference row by row
template <typename A, typename B>
int run_rmi_combiner(const char* name, std::mt19937& rng) {
    auto as = make_rows<A>(rng, 120, 150, 40);
    auto bs = make_rows<B>(rng, 120, 150, 40);
    auto a = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 150);
    auto b = sdr_amd::SDRBatch::pack(bs.begin(), bs.end(), 0, 150);
    constexpr bool arg_has_value = sdr_amd::detail::has_value<typename B::value_type::data_type>::value;
    a.rmi(b, [](float mine, const float* theirs) { return theirs ? mine * (1.0f - *theirs * 0.5f) : mine * 0.5f; });
    auto h = a.to_host();
    for (size_t j = 0; j < as.size(); ++j) {
        A want = as[j];
        want.rmi(bs[j], [](typename A::iterator this_pos, typename B::iterator arg_pos) {
            auto& mine = const_cast<typename A::value_type::data_type&>(this_pos->data());
            float v;
            if constexpr (arg_has_value) v = mine.value() * (1.0f - arg_pos->data().value() * 0.5f);
            else v = mine.value() * 0.5f;
            (void)arg_pos;
            mine = typename A::value_type::data_type(v);
        });
        REQUIRE(sdr_amd::Elements::unpack<A>(h, 1, j, 0) == want);
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// all-pairs top-k of a database with itself: every count is the reference's ands, the order is
// overlap descending then row ascending, the self match is excluded
int run_allpairs(std::mt19937& rng, unsigned flags, const char* name) {
    using E = SDRElem<>;
    auto ds = make_rows<SDR<E>>(rng, 333, 512, 40);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 512);
    const uint32_t k = 7;
    auto top = d.allpairs_topk(d, k, 0, flags);
    auto counts = top.counts_to_host();
    auto rows = top.rows_to_host();
    for (size_t i = 0; i < ds.size(); ++i) {
        std::vector<std::pair<long, long>> cand;   // (-overlap, row)
        for (size_t j = 0; j < ds.size(); ++j)
            if (j != i) cand.push_back({-(long)ds[i].ands(ds[j]), (long)j});
        std::sort(cand.begin(), cand.end());
        for (uint32_t t = 0; t < k; ++t) {
            REQUIRE(rows[i * k + t] == cand[t].second);
            REQUIRE((long)counts[i * k + t] == -cand[t].first);
        }
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// sparse dot products on ArithData<float> rows: every entry equals the reference's inner() bit for bit,
// and so do the entries of row_major_mul_vec (one vector against the rows of a matrix)
template <typename B>
int run_inner(std::mt19937& rng, const char* name) {
    using AE = SDRElem<int, ArithData<float>>;
    auto qs = make_rows<SDR<AE>>(rng, 23, 256, 40);
    auto ds = make_rows<B>(rng, 211, 256, 40);
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, 256);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 256);
    REQUIRE(q.data_kind() == SDRB_DATA_ARITH);
    auto v = q.inner(d).to_host();
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            const float want = qs[i].template inner<ArithData<float>>(ds[j]).value();
            REQUIRE(std::memcmp(&v[i * ds.size() + j], &want, 4) == 0);
        }
    if constexpr (std::is_same_v<typename B::value_type, AE>) {
        // matrix (rows ds, row ids 0..n-1) times vector qs[0]: row_major_mul_vec keeps every row (ArithData is always relevant)
        using Row = SDRElem<int, SDR<AE>>;
        SDR<Row> m;
        for (size_t j = 0; j < ds.size(); ++j) m.push_back(Row((int)j, SDR<AE>(ds[j])));
        auto prod = m.row_major_mul_vec(qs[0]);
        auto d_as_this = d.inner(q).to_host();   // d_j.inner(q_i) at [j][i]
        size_t j = 0;
        for (auto it = prod.cbegin(); it != prod.cend(); ++it, ++j) {
            REQUIRE((size_t)it->id() == j);
            const float want = it->data().value();
            REQUIRE(std::memcmp(&d_as_this[j * qs.size() + 0], &want, 4) == 0);
        }
        REQUIRE(j == ds.size());
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// row-wise and in-place forms: b.andi(c) ... on whole batches equal the reference's in-place results, a.op(a) included
template <typename A, typename B>
int run_inplace(std::mt19937& rng, const char* name) {
    using R = SDR<typename A::value_type, std::vector<typename A::value_type>>;
    const int ops[4] = {SDRB_OP_AND, SDRB_OP_OR, SDRB_OP_XOR, SDRB_OP_RM};
    for (int op : ops) {
        auto as = make_rows<A>(rng, 150, 128, 40);
        auto bs = make_rows<B>(rng, 150, 128, 40);
        auto a = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 128);
        auto b = sdr_amd::SDRBatch::pack(bs.begin(), bs.end(), 0, 128);
        auto sizes = a.rowwise_size(op, b).to_host();
        switch (op) {
            case SDRB_OP_AND: a.andi(b); break;
            case SDRB_OP_OR: a.ori(b); break;
            case SDRB_OP_XOR: a.xori(b); break;
            default: a.rmi(b); break;
        }
        auto h = a.to_host();
        for (size_t j = 0; j < as.size(); ++j) {
            size_t want_size = op == SDRB_OP_AND ? as[j].ands(bs[j]) : op == SDRB_OP_OR ? as[j].ors(bs[j])
                             : op == SDRB_OP_XOR ? as[j].xors(bs[j]) : as[j].rms(bs[j]);
            REQUIRE(sizes[j] == want_size);
            R want(as[j].cbegin(), as[j].cend());
            switch (op) {
                case SDRB_OP_AND: want.andi(bs[j]); break;
                case SDRB_OP_OR: want.ori(bs[j]); break;
                case SDRB_OP_XOR: want.xori(bs[j]); break;
                default: want.rmi(bs[j]); break;
            }
            REQUIRE(sdr_amd::Elements::unpack<R>(h, as.size(), 0, j) == want);
        }
    }
    // a.op(a) (src/test.cpp:241-272)
    auto as = make_rows<A>(rng, 60, 128, 40);
    for (int op : ops) {
        auto a = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 128);
        switch (op) {
            case SDRB_OP_AND: a.andi(a); break;
            case SDRB_OP_OR: a.ori(a); break;
            case SDRB_OP_XOR: a.xori(a); break;
            default: a.rmi(a); break;
        }
        auto h = a.to_host();
        for (size_t j = 0; j < as.size(); ++j) {
            R want(as[j].cbegin(), as[j].cend());
            switch (op) {
                case SDRB_OP_AND: want.andi(want); break;
                case SDRB_OP_OR: want.ori(want); break;
                case SDRB_OP_XOR: want.xori(want); break;
                default: want.rmi(want); break;
            }
            REQUIRE(sdr_amd::Elements::unpack<R>(h, as.size(), 0, j) == want);
        }
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// range / point queries and the transpose against the reference's members
int run_queries(std::mt19937& rng, const char* name) {
    using U = SDRElem<int, UnitData>;
    using R = SDR<U>;
    auto ds = make_rows<R>(rng, 180, 300, 40);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 300);
    const std::vector<int32_t> lo{0, 10, 150, 299, 77, 0}, hi{300, 20, 150, 300, 260, 1};
    auto sizes = d.range_size(lo, hi).to_host();
    auto elems = d.range_elems(lo, hi).to_host();
    for (size_t r = 0; r < lo.size(); ++r)
        for (size_t j = 0; j < ds.size(); ++j) {
            REQUIRE(sizes[r * ds.size() + j] == ds[j].ands(lo[r], hi[r]));
            REQUIRE(sdr_amd::Elements::unpack<R>(elems, ds.size(), r, j) == ds[j].ande(lo[r], hi[r]));
        }
    const std::vector<int32_t> probe{0, 5, 123, 299, 300, -4};
    auto pos = d.point_query(probe).to_host();
    auto host = d.to_host();
    for (size_t q = 0; q < probe.size(); ++q)
        for (size_t j = 0; j < ds.size(); ++j) {
            const UnitData* want = ds[j].ande(probe[q]);
            const int32_t p = pos[q * ds.size() + j];
            REQUIRE((want != nullptr) == (p >= 0));
            if (want) REQUIRE(host.data[host.offsets[j] + p] == want->value());
        }
    // transpose: row c of the result lists, ascending, exactly the rows j with ds[j].ande(c) != nullptr and
    // carries their payloads.  (The reference's own transpose() fills columns in heap order and trips its
    // push_back assertion in a build with asserts on, so its point lookup is the pin here; the committed
    // fixture of the Python suite holds transpose() outputs from a build without asserts.)
    auto t = d.transpose();
    REQUIRE(t.size() == 300);
    auto th = t.to_host();
    for (int c = 0; c < 300; ++c) {
        uint32_t k = th.offsets[c];
        for (size_t j = 0; j < ds.size(); ++j) {
            const UnitData* want = ds[j].ande(c);
            if (!want) continue;
            REQUIRE(k < th.offsets[c + 1]);
            REQUIRE(th.ids[k] == (int64_t)j);
            REQUIRE(th.data[k] == want->value());
            ++k;
        }
        REQUIRE(k == th.offsets[c + 1]);
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// the set algebra on ArithData<float> rows (and = product, or = sum, rm = difference; no xor): every pair
// against the reference's own calls
template <typename A, typename B>
int run_arith(std::mt19937& rng, const char* name) {
    auto qs = make_rows<A>(rng, 19, 200, 40);
    auto ds = make_rows<B>(rng, 150, 200, 40);
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, 200);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 200);
    auto c_and = q.ands(d).to_host(), c_or = q.ors(d).to_host(), c_rm = q.rms(d).to_host();
    auto h_and = q.ande(d).to_host(), h_or = q.ore(d).to_host(), h_rm = q.rme(d).to_host();
    using R = SDR<typename A::value_type, std::vector<typename A::value_type>>;
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            const size_t k = i * ds.size() + j;
            REQUIRE(c_and[k] == qs[i].ands(ds[j]));
            REQUIRE(c_or[k] == qs[i].ors(ds[j]));
            REQUIRE(c_rm[k] == qs[i].rms(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_and, ds.size(), i, j) == qs[i].ande(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_or, ds.size(), i, j) == qs[i].ore(ds[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_rm, ds.size(), i, j) == qs[i].rme(ds[j]));
        }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// batched visitors: the records equal, call by call, what the reference's andv / orv hand to visitors
// (positions = iterator - begin()), for vector rows with payloads on one side
int run_visitors(std::mt19937& rng, const char* name) {
    using U = SDRElem<int, UnitData>;
    using E = SDRElem<>;
    auto qs = make_rows<SDR<U>>(rng, 17, 300, 80);
    auto ds = make_rows<SDR<E>>(rng, 120, 300, 80);
    auto q = sdr_amd::SDRBatch::pack(qs.begin(), qs.end(), 0, 300);
    auto d = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 300);
    auto av = q.andv(d).to_host(), ov = q.orv(d).to_host();
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            const size_t p = i * ds.size() + j;
            uint32_t k = av.offsets[p];
            bool ok = true;
            qs[i].andv(ds[j], [&](SDR<U>::iterator tp, SDR<E>::iterator ap) {
                ok = ok && k < av.offsets[p + 1] && av.this_pos[k] == (uint32_t)(tp - qs[i].begin()) &&
                     av.arg_pos[k] == (uint32_t)(ap - ds[j].begin());
                ++k;
            });
            REQUIRE(ok && k == av.offsets[p + 1]);
            k = ov.offsets[p];
            auto rec = [&](uint32_t tp, uint32_t ap) {
                ok = ok && k < ov.offsets[p + 1] && ov.this_pos[k] == tp && ov.arg_pos[k] == ap;
                ++k;
            };
            qs[i].orv(ds[j],
                      [&](SDR<U>::iterator tp) { rec((uint32_t)(tp - qs[i].begin()), SDRB_POS_NONE); },
                      [&](SDR<E>::iterator ap) { rec(SDRB_POS_NONE, (uint32_t)(ap - ds[j].begin())); },
                      [&](SDR<U>::iterator tp, SDR<E>::iterator ap) {
                          rec((uint32_t)(tp - qs[i].begin()), (uint32_t)(ap - ds[j].begin()));
                      });
            REQUIRE(ok && k == ov.offsets[p + 1]);
            // the five-callback form (SDR.hpp:284-285): visitor calls up to the point where one side is exhausted, then
            // ONE position call for the rest of the other side
            k = ov.offsets[p];
            int tail_calls = 0;
            bool tail_arg_only = false;
            uint32_t tail_pos = 0;
            qs[i].orv(ds[j],
                      [&](SDR<U>::iterator tp) { rec((uint32_t)(tp - qs[i].begin()), SDRB_POS_NONE); },
                      [&](SDR<E>::iterator ap) { rec(SDRB_POS_NONE, (uint32_t)(ap - ds[j].begin())); },
                      [&](SDR<U>::iterator tp, SDR<E>::iterator ap) {
                          rec((uint32_t)(tp - qs[i].begin()), (uint32_t)(ap - ds[j].begin()));
                      },
                      [&](SDR<U>::iterator tp) { ++tail_calls; tail_arg_only = false; tail_pos = (uint32_t)(tp - qs[i].begin()); },
                      [&](SDR<E>::iterator ap) { ++tail_calls; tail_arg_only = true; tail_pos = (uint32_t)(ap - ds[j].begin()); });
            const auto tail = sdr_amd::Visits::orv_tail(ov, p, (uint32_t)qs[i].size(), (uint32_t)ds[j].size());
            REQUIRE(ok && tail_calls == 1 && k == tail.head_end && tail.arg_only == tail_arg_only && tail.position == tail_pos);
        }
    // rows that end together, an empty `this`, an empty `arg`
    {
        std::vector<SDR<U>> ts(3);
        std::vector<SDR<E>> as(3);
        for (int id : {2, 5, 9}) { ts[0].push_back(U(id, UnitData(0.5f))); as[0].push_back(E(id)); }
        for (int id : {1, 4}) as[1].push_back(E(id));          // ts[1] empty
        for (int id : {3, 8}) ts[2].push_back(U(id, UnitData(0.5f)));   // as[2] empty
        auto tb = sdr_amd::SDRBatch::pack(ts.begin(), ts.end(), 0, 16), ab = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 16);
        auto h = tb.orv(ab, SDRB_FLAG_ROWWISE).to_host();
        for (size_t r = 0; r < 3; ++r) {
            int tail_calls = 0;
            bool tail_arg_only = false;
            uint32_t tail_pos = 0, visits = 0;
            ts[r].orv(as[r], [&](SDR<U>::iterator) { ++visits; }, [&](SDR<E>::iterator) { ++visits; },
                      [&](SDR<U>::iterator, SDR<E>::iterator) { ++visits; },
                      [&](SDR<U>::iterator tp) { ++tail_calls; tail_arg_only = false; tail_pos = (uint32_t)(tp - ts[r].begin()); },
                      [&](SDR<E>::iterator ap) { ++tail_calls; tail_arg_only = true; tail_pos = (uint32_t)(ap - as[r].begin()); });
            const auto tail = sdr_amd::Visits::orv_tail(h, r, (uint32_t)ts[r].size(), (uint32_t)as[r].size());
            REQUIRE(tail_calls == 1 && tail.head_end - h.offsets[r] == visits && tail.arg_only == tail_arg_only && tail.position == tail_pos);
        }
    }
    // operator== for every pair (a few rows are made equal on purpose)
    for (size_t r : {size_t(1), size_t(2)}) {
        SDR<E> same;
        for (auto it = qs[r].cbegin(); it != qs[r].cend(); ++it) same.push_back(E(it->id()));
        ds[2 * r + 1] = same;
    }
    auto d2 = sdr_amd::SDRBatch::pack(ds.begin(), ds.end(), 0, 300);
    auto eq = q.equals(d2).to_host();
    long n_equal = 0;
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < ds.size(); ++j) {
            REQUIRE((eq[i * ds.size() + j] != 0) == (qs[i] == ds[j]));
            n_equal += eq[i * ds.size() + j];
        }
    REQUIRE(n_equal >= 2);
    auto self = q.equals(q).to_host();
    for (size_t i = 0; i < qs.size(); ++i)
        for (size_t j = 0; j < qs.size(); ++j) REQUIRE((self[i * qs.size() + j] != 0) == (qs[i] == qs[j]));
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// operator sugar (SDR.hpp:431-460) on batches equals the reference's operators on every pair / row
int run_operators(std::mt19937& rng, const char* name) {
    using U = SDRElem<int, UnitData>;
    using R = SDR<U>;
    auto as = make_rows<R>(rng, 25, 90, 30), bs = make_rows<R>(rng, 25, 90, 30);
    auto a = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 90);
    auto b = sdr_amd::SDRBatch::pack(bs.begin(), bs.end(), 0, 90);
    auto h_and = (a & b).to_host(), h_mul = (a * b).to_host(), h_or = (a | b).to_host(), h_add = (a + b).to_host();
    auto h_xor = (a ^ b).to_host(), h_sub = (a - b).to_host();
    auto c_and = (a && b).to_host(), c_or = (a || b).to_host();
    for (size_t i = 0; i < as.size(); ++i)
        for (size_t j = 0; j < bs.size(); ++j) {
            const size_t k = i * bs.size() + j;
            REQUIRE(sdr_amd::Elements::unpack<R>(h_and, bs.size(), i, j) == (as[i] & bs[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_mul, bs.size(), i, j) == (as[i] * bs[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_or, bs.size(), i, j) == (as[i] | bs[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_add, bs.size(), i, j) == (as[i] + bs[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_xor, bs.size(), i, j) == (as[i] ^ bs[j]));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_sub, bs.size(), i, j) == (as[i] - bs[j]));
            REQUIRE(c_and[k] == (as[i] && bs[j]));
            REQUIRE(c_or[k] == (as[i] || bs[j]));
        }
    std::vector<R> want(as);
    for (size_t j = 0; j < want.size(); ++j) {
        want[j] &= bs[j];
        want[j] |= as[j];
        want[j] ^= bs[j];
        want[j] -= as[j];
        want[j] += bs[j];
        want[j] *= as[j];
    }
    auto c = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 90);
    c &= b;
    c |= a;
    c ^= b;
    c -= a;
    c += b;
    c *= a;
    auto h = c.to_host();
    for (size_t j = 0; j < want.size(); ++j) REQUIRE(sdr_amd::Elements::unpack<R>(h, want.size(), 0, j) == want[j]);
    // shift and its operators (SDR.hpp:461-468, 1579-1590; this file is built with the reference's asserts on, so its
    // shift adds): in place, and as shifted copies that leave the batch alone; a batch without a declared width may
    // go below zero, one with a width may not leave it
    {
        auto free_rows = sdr_amd::SDRBatch::pack(as.begin(), as.end(), 0, 0);
        auto up = free_rows << 1000, down = free_rows >> 200;
        free_rows <<= 7;
        free_rows >>= 50;
        auto h_up = up.to_host(), h_down = down.to_host(), h_in = free_rows.to_host();
        for (size_t j = 0; j < as.size(); ++j) {
            REQUIRE(sdr_amd::Elements::unpack<R>(h_up, as.size(), 0, j) == (as[j] << 1000));
            REQUIRE(sdr_amd::Elements::unpack<R>(h_down, as.size(), 0, j) == (as[j] >> 200));
            R moved(as[j]);
            moved <<= 7;
            moved >>= 50;
            REQUIRE(sdr_amd::Elements::unpack<R>(h_in, as.size(), 0, j) == moved);
        }
        // the shifted rows still compute: overlaps with rows shifted the same way are those of the rows themselves
        auto b_down = sdr_amd::SDRBatch::pack(bs.begin(), bs.end(), 0, 0) >> 200;
        auto c_down = (down && b_down).to_host();
        for (size_t i = 0; i < as.size(); ++i)
            for (size_t j = 0; j < bs.size(); ++j) REQUIRE(c_down[i * bs.size() + j] == (as[i] && bs[j]));
        bool refused = false;
        try { a <<= 1000; } catch (const sdr_amd::Error& e) { refused = e.code() == SDRB_ERR_ID_RANGE; }
        REQUIRE(refused);
        refused = false;
        try { up <<= 2147483000; } catch (const sdr_amd::Error& e) { refused = e.code() == SDRB_ERR_ID_RANGE; }
        REQUIRE(refused);
    }
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

// database files: SDR objects -> file -> SDR objects (reference operator==), and a batch loaded from
// a row range of the file gives the reference's counts
template <typename SDR_t>
int run_file(std::mt19937& rng, const char* name) {
    auto ds = make_rows<SDR_t>(rng, 500, 400, 40);
    const std::string path = std::string("/tmp/sdrb_cpp_test_") + name + ".sdrb";
    sdr_amd::save(ds.begin(), ds.end(), path, 400, 64);   // 8 chunks
    using R = SDR<typename SDR_t::value_type, std::vector<typename SDR_t::value_type>>;
    auto back = sdr_amd::load_rows<R>(path);
    REQUIRE(back.size() == ds.size());
    for (size_t r = 0; r < ds.size(); ++r) REQUIRE(back[r] == ds[r]);
    auto part = sdr_amd::load_rows<R>(path, 100, 130);
    for (size_t r = 0; r < part.size(); ++r) REQUIRE(part[r] == ds[100 + r]);
    auto shard = sdr_amd::SDRBatch::from_file(path, 0, 100, 400);
    auto q = sdr_amd::SDRBatch::pack(ds.begin(), ds.begin() + 20, 0, 400);
    if constexpr (std::is_same_v<typename SDR_t::value_type::data_type, ArithData<float>>) {
        auto dots = q.inner(shard).to_host();
        for (size_t i = 0; i < 20; ++i)
            for (size_t j = 0; j < 300; ++j)
                REQUIRE(dots[i * 300 + j] == ds[i].template inner<ArithData<float>>(ds[100 + j]).value());
    } else {
        auto counts = q.ands(shard).to_host();
        for (size_t i = 0; i < 20; ++i)
            for (size_t j = 0; j < 300; ++j) REQUIRE(counts[i * 300 + j] == ds[i].ands(ds[100 + j]));
    }
    std::remove(path.c_str());
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

int main() {
    std::setvbuf(stdout, nullptr, _IOLBF, 0);   // progress survives an assert of the reference
    std::mt19937 rng(12345);
    using E = SDRElem<>;
    using U = SDRElem<int, UnitData>;
    using AE = SDRElem<int, ArithData<float>>;
    using LE = SDRElem<long>;
    using LU = SDRElem<long, UnitData>;
    try {
        if (run<SDR<E>, SDR<E>>("vec x vec (table kernel)", rng, 512, 512)) return 1;
        if (run<SDR<E>, SDR<E>>("vec x vec (merge kernel)", rng, 512, 0)) return 1;
        if (run<SDR<E, std::set<E, std::less<>>>, SDR<E, std::forward_list<E>>>("set x flist", rng, 300, 300)) return 1;
        if (run<SDR<U>, SDR<U>>("unit x unit", rng, 200, 0)) return 1;
        if (run<SDR<U>, SDR<E>>("unit x empty", rng, 200, 200)) return 1;
        if (run<SDR<E>, SDR<U>>("empty x unit", rng, 200, 200)) return 1;
        if (run<SDR<U, IDContiguousContainer<U>>, SDR<U, IDContiguousContainer<U>>>("idcontiguous x idcontiguous", rng, 200, 0)) return 1;
        if (run<SDR<LE>, SDR<LE>>("long ids", rng, 512, 0)) return 1;
        if (run<SDR<LU>, SDR<LU>>("long ids, unit x unit", rng, 200, 0)) return 1;
        if (run<SDR<LU>, SDR<LE>>("long ids, unit x empty", rng, 200, 0)) return 1;
        // ids of different widths in one call, the pairing the reference's fuzzer runs (src/fuzz.cpp:413-423)
        g_small_long = true;
        if (run<SDR<LU>, SDR<U>>("long x int ids (all fit)", rng, 200, 0)) return 1;
        if (run<SDR<U>, SDR<LU>>("int x long ids (all fit)", rng, 200, 0)) return 1;
        if (run<SDR<LE>, SDR<E>>("long x int ids, empty", rng, 512, 0)) return 1;
        if (run<SDR<LE>, SDR<U>>("long x int ids, e x u", rng, 200, 0)) return 1;
        // (long ids BEYOND 32 bits against int ids are not compared here: the reference's search hands the long id to
        // SDRElem<int>::operator<(int), SDRElem.hpp:73-75, which truncates it; the engine compares ids by value)
        g_small_long = false;
        // ids no table indexes as they are: close together (moved to zero for the call), far apart (searched)
        g_id_mode = 1;
        if (run<SDR<LE>, SDR<LE>>("long ids near 2^40", rng, 512, 0)) return 1;
        if (run<SDR<LU>, SDR<LE>>("long ids near 2^40, u x e", rng, 200, 0)) return 1;
        if (run<SDR<E>, SDR<E>>("int ids from -1700", rng, 512, 0)) return 1;
        if (run<SDR<U>, SDR<U>>("int ids from -1700, unit", rng, 200, 0)) return 1;
        if (run<SDR<E>, SDR<U>>("int ids from -1700, e x u", rng, 200, 0)) return 1;
        g_id_mode = 2;
        if (run<SDR<E>, SDR<E>>("int ids far apart", rng, 512, 0)) return 1;
        if (run<SDR<U>, SDR<U>>("int ids far apart, unit", rng, 200, 0)) return 1;
        if (run<SDR<U>, SDR<E>>("int ids far apart, u x e", rng, 200, 0)) return 1;
        g_id_mode = 0;
        if (run_few<SDR<E>>("few rows, ande", rng, 512, 40, false)) return 1;
        if (run_few<SDR<U>>("few rows, ande, unit", rng, 300, 40, false)) return 1;
        if (run_few<SDR<U>>("few rows, ande, refill", rng, 64, 40, true)) return 1;
        if (run_retarget<SDR<U>, SDR<U>, E, true>("unit x unit -> empty", rng, SDRB_DATA_EMPTY)) return 1;
        if (run_retarget<SDR<U>, SDR<E>, E, true>("unit x empty -> empty", rng, SDRB_DATA_EMPTY)) return 1;
        if (run_retarget<SDR<E>, SDR<U>, U, false>("empty x unit -> unit", rng, SDRB_DATA_UNIT)) return 1;
        if (run_retarget<SDR<E>, SDR<E>, U, false>("empty x empty -> unit", rng, SDRB_DATA_UNIT)) return 1;
        if (run_rmi_combiner<SDR<U>, SDR<U>>("rmi combiner unit x unit", rng)) return 1;
        if (run_rmi_combiner<SDR<U>, SDR<E>>("rmi combiner unit x empty", rng)) return 1;
        if (run_inner<SDR<AE>>(rng, "inner arith x arith")) return 1;
        if (run_inner<SDR<E>>(rng, "inner arith x empty")) return 1;
        if (run_inplace<SDR<E>, SDR<E>>(rng, "in-place empty x empty")) return 1;
        if (run_inplace<SDR<U>, SDR<U>>(rng, "in-place unit x unit")) return 1;
        if (run_inplace<SDR<U>, SDR<E>>(rng, "in-place unit x empty")) return 1;
        if (run_queries(rng, "range/point/transpose")) return 1;
        if (run_operators(rng, "operator sugar")) return 1;
        if (run_visitors(rng, "andv / orv visitors")) return 1;
        if (run_arith<SDR<AE>, SDR<AE>>(rng, "arith x arith")) return 1;
        if (run_arith<SDR<AE>, SDR<E>>(rng, "arith x empty")) return 1;
        if (run_arith<SDR<E>, SDR<AE>>(rng, "empty x arith")) return 1;
        if (run_file<SDR<E>>(rng, "file_empty")) return 1;
        if (run_file<SDR<U>>(rng, "file_unit")) return 1;
        if (run_file<SDR<AE>>(rng, "file_arith")) return 1;
        if (run_allpairs(rng, 0, "all-pairs top-k (default)")) return 1;
        if (run_allpairs(rng, SDRB_FLAG_FORCE_MFMA, "all-pairs top-k (matrix cores)")) return 1;
    } catch (const sdr_amd::Error& e) {
        std::printf("FAIL sdr_amd::Error %d: %s\n", e.code(), e.what());
        return 1;
    }
    std::printf("PASS %ld checks\n", g_checks);
    return 0;
}
