// Drop-in check of include/sdr_amd/SDRBatch.hpp against the REAL reference: SDR objects are built
// with the reference's own classes and containers, packed, evaluated on the GPU, and every pair
// is compared with the reference's per-pair call using the reference's operator== (ids and
// payloads, SDR.hpp:470-483).  Built only where /root/reference exists (oracle/Makefile ->
// oracle/_ref/test_sdrbatch); the GPU test-suite runs the binary when it is present.
#include <cstdio>
#include <forward_list>
#include <random>
#include <set>
#include <vector>

#include "SparseDistributedRepresentation/SDR.hpp"
#include "SparseDistributedRepresentation/DataTypes/UnitData.hpp"
#include "SparseDistributedRepresentation/IDContiguousContainer.hpp"
#include "sdr_amd/SDRBatch.hpp"

using namespace sparse_distributed_representation;

static long g_checks = 0;
#define REQUIRE(x) do { ++g_checks; if (!(x)) { std::printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #x); return 1; } } while (0)

template <typename SDR_t>
std::vector<SDR_t> make_rows(std::mt19937& rng, size_t n, int width, int max_len) {
    using Elem = typename SDR_t::value_type;
    using Data = typename Elem::data_type;
    std::vector<SDR_t> rows(n);
    std::uniform_real_distribution<float> val(0.0f, 1.0f);
    for (auto& r : rows) {
        int len = (int)(rng() % (unsigned)(max_len + 1));
        std::set<int> ids;
        while ((int)ids.size() < len) ids.insert((int)(rng() % (unsigned)width));
        std::vector<Elem> elems;
        for (int id : ids) {
            if constexpr (std::is_same_v<Data, UnitData>) elems.push_back(Elem(id, UnitData(val(rng))));
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
    std::printf("ok %-28s %ld checks so far\n", name, g_checks);
    return 0;
}

int main() {
    std::mt19937 rng(12345);
    using E = SDRElem<>;
    using U = SDRElem<int, UnitData>;
    try {
        if (run<SDR<E>, SDR<E>>("vec x vec (table kernel)", rng, 512, 512)) return 1;
        if (run<SDR<E>, SDR<E>>("vec x vec (merge kernel)", rng, 512, 0)) return 1;
        if (run<SDR<E, std::set<E, std::less<>>>, SDR<E, std::forward_list<E>>>("set x flist", rng, 300, 300)) return 1;
        if (run<SDR<U>, SDR<U>>("unit x unit", rng, 200, 0)) return 1;
        if (run<SDR<U>, SDR<E>>("unit x empty", rng, 200, 200)) return 1;
        if (run<SDR<E>, SDR<U>>("empty x unit", rng, 200, 200)) return 1;
        if (run<SDR<U, IDContiguousContainer<U>>, SDR<U, IDContiguousContainer<U>>>("idcontiguous x idcontiguous", rng, 200, 0)) return 1;
    } catch (const sdr_amd::Error& e) {
        std::printf("FAIL sdr_amd::Error %d: %s\n", e.code(), e.what());
        return 1;
    }
    std::printf("PASS %ld checks\n", g_checks);
    return 0;
}
