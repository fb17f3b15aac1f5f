// host_test.cpp -- the reference's own unit tests for the hot-path searchers, restated
// against the C++ mirror (scann.hpp) so they run through libscann_hip.so on the GPU.
//   brute_force/searcher.rs:280-376, tests/unit_tests.rs:204-259
//   hashes/hasher.rs:322-380
//   tree_x_hybrid/mod.rs:436-468
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdio>
#include <string>
#include <unistd.h>

#include "scann.hpp"

using namespace scann;

static int g_fail = 0;
#define EXPECT(cond)                                                         \
    do {                                                                     \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
    } while (0)

static DenseDataset cube() {
    return DenseDataset::from_vecs({{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 1}});
}

static void sorted(const NNResultsVector &r) {
    for (size_t i = 1; i < r.size(); ++i) EXPECT(r[i].second >= r[i - 1].second);
}

static void brute_force_tests() {
    BruteForceSearcher s(cube(), DistanceMeasure::SquaredL2);
    auto r = s.search({0, 0, 0}, 3);                      // test_brute_force_search
    EXPECT(r.size() == 3 && r[0].first == 0 && std::fabs(r[0].second) < 1e-6);
    r = s.search({0.5f, 0.5f, 0.5f}, 5);                  // test_brute_force_search_all
    EXPECT(r.size() == 5);
    sorted(r);
    auto b = s.search_batched({{0, 0, 0}, {1, 1, 1}}, 2); // test_brute_force_batched
    EXPECT(b.size() == 2 && b[0].size() == 2 && b[1].size() == 2);
    auto rr = s.search_radius({0, 0, 0}, 1.5f);           // test_brute_force_radius: origin + 3 unit points
    EXPECT(rr.size() == 4 && rr[0].first == 0);
    sorted(rr);
    bool threw = false;                                   // test_brute_force_dimension_mismatch
    try { s.search({1, 2}, 5); } catch (const ScannError &e) { threw = e.code == ErrorCode::InvalidArgument; }
    EXPECT(threw);
    BruteForceSearcher empty(DenseDataset(), DistanceMeasure::SquaredL2);   // test_brute_force_empty_dataset
    EXPECT(empty.search({1, 2, 3}, 5).empty());
    BruteForceSearcher dot(DenseDataset::from_vecs({{1, 0}, {0, 1}, {1, 1}}), DistanceMeasure::DotProduct);
    r = dot.search({1, 0}, 3);                            // test_brute_force_search_dot_product
    EXPECT(r.size() == 3 && r[0].second <= r[1].second && r[0].second == -1.0f && r[2].second == 0.0f);
}

static DenseDataset sin_dataset(size_t n, size_t dim) {  // hasher.rs:326-336, tree_x_hybrid/mod.rs:427-434
    std::vector<std::vector<float>> v(n, std::vector<float>(dim));
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < dim; ++j) v[i][j] = std::sin((float)(i * j) / 100.0f);
    return DenseDataset::from_vecs(v);
}

static void hasher_tests() {
    AsymmetricHasher h(AsymmetricHasherConfig(16, 8).with_seed(42));
    h.build(sin_dataset(200, 32));                        // test_asymmetric_hasher_build
    EXPECT(h.num_datapoints() == 200 && h.dimensionality() == 32);
    EXPECT(h.encoded_database().size() == 200 * 8);
    std::vector<float> q(32, 0.5f);
    auto r = h.search(q, 10);                             // test_asymmetric_hasher_search
    EXPECT(r.size() == 10);
    sorted(r);
    r = h.search_with_reordering(q, 10, 50);              // test_asymmetric_hasher_reordering
    EXPECT(r.size() == 10);
    sorted(r);
    AsymmetricHasher ns(AsymmetricHasherConfig(16, 8));
    ns.build_no_store(sin_dataset(200, 32));
    bool threw = false;
    try { ns.search_with_reordering(q, 10, 50); }
    catch (const ScannError &e) { threw = e.code == ErrorCode::FailedPrecondition; }
    EXPECT(threw);
    bool bad = false;                                     // codebook.rs:154-159
    try { AsymmetricHasher x(AsymmetricHasherConfig(16, 24)); x.build(sin_dataset(50, 32)); }
    catch (const ScannError &e) { bad = e.code == ErrorCode::InvalidArgument; }
    EXPECT(bad);
}

static void tree_x_hybrid_tests() {
    auto cfg = TreeXHybridConfig(10, 3).with_hash(AsymmetricHasherConfig(16, 8).with_seed(42)).with_residuals(true);
    TreeXHybridSearcher s(cfg);
    s.build(sin_dataset(500, 32));                        // test_tree_x_hybrid_build
    EXPECT(s.num_partitions() == 10 && s.num_datapoints() == 500);
    std::vector<float> q(32);
    for (int i = 0; i < 32; ++i) q[i] = std::sin((float)i / 10.0f);
    auto r = s.search(q, 10);                             // test_tree_x_hybrid_search
    EXPECT(r.size() == 10);
    sorted(r);
    auto b = s.search_batched({q, q}, 10);
    EXPECT(b.size() == 2 && b[0] == r && b[1] == r);
    // Searcher::search_batched_with_params (mod.rs:399-409): one num_neighbors per query == single searches
    auto bp = s.search_batched_with_params({q, q, q}, {10, 3, 1});
    EXPECT(bp.size() == 3 && bp[0] == r && bp[1] == s.search(q, 3) && bp[2] == s.search(q, 1));
    bool threw = false;
    try { s.search({1, 2, 3}, 5); } catch (const ScannError &e) { threw = e.code == ErrorCode::InvalidArgument; }
    EXPECT(threw);
    // search_with_filter (mod.rs:245-294) with restricts/allowlist.rs filters
    std::vector<DatapointIndex> even;
    for (DatapointIndex i = 0; i < 500; i += 2) even.push_back(i);
    auto allow = RestrictAllowlist::from_indices(even, 500);
    EXPECT(allow.num_allowed() == 250 && allow.is_allowed(4) && !allow.is_allowed(5) && !allow.is_allowed(500));
    auto fr = s.search_with_filter(q, 10, &allow);
    EXPECT(fr.size() == 10);
    sorted(fr);
    for (auto &p : fr) EXPECT(p.first % 2 == 0);
    auto deny = RestrictDenylist::from_indices({r[0].first, r[1].first}, 500);
    auto dr = s.search_with_filter(q, 10, &deny);
    EXPECT(dr.size() == 10);
    for (auto &p : dr) EXPECT(p.first != r[0].first && p.first != r[1].first);
    EXPECT(s.search_with_filter(q, 10, nullptr) == r);
    auto two = RestrictAllowlist::from_indices({7, 11}, 500);   // fewer allowed than k
    auto tr = s.search_with_filter(q, 10, &two);
    EXPECT(tr.size() <= 2);
    for (auto &p : tr) EXPECT(p.first == 7 || p.first == 11);
    {   // index file round trip: the loaded searcher answers exactly like the built one
        const std::string path = "/tmp/scann_host_test_" + std::to_string((long)getpid()) + ".scannidx";
        s.save(path);
        auto loaded = TreeXHybridSearcher::load(path);
        EXPECT(loaded.num_partitions() == 10 && loaded.num_datapoints() == 500 && loaded.dimensionality() == 32);
        EXPECT(loaded.search(q, 10) == r);
        EXPECT(loaded.search_with_filter(q, 10, &allow) == fr);
        std::remove(path.c_str());
        bool missing = false;
        try { TreeXHybridSearcher::load(path); } catch (const ScannError &e) { missing = e.code == ErrorCode::NotFound; }
        EXPECT(missing);
    }
    // TreeXHybridConfig::default(): 100 partitions, search 10, 256 codes x 8 subspaces (mod.rs:37-48)
    TreeXHybridSearcher dflt{TreeXHybridConfig()};
    dflt.build(sin_dataset(3000, 32));
    EXPECT(dflt.num_partitions() == 100);
    auto dr2 = dflt.search(q, 10);
    EXPECT(dr2.size() == 10);
    sorted(dr2);
    bool empty = false;
    try { TreeXHybridSearcher e(cfg); e.build(DenseDataset()); }
    catch (const ScannError &e) { empty = e.code == ErrorCode::InvalidArgument; }
    EXPECT(empty);
}

static DenseDataset scann_test_dataset() {   // scann.rs:438-448
    std::vector<std::vector<float>> v(100, std::vector<float>(16));
    for (int i = 0; i < 100; ++i)
        for (int j = 0; j < 16; ++j) v[i][j] = std::sin((float)i + (float)j * 0.1f);
    return DenseDataset::from_vecs(v);
}

static void scann_facade_tests() {   // scann.rs:450-496 + the four search modes (:181-294)
    auto bf = Scann::brute_force(scann_test_dataset());   // test_scann_brute_force
    EXPECT(bf.search_mode() == SearchMode::BruteForce && bf.size() == 100);
    EXPECT(bf.search(std::vector<float>(16, 0.5f), 10).size() == 10);
    auto b = ScannBuilder().num_neighbors(5).distance_measure(DistanceMeasure::SquaredL2).brute_force()
                 .build(scann_test_dataset());            // test_scann_builder
    EXPECT(b.config().num_neighbors == 5 && b.distance_measure() == DistanceMeasure::SquaredL2);
    auto br = bf.search_batched({std::vector<float>(16, 0.5f), std::vector<float>(16, 0.3f),
                                 std::vector<float>(16, 0.7f)}, 5);   // test_scann_batched
    EXPECT(br.size() == 3);
    for (auto &r : br) EXPECT(r.size() == 5);
    EXPECT(ScannBuilder().num_neighbors(3).brute_force().build(cube()).search({0, 0, 0}, 3).size() == 3);

    const std::vector<float> q(32, 0.25f);
    auto exact = Scann::brute_force(sin_dataset(400, 32)).search(q, 400);
    // Partitioned: every partition searched == brute force over the same rows (same kernel maths)
    auto part = Scann::partitioned(sin_dataset(400, 32), 8, 8);
    EXPECT(part.search_mode() == SearchMode::Partitioned);
    auto pr = part.search(q, 10);
    EXPECT(pr.size() == 10);
    sorted(pr);
    for (size_t i = 0; i < pr.size(); ++i) EXPECT(pr[i].second == exact[i].second);
    auto few = Scann::partitioned(sin_dataset(400, 32), 8, 2).search(q, 10);
    EXPECT(few.size() == 10 && few[0].second >= exact[0].second);
    sorted(few);
    // DotProduct through the builder: distances are negated dots
    auto dp = ScannBuilder().distance_measure(DistanceMeasure::DotProduct).tree(8, 8).build(sin_dataset(400, 32));
    auto dexact = ScannBuilder().distance_measure(DistanceMeasure::DotProduct).brute_force().build(sin_dataset(400, 32));
    auto d1 = dp.search(q, 5), d2 = dexact.search(q, 5);
    for (size_t i = 0; i < 5; ++i) EXPECT(d1[i].second == d2[i].second);
    // Hashed (256 buckets x num_blocks) and TreeAH, without and with the exact reordering
    auto hashed = Scann::hashed(sin_dataset(400, 32), 8);
    EXPECT(hashed.search_mode() == SearchMode::Hashed);
    auto hr = hashed.search(q, 10);
    EXPECT(hr.size() == 10);
    sorted(hr);
    auto t = ScannBuilder().num_neighbors(10).tree(8, 4).hash(8).build(sin_dataset(400, 32));
    EXPECT(t.search_mode() == SearchMode::TreeAH);
    auto r = t.search(q, 10);
    EXPECT(r.size() == 10);
    sorted(r);
    auto tr = ScannBuilder().num_neighbors(10).tree(8, 4).hash(8).reorder(40).build(sin_dataset(400, 32));
    auto rr = tr.search(q, 10);
    EXPECT(rr.size() == 10);
    sorted(rr);
    {   // the reordering re-scores the SAME k rows exactly (scann.rs:199-209)
        std::vector<uint32_t> a, c;
        for (auto &p : r) a.push_back(p.first);
        for (auto &p : rr) c.push_back(p.first);
        std::sort(a.begin(), a.end());
        std::sort(c.begin(), c.end());
        EXPECT(a == c);
        for (auto &p : rr) {
            bool found = false;
            for (auto &e : exact) if (e.first == p.first) { found = e.second == p.second; break; }
            EXPECT(found);
        }
    }
    auto tb = t.search_batched({q, q}, 10);
    EXPECT(tb.size() == 2 && tb[0] == r && tb[1] == r);
    bool threw = false;
    try { ScannBuilder().build(DenseDataset()); } catch (const ScannError &e) { threw = e.code == ErrorCode::InvalidArgument; }
    EXPECT(threw);
}

// tests/stress_tests.rs:256-297 (stress_test_concurrent_queries) and :300-323 (high-dimensional)
static void stress_tests() {
    std::vector<std::vector<float>> v(5000, std::vector<float>(32));
    uint64_t st = 42;
    for (auto &r : v)
        for (auto &x : r) x = (float)(detail::splitmix(st) >> 40) * (1.0f / 16777216.0f);
    const Scann scann = Scann::brute_force(DenseDataset::from_vecs(v));
    auto run = [&](int threads) {
        std::vector<std::thread> pool;
        std::atomic<int> bad{0};
        const auto t0 = std::chrono::steady_clock::now();
        for (int t = 0; t < threads; ++t)
            pool.emplace_back([&, t]() {
                uint64_t s2 = 100 + (uint64_t)t;
                std::vector<float> q(32);
                for (int i = 0; i < 50; ++i) {
                    for (auto &x : q) x = (float)(detail::splitmix(s2) >> 40) * (1.0f / 16777216.0f);
                    auto r = scann.search(q, 10);
                    if (r.size() != 10) ++bad;
                    for (size_t j = 1; j < r.size(); ++j)
                        if (r[j].second < r[j - 1].second) ++bad;
                }
            });
        for (auto &th : pool) th.join();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        EXPECT(bad.load() == 0);
        return threads * 50 / sec;
    };
    run(1);   // warm-up (workspaces of the first slot)
    const double q1 = run(1), q4 = run(4);
    run(4);
    std::printf("concurrent queries: 1 thread %.0f QPS, 4 threads %.0f QPS\n", q1, q4);

    std::vector<std::vector<float>> hd(1000, std::vector<float>(512));
    for (auto &r : hd)
        for (auto &x : r) x = (float)(detail::splitmix(st) >> 40) * (1.0f / 16777216.0f);
    BruteForceSearcher hs(DenseDataset::from_vecs(hd), DistanceMeasure::SquaredL2);
    for (int i = 0; i < 20; ++i) {
        auto r = hs.search(hd[(size_t)i * 7], 10);
        EXPECT(r.size() == 10 && r[0].first == (uint32_t)(i * 7) && r[0].second == 0.0f);
        sorted(r);
    }
}

// quantization/fp8.rs tests (:278-343) through the C++ mirror
static void fp8_tests() {
    const float e4[] = {0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 100.0f, -0.1f};        // test_fp8_e4m3_roundtrip
    {
        Fp8Quantizer qz = Fp8Quantizer::e4m3();
        std::vector<float> v(e4, e4 + 7);
        const std::vector<float> r = qz.dequantize(qz.quantize(v));
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] != 0.0f) EXPECT(std::fabs((v[i] - r[i]) / v[i]) < 0.2f);
    }
    const float e5[] = {0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 4.0f, -0.125f};        // test_fp8_e5m2_roundtrip
    {
        Fp8Quantizer qz = Fp8Quantizer::e5m2();
        std::vector<float> v(e5, e5 + 7);
        const std::vector<float> r = qz.dequantize(qz.quantize(v));
        for (size_t i = 0; i < v.size(); ++i)
            if (std::fabs(v[i]) > 1e-6f) EXPECT(std::fabs((v[i] - r[i]) / v[i]) < 0.5f);
    }
    {                                                                            // test_fp8_quantizer
        Fp8Quantizer qz = Fp8Quantizer::e4m3();
        std::vector<float> v = {1.0f, 2.0f, 3.0f, 4.0f};
        const std::vector<uint8_t> b = qz.quantize(v);
        const std::vector<float> r = qz.dequantize(b);
        for (size_t i = 0; i < v.size(); ++i) EXPECT(std::fabs(v[i] - r[i]) < 0.5f);
        // test_simd_operations: dot and squared L2 of the query against its own FP8 image
        const float dot = -one_to_many_fp8(v, b, 4, 1, DistanceMeasure::DotProduct)[0];
        EXPECT(std::fabs(dot - 30.0f) < 1.0f);
        EXPECT(one_to_many_fp8(v, b, 4, 1, DistanceMeasure::SquaredL2)[0] < 1.0f);
        qz.calibrate_scale(2.0f);
        EXPECT(qz.scale() == 224.0f && qz.bits() == 8 && qz.format() == Fp8Format::E4M3);
    }
}

int main() {
    try {
        fp8_tests();
        brute_force_tests();
        hasher_tests();
        tree_x_hybrid_tests();
        scann_facade_tests();
        stress_tests();
    } catch (const ScannError &e) {
        std::printf("ScannError %d: %s\n", (int)e.code, e.what());
        return 2;
    }
    if (g_fail) return 1;
    std::printf("host_test ok\n");
    return 0;
}
