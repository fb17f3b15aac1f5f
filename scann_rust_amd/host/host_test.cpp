// host_test.cpp -- the reference's own unit tests for the hot-path searchers, restated
// against the C++ mirror (scann.hpp) so they run through libscann_hip.so on the GPU.
//   brute_force/searcher.rs:280-376, tests/unit_tests.rs:204-259
//   hashes/hasher.rs:322-380
//   tree_x_hybrid/mod.rs:436-468
#include <cmath>
#include <cstdio>

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

static void builder_tests() {   // scann.rs:364-426
    auto bf = ScannBuilder().num_neighbors(3).brute_force().build(cube());
    EXPECT(bf.search({0, 0, 0}, 3).size() == 3);
    auto t = ScannBuilder().num_neighbors(10).tree(8, 4).hash(8).reorder(40).build(sin_dataset(400, 32));
    auto r = t.search(std::vector<float>(32, 0.25f), 10);
    EXPECT(r.size() == 10);
    sorted(r);
    bool threw = false;
    try { ScannBuilder().build(DenseDataset()); } catch (const ScannError &e) { threw = e.code == ErrorCode::InvalidArgument; }
    EXPECT(threw);
}

int main() {
    try {
        brute_force_tests();
        hasher_tests();
        tree_x_hybrid_tests();
        builder_tests();
    } catch (const ScannError &e) {
        std::printf("ScannError %d: %s\n", (int)e.code, e.what());
        return 2;
    }
    if (g_fail) return 1;
    std::printf("host_test ok\n");
    return 0;
}
