/*
 * scann_oracle.c -- CPU restatement of the reference hot path.
 * TEST INFRASTRUCTURE ONLY (see scann_oracle.h for the rules and the parity pin).
 *
 * Build: gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp -fPIC -shared
 *        (-ffp-contract=off is REQUIRED: Rust never contracts a*b+c; FMA appears
 *         only where simd/x86.rs calls _mm256_fmadd_ps explicitly.)
 */
#include "scann_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OR_ERR_INVALID_ARGUMENT (-3)

int or_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* L1 kernels                                                               */
/* ------------------------------------------------------------------------ */

/* simd/x86.rs:31-44 horizontal_sum_f32_avx2 */
static inline float hsum256(__m256 v) {
    __m128 hi = _mm256_extractf128_ps(v, 1);
    __m128 lo = _mm256_castps256_ps128(v);
    __m128 sum128 = _mm_add_ps(lo, hi);
    __m128 shuf = _mm_movehdup_ps(sum128);
    __m128 sums = _mm_add_ps(sum128, shuf);
    shuf = _mm_movehl_ps(sums, sums);
    sums = _mm_add_ss(sums, shuf);
    return _mm_cvtss_f32(sums);
}

/* simd/x86.rs:139-165 */
float or_squared_l2_avx2(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    __m256 sum = _mm256_setzero_ps();
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8);
        __m256 vb = _mm256_loadu_ps(b + i * 8);
        __m256 diff = _mm256_sub_ps(va, vb);
        sum = _mm256_fmadd_ps(diff, diff, sum);
    }
    float result = hsum256(sum);
    for (size_t i = n - rem; i < n; ++i) {
        float diff = a[i] - b[i];
        result += diff * diff; /* not fused: -ffp-contract=off */
    }
    return result;
}

/* simd/x86.rs:72-96 */
float or_dot_product_avx2(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    __m256 sum = _mm256_setzero_ps();
    for (size_t i = 0; i < chunks; ++i) {
        __m256 va = _mm256_loadu_ps(a + i * 8);
        __m256 vb = _mm256_loadu_ps(b + i * 8);
        sum = _mm256_fmadd_ps(va, vb, sum);
    }
    float result = hsum256(sum);
    for (size_t i = n - rem; i < n; ++i) result += a[i] * b[i];
    return result;
}

/* wide::f32x8::reduce_add as used by the portable fallback
 * (simd/dispatch.rs:149,174).  `wide` 0.7 on x86-64/AVX reduces exactly like
 * the hsum above; on other targets it is a pairwise tree.  The portable path
 * is never taken on AVX2+FMA hosts (dispatch.rs:84-91,116-123), so this
 * function is provided for completeness and pinned only by tolerance tests. */
float or_squared_l2_portable(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) {
            float d = a[i * 8 + j] - b[i * 8 + j];
            float p = d * d;
            lane[j] = lane[j] + p;
        }
    float s0 = lane[0] + lane[4], s1 = lane[1] + lane[5];
    float s2 = lane[2] + lane[6], s3 = lane[3] + lane[7];
    float result = (s0 + s1) + (s2 + s3);
    for (size_t i = n - rem; i < n; ++i) {
        float d = a[i] - b[i];
        result += d * d;
    }
    return result;
}

float or_dot_product_portable(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8, rem = n % 8;
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) {
            float p = a[i * 8 + j] * b[i * 8 + j];
            lane[j] = lane[j] + p;
        }
    float s0 = lane[0] + lane[4], s1 = lane[1] + lane[5];
    float s2 = lane[2] + lane[6], s3 = lane[3] + lane[7];
    float result = (s0 + s1) + (s2 + s3);
    for (size_t i = n - rem; i < n; ++i) result += a[i] * b[i];
    return result;
}

/* partitioning/tree_partitioner.rs:184-192; hashes/codebook.rs:107-115;
 * hashes/lut16.rs:247-255; bin/ann_benchmark.rs:442-450.
 * Iterator::sum::<f32>() folds left from 0.0 in element order. */
float or_squared_l2_sequential(const float *a, const float *b, size_t n) {
    float sum = 0.0f;
    for (size_t i = 0; i < n; ++i) {
        float d = a[i] - b[i];
        float p = d * d;
        sum = sum + p;
    }
    return sum;
}

/* simd/x86.rs:267-346.  The 4-point batching only interleaves independent
 * points; per point the arithmetic is squared_l2_avx2 with diff = q - db. */
void or_one_to_many_squared_l2(const float *q, size_t dim, const float *db,
                               size_t stride, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = or_squared_l2_avx2(q, db + i * stride, dim);
}

/* simd/x86.rs:195-258: per point dot_product_avx2(q, x), stored negated. */
void or_one_to_many_dot_product(const float *q, size_t dim, const float *db,
                                size_t stride, size_t n, float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = -or_dot_product_avx2(q, db + i * stride, dim);
}

/* data_format/dataset.rs:90-96 */
size_t or_compute_stride(size_t dim) {
    const size_t per_line = 64 / sizeof(float);
    return (dim + per_line - 1) / per_line * per_line;
}

/* ------------------------------------------------------------------------ */
/* ordering helpers                                                         */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t idx;
    float dist;
} pair_t;

/* a.1.partial_cmp(&b.1).unwrap_or(Equal): NaN compares Equal to everything. */
static inline int partial_less(float a, float b) { return a < b; }

/* OrderedFloat total order: NaN is greater than every number and equal to
 * itself; -0.0 == +0.0 (ordered-float 4.x, Ord impl). */
static inline int ordered_cmp(float a, float b) {
    int an = isnan(a), bn = isnan(b);
    if (an || bn) return an - bn; /* NaN > number; NaN == NaN */
    return (a > b) - (a < b);
}

/* Stable merge sort by dist using partial_cmp semantics (slice::sort_by is
 * stable; any stable sort yields the same permutation). */
static void stable_sort_pairs(pair_t *v, size_t n, pair_t *tmp) {
    if (n < 2) return;
    if (n <= 16) { /* insertion sort is stable */
        for (size_t i = 1; i < n; ++i) {
            pair_t x = v[i];
            size_t j = i;
            while (j > 0 && partial_less(x.dist, v[j - 1].dist)) {
                v[j] = v[j - 1];
                --j;
            }
            v[j] = x;
        }
        return;
    }
    size_t mid = n / 2;
    stable_sort_pairs(v, mid, tmp);
    stable_sort_pairs(v + mid, n - mid, tmp);
    size_t i = 0, j = mid, o = 0;
    while (i < mid && j < n) {
        if (partial_less(v[j].dist, v[i].dist)) tmp[o++] = v[j++];
        else tmp[o++] = v[i++];
    }
    while (i < mid) tmp[o++] = v[i++];
    while (j < n) tmp[o++] = v[j++];
    memcpy(v, tmp, n * sizeof(pair_t));
}

static void sort_pairs(pair_t *v, size_t n) {
    if (n < 2) return;
    pair_t *tmp = (pair_t *)malloc(n * sizeof(pair_t));
    stable_sort_pairs(v, n, tmp);
    free(tmp);
}

/* Stable sort by OrderedFloat key (sort_by_key, tree_partitioner.rs:212). */
static void stable_sort_pairs_ordered(pair_t *v, size_t n, pair_t *tmp) {
    if (n < 2) return;
    if (n <= 16) {
        for (size_t i = 1; i < n; ++i) {
            pair_t x = v[i];
            size_t j = i;
            while (j > 0 && ordered_cmp(x.dist, v[j - 1].dist) < 0) {
                v[j] = v[j - 1];
                --j;
            }
            v[j] = x;
        }
        return;
    }
    size_t mid = n / 2;
    stable_sort_pairs_ordered(v, mid, tmp);
    stable_sort_pairs_ordered(v + mid, n - mid, tmp);
    size_t i = 0, j = mid, o = 0;
    while (i < mid && j < n) {
        if (ordered_cmp(v[j].dist, v[i].dist) < 0) tmp[o++] = v[j++];
        else tmp[o++] = v[i++];
    }
    while (i < mid) tmp[o++] = v[i++];
    while (j < n) tmp[o++] = v[j++];
    memcpy(v, tmp, n * sizeof(pair_t));
}

/* ------------------------------------------------------------------------ */
/* TopK: brute_force/top_k.rs:20-113 over std::collections::BinaryHeap       */
/* ------------------------------------------------------------------------ */

/* (OrderedFloat<f32>, u32) tuple ordering */
static inline int heap_le(pair_t a, pair_t b) { /* a <= b */
    int c = ordered_cmp(a.dist, b.dist);
    if (c != 0) return c < 0;
    return a.idx <= b.idx;
}

typedef struct {
    pair_t *data;
    size_t len, k;
} topk_t;

/* BinaryHeap::sift_up(start,pos) (alloc::collections::binary_heap) */
static size_t heap_sift_up(pair_t *d, size_t start, size_t pos) {
    pair_t elt = d[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (heap_le(elt, d[parent])) break;
        d[pos] = d[parent];
        pos = parent;
    }
    d[pos] = elt;
    return pos;
}

/* BinaryHeap::sift_down_to_bottom(0) */
static void heap_sift_down_to_bottom(pair_t *d, size_t end) {
    size_t pos = 0, start = 0;
    pair_t elt = d[pos];
    size_t child = 1;
    size_t limit = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
    while (child <= limit && end >= 2) {
        if (heap_le(d[child], d[child + 1])) child += 1;
        d[pos] = d[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (end >= 1 && child == end - 1) {
        d[pos] = d[child];
        pos = child;
    }
    d[pos] = elt;
    heap_sift_up(d, start, pos);
}

static void topk_init(topk_t *t, size_t k) {
    t->data = (pair_t *)malloc((k + 1) * sizeof(pair_t));
    t->len = 0;
    t->k = k;
}

static void heap_push(topk_t *t, pair_t p) {
    t->data[t->len] = p;
    heap_sift_up(t->data, 0, t->len);
    t->len++;
}

static void heap_pop(topk_t *t) {
    /* data.pop().map(|mut item| { if !empty { swap(item, data[0]); sift_down_to_bottom(0) } }) */
    t->len--;
    if (t->len > 0) {
        t->data[0] = t->data[t->len];
        heap_sift_down_to_bottom(t->data, t->len);
    }
}

/* TopK::push top_k.rs:66-81 */
static inline void topk_push(topk_t *t, uint32_t idx, float dist) {
    pair_t p = {idx, dist};
    if (t->len < t->k) {
        heap_push(t, p);
    } else if (t->len > 0) { /* heap.peek() is Some */
        if (dist < t->data[0].dist) {
            heap_pop(t);
            heap_push(t, p);
        }
    }
}

/* TopK::drain_sorted top_k.rs:105-112: drain() yields the backing vec in
 * order; then a stable sort by distance only. */
static size_t topk_drain_sorted(topk_t *t, uint32_t *out_idx, float *out_dist) {
    sort_pairs(t->data, t->len);
    size_t n = t->len;
    for (size_t i = 0; i < n; ++i) {
        out_idx[i] = t->data[i].idx;
        out_dist[i] = t->data[i].dist;
    }
    t->len = 0;
    return n;
}

size_t or_topk_run(size_t k, const uint32_t *idx, const float *dist, size_t n,
                   uint32_t *out_idx, float *out_dist) {
    topk_t t;
    topk_init(&t, k);
    for (size_t i = 0; i < n; ++i) topk_push(&t, idx[i], dist[i]);
    size_t r = topk_drain_sorted(&t, out_idx, out_dist);
    free(t.data);
    return r;
}

/* ------------------------------------------------------------------------ */
/* FastTopNeighbors: brute_force/top_k.rs:264-393                           */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t *indices;
    float *distances;
    size_t size, capacity;
} ftn_t;

static void ftn_init(ftn_t *f, size_t cap) {
    f->indices = (uint32_t *)calloc(cap ? cap : 1, sizeof(uint32_t));
    f->distances = (float *)malloc((cap ? cap : 1) * sizeof(float));
    for (size_t i = 0; i < cap; ++i) f->distances[i] = INFINITY;
    f->size = 0;
    f->capacity = cap;
}
static void ftn_free(ftn_t *f) {
    free(f->indices);
    free(f->distances);
}

/* push :333-355.  NOTE: with capacity 0 the reference indexes distances[0]
 * on an empty Vec and panics; the oracle treats capacity 0 as "keep nothing". */
static inline void ftn_push(ftn_t *f, uint32_t index, float distance) {
    if (f->size < f->capacity) {
        f->indices[f->size] = index;
        f->distances[f->size] = distance;
        f->size++;
    } else {
        if (f->capacity == 0) return;
        size_t max_idx = 0;
        float max_dist = f->distances[0];
        for (size_t i = 1; i < f->size; ++i) {
            if (f->distances[i] > max_dist) {
                max_dist = f->distances[i];
                max_idx = i;
            }
        }
        if (distance < max_dist) {
            f->indices[max_idx] = index;
            f->distances[max_idx] = distance;
        }
    }
}

/* threshold :320-330 (epsilon = 0) */
static inline float ftn_threshold(const ftn_t *f) {
    if (f->size >= f->capacity) {
        float m = -INFINITY;
        for (size_t i = 0; i < f->size; ++i) m = fmaxf(m, f->distances[i]);
        return m * (1.0f + 0.0f);
    }
    return INFINITY;
}

/* results :374-382 */
static size_t ftn_results(const ftn_t *f, pair_t *out) {
    for (size_t i = 0; i < f->size; ++i) {
        out[i].idx = f->indices[i];
        out[i].dist = f->distances[i];
    }
    sort_pairs(out, f->size);
    return f->size;
}

size_t or_fast_top_neighbors_run(size_t cap, const uint32_t *idx, const float *dist,
                                 size_t n, uint32_t *out_idx, float *out_dist) {
    ftn_t f;
    ftn_init(&f, cap);
    for (size_t i = 0; i < n; ++i) ftn_push(&f, idx[i], dist[i]);
    pair_t *tmp = (pair_t *)malloc((cap ? cap : 1) * sizeof(pair_t));
    size_t r = ftn_results(&f, tmp);
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = tmp[i].idx;
        out_dist[i] = tmp[i].dist;
    }
    free(tmp);
    ftn_free(&f);
    return r;
}

size_t or_fast_top_neighbors_push_batch(size_t cap, const uint32_t *idx,
                                        const float *dist, size_t n,
                                        uint32_t *out_idx, float *out_dist) {
    ftn_t f;
    ftn_init(&f, cap);
    for (size_t i = 0; i < n; ++i)
        if (dist[i] < ftn_threshold(&f)) ftn_push(&f, idx[i], dist[i]);
    pair_t *tmp = (pair_t *)malloc((cap ? cap : 1) * sizeof(pair_t));
    size_t r = ftn_results(&f, tmp);
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = tmp[i].idx;
        out_dist[i] = tmp[i].dist;
    }
    free(tmp);
    ftn_free(&f);
    return r;
}

/* l1_distance_avx2: simd/x86.rs:103-132 */
float or_l1_avx2(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8;
    float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) lane[j] = lane[j] + fabsf(a[8 * i + j] - b[8 * i + j]);
    /* horizontal_sum_f32_avx2 x86.rs:31-44: (lo + hi), + movehdup, + movehl */
    float s0 = lane[0] + lane[4], s1 = lane[1] + lane[5], s2 = lane[2] + lane[6], s3 = lane[3] + lane[7];
    float r = (s0 + s1) + (s2 + s3);
    for (size_t i = chunks * 8; i < n; ++i) r += fabsf(a[i] - b[i]);
    return r;
}

/* wide 0.7 f32x8::reduce_add without target_feature = "avx" (see scann_oracle.h): two f32x4 halves,
 * each summed sequentially */
static float wide07_reduce_add(const float *v) {
    float lo = ((v[0] + v[1]) + v[2]) + v[3];
    float hi = ((v[4] + v[5]) + v[6]) + v[7];
    return lo + hi;
}

/* cosine_similarity_f32_simd one_to_one.rs:559-604, cosine_distance :607-612 */
float or_cosine_distance(const float *a, const float *b, size_t n) {
    size_t chunks = n / 8;
    float ab[8] = {0}, aa[8] = {0}, bb[8] = {0};
    for (size_t i = 0; i < chunks; ++i)
        for (int j = 0; j < 8; ++j) {
            float x = a[8 * i + j], y = b[8 * i + j];
            ab[j] = ab[j] + x * y; /* add(mul): two roundings */
            aa[j] = aa[j] + x * x;
            bb[j] = bb[j] + y * y;
        }
    float sab = wide07_reduce_add(ab), saa = wide07_reduce_add(aa), sbb = wide07_reduce_add(bb);
    for (size_t i = chunks * 8; i < n; ++i) {
        sab += a[i] * b[i];
        saa += a[i] * a[i];
        sbb += b[i] * b[i];
    }
    float na = sqrtf(saa), nb = sqrtf(sbb);
    float sim = (na == 0.0f || nb == 0.0f) ? 0.0f : sab / (na * nb);
    return 1.0f - sim;
}

/* ------------------------------------------------------------------------ */
/* brute force: brute_force/searcher.rs:77-208                              */
/* ------------------------------------------------------------------------ */

/* compute_distances :113-139 */
static void bf_compute_distances(const float *data, size_t n, size_t dim,
                                 size_t stride, int measure, const float *q,
                                 float *dist) {
    if (measure == OR_SQUARED_L2 || measure == OR_L2) {
        or_one_to_many_squared_l2(q, dim, data, stride, n, dist);
        if (measure == OR_L2)
            for (size_t i = 0; i < n; ++i) dist[i] = sqrtf(dist[i]);
    } else if (measure == OR_DOT_PRODUCT) {
        or_one_to_many_dot_product(q, dim, data, stride, n, dist);
    } else { /* :131-137 fallback: DistanceMeasure::distance one by one */
        for (size_t i = 0; i < n; ++i) dist[i] = or_measure_distance(measure, q, data + i * stride, dim);
    }
}

int or_bf_search(const float *data, size_t n, size_t dim, size_t stride, int measure,
                 const float *q, size_t qdim, size_t k, uint32_t *out_idx,
                 float *out_dist) {
    if (n == 0) return 0;                               /* :78-80 */
    if (qdim != dim) return OR_ERR_INVALID_ARGUMENT;    /* :83-89 */
    if (k > n) k = n;                                   /* :91 */
    float *dist = (float *)malloc(n * sizeof(float));   /* :100 */
    bf_compute_distances(data, n, dim, stride, measure, q, dist);
    topk_t t;
    topk_init(&t, k);
    for (size_t i = 0; i < n; ++i) topk_push(&t, (uint32_t)i, dist[i]); /* :104-107 */
    size_t r = topk_drain_sorted(&t, out_idx, out_dist);
    free(t.data);
    free(dist);
    return (int)r;
}

int or_bf_search_batched(const float *data, size_t n, size_t dim, size_t stride,
                         int measure, const float *queries, size_t nq,
                         size_t q_stride, size_t k, uint32_t *out_idx,
                         float *out_dist, uint32_t *out_count, int nthreads) {
    int err = 0;
    if (nthreads <= 0) nthreads = or_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long qi = 0; qi < (long)nq; ++qi) {
        int r = or_bf_search(data, n, dim, stride, measure, queries + qi * q_stride,
                             dim, k, out_idx + qi * k, out_dist + qi * k);
        if (r < 0) {
#pragma omp atomic write
            err = r;
            r = 0;
        }
        out_count[qi] = (uint32_t)r;
    }
    return err;
}

size_t or_bf_search_radius(const float *data, size_t n, size_t dim, size_t stride,
                           int measure, const float *q, float radius,
                           uint32_t *out_idx, float *out_dist) {
    if (n == 0) return 0;
    float *dist = (float *)malloc(n * sizeof(float));
    bf_compute_distances(data, n, dim, stride, measure, q, dist);
    pair_t *res = (pair_t *)malloc(n * sizeof(pair_t));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i)
        if (dist[i] <= radius) {
            res[m].idx = (uint32_t)i;
            res[m].dist = dist[i];
            ++m;
        }
    sort_pairs(res, m);
    for (size_t i = 0; i < m; ++i) {
        out_idx[i] = res[i].idx;
        out_dist[i] = res[i].dist;
    }
    free(res);
    free(dist);
    return m;
}

/* ------------------------------------------------------------------------ */
/* partitioner: partitioning/tree_partitioner.rs:175-229                    */
/* ------------------------------------------------------------------------ */

static size_t partition_impl(const float *centers, size_t L, size_t dim,
                             const float *q, size_t num_partitions, pair_t *scratch,
                             pair_t *tmp, uint32_t *out_tokens, float *out_dists) {
    for (size_t c = 0; c < L; ++c) { /* compute_center_distances :175-180 */
        scratch[c].idx = (uint32_t)c;
        scratch[c].dist = or_squared_l2_sequential(q, centers + c * dim, dim);
    }
    stable_sort_pairs_ordered(scratch, L, tmp); /* :212 */
    size_t r = num_partitions < L ? num_partitions : L; /* :214 */
    for (size_t i = 0; i < r; ++i) {
        if (out_tokens) out_tokens[i] = scratch[i].idx;
        if (out_dists) out_dists[i] = scratch[i].dist;
    }
    return r;
}

size_t or_partition(const float *centers, size_t L, size_t dim, const float *q,
                    size_t num_partitions, uint32_t *out_tokens, float *out_dists) {
    pair_t *scratch = (pair_t *)malloc((L ? L : 1) * sizeof(pair_t));
    pair_t *tmp = (pair_t *)malloc((L ? L : 1) * sizeof(pair_t));
    size_t r = partition_impl(centers, L, dim, q, num_partitions, scratch, tmp,
                              out_tokens, out_dists);
    free(scratch);
    free(tmp);
    return r;
}

/* ------------------------------------------------------------------------ */
/* codebook / LUT                                                           */
/* ------------------------------------------------------------------------ */

/* SubspaceCodebook::encode codebook.rs:82-95 (strict <, lowest index wins);
 * Codebook::encode :205-215. */
void or_encode(const float *codebook, size_t S, size_t K, size_t dsub, const float *x,
               uint8_t *codes) {
    for (size_t s = 0; s < S; ++s) {
        float min_dist = INFINITY;
        uint8_t min_idx = 0;
        const float *sub = x + s * dsub;
        for (size_t c = 0; c < K; ++c) {
            float d = or_squared_l2_sequential(sub, codebook + (s * K + c) * dsub, dsub);
            if (d < min_dist) {
                min_dist = d;
                min_idx = (uint8_t)c;
            }
        }
        codes[s] = min_idx;
    }
}

/* LookupTable::from_query lut.rs:47-70 */
void or_lut_from_query(const float *codebook, size_t S, size_t K, size_t dsub,
                       const float *q, float *lut) {
    for (size_t s = 0; s < S; ++s)
        for (size_t c = 0; c < K; ++c)
            lut[s * K + c] =
                or_squared_l2_sequential(q + s * dsub, codebook + (s * K + c) * dsub, dsub);
}

/* LookupTable::compute_distance lut.rs:74-82 */
float or_lut_distance(const float *lut, size_t S, size_t K, const uint8_t *codes) {
    float sum = 0.0f;
    for (size_t s = 0; s < S; ++s) sum += lut[s * K + codes[s]];
    return sum;
}

/* ------------------------------------------------------------------------ */
/* LUT16                                                                    */
/* ------------------------------------------------------------------------ */

size_t or_pack4_bytes_per_point(size_t S) { return (S + 1) / 2; }

/* lut16.rs:43-61 */
void or_pack4(const uint8_t *codes, size_t n, size_t S, uint8_t *packed) {
    size_t bpp = (S + 1) / 2;
    for (size_t p = 0; p < n; ++p)
        for (size_t j = 0; j < bpp; ++j) {
            uint8_t lo = codes[p * S + 2 * j] & 0x0F;
            uint8_t hi = (2 * j + 1 < S) ? (uint8_t)((codes[p * S + 2 * j + 1] & 0x0F) << 4) : 0;
            packed[p * bpp + j] = lo | hi;
        }
}

/* lut16.rs:64-77 */
void or_unpack4(const uint8_t *packed, size_t n, size_t S, uint8_t *codes) {
    size_t bpp = (S + 1) / 2;
    for (size_t p = 0; p < n; ++p) {
        size_t o = 0;
        for (size_t i = 0; i < bpp; ++i) {
            uint8_t byte = packed[p * bpp + i];
            codes[p * S + o++] = byte & 0x0F;
            if (i * 2 + 1 < S) codes[p * S + o++] = (byte >> 4) & 0x0F;
        }
    }
}

/* lut16.rs:186-203 */
float or_lut16_distance_packed_f32(const float *tables, size_t S, const uint8_t *packed) {
    float sum = 0.0f;
    size_t t = 0, bpp = (S + 1) / 2;
    for (size_t i = 0; i < bpp; ++i) {
        uint8_t byte = packed[i];
        if (t < S) { sum += tables[t * 16 + (byte & 0x0F)]; ++t; }
        if (t < S) { sum += tables[t * 16 + ((byte >> 4) & 0x0F)]; ++t; }
    }
    return sum;
}

/* f32::round: half away from zero; `as u8` saturates. */
static inline uint8_t round_sat_u8(float v) {
    float r = roundf(v);
    if (!(r > 0.0f)) return 0; /* also NaN -> 0 */
    if (r >= 255.0f) return 255;
    return (uint8_t)r;
}

/* lut16_simd.rs:39-90 */
void or_lut16_quantize(const float *tables, size_t S, uint8_t *lut8, float *bias,
                       float *multiplier) {
    if (S == 0) {
        *bias = 0.0f;
        *multiplier = 1.0f;
        return;
    }
    float gmin = 3.40282347e+38f, gmax = -3.40282347e+38f; /* f32::MAX / f32::MIN */
    for (size_t i = 0; i < S * 16; ++i) {
        gmin = fminf(gmin, tables[i]); /* f32::min / f32::max ignore NaN like fminf */
        gmax = fmaxf(gmax, tables[i]);
    }
    float range = gmax - gmin;
    float scale;
    if (range < 1e-10f) {
        *multiplier = 1.0f;
        *bias = gmin;
        scale = 1.0f;
    } else {
        scale = 255.0f / range;
        *multiplier = 1.0f / scale;
        *bias = gmin;
    }
    for (size_t i = 0; i < S * 16; ++i) lut8[i] = round_sat_u8((tables[i] - gmin) * scale);
}

/* simd/dispatch.rs:259-295 */
void or_lut16_distances_batch_raw(const uint8_t *packed, const uint8_t *lut8, size_t S,
                                  size_t n, float *out) {
    size_t bpp = (S + 1) / 2;
    for (size_t dp = 0; dp < n; ++dp) {
        const uint8_t *row = packed + dp * bpp;
        uint32_t sum = 0;
        size_t sub = 0;
        for (size_t b = 0; b < bpp; ++b) {
            uint8_t byte = row[b];
            if (sub < S) { sum += lut8[sub * 16 + (byte & 0x0F)]; ++sub; }
            if (sub < S) { sum += lut8[sub * 16 + ((byte >> 4) & 0x0F)]; ++sub; }
        }
        out[dp] = (float)sum;
    }
}

/* lut16_simd.rs:119-141 */
void or_lut16_distances_batch(const uint8_t *packed, const uint8_t *lut8, size_t S,
                              size_t n, float bias, float multiplier, float *out) {
    or_lut16_distances_batch_raw(packed, lut8, S, n, out);
    float bias_total = bias * (float)S;
    for (size_t i = 0; i < n; ++i) {
        float m = out[i] * multiplier;
        out[i] = m + bias_total;
    }
}

/* lut16_simd.rs:144-154 */
float or_lut16_distance_single(const uint8_t *lut8, size_t S, float bias,
                               float multiplier, const uint8_t *codes) {
    uint32_t sum = 0;
    for (size_t s = 0; s < S; ++s) sum += lut8[s * 16 + (codes[s] & 0x0F)];
    float a = (float)sum * multiplier;
    float b = bias * (float)S;
    return a + b;
}

/* ------------------------------------------------------------------------ */
/* re-rank: tree_x_hybrid/mod.rs:342-364, utils/reordering.rs:23-54,         */
/* hashes/hasher.rs:206-228                                                 */
/* ------------------------------------------------------------------------ */
static size_t reorder_impl(const float *data, size_t stride, size_t dim, const float *q,
                           pair_t *cand, size_t n_cand, size_t k) {
    for (size_t i = 0; i < n_cand; ++i)
        cand[i].dist = or_squared_l2_avx2(q, data + (size_t)cand[i].idx * stride, dim);
    sort_pairs(cand, n_cand);
    return n_cand < k ? n_cand : k;
}

size_t or_reorder(const float *data, size_t stride, size_t dim, const float *q,
                  const uint32_t *cand_idx, size_t n_cand, size_t k, uint32_t *out_idx,
                  float *out_dist) {
    pair_t *c = (pair_t *)malloc((n_cand ? n_cand : 1) * sizeof(pair_t));
    for (size_t i = 0; i < n_cand; ++i) c[i].idx = cand_idx[i];
    size_t r = reorder_impl(data, stride, dim, q, c, n_cand, k);
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = c[i].idx;
        out_dist[i] = c[i].dist;
    }
    free(c);
    return r;
}

/* ------------------------------------------------------------------------ */
/* AsymmetricHasher: hashes/hasher.rs:162-229                               */
/* ------------------------------------------------------------------------ */
static int ah_search_pairs(const float *codebook, size_t S, size_t K, size_t dsub,
                           const uint8_t *codes, size_t n, const float *q, size_t k,
                           pair_t *out) {
    float *lut = (float *)malloc(S * K * sizeof(float));
    or_lut_from_query(codebook, S, K, dsub, q, lut); /* :174 */
    ftn_t f;
    ftn_init(&f, k); /* :177 */
    for (size_t i = 0; i < n; ++i)
        ftn_push(&f, (uint32_t)i, or_lut_distance(lut, S, K, codes + i * S)); /* :179-182 */
    size_t r = ftn_results(&f, out);
    ftn_free(&f);
    free(lut);
    return (int)r;
}

int or_ah_search(const float *codebook, size_t S, size_t K, size_t dsub,
                 const uint8_t *codes, size_t n, const float *q, size_t qdim, size_t k,
                 uint32_t *out_idx, float *out_dist) {
    if (n == 0) return 0;                                    /* :163-165 */
    if (qdim != S * dsub) return OR_ERR_INVALID_ARGUMENT;    /* :167-171 */
    pair_t *res = (pair_t *)malloc((k ? k : 1) * sizeof(pair_t));
    int r = ah_search_pairs(codebook, S, K, dsub, codes, n, q, k, res);
    for (int i = 0; i < r; ++i) {
        out_idx[i] = res[i].idx;
        out_dist[i] = res[i].dist;
    }
    free(res);
    return r;
}

int or_ah_search_with_reordering(const float *codebook, size_t S, size_t K, size_t dsub,
                                 const uint8_t *codes, size_t n, const float *data,
                                 size_t stride, const float *q, size_t qdim, size_t k,
                                 size_t pre_reorder_k, uint32_t *out_idx,
                                 float *out_dist) {
    if (n == 0) return 0;
    if (qdim != S * dsub) return OR_ERR_INVALID_ARGUMENT;
    pair_t *cand = (pair_t *)malloc((pre_reorder_k ? pre_reorder_k : 1) * sizeof(pair_t));
    int nc = ah_search_pairs(codebook, S, K, dsub, codes, n, q, pre_reorder_k, cand); /* :200 */
    size_t r = reorder_impl(data, stride, S * dsub, q, cand, (size_t)nc, k);      /* :206-227 */
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = cand[i].idx;
        out_dist[i] = cand[i].dist;
    }
    free(cand);
    return (int)r;
}

/* AsymmetricHasher::search_batched (hasher.rs:232-238) is a plain sequential map in
 * the reference; nthreads > 1 runs one query per task the way the other searchers'
 * par_iter does (reported as such by the CPU baseline). */
int or_ah_search_batched(const float *codebook, size_t S, size_t K, size_t dsub,
                         const uint8_t *codes, size_t n, const float *data, size_t stride,
                         const float *queries, size_t nq, size_t q_stride, size_t k,
                         size_t pre_reorder_k, int reorder, uint32_t *out_idx,
                         float *out_dist, uint32_t *out_count, int nthreads) {
    int err = 0;
    if (nthreads <= 0) nthreads = or_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long qi = 0; qi < (long)nq; ++qi) {
        int r;
        if (reorder)
            r = or_ah_search_with_reordering(codebook, S, K, dsub, codes, n, data, stride,
                                             queries + qi * q_stride, S * dsub, k, pre_reorder_k,
                                             out_idx + qi * k, out_dist + qi * k);
        else
            r = or_ah_search(codebook, S, K, dsub, codes, n, queries + qi * q_stride, S * dsub, k,
                             out_idx + qi * k, out_dist + qi * k);
        if (r < 0) {
#pragma omp atomic write
            err = r;
            r = 0;
        }
        out_count[qi] = (uint32_t)r;
    }
    return err;
}

/* ------------------------------------------------------------------------ */
/* Tree-X-Hybrid: tree_x_hybrid/mod.rs:245-364                              */
/* ------------------------------------------------------------------------ */

/* search_partition :297-339.  Writes <= m pairs to out, returns the count. */
static size_t txh_search_partition(const or_txh_index *ix, const float *q, uint32_t pid,
                                   size_t m, float *qres, float *lut, ftn_t *f,
                                   pair_t *out) {
    size_t dim = ix->dim;
    const float *qlut = q;
    (void)m; /* capacity already set on f */
    if (ix->use_residuals) { /* :309-316 */
        const float *c = ix->centers + (size_t)pid * dim;
        for (size_t j = 0; j < dim; ++j) qres[j] = q[j] - c[j];
        qlut = qres;
    }
    or_lut_from_query(ix->codebook, ix->S, ix->K, ix->dsub, qlut, lut); /* :319 */
    f->size = 0; /* FastTopNeighbors::new(k) :322 */
    for (size_t i = 0; i < f->capacity; ++i) f->distances[i] = INFINITY;
    uint32_t b = ix->leaf_off[pid], e = ix->leaf_off[pid + 1];
    for (uint32_t i = b; i < e; ++i) { /* :324-336 */
        uint32_t id = ix->leaf_ids[i];
        if (ix->allow && !((ix->allow[id >> 6] >> (id & 63u)) & 1u)) continue; /* :328-332 */
        float d = or_lut_distance(lut, ix->S, ix->K, ix->codes + (size_t)i * ix->S);
        ftn_push(f, ix->leaf_ids[i], d);
    }
    return ftn_results(f, out); /* :338 */
}

int or_txh_search(const or_txh_index *ix, const float *q, size_t qdim, size_t k,
                  uint32_t *out_idx, float *out_dist, uint32_t *tokens,
                  float *token_dists, size_t *n_tokens, uint32_t *cand_idx,
                  float *cand_dist, size_t *n_cand) {
    if (qdim != ix->dim) return OR_ERR_INVALID_ARGUMENT; /* :251-253 */
    size_t L = ix->L;
    size_t P = ix->partitions_to_search < L ? ix->partitions_to_search : L;
    uint32_t *tok = (uint32_t *)malloc((P ? P : 1) * sizeof(uint32_t));
    float *tokd = (float *)malloc((P ? P : 1) * sizeof(float));
    size_t np = or_partition(ix->centers, L, ix->dim, q, ix->partitions_to_search, tok, tokd); /* :258-261 */
    if (n_tokens) *n_tokens = np;
    for (size_t i = 0; i < np; ++i) {
        if (tokens) tokens[i] = tok[i];
        if (token_dists) token_dists[i] = tokd[i];
    }

    /* pre_reorder_k = (k as f32 * multiplier) as usize  :263 (saturating, truncating) */
    float mf = (float)k * ix->pre_reorder_multiplier;
    size_t m = (mf > 0.0f) ? (size_t)mf : 0;

    pair_t *all = (pair_t *)malloc((np * m > 0 ? np * m : 1) * sizeof(pair_t));
    size_t n_all = 0;
    float *qres = (float *)malloc(ix->dim * sizeof(float));
    float *lut = (float *)malloc((size_t)ix->S * ix->K * sizeof(float));
    ftn_t f;
    ftn_init(&f, m);
    for (size_t t = 0; t < np; ++t) /* :266-280: collect preserves token order */
        n_all += txh_search_partition(ix, q, tok[t], m, qres, lut, &f, all + n_all);
    ftn_free(&f);
    free(lut);
    free(qres);

    sort_pairs(all, n_all);          /* :289 */
    if (n_all > m) n_all = m;        /* :290 */
    if (n_cand) *n_cand = n_all;
    for (size_t i = 0; i < n_all; ++i) {
        if (cand_idx) cand_idx[i] = all[i].idx;
        if (cand_dist) cand_dist[i] = all[i].dist;
    }
    size_t r = reorder_impl(ix->data, ix->stride, ix->dim, q, all, n_all, k); /* :293 */
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = all[i].idx;
        out_dist[i] = all[i].dist;
    }
    free(all);
    free(tok);
    free(tokd);
    return (int)r;
}

int or_txh_search_batched(const or_txh_index *ix, const float *queries, size_t nq,
                          size_t q_stride, size_t k, uint32_t *out_idx, float *out_dist,
                          uint32_t *out_count, int nthreads) {
    int err = 0;
    if (nthreads <= 0) nthreads = or_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long qi = 0; qi < (long)nq; ++qi) {
        int r = or_txh_search(ix, queries + qi * q_stride, ix->dim, k, out_idx + qi * k,
                              out_dist + qi * k, NULL, NULL, NULL, NULL, NULL, NULL);
        if (r < 0) {
#pragma omp atomic write
            err = r;
            r = 0;
        }
        out_count[qi] = (uint32_t)r;
    }
    return err;
}

/* ------------------------------------------------------------------------ */
/* Scann facade modes: scann.rs:181-294                                     */
/* ------------------------------------------------------------------------ */
float or_measure_distance(int measure, const float *a, const float *b, size_t dim) {
    if (measure == OR_DOT_PRODUCT) return -or_dot_product_avx2(a, b, dim); /* one_to_one.rs:464-469 */
    if (measure == OR_L1) return or_l1_avx2(a, b, dim);                    /* distance_measures/mod.rs:71 */
    if (measure == OR_COSINE) return or_cosine_distance(a, b, dim);        /* :74 */
    float d = or_squared_l2_avx2(a, b, dim);                               /* :162-171 */
    return measure == OR_L2 ? sqrtf(d) : d;                                /* :156-158 */
}

size_t or_reorder_measure(const float *data, size_t stride, size_t dim, int measure, const float *q,
                          const uint32_t *cand_idx, size_t n_cand, size_t k, uint32_t *out_idx,
                          float *out_dist) {
    pair_t *c = (pair_t *)malloc((n_cand ? n_cand : 1) * sizeof(pair_t));
    for (size_t i = 0; i < n_cand; ++i) { /* reordering.rs:35-44 */
        c[i].idx = cand_idx[i];
        c[i].dist = or_measure_distance(measure, q, data + (size_t)cand_idx[i] * stride, dim);
    }
    sort_pairs(c, n_cand);                 /* :47-50 */
    size_t r = n_cand < k ? n_cand : k;    /* :53 */
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = c[i].idx;
        out_dist[i] = c[i].dist;
    }
    free(c);
    return r;
}

int or_scann_search_partitioned(const float *centers, size_t L, size_t dim, const uint32_t *leaf_off,
                                const uint32_t *leaf_ids, const float *data, size_t stride, int measure,
                                const float *q, size_t P, size_t k, uint32_t *out_idx, float *out_dist) {
    uint32_t *tokens = (uint32_t *)malloc((L ? L : 1) * sizeof(uint32_t));
    float *tdist = (float *)malloc((L ? L : 1) * sizeof(float));
    size_t np = or_partition(centers, L, dim, q, P, tokens, tdist); /* scann.rs:222-225 */
    size_t total = 0;
    for (size_t t = 0; t < np; ++t) total += leaf_off[tokens[t] + 1] - leaf_off[tokens[t]];
    pair_t *c = (pair_t *)malloc((total ? total : 1) * sizeof(pair_t));
    size_t o = 0;
    for (size_t t = 0; t < np; ++t) /* :228-233 candidates in token order, then leaf order */
        for (uint32_t r = leaf_off[tokens[t]]; r < leaf_off[tokens[t] + 1]; ++r) {
            c[o].idx = leaf_ids[r];
            c[o].dist = or_measure_distance(measure, q, data + (size_t)leaf_ids[r] * stride, dim); /* :237-246 */
            ++o;
        }
    sort_pairs(c, total);                  /* :249 */
    size_t r = total < k ? total : k;      /* :250 */
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = c[i].idx;
        out_dist[i] = c[i].dist;
    }
    free(c);
    free(tokens);
    free(tdist);
    return (int)r;
}

int or_scann_search_tree_ah(const float *centers, size_t L, size_t dim, const uint32_t *leaf_off,
                            const uint32_t *leaf_ids, const float *codebook, size_t S, size_t K,
                            size_t dsub, const uint8_t *codes, const float *data, size_t stride,
                            int measure, int reorder, const float *q, size_t P, size_t k,
                            uint32_t *out_idx, float *out_dist) {
    uint32_t *tokens = (uint32_t *)malloc((L ? L : 1) * sizeof(uint32_t));
    float *tdist = (float *)malloc((L ? L : 1) * sizeof(float));
    size_t np = or_partition(centers, L, dim, q, P, tokens, tdist); /* scann.rs:266-269 */
    size_t total = 0;
    for (size_t t = 0; t < np; ++t) total += leaf_off[tokens[t] + 1] - leaf_off[tokens[t]];
    pair_t *c = (pair_t *)malloc((total ? total : 1) * sizeof(pair_t));
    float *lut = (float *)malloc(S * K * sizeof(float));
    or_lut_from_query(codebook, S, K, dsub, q, lut); /* :277-280: the same table for every token */
    size_t o = 0;
    for (size_t t = 0; t < np; ++t)
        for (uint32_t r = leaf_off[tokens[t]]; r < leaf_off[tokens[t] + 1]; ++r) { /* :282-286 */
            c[o].idx = leaf_ids[r];
            c[o].dist = or_lut_distance(lut, S, K, codes + (size_t)leaf_ids[r] * S);
            ++o;
        }
    sort_pairs(c, total);                  /* :291 */
    size_t r = total < k ? total : k;      /* :292 */
    if (reorder) {                         /* scann.rs:199-209 on the truncated list */
        for (size_t i = 0; i < r; ++i)
            c[i].dist = or_measure_distance(measure, q, data + (size_t)c[i].idx * stride, dim);
        sort_pairs(c, r);
    }
    for (size_t i = 0; i < r; ++i) {
        out_idx[i] = c[i].idx;
        out_dist[i] = c[i].dist;
    }
    free(lut);
    free(c);
    free(tokens);
    free(tdist);
    return (int)r;
}

/* bin/ann_benchmark.rs:427-450: sequential scalar squared_l2, stable sort, take k. */
void or_exact_ground_truth(const float *train, size_t n, size_t dim, size_t stride,
                           const float *queries, size_t nq, size_t q_stride, size_t k,
                           uint32_t *gt, int nthreads) {
    if (nthreads <= 0) nthreads = or_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
    for (long qi = 0; qi < (long)nq; ++qi) {
        pair_t *d = (pair_t *)malloc(n * sizeof(pair_t));
        for (size_t i = 0; i < n; ++i) {
            d[i].idx = (uint32_t)i;
            d[i].dist = or_squared_l2_sequential(queries + qi * q_stride, train + i * stride, dim);
        }
        sort_pairs(d, n);
        for (size_t i = 0; i < k && i < n; ++i) gt[qi * k + i] = d[i].idx;
        free(d);
    }
}

/* ------------------------------------------------------------------------ */
/* K-means Lloyd loop: trees/kmeans.rs:210-263 (fit_single from given        */
/* centres), :352-379 assign_clusters, :382-414 update_centers, :419-429     */
/* squared_distance_with_threshold (AVX2 order when dim >= simd_threshold).  */
/* Rows are the window [col_offset, col_offset + dim) of data[n][stride].    */
/* ------------------------------------------------------------------------ */
static double km_assign(const float *data, size_t n, size_t stride, size_t col_offset, size_t dim,
                        const float *centers, size_t k, size_t simd_threshold, uint32_t *assign) {
    double inertia = 0.0;
    for (size_t i = 0; i < n; ++i) {
        const float *p = data + i * stride + col_offset;
        float min_dist = INFINITY;
        uint32_t min_idx = 0;
        for (size_t c = 0; c < k; ++c) {
            const float *cc = centers + c * dim;
            float d = (dim >= simd_threshold) ? or_squared_l2_avx2(p, cc, dim)
                                              : or_squared_l2_sequential(p, cc, dim);
            if (d < min_dist) { /* strict: lowest index on ties */
                min_dist = d;
                min_idx = (uint32_t)c;
            }
        }
        assign[i] = min_idx;
        inertia += (double)min_dist; /* :376 sum in datapoint order */
    }
    return inertia;
}

int or_kmeans_lloyd(const float *data, size_t n, size_t stride, size_t col_offset, size_t dim,
                    float *centers, size_t k, size_t max_iterations, double convergence_threshold,
                    size_t simd_threshold, uint32_t *assign, uint32_t *sizes, double *inertia_out,
                    uint32_t *iterations_out, int *converged_out) {
    if (n == 0) return OR_ERR_INVALID_ARGUMENT; /* :167-169 */
    double *sums = (double *)malloc(k * dim * sizeof(double));
    size_t *counts = (size_t *)malloc(k * sizeof(size_t));
    double prev = INFINITY;
    uint32_t iters = 0;
    int converged = 0;
    for (size_t it = 0; it < max_iterations; ++it) { /* :226-246 */
        iters = (uint32_t)(it + 1);
        double inertia = km_assign(data, n, stride, col_offset, dim, centers, k, simd_threshold, assign);
        double rel = fabs(prev - inertia) / (prev + 1e-10);
        if (rel < convergence_threshold) {
            converged = 1;
            break;
        }
        prev = inertia;
        for (size_t e = 0; e < k * dim; ++e) sums[e] = 0.0; /* update_centers :382-414 */
        for (size_t c = 0; c < k; ++c) counts[c] = 0;
        for (size_t i = 0; i < n; ++i) {
            const float *p = data + i * stride + col_offset;
            size_t c = assign[i];
            counts[c] += 1;
            for (size_t j = 0; j < dim; ++j) sums[c * dim + j] += (double)p[j];
        }
        for (size_t c = 0; c < k; ++c) {
            if (counts[c] > 0) {
                for (size_t j = 0; j < dim; ++j)
                    centers[c * dim + j] = (float)(sums[c * dim + j] / (double)counts[c]);
            } else { /* empty cluster: data[c % n] :405-408 */
                const float *p = data + (c % n) * stride + col_offset;
                for (size_t j = 0; j < dim; ++j) centers[c * dim + j] = p[j];
            }
        }
    }
    double fin = km_assign(data, n, stride, col_offset, dim, centers, k, simd_threshold, assign); /* :248-249 */
    if (sizes) {
        for (size_t c = 0; c < k; ++c) sizes[c] = 0;
        for (size_t i = 0; i < n; ++i) sizes[assign[i]] += 1;
    }
    if (inertia_out) *inertia_out = fin;
    if (iterations_out) *iterations_out = iters;
    if (converged_out) *converged_out = converged;
    free(sums);
    free(counts);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * quantization/fp8.rs -- the reference's FP8 codec and quantizer, and the FP8 one-to-many kernels
 * of distance_measures/one_to_many_asymmetric.rs:327-377.
 * ------------------------------------------------------------------------------------------- */
uint8_t or_fp8_from_f32(float value, int format) {
    /* fp8.rs:80-117 (E4M3, bias 7, 3 mantissa bits) / :146-182 (E5M2, bias 15, 2 mantissa bits) */
    const int mbits = format == OR_FP8_E5M2 ? 2 : 3;
    const int bias = format == OR_FP8_E5M2 ? 15 : 7;
    const int emax = format == OR_FP8_E5M2 ? 31 : 15;
    const uint8_t maxcode = format == OR_FP8_E5M2 ? 0x7C : 0x7E;
    if (value == 0.0f) return 0;                                  /* :85-87 (also -0.0) */
    uint32_t bits;
    memcpy(&bits, &value, 4);
    const uint32_t sign = (bits >> 31) & 1u;
    const int exp = (int)((bits >> 23) & 0xFFu);
    const uint32_t mantissa = bits & 0x7FFFFFu;
    if (exp == 0xFF) return (uint8_t)((sign << 7) | maxcode);     /* :95-98 infinity or NaN -> max */
    const int fp8_exp = exp - 127 + bias;                         /* :101 */
    if (fp8_exp <= 0) return (uint8_t)(sign << 7);                /* :103-106 underflow */
    if (fp8_exp >= emax) return (uint8_t)((sign << 7) | maxcode); /* :108-111 overflow */
    const uint32_t m = ((mantissa >> (23 - mbits)) + ((mantissa >> (22 - mbits)) & 1u)) & ((1u << mbits) - 1u); /* :114 */
    return (uint8_t)((sign << 7) | ((uint32_t)fp8_exp << mbits) | m);
}

float or_fp8_to_f32(uint8_t b, int format) {
    /* fp8.rs:120-143 / :185-203 */
    const int mbits = format == OR_FP8_E5M2 ? 2 : 3;
    const int bias = format == OR_FP8_E5M2 ? 15 : 7;
    const uint32_t sign = (b >> 7) & 1u;
    const int exp = (b >> mbits) & (format == OR_FP8_E5M2 ? 0x1F : 0xF);
    const uint32_t mantissa = b & ((1u << mbits) - 1u);
    if (exp == 0 && mantissa == 0) return sign ? -0.0f : 0.0f;
    const int fp32_exp = exp == 0 ? 126 - bias : exp - bias + 127;   /* (the reference's "subnormal" exponent) */
    const uint32_t bits = (sign << 31) | ((uint32_t)fp32_exp << 23) | (mantissa << (23 - mbits));
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

float or_fp8_calibrate_scale(float max_abs_value, int format) {
    const float fp8_max = format == OR_FP8_E5M2 ? 57344.0f : 448.0f;
    const float d = max_abs_value > 1e-10f ? max_abs_value : 1e-10f;   /* f32::max: NaN operand ignored */
    return fp8_max / d;
}

void or_fp8_quantize(const float *values, size_t n, float scale, int format, uint8_t *out) {
    for (size_t i = 0; i < n; ++i) out[i] = or_fp8_from_f32(values[i] * scale, format);
}

void or_fp8_dequantize(const uint8_t *bits, size_t n, float scale, int format, float *out) {
    for (size_t i = 0; i < n; ++i) out[i] = or_fp8_to_f32(bits[i], format) / scale;
}

void or_one_to_many_fp8(const float *query, size_t dim, const uint8_t *database, size_t stride,
                        size_t num_points, int measure, float *results) {
    for (size_t i = 0; i < num_points; ++i) {
        const uint8_t *row = database + i * stride;
        float sum = 0.0f;
        if (measure == OR_DOT_PRODUCT) {
            for (size_t j = 0; j < dim; ++j) sum += query[j] * or_fp8_to_f32(row[j], OR_FP8_E4M3);
            results[i] = -sum;
        } else {
            for (size_t j = 0; j < dim; ++j) {
                const float diff = query[j] - or_fp8_to_f32(row[j], OR_FP8_E4M3);
                sum += diff * diff;
            }
            results[i] = sum;
        }
    }
}
