/*
 * sybil_oracle.c -- CPU ORACLE (test infrastructure, never the product path).
 * See sybil_oracle.h for scope, the reference file:line map and how it is pinned.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: Go on amd64 never fuses a*b+c, and the running-mean
 * arithmetic below has to round exactly like the reference's.
 */
#include "sybil_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define INTERNAL_RESULT_LIMIT 100000 /* aggregate.go:15 */
#define NUM_BUCKETS 1000             /* hist.go:3 */

/* ------------------------------------------------------------------ */
/* BasicHist (hist_basic.go)                                            */
/* ------------------------------------------------------------------ */

typedef struct {
    int64_t *v;
    int64_t n, cap;
} i64vec;

static void vec_push(i64vec *a, int64_t x) {
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 8;
        a->v = (int64_t *)realloc(a->v, (size_t)a->cap * sizeof(int64_t));
    }
    a->v[a->n++] = x;
}

struct orc_hist {
    /* BasicHistCachedInfo, hist_basic.go:10-26 */
    int64_t num_buckets, bucket_size;
    int64_t *values;
    double *averages;
    int64_t n_values;
    int percentile_mode;
    i64vec outliers, underliers; /* as the reference holds them (not merged by Combine) */
    int64_t max, min;
    int64_t samples, count;
    double avg;
    int64_t info_min, info_max;
    /* construction parameters */
    int op;
    int64_t hist_bucket;
    int weight_mode;
    /* exact side-state (what an order-independent integer engine computes) */
    uint64_t sum_exact;
    int64_t true_min, true_max;
    i64vec all_outliers, all_underliers; /* merged across blocks */
    /* MultiHist (hist_multi.go:6-19): the fields above are its Max / Min / Samples / Count / Avg / Info; values,
     * buckets and outliers live in the sub-histograms */
    int multi;
    int n_sub;
    struct orc_hist **sub;
};

/* hist_basic.go:34-70 SetupBuckets */
void orc_setup_buckets(int64_t info_min, int64_t info_max, int64_t hist_bucket,
                       int64_t *bucket_size, int64_t *num_buckets, int64_t *n_values) {
    int64_t size = info_max - info_min;
    int64_t nb = NUM_BUCKETS;
    int64_t bs = size / nb;
    if (hist_bucket > 0) bs = hist_bucket;
    if (bs == 0) {
        if (size < 100) {
            bs = 1;
            nb = size;
        } else {
            bs = size / 100;
            nb = size / bs;
        }
    }
    nb += 1;
    *bucket_size = bs;
    *num_buckets = nb;
    *n_values = nb + 1 > 0 ? nb + 1 : 0; /* Go would panic on a negative make() */
}

/* hist_basic.go:72-93 newBasicHist + TrackPercentiles */
orc_hist *orc_hist_new(int64_t info_min, int64_t info_max, int op, int64_t hist_bucket, int weight_mode) {
    orc_hist *h = (orc_hist *)calloc(1, sizeof(*h));
    h->info_min = info_min;
    h->info_max = info_max;
    h->op = op;
    h->hist_bucket = hist_bucket;
    h->weight_mode = weight_mode;
    h->true_min = INT64_MAX;
    h->true_max = INT64_MIN;
    if (op == ORC_AGG_HIST) {
        h->percentile_mode = 1;
        h->min = info_min; /* SetupBuckets: h.Min = min; h.Max = max */
        h->max = info_max;
        orc_setup_buckets(info_min, info_max, hist_bucket, &h->bucket_size, &h->num_buckets, &h->n_values);
        h->values = (int64_t *)calloc((size_t)(h->n_values ? h->n_values : 1), sizeof(int64_t));
        h->averages = (double *)calloc((size_t)(h->n_values ? h->n_values : 1), sizeof(double));
    }
    /* avg mode: SetupBuckets never runs, so Min = Max = 0 (Go zero values) */
    return h;
}

/* hist_multi.go:23-38 newMultiHist + :223-257 TrackPercentiles */
orc_hist *orc_hist_new_multi(int64_t info_min, int64_t info_max, int op, int64_t hist_bucket, int weight_mode) {
    orc_hist *h = (orc_hist *)calloc(1, sizeof(*h));
    h->multi = 1;
    h->info_min = info_min;
    h->info_max = info_max;
    h->op = op;
    h->hist_bucket = hist_bucket;
    h->weight_mode = weight_mode;
    h->true_min = INT64_MAX;
    h->true_max = INT64_MIN;
    h->min = info_min; /* :31-32, in avg mode too (unlike BasicHist) */
    h->max = info_max;
    if (op == ORC_AGG_HIST) {
        h->percentile_mode = 1;
        int64_t bucket_size = h->max - h->min;
        int num_hists = 0;
        /* 1:1 buckets for the smallest range, then logarithmically wider ones (HIST_FACTOR_POW = 1) */
        for (int64_t t = bucket_size; t > NUM_BUCKETS; t >>= 1) num_hists++;
        h->n_sub = num_hists + 1;
        h->sub = (orc_hist **)calloc((size_t)h->n_sub, sizeof(orc_hist *));
        int64_t right_edge = h->max;
        for (int i = 0; i < num_hists; i++) {
            bucket_size >>= 1;
            const int64_t smin = right_edge - bucket_size, smax = right_edge;
            right_edge = smin;
            h->sub[i] = orc_hist_new(smin, smax, ORC_AGG_HIST, hist_bucket, weight_mode);
        }
        /* the smallest hist at the end: h.Min -> the last bucket's left edge */
        h->sub[num_hists] = orc_hist_new(h->min, right_edge, ORC_AGG_HIST, hist_bucket, weight_mode);
    }
    return h;
}

int orc_hist_n_sub(const orc_hist *h) { return h->multi ? h->n_sub : 0; }

int orc_hist_sub(const orc_hist *h, int k, int64_t *out6) {
    if (!h->multi || k < 0 || k >= h->n_sub) return -1;
    int64_t off = 0;
    for (int i = 0; i < k; i++) off += h->sub[i]->n_values;
    const orc_hist *s = h->sub[k];
    out6[0] = s->info_min;
    out6[1] = s->info_max;
    out6[2] = s->bucket_size;
    out6[3] = s->num_buckets;
    out6[4] = s->n_values;
    out6[5] = off;
    return 0;
}

typedef struct {
    int64_t key, count;
} kc_pair;
static int cmp_kc(const void *a, const void *b) {
    const int64_t x = ((const kc_pair *)a)->key, y = ((const kc_pair *)b)->key;
    return x < y ? -1 : x > y;
}

/* MultiHist.GetSparseBuckets, hist_multi.go:190-207, over BasicHist.GetSparseBuckets (hist_basic.go:221-237): non-zero
 * buckets keyed by their lower edge, every outlier / underlier (of every block: the exact variant) +1 under its own
 * value, equal keys of different sub-histograms added up.  Sorted by key; the caller frees *out. */
static int64_t multi_sparse(const orc_hist *h, kc_pair **out) {
    int64_t cap = 0;
    for (int i = 0; i < h->n_sub; i++) cap += h->sub[i]->n_values + h->sub[i]->all_outliers.n + h->sub[i]->all_underliers.n;
    kc_pair *v = (kc_pair *)malloc(sizeof(kc_pair) * (size_t)(cap ? cap : 1));
    int64_t n = 0;
    for (int i = 0; i < h->n_sub; i++) {
        const orc_hist *s = h->sub[i];
        for (int64_t k = 0; k < s->n_values; k++)
            if (s->values[k] > 0) {
                v[n].key = k * s->bucket_size + s->min;
                v[n++].count = s->values[k];
            }
        for (int64_t j = 0; j < s->all_outliers.n; j++) {
            v[n].key = s->all_outliers.v[j];
            v[n++].count = 1;
        }
        for (int64_t j = 0; j < s->all_underliers.n; j++) {
            v[n].key = s->all_underliers.v[j];
            v[n++].count = 1;
        }
    }
    qsort(v, (size_t)n, sizeof(kc_pair), cmp_kc);
    int64_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        if (m > 0 && v[m - 1].key == v[i].key) v[m - 1].count += v[i].count;
        else v[m++] = v[i];
    }
    *out = v;
    return m;
}

int64_t orc_hist_sparse(const orc_hist *h, int64_t *keys, int64_t *counts, int64_t cap) {
    if (!h->multi) return -1;
    kc_pair *v;
    const int64_t n = multi_sparse(h, &v);
    for (int64_t i = 0; i < n && i < cap; i++) {
        keys[i] = v[i].key;
        counts[i] = v[i].count;
    }
    free(v);
    return n;
}

/* MultiHist.GetPercentiles, hist_multi.go:93-128, literally -- over the sorted keys of the union */
static int multi_percentiles(const orc_hist *h, int64_t *out100) {
    if (h->count == 0) return 0;
    kc_pair *v;
    const int64_t n = multi_sparse(h, &v);
    int64_t total = 0;
    for (int64_t i = 0; i < n; i++) total += v[i].count;
    int64_t pct[101];
    memset(pct, 0, sizeof(pct));
    int64_t prev_p = 0, count = 0;
    for (int64_t i = 0; i < n && total > 0; i++) {
        count += v[i].count;
        const int64_t p = (100 * count) / total;
        for (int64_t ip = prev_p; ip <= p; ip++)
            if (ip <= 100) pct[ip] = v[i].key;
        if (p <= 100) pct[p] = v[i].key;
        prev_p = p;
    }
    free(v);
    memcpy(out100, pct, 100 * sizeof(int64_t));
    return 100;
}

/* MultiHist.GetStdDev, hist_multi.go:140-155: sqrt(sum over the union's keys of (key - Avg)^2 * count / Count) */
static double multi_stddev(const orc_hist *h, double avg) {
    kc_pair *v;
    const int64_t n = multi_sparse(h, &v);
    double sum_variance = 0;
    for (int64_t i = 0; i < n; i++) {
        const double delta = (double)v[i].key - avg;
        const double ratio = (double)v[i].count / (double)h->count;
        sum_variance += (delta * delta) * ratio;
    }
    free(v);
    return sqrt(sum_variance);
}

void orc_hist_free(orc_hist *h) {
    if (!h) return;
    for (int i = 0; i < h->n_sub; i++) orc_hist_free(h->sub[i]);
    free(h->sub);
    free(h->values);
    free(h->averages);
    free(h->outliers.v);
    free(h->underliers.v);
    free(h->all_outliers.v);
    free(h->all_underliers.v);
    free(h);
}

/* hist_basic.go:101-151 AddWeightedValue */
void orc_hist_add(orc_hist *h, int64_t value, int64_t weight) {
    /* :104 -- note Info.Max*10 wraps exactly like Go's int64 multiply */
    int64_t max10 = (int64_t)((uint64_t)h->info_max * 10u);
    if (value > max10 || value < h->info_min) return;

    if (h->weight_mode || weight > 1) { /* :111-116 */
        h->samples++;
        h->count += weight;
    } else {
        h->count++;
    }
    /* :118 */
    h->avg = h->avg + (((double)value - h->avg) / (double)h->count) * (double)weight;
    if (value > h->max) h->max = value;
    if (value < h->min) h->min = value;

    /* exact side-state: the weight actually applied to Count */
    {
        int64_t w_applied = (h->weight_mode || weight > 1) ? weight : 1;
        h->sum_exact += (uint64_t)value * (uint64_t)w_applied;
        if (value < h->true_min) h->true_min = value;
        if (value > h->true_max) h->true_max = value;
    }

    if (!h->percentile_mode) return;

    if (h->multi) {
        /* hist_multi.go:84-89: the first sub-histogram whose range holds the value takes it -- through ITS
         * AddWeightedValue, reject gate (Info.Max*10 of the sub-range: negative maxima reject) included */
        for (int i = 0; i < h->n_sub; i++)
            if (value >= h->sub[i]->info_min && value <= h->sub[i]->info_max) {
                orc_hist_add(h->sub[i], value, weight);
                break;
            }
        return;
    }

    int64_t b = (value - h->min) / h->bucket_size; /* :130, truncating */
    if (b >= h->n_values) {
        vec_push(&h->outliers, value);
        vec_push(&h->all_outliers, value);
        b = h->n_values - 1;
    }
    if (b < 0) {
        vec_push(&h->underliers, value);
        vec_push(&h->all_underliers, value);
        b = 0;
    }
    double partial = h->averages[b];
    h->values[b] += weight; /* :147 */
    h->averages[b] = partial + (((double)value - partial) / (double)h->values[b] * (double)weight);
}

double orc_combine_avg(double avg_a, int64_t count_a, double avg_b, int64_t count_b) {
    int64_t total = count_a + count_b;
    return (avg_a * ((double)count_a / (double)total)) + (avg_b * ((double)count_b / (double)total));
}

/* hist_basic.go:259-279 Combine.  Averages/Outliers/Underliers are NOT merged there. */
void orc_hist_combine(orc_hist *h, const orc_hist *o) {
    if (h->multi) /* hist_multi.go:209-212: the sub-histograms pairwise, then the outer fields like BasicHist */
        for (int i = 0; i < h->n_sub && i < o->n_sub; i++) orc_hist_combine(h->sub[i], o->sub[i]);
    for (int64_t k = 0; k < o->n_values && k < h->n_values; k++) h->values[k] += o->values[k];
    int64_t total = h->count + o->count;
    h->avg = orc_combine_avg(h->avg, h->count, o->avg, o->count);
    if (h->min > o->min) h->min = o->min;
    if (h->max < o->max) h->max = o->max;
    h->samples += o->samples;
    h->count = total;
    /* exact side-state */
    h->sum_exact += o->sum_exact;
    if (o->true_min < h->true_min) h->true_min = o->true_min;
    if (o->true_max > h->true_max) h->true_max = o->true_max;
    for (int64_t i = 0; i < o->all_outliers.n; i++) vec_push(&h->all_outliers, o->all_outliers.v[i]);
    for (int64_t i = 0; i < o->all_underliers.n; i++) vec_push(&h->all_underliers, o->all_underliers.v[i]);
}

/* hist_basic.go:153-183 GetPercentiles, literally */
int orc_percentiles_from_values(const int64_t *values, int64_t n_values, int64_t bucket_size,
                                int64_t hmin, int64_t count, int64_t *out100) {
    if (count == 0) return 0;
    int64_t pct[101];
    memset(pct, 0, sizeof(pct));
    pct[0] = hmin;
    int64_t c = 0, prev_p = 0;
    for (int64_t k = 0; k < n_values; k++) {
        c += values[k];
        int64_t p = (100 * c) / count;
        if (p > 100) p = 100; /* Go would index out of range; unreachable when sum(Values)==Count */
        if (p < 0) p = 0;
        for (int64_t ip = prev_p; ip <= p; ip++) pct[ip] = k * bucket_size + hmin;
        pct[p] = k;
        prev_p = p;
    }
    memcpy(out100, pct, 100 * sizeof(int64_t));
    return 100;
}

/* hist_basic.go:192-219 GetStdDev, literally */
double orc_stddev_from_values(const int64_t *values, int64_t n_values, int64_t bucket_size,
                              int64_t hmin, int64_t count, double avg,
                              const int64_t *outliers, int64_t n_out,
                              const int64_t *underliers, int64_t n_under) {
    double sum_variance = 0;
    for (int64_t b = 0; b < n_values; b++) {
        int64_t val = b * bucket_size + hmin;
        double delta = (double)val - avg;
        double ratio = (double)values[b] / (double)count;
        sum_variance += (delta * delta) * ratio;
    }
    for (int64_t i = 0; i < n_out; i++) {
        double d = (double)outliers[i] - avg;
        double delta = d * d; /* math.Pow(x,2) == x*x bit-for-bit */
        double ratio = 1 / (double)count;
        sum_variance += delta * ratio;
    }
    for (int64_t i = 0; i < n_under; i++) {
        double d = (double)underliers[i] - avg;
        double delta = d * d;
        double ratio = 1 / (double)count;
        sum_variance += delta * ratio;
    }
    return sqrt(sum_variance);
}

void orc_hist_info_get(const orc_hist *h, orc_hist_info *out) {
    memset(out, 0, sizeof(*out));
    out->present = 1;
    out->percentile_mode = h->percentile_mode;
    out->num_buckets = h->num_buckets;
    out->bucket_size = h->bucket_size;
    out->n_values = h->n_values;
    out->count = h->count;
    out->samples = h->samples;
    out->min = h->min;
    out->max = h->max;
    out->avg = h->avg;
    out->sum_exact = (int64_t)h->sum_exact;
    out->true_min = h->true_min;
    out->true_max = h->true_max;
    out->n_outliers = h->all_outliers.n;
    out->n_underliers = h->all_underliers.n;
    if (h->multi) {
        out->num_buckets = 0;
        out->bucket_size = 0;
        out->n_values = 0;
        for (int i = 0; i < h->n_sub; i++) {
            out->n_values += h->sub[i]->n_values;
            out->n_outliers += h->sub[i]->all_outliers.n;
            out->n_underliers += h->sub[i]->all_underliers.n;
        }
        const double avg_exact = h->count ? (double)((long double)(int64_t)h->sum_exact / (long double)h->count) : 0.0;
        out->stddev_ref = h->percentile_mode ? multi_stddev(h, h->avg) : 0.0;
        out->stddev_exact = h->percentile_mode ? multi_stddev(h, avg_exact) : 0.0;
        return;
    }
    out->stddev_ref = orc_stddev_from_values(h->values, h->n_values, h->bucket_size, h->min, h->count, h->avg,
                                             h->outliers.v, h->outliers.n, h->underliers.v, h->underliers.n);
    {
        double avg_exact = h->count ? (double)((long double)(int64_t)h->sum_exact / (long double)h->count) : 0.0;
        out->stddev_exact = orc_stddev_from_values(h->values, h->n_values, h->bucket_size, h->min, h->count,
                                                   avg_exact, h->all_outliers.v, h->all_outliers.n,
                                                   h->all_underliers.v, h->all_underliers.n);
    }
}

int64_t orc_hist_values(const orc_hist *h, int64_t *out, int64_t cap) {
    if (h->multi) { /* the sub-histograms' Values, Subhists[0] first */
        int64_t n = 0;
        for (int i = 0; i < h->n_sub; i++) n += h->sub[i]->n_values;
        if (cap < n) return -1;
        n = 0;
        for (int i = 0; i < h->n_sub; i++) {
            memcpy(out + n, h->sub[i]->values, (size_t)h->sub[i]->n_values * sizeof(int64_t));
            n += h->sub[i]->n_values;
        }
        return n;
    }
    if (cap < h->n_values) return -1;
    memcpy(out, h->values, (size_t)h->n_values * sizeof(int64_t));
    return h->n_values;
}

int orc_hist_percentiles(const orc_hist *h, int64_t *out100) {
    if (h->multi) return multi_percentiles(h, out100);
    return orc_percentiles_from_values(h->values, h->n_values, h->bucket_size, h->min, h->count, out100);
}

int64_t orc_hist_outliers(const orc_hist *h, int64_t *out, int64_t cap) {
    if (cap < h->outliers.n) return -1;
    if (h->outliers.n) memcpy(out, h->outliers.v, (size_t)h->outliers.n * sizeof(int64_t));
    return h->outliers.n;
}

int64_t orc_time_bucket(int64_t t, int64_t bucket) { return t / bucket * bucket; } /* aggregate.go:174 */

/* ------------------------------------------------------------------ */
/* Result / ResultMap (query_spec.go)                                   */
/* ------------------------------------------------------------------ */

#define KEY_BYTES (ORC_GROUP_BY_WIDTH * ORC_MAX_GROUPS)

struct orc_result {
    uint8_t key[KEY_BYTES];
    int64_t time_bucket;
    int64_t count, samples;
    orc_hist *hists[ORC_MAX_AGGS];
    uint8_t *llb; /* Result.Distinct (query_spec.go:87): ORC_LLB_M registers, NULL = none added yet (all zero) */
};

typedef struct {
    orc_result **items; /* insertion order */
    int64_t n, cap;
    int64_t *slots; /* open addressing: index into items, -1 empty */
    int64_t nslots;
    int key_len;
} rmap;

static uint64_t hash_key(const uint8_t *k, int len, int64_t tb) {
    uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t)tb * 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < len; i++) {
        h ^= k[i];
        h *= 0x100000001b3ull;
    }
    h ^= h >> 29;
    return h;
}

static void rmap_init(rmap *m, int key_len) {
    memset(m, 0, sizeof(*m));
    m->key_len = key_len;
    m->nslots = 64;
    m->slots = (int64_t *)malloc((size_t)m->nslots * sizeof(int64_t));
    for (int64_t i = 0; i < m->nslots; i++) m->slots[i] = -1;
}

static void rmap_grow(rmap *m) {
    int64_t ns = m->nslots * 2;
    int64_t *s = (int64_t *)malloc((size_t)ns * sizeof(int64_t));
    for (int64_t i = 0; i < ns; i++) s[i] = -1;
    for (int64_t i = 0; i < m->n; i++) {
        uint64_t h = hash_key(m->items[i]->key, m->key_len, m->items[i]->time_bucket);
        int64_t p = (int64_t)(h & (uint64_t)(ns - 1));
        while (s[p] >= 0) p = (p + 1) & (ns - 1);
        s[p] = i;
    }
    free(m->slots);
    m->slots = s;
    m->nslots = ns;
}

static orc_result *rmap_find(const rmap *m, const uint8_t *key, int64_t tb) {
    uint64_t h = hash_key(key, m->key_len, tb);
    int64_t p = (int64_t)(h & (uint64_t)(m->nslots - 1));
    while (m->slots[p] >= 0) {
        orc_result *r = m->items[m->slots[p]];
        if (r->time_bucket == tb && memcmp(r->key, key, (size_t)m->key_len) == 0) return r;
        p = (p + 1) & (m->nslots - 1);
    }
    return NULL;
}

static void rmap_put(rmap *m, orc_result *r) {
    if ((m->n + 1) * 2 > m->nslots) rmap_grow(m);
    if (m->n == m->cap) {
        m->cap = m->cap ? m->cap * 2 : 16;
        m->items = (orc_result **)realloc(m->items, (size_t)m->cap * sizeof(*m->items));
    }
    m->items[m->n] = r;
    uint64_t h = hash_key(r->key, m->key_len, r->time_bucket);
    int64_t p = (int64_t)(h & (uint64_t)(m->nslots - 1));
    while (m->slots[p] >= 0) p = (p + 1) & (m->nslots - 1);
    m->slots[p] = m->n;
    m->n++;
}

static orc_result *result_new(const uint8_t *key, int key_len, int64_t tb) {
    orc_result *r = (orc_result *)calloc(1, sizeof(*r));
    if (key) memcpy(r->key, key, (size_t)key_len);
    r->time_bucket = tb;
    return r;
}

static void result_free(orc_result *r) {
    if (!r) return;
    for (int a = 0; a < ORC_MAX_AGGS; a++) orc_hist_free(r->hists[a]);
    free(r->llb);
    free(r);
}

static void rmap_free(rmap *m, int free_items) {
    if (free_items)
        for (int64_t i = 0; i < m->n; i++) result_free(m->items[i]);
    free(m->items);
    free(m->slots);
    memset(m, 0, sizeof(*m));
}

/* query_spec.go:138-193 Result.Combine (MERGE_TABLE == nil path) */
static void result_combine(orc_result *rs, const orc_result *next, const orc_query *q) {
    if (!next) return;
    if (next->count == 0) return;
    for (int a = 0; a < q->n_aggs; a++) {
        const orc_hist *h = next->hists[a];
        if (!h) continue;
        if (!rs->hists[a]) {
            /* nh := h.NewHist(); nh.Combine(h) */
            rs->hists[a] = h->multi ? orc_hist_new_multi(h->info_min, h->info_max, h->op, h->hist_bucket, h->weight_mode)
                                    : orc_hist_new(h->info_min, h->info_max, h->op, h->hist_bucket, h->weight_mode);
        }
        orc_hist_combine(rs->hists[a], h);
    }
    /* query_spec.go:180-188 combine count distincts */
    if (next->llb) {
        if (!rs->llb) rs->llb = (uint8_t *)calloc(ORC_LLB_M, 1);
        orc_llb_merge(rs->llb, next->llb);
    }
    rs->samples += next->samples;
    rs->count += next->count;
}

/* ------------------------------------------------------------------ */
/* Count distinct: github.com/logv/loglogbeta (see the header: unpinned) */
/* ------------------------------------------------------------------ */

static inline int col_populated(const orc_col *c, int64_t row) { return c->populated ? c->populated[row] != 0 : 1; }

static inline uint64_t rotr64(uint64_t v, unsigned k) { return (v >> k) | (v << (64 - k)); }
static inline uint64_t rd_le(const uint8_t *p, int n) {
    uint64_t v = 0;
    for (int i = 0; i < n; i++) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

/* MetroHash64 v1 (metrohash64.cpp, MetroHash64::Hash) == go-metro Hash64(buffer, seed) */
uint64_t orc_metro64(const uint8_t *ptr, int64_t len, uint64_t seed) {
    const uint64_t k0 = 0xD6D018F5ull, k1 = 0xA2AA033Bull, k2 = 0x62992FC1ull, k3 = 0x30BC5B29ull;
    const uint8_t *end = ptr + len;
    uint64_t hash = (seed + k2) * k0;
    if (len >= 32) {
        uint64_t v0 = hash, v1 = hash, v2 = hash, v3 = hash;
        do {
            v0 += rd_le(ptr, 8) * k0; ptr += 8; v0 = rotr64(v0, 29) + v2;
            v1 += rd_le(ptr, 8) * k1; ptr += 8; v1 = rotr64(v1, 29) + v3;
            v2 += rd_le(ptr, 8) * k2; ptr += 8; v2 = rotr64(v2, 29) + v0;
            v3 += rd_le(ptr, 8) * k3; ptr += 8; v3 = rotr64(v3, 29) + v1;
        } while (ptr <= end - 32);
        v2 ^= rotr64(((v0 + v3) * k0) + v1, 37) * k1;
        v3 ^= rotr64(((v1 + v2) * k1) + v0, 37) * k0;
        v0 ^= rotr64(((v0 + v2) * k0) + v3, 37) * k1;
        v1 ^= rotr64(((v1 + v3) * k1) + v2, 37) * k0;
        hash += v0 ^ v1;
    }
    if (end - ptr >= 16) {
        uint64_t v0 = hash + rd_le(ptr, 8) * k2; ptr += 8; v0 = rotr64(v0, 29) * k3;
        uint64_t v1 = hash + rd_le(ptr, 8) * k2; ptr += 8; v1 = rotr64(v1, 29) * k3;
        v0 ^= rotr64(v0 * k0, 21) + v1;
        v1 ^= rotr64(v1 * k3, 21) + v0;
        hash += v1;
    }
    if (end - ptr >= 8) { hash += rd_le(ptr, 8) * k3; ptr += 8; hash ^= rotr64(hash, 55) * k1; }
    if (end - ptr >= 4) { hash += rd_le(ptr, 4) * k3; ptr += 4; hash ^= rotr64(hash, 26) * k1; }
    if (end - ptr >= 2) { hash += rd_le(ptr, 2) * k3; ptr += 2; hash ^= rotr64(hash, 48) * k1; }
    if (end - ptr >= 1) { hash += rd_le(ptr, 1) * k3; hash ^= rotr64(hash, 37) * k1; }
    hash ^= rotr64(hash, 28);
    hash *= k0;
    hash ^= rotr64(hash, 29);
    return hash;
}

/* loglogbeta.go AddHash: k = top 14 bits; val = leading zeros of the rest (a guard of 14 one-bits below it) + 1 */
void orc_llb_add_hash(uint8_t *regs, uint64_t x) {
    const uint64_t k = x >> (64 - ORC_LLB_P);
    const uint64_t rest = (x << ORC_LLB_P) ^ (UINT64_MAX >> (64 - ORC_LLB_P));
    const uint8_t val = (uint8_t)(__builtin_clzll(rest) + 1); /* rest != 0: its low 14 bits are ones */
    if (regs[k] < val) regs[k] = val;
}

void orc_llb_add(uint8_t *regs, const uint8_t *value, int64_t len) { orc_llb_add_hash(regs, orc_metro64(value, len, 1337)); }

void orc_llb_merge(uint8_t *regs, const uint8_t *other) {
    for (int i = 0; i < ORC_LLB_M; i++)
        if (regs[i] < other[i]) regs[i] = other[i];
}

/* loglogbeta.go Cardinality: alpha m (m - ez) / (beta(ez) + sum 2^-reg), registers summed in index order */
uint64_t orc_llb_cardinality(const uint8_t *regs) {
    const double m = (double)ORC_LLB_M;
    const double alpha = 0.7213 / (1.0 + 1.079 / m);
    double sum = 0.0, ez = 0.0;
    for (int i = 0; i < ORC_LLB_M; i++) {
        if (regs[i] == 0) ez += 1.0;
        sum += 1.0 / pow(2.0, (double)regs[i]);
    }
    const double zl = log(ez + 1.0);
    const double beta = -0.370393911 * ez + 0.070471823 * zl + 0.17393686 * pow(zl, 2) + 0.16339839 * pow(zl, 3) +
                        -0.09237745 * pow(zl, 4) + 0.03738027 * pow(zl, 5) + -0.005384159 * pow(zl, 6) + 0.00042419 * pow(zl, 7);
    return (uint64_t)(alpha * m * (m - ez) / (beta + sum));
}

/* aggregate.go:205-243: the row's distinct value goes into the Result's sketch */
static void distinct_add(const orc_query *q, const orc_col *cols, int64_t row, orc_result *r) {
    if (!r->llb) r->llb = (uint8_t *)calloc(ORC_LLB_M, 1);
    int only_ints = 1; /* :84-91, by column type */
    for (int g = 0; g < q->n_distincts; g++)
        if (cols[q->distinct_cols[g]].type != ORC_INT_VAL) only_ints = 0;
    if (only_ints) { /* :208-222 fast path: 8 bytes per column, little endian, MISSING_VALUE when unpopulated */
        uint8_t buf[ORC_GROUP_BY_WIDTH * ORC_MAX_GROUPS];
        for (int g = 0; g < q->n_distincts; g++) {
            const orc_col *c = &cols[q->distinct_cols[g]];
            const uint64_t v = col_populated(c, row) ? (uint64_t)c->ints[row] : UINT64_MAX;
            for (int b = 0; b < 8; b++) buf[g * 8 + b] = (uint8_t)(v >> (8 * b));
        }
        orc_llb_add(r->llb, buf, (int64_t)q->n_distincts * ORC_GROUP_BY_WIDTH);
        return;
    }
    /* :224-239 slow path: decimal ints / dictionary strings, each followed by GROUP_DELIMITER ("\t") */
    size_t cap = 64, n = 0;
    char *buf = (char *)malloc(cap);
    for (int g = 0; g < q->n_distincts; g++) {
        const orc_col *c = &cols[q->distinct_cols[g]];
        char num[32];
        const char *piece = "";
        if (col_populated(c, row)) {
            if (c->type == ORC_INT_VAL) {
                snprintf(num, sizeof(num), "%lld", (long long)c->ints[row]);
                piece = num;
            } else if (c->type == ORC_STR_VAL) {
                const int64_t id = c->strs[row];
                if (q->distinct_dicts[g] && id >= 0 && id < q->distinct_dict_len[g]) piece = q->distinct_dicts[g][id];
            }
        }
        const size_t len = strlen(piece);
        if (n + len + 1 > cap) {
            while (n + len + 1 > cap) cap *= 2;
            buf = (char *)realloc(buf, cap);
        }
        memcpy(buf + n, piece, len);
        n += len;
        buf[n++] = '\t';
    }
    orc_llb_add(r->llb, (const uint8_t *)buf, (int64_t)n);
    free(buf);
}

/* ------------------------------------------------------------------ */
/* Filters (filter.go)                                                  */
/* ------------------------------------------------------------------ */


static int filter_row(const orc_filter *f, const orc_col *cols, int64_t row) {
    const orc_col *c = &cols[f->col];
    if (c->type == ORC_INT_VAL) { /* filter.go:171-195 */
        if (!col_populated(c, row)) return 0;
        int64_t field = c->ints[row];
        switch (f->op) {
        case ORC_OP_GT: return field > f->value;
        case ORC_OP_LT: return field < f->value;
        case ORC_OP_EQ: return field == f->value;
        case ORC_OP_NEQ: return field != f->value;
        default: return 0;
        }
    } else if (c->type == ORC_STR_VAL) { /* filter.go:199-250 */
        if (!col_populated(c, row)) return 0;
        int64_t val = c->strs[row];
        switch (f->op) {
        case ORC_OP_NRE:
        case ORC_OP_RE: {
            int ret = (f->idtable && val >= 0 && val < f->idtable_len) ? f->idtable[val] != 0 : 0;
            return f->op == ORC_OP_NRE ? !ret : ret;
        }
        case ORC_OP_EQ: return val == f->value;
        case ORC_OP_NEQ: return val != f->value;
        default: return 0;
        }
    } else if (c->type == ORC_SET_VAL) { /* filter.go:252-285 */
        if (!col_populated(c, row)) return 0;
        int64_t lo = c->set_off[row], hi = c->set_off[row + 1];
        switch (f->op) {
        case ORC_OP_IN:
            for (int64_t i = lo; i < hi; i++)
                if (c->set_vals[i] == f->value) return 1;
            return 0;
        case ORC_OP_NIN:
            for (int64_t i = lo; i < hi; i++)
                if (c->set_vals[i] == f->value) return 0;
            return 1;
        default: return 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* FilterAndAggRecords over one block (aggregate.go:56-282)             */
/* ------------------------------------------------------------------ */

typedef struct {
    rmap results;      /* querySpec.Results */
    rmap time_results; /* querySpec.TimeResults, flattened: key = (bucket, group key) */
    int64_t matched;
    int64_t weight_carry_out;
} block_spec;

/* per time-bucket key count, only to honour INTERNAL_RESULT_LIMIT per map */
typedef struct {
    int64_t tb, n;
} tbcount;

static int64_t *tb_count_ref(tbcount **arr, int64_t *n, int64_t *cap, int64_t tb) {
    for (int64_t i = *n - 1; i >= 0; i--)
        if ((*arr)[i].tb == tb) return &(*arr)[i].n;
    if (*n == *cap) {
        *cap = *cap ? *cap * 2 : 16;
        *arr = (tbcount *)realloc(*arr, (size_t)*cap * sizeof(tbcount));
    }
    (*arr)[*n].tb = tb;
    (*arr)[*n].n = 0;
    (*n)++;
    return &(*arr)[*n - 1].n;
}

static void scan_block(const orc_query *q, const orc_col *cols, int64_t row0, int64_t row1,
                       int64_t weight_in, block_spec *spec) {
    int key_len = ORC_GROUP_BY_WIDTH * q->n_groups;
    rmap_init(&spec->results, key_len);
    rmap_init(&spec->time_results, key_len);
    spec->matched = 0;

    uint8_t keybuf[KEY_BYTES];
    memset(keybuf, 0, sizeof(keybuf));
    int64_t weight = weight_in; /* aggregate.go:68 -- declared outside the row loop */
    int weight_mode = q->weight_col >= 0;
    tbcount *tbc = NULL;
    int64_t tbc_n = 0, tbc_cap = 0;

    for (int64_t i = row0; i < row1; i++) {
        /* :100-102 */
        if (weight_mode && cols[q->weight_col].type == ORC_INT_VAL && col_populated(&cols[q->weight_col], i))
            weight = cols[q->weight_col].ints[i];

        /* :105-116 filters, ANDed, short-circuit */
        int add = 1;
        for (int j = 0; j < q->n_filters; j++) {
            if (!filter_row(&q->filters[j], cols, i)) {
                add = 0;
                break;
            }
        }
        if (!add) continue;
        spec->matched++; /* :117 */

        /* :125-143 group key: 8 LE bytes per group column */
        for (int g = 0; g < q->n_groups; g++) {
            const orc_col *c = &cols[q->group_cols[g]];
            uint64_t v = 0; /* copy(bs, zero): a populated set column leaves zeros */
            if (!col_populated(c, i)) {
                v = UINT64_MAX; /* MISSING_VALUE */
            } else if (c->type == ORC_INT_VAL) {
                v = (uint64_t)c->ints[i];
            } else if (c->type == ORC_STR_VAL) {
                v = (uint64_t)(int64_t)c->strs[i]; /* uint64(r.Strs[..]) sign-extends int32 */
            }
            for (int b = 0; b < 8; b++) keybuf[g * 8 + b] = (uint8_t)(v >> (8 * b));
        }

        rmap *result_map = &spec->results;
        int64_t tb = 0;
        int time_mode = q->time_bucket > 0;
        if (time_mode) { /* :146-183 */
            const orc_col *tc = &cols[q->time_col];
            if (tc->type != ORC_INT_VAL || !col_populated(tc, i)) continue;
            int64_t val = tc->ints[i];
            orc_result *big = rmap_find(&spec->results, keybuf, 0);
            if (!big && spec->results.n < INTERNAL_RESULT_LIMIT) {
                big = result_new(keybuf, key_len, 0);
                rmap_put(&spec->results, big);
            }
            if (big) {
                big->samples++;
                big->count += weight;
            }
            tb = orc_time_bucket(val, q->time_bucket);
            result_map = &spec->time_results;
        }

        /* :186-200 find or create */
        orc_result *r = rmap_find(result_map, keybuf, tb);
        if (!r) {
            if (time_mode) {
                int64_t *cnt = tb_count_ref(&tbc, &tbc_n, &tbc_cap, tb);
                if (*cnt >= INTERNAL_RESULT_LIMIT) continue;
                (*cnt)++;
            } else if (result_map->n >= INTERNAL_RESULT_LIMIT) {
                continue;
            }
            r = result_new(keybuf, key_len, tb);
            rmap_put(result_map, r);
        }
        r->samples++; /* :202-203 */
        r->count += weight;
        if (q->n_distincts > 0) distinct_add(q, cols, i, r); /* :205-243 */

        /* :246-261 aggregations */
        for (int a = 0; a < q->n_aggs; a++) {
            const orc_col *c = &cols[q->aggs[a].col];
            if (c->type != ORC_INT_VAL || !col_populated(c, i)) continue;
            if (!r->hists[a])
                r->hists[a] = q->loghist ? orc_hist_new_multi(q->aggs[a].info_min, q->aggs[a].info_max, q->op, q->hist_bucket, weight_mode)
                                         : orc_hist_new(q->aggs[a].info_min, q->aggs[a].info_max, q->op, q->hist_bucket, weight_mode);
            orc_hist_add(r->hists[a], c->ints[i], weight);
        }
    }
    free(tbc);
    spec->weight_carry_out = weight;
}

/* table_block_io.go:110-182 with exact block min/max.  min_record/max_record carry the
 * block's IntInfo; a gt/lt filter that fails on BOTH extremes, or an eq value outside
 * [min,max], skips the block.  A filter column with no populated row in the block has no
 * IntInfoMap entry, so Filter() is false on both pseudo-records and the block is skipped
 * (the reference loads it only when the block holds no int value at all, in which case
 * every row fails the filter anyway -- results are identical). */
static int should_load_block(const orc_query *q, const orc_col *cols, int64_t row0, int64_t row1) {
    if (row1 <= row0) return 1;
    for (int j = 0; j < q->n_filters; j++) {
        const orc_filter *f = &q->filters[j];
        const orc_col *c = &cols[f->col];
        if (c->type != ORC_INT_VAL) continue;
        if (f->op != ORC_OP_GT && f->op != ORC_OP_LT && f->op != ORC_OP_EQ) continue;
        int64_t mn = INT64_MAX, mx = INT64_MIN;
        int pop = 0;
        for (int64_t i = row0; i < row1; i++) {
            if (!col_populated(c, i)) continue;
            pop = 1;
            if (c->ints[i] < mn) mn = c->ints[i];
            if (c->ints[i] > mx) mx = c->ints[i];
        }
        if (!pop) return 0;
        if (f->op == ORC_OP_GT && !(mn > f->value) && !(mx > f->value)) return 0;
        if (f->op == ORC_OP_LT && !(mn < f->value) && !(mx < f->value)) return 0;
        if (f->op == ORC_OP_EQ && (mn > f->value || mx < f->value)) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ */
/* CombineResults over blocks, in block index order (aggregate.go:414-467) */
/* ------------------------------------------------------------------ */

struct orc_results {
    orc_query q;
    rmap results, time_results;
    orc_result *cumulative;
    int64_t matched, blocks_scanned, blocks_skipped;
    orc_result **sorted[2]; /* canonical order views */
};

static void merge_block(orc_results *R, block_spec *spec) {
    const orc_query *q = &R->q;
    int key_len = ORC_GROUP_BY_WIDTH * q->n_groups;
    R->matched += spec->matched;
    /* master_result.Combine(&spec.Results); cumulative_result.Combine(result) */
    for (int64_t i = 0; i < spec->results.n; i++) {
        orc_result *r = spec->results.items[i];
        result_combine(R->cumulative, r, q);
        orc_result *m = rmap_find(&R->results, r->key, 0);
        if (!m) {
            rmap_put(&R->results, r); /* adopted by pointer */
            spec->results.items[i] = NULL;
        } else {
            result_combine(m, r, q);
        }
    }
    (void)key_len;
    for (int64_t i = 0; i < spec->time_results.n; i++) {
        orc_result *r = spec->time_results.items[i];
        orc_result *m = rmap_find(&R->time_results, r->key, r->time_bucket);
        if (!m) {
            rmap_put(&R->time_results, r);
            spec->time_results.items[i] = NULL;
        } else {
            result_combine(m, r, q);
        }
    }
    rmap_free(&spec->results, 1);
    rmap_free(&spec->time_results, 1);
}

static int g_cmp_groups;
static int cmp_result(const void *pa, const void *pb) {
    const orc_result *a = *(orc_result *const *)pa, *b = *(orc_result *const *)pb;
    if (a->time_bucket != b->time_bucket) return a->time_bucket < b->time_bucket ? -1 : 1;
    for (int g = 0; g < g_cmp_groups; g++) {
        uint64_t x = 0, y = 0;
        for (int k = 7; k >= 0; k--) {
            x = (x << 8) | a->key[g * 8 + k];
            y = (y << 8) | b->key[g * 8 + k];
        }
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

orc_results *orc_query_run(const orc_query *q, const orc_col *cols, int32_t ncols, int64_t nrows) {
    (void)ncols;
    orc_results *R = (orc_results *)calloc(1, sizeof(*R));
    R->q = *q;
    int key_len = ORC_GROUP_BY_WIDTH * q->n_groups;
    rmap_init(&R->results, key_len);
    rmap_init(&R->time_results, key_len);
    R->cumulative = result_new(NULL, 0, 0);

    int64_t block_rows = q->block_rows > 0 ? q->block_rows : 65536;
    int64_t nblocks = (nrows + block_rows - 1) / block_rows;
    int nthreads = q->n_threads > 0 ? q->n_threads : 1;
    /* the weight-inheritance quirk (aggregate.go:68) makes blocks order dependent only
     * through a weight column with unpopulated rows; each block starts from weight 1
     * because every block is its own FilterAndAggRecords call. */
    int64_t batch = nthreads * 4;
    block_spec *specs = (block_spec *)calloc((size_t)batch, sizeof(block_spec));
    uint8_t *live = (uint8_t *)calloc((size_t)batch, 1);
    for (int64_t b0 = 0; b0 < nblocks; b0 += batch) {
        int64_t bn = nblocks - b0 < batch ? nblocks - b0 : batch;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
        for (int64_t k = 0; k < bn; k++) {
            int64_t row0 = (b0 + k) * block_rows;
            int64_t row1 = row0 + block_rows < nrows ? row0 + block_rows : nrows;
            live[k] = 0;
            if (q->block_skip && !should_load_block(q, cols, row0, row1)) continue;
            scan_block(q, cols, row0, row1, 1, &specs[k]);
            live[k] = 1;
        }
        for (int64_t k = 0; k < bn; k++) {
            if (!live[k]) {
                R->blocks_skipped++;
                continue;
            }
            R->blocks_scanned++;
            merge_block(R, &specs[k]);
        }
    }
    free(specs);
    free(live);

    for (int w = 0; w < 2; w++) {
        rmap *m = w == 0 ? &R->results : &R->time_results;
        R->sorted[w] = (orc_result **)malloc((size_t)(m->n ? m->n : 1) * sizeof(orc_result *));
        if (m->n) memcpy(R->sorted[w], m->items, (size_t)m->n * sizeof(orc_result *));
        g_cmp_groups = q->n_groups;
        qsort(R->sorted[w], (size_t)m->n, sizeof(orc_result *), cmp_result);
    }
    return R;
}

void orc_results_free(orc_results *R) {
    if (!R) return;
    rmap_free(&R->results, 1);
    rmap_free(&R->time_results, 1);
    result_free(R->cumulative);
    free(R->sorted[0]);
    free(R->sorted[1]);
    free(R);
}

int64_t orc_matched_count(const orc_results *r) { return r->matched; }
int64_t orc_blocks_scanned(const orc_results *r) { return r->blocks_scanned; }
int64_t orc_blocks_skipped(const orc_results *r) { return r->blocks_skipped; }

int64_t orc_num_results(const orc_results *r, int which) {
    if (which == 0) return r->results.n;
    if (which == 1) return r->time_results.n;
    return 1;
}

static const orc_result *get_result(const orc_results *R, int which, int64_t idx) {
    if (which == 2) return idx == 0 ? R->cumulative : NULL;
    if (which != 0 && which != 1) return NULL;
    const rmap *m = which == 0 ? &R->results : &R->time_results;
    if (idx < 0 || idx >= m->n) return NULL;
    return R->sorted[which][idx];
}

int orc_result_get(const orc_results *R, int which, int64_t idx, uint8_t *key, int64_t *time_bucket,
                   int64_t *count, int64_t *samples) {
    const orc_result *r = get_result(R, which, idx);
    if (!r) return -1;
    if (key) memcpy(key, r->key, (size_t)(ORC_GROUP_BY_WIDTH * R->q.n_groups));
    if (time_bucket) *time_bucket = r->time_bucket;
    if (count) *count = r->count;
    if (samples) *samples = r->samples;
    return 0;
}

int64_t orc_result_distinct(const orc_results *R, int which, int64_t idx, uint8_t *regs_out) {
    const orc_result *r = get_result(R, which, idx);
    if (!r) return -1;
    static const uint8_t empty[ORC_LLB_M] = {0};
    const uint8_t *regs = r->llb ? r->llb : empty; /* NewResult: hll.New() */
    if (regs_out) memcpy(regs_out, regs, ORC_LLB_M);
    return (int64_t)orc_llb_cardinality(regs);
}

int orc_result_hist(const orc_results *R, int which, int64_t idx, int agg, orc_hist_info *out) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs) return -1;
    if (!r->hists[agg]) {
        memset(out, 0, sizeof(*out));
        return 0;
    }
    orc_hist_info_get(r->hists[agg], out);
    return 0;
}

int64_t orc_result_hist_values(const orc_results *R, int which, int64_t idx, int agg, int64_t *out, int64_t cap) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    return orc_hist_values(r->hists[agg], out, cap);
}

/* every outlier and underlier value of the hist, merged across blocks (the "exact" side-state), ascending */
static int cmp_i64(const void *a, const void *b) {
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}
int64_t orc_result_outliers(const orc_results *R, int which, int64_t idx, int agg, int64_t *out, int64_t cap) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    const orc_hist *h = r->hists[agg];
    int64_t n = h->all_outliers.n + h->all_underliers.n;
    for (int i = 0; i < h->n_sub; i++) n += h->sub[i]->all_outliers.n + h->sub[i]->all_underliers.n;
    if (cap < n) return n;
    int64_t at = 0;
    for (int i = -1; i < h->n_sub; i++) {
        const orc_hist *s = i < 0 ? h : h->sub[i];
        if (s->all_outliers.n) memcpy(out + at, s->all_outliers.v, (size_t)s->all_outliers.n * sizeof(int64_t));
        at += s->all_outliers.n;
        if (s->all_underliers.n) memcpy(out + at, s->all_underliers.v, (size_t)s->all_underliers.n * sizeof(int64_t));
        at += s->all_underliers.n;
    }
    qsort(out, (size_t)n, sizeof(int64_t), cmp_i64);
    return n;
}

int orc_result_n_sub(const orc_results *R, int which, int64_t idx, int agg) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    return orc_hist_n_sub(r->hists[agg]);
}
int orc_result_sub(const orc_results *R, int which, int64_t idx, int agg, int k, int64_t *out6) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    return orc_hist_sub(r->hists[agg], k, out6);
}
int64_t orc_result_sparse(const orc_results *R, int which, int64_t idx, int agg, int64_t *keys, int64_t *counts, int64_t cap) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    return orc_hist_sparse(r->hists[agg], keys, counts, cap);
}

int orc_result_percentiles(const orc_results *R, int which, int64_t idx, int agg, int64_t *out100) {
    const orc_result *r = get_result(R, which, idx);
    if (!r || agg < 0 || agg >= R->q.n_aggs || !r->hists[agg]) return -1;
    return orc_hist_percentiles(r->hists[agg], out100);
}

/* ------------------------------------------------------------------ */
/* Synthetic table generator (ours; mirrored bit-for-bit by the HIP generator) */
/* ------------------------------------------------------------------ */

uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static inline uint64_t mulhi64(uint64_t a, uint64_t b) { return (uint64_t)(((unsigned __int128)a * b) >> 64); }

void orc_synth_fill(int kind, int64_t a, int64_t b, uint64_t seed, int32_t col_index,
                    int64_t row0, int64_t n, int64_t total_rows, int64_t *out) {
    uint64_t cs = seed ^ ((uint64_t)(col_index + 1) * 0x9E3779B97F4A7C15ull);
    for (int64_t k = 0; k < n; k++) {
        uint64_t i = (uint64_t)(row0 + k);
        int64_t v;
        if (kind == ORC_SYN_TIME) {
            /* a + floor(i*b/N); i*b is required to stay below 2^64 (true for 2^33 rows x 30 days) */
            v = a + (int64_t)((i * (uint64_t)b) / (uint64_t)total_rows);
        } else if (kind == ORC_SYN_BELL) {
            uint64_t h = orc_splitmix64(cs ^ i);
            uint64_t s = 0;
            /* four 16-bit lanes of one hash, each scaled to [0,b) */
            for (int j = 0; j < 4; j++) s += (((h >> (16 * j)) & 0xFFFFu) * (uint64_t)b) >> 16;
            v = a + (int64_t)s;
        } else {
            uint64_t h = orc_splitmix64(cs ^ i);
            v = a + (int64_t)mulhi64(h, (uint64_t)b); /* unbiased-enough range reduction */
        }
        out[k] = v;
    }
}

/* ------------------------------------------------------------------ */
/* columnar CPU baseline (see sybil_oracle.h)                           */
/* ------------------------------------------------------------------ */
int64_t orc_columnar_scan(int64_t nrows, int32_t nf, const int64_t *const *fcols, const int64_t *lo, const int64_t *hi,
                          int32_t ng, const int64_t *const *gcols, const int64_t *gmin, const int64_t *gcard, int32_t na,
                          const int64_t *const *acols, const int64_t *hmin, const int64_t *bucket_size, int32_t n_threads,
                          int64_t *out) {
    int64_t cells = 1;
    for (int g = 0; g < ng; g++) cells *= gcard[g];
    const int64_t fields = 1 + 3 * (int64_t)na, words = fields * cells;
    memset(out, 0, (size_t)words * sizeof(int64_t));
    int64_t matched = 0;
    int bad = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel reduction(+ : matched) reduction(| : bad)
    {
        int64_t *tab = (int64_t *)calloc((size_t)words, sizeof(int64_t));
#pragma omp for schedule(static)
        for (int64_t blk = 0; blk < (nrows + 65535) / 65536; blk++) {
            const int64_t r0 = blk * 65536, r1 = r0 + 65536 < nrows ? r0 + 65536 : nrows;
            for (int64_t r = r0; r < r1; r++) {
                int pass = 1;
                for (int f = 0; f < nf; f++) pass &= fcols[f][r] >= lo[f] && fcols[f][r] <= hi[f];
                if (!pass) continue;
                matched++;
                int64_t cell = 0;
                for (int g = 0; g < ng; g++) {
                    const int64_t d = gcols[g][r] - gmin[g];
                    if (d < 0 || d >= gcard[g]) bad = 1;
                    cell = cell * gcard[g] + d;
                }
                if (bad) continue;
                tab[cell] += 1;
                for (int a = 0; a < na; a++) {
                    const int64_t v = acols[a][r], b = (v - hmin[a]) / bucket_size[a];
                    tab[(1 + 3 * a) * cells + cell] += v;
                    tab[(2 + 3 * a) * cells + cell] += b;
                    tab[(3 + 3 * a) * cells + cell] += b * b;
                }
            }
        }
#pragma omp critical
        for (int64_t i = 0; i < words; i++) out[i] += tab[i];
        free(tab);
    }
    return bad ? -1 : matched;
}

/* ---- full-size checker: generator + direct-mapped row loop, one block at a time (see sybil_oracle.h) ---- */
int64_t orc_synth_scan(const orc_synth_query *q, int64_t *out_fields, int64_t *out_hist) {
    int64_t gcells = 1;
    for (int g = 0; g < q->ng; g++) gcells *= q->gcard[g];
    const int64_t cells = gcells * (q->has_time ? q->n_tb : 1);
    const int64_t fields = 1 + 3 * (int64_t)q->na, words = fields * cells;
    memset(out_fields, 0, (size_t)words * sizeof(int64_t));
    if (out_hist) memset(out_hist, 0, (size_t)(cells * q->na * q->nv_max) * sizeof(int64_t));
    int64_t matched = 0;
    int bad = 0;
    const int64_t B = 65536;
    const int64_t nblk = (q->nrows + B - 1) / B;
#ifdef _OPENMP
    if (q->n_threads > 0) omp_set_num_threads(q->n_threads);
#endif
#pragma omp parallel reduction(+ : matched) reduction(| : bad)
    {
        int64_t *tab = (int64_t *)calloc((size_t)words, sizeof(int64_t));
        const int ncol = q->nf + q->ng + q->na + (q->has_time ? 1 : 0);
        int64_t *buf = (int64_t *)malloc((size_t)(ncol > 0 ? ncol : 1) * (size_t)B * sizeof(int64_t));
        const int64_t *fc[4], *gc[4], *ac[4], *tc = NULL;
#pragma omp for schedule(dynamic, 4)
        for (int64_t blk = 0; blk < nblk; blk++) {
            const int64_t r0 = blk * B, n = r0 + B < q->nrows ? B : q->nrows - r0;
            int k = 0;
            for (int f = 0; f < q->nf; f++, k++) {
                orc_synth_fill(q->fcol[f].kind, q->fcol[f].a, q->fcol[f].b, q->seed, q->fcol[f].col_index, q->row0 + r0, n,
                               q->total_rows, buf + (int64_t)k * B);
                fc[f] = buf + (int64_t)k * B;
            }
            for (int g = 0; g < q->ng; g++, k++) {
                orc_synth_fill(q->gcol[g].kind, q->gcol[g].a, q->gcol[g].b, q->seed, q->gcol[g].col_index, q->row0 + r0, n,
                               q->total_rows, buf + (int64_t)k * B);
                gc[g] = buf + (int64_t)k * B;
            }
            for (int a = 0; a < q->na; a++, k++) {
                orc_synth_fill(q->acol[a].kind, q->acol[a].a, q->acol[a].b, q->seed, q->acol[a].col_index, q->row0 + r0, n,
                               q->total_rows, buf + (int64_t)k * B);
                ac[a] = buf + (int64_t)k * B;
            }
            if (q->has_time) {
                orc_synth_fill(q->tcol.kind, q->tcol.a, q->tcol.b, q->seed, q->tcol.col_index, q->row0 + r0, n, q->total_rows,
                               buf + (int64_t)k * B);
                tc = buf + (int64_t)k * B;
            }
            for (int64_t r = 0; r < n; r++) {
                int pass = 1;
                for (int f = 0; f < q->nf; f++) pass &= fc[f][r] >= q->lo[f] && fc[f][r] <= q->hi[f];
                if (!pass) continue;
                matched++;
                int64_t cell = 0;
                int oob = 0;
                if (q->has_time) {
                    const int64_t tb = orc_time_bucket(tc[r], q->time_bucket) / q->time_bucket - q->tb_min;
                    if (tb < 0 || tb >= q->n_tb) oob = 1;
                    cell = tb;
                }
                for (int g = 0; g < q->ng; g++) {
                    const int64_t d = gc[g][r] - q->gmin[g];
                    if (d < 0 || d >= q->gcard[g]) oob = 1;
                    cell = cell * q->gcard[g] + d;
                }
                if (oob) {
                    bad = 1;
                    continue;
                }
                tab[cell] += 1;
                for (int a = 0; a < q->na; a++) {
                    const int64_t v = ac[a][r];
                    int64_t b = (v - q->hmin[a]) / q->bucket_size[a];
                    if (b >= q->n_values[a]) b = q->n_values[a] - 1; /* hist_basic.go:132-142 */
                    if (b < 0) b = 0;
                    tab[(1 + 3 * a) * cells + cell] += v;
                    tab[(2 + 3 * a) * cells + cell] += b;
                    tab[(3 + 3 * a) * cells + cell] += b * b;
                    if (out_hist) __atomic_fetch_add(&out_hist[(cell * q->na + a) * q->nv_max + b], 1, __ATOMIC_RELAXED);
                }
            }
        }
#pragma omp critical
        for (int64_t i = 0; i < words; i++) out_fields[i] += tab[i];
        free(tab);
        free(buf);
    }
    return bad ? -1 : matched;
}
