/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the reference's
 * selector arithmetic.  Nothing under oracle/ is imported, linked or executed
 * by the product path; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * Parity status: PINNED for selected-index lists (tests/golden/selector_*.npz,
 * produced by running the reference's own selector classes in the build
 * container, oracle/gen_golden_selectors.py).  The f64/f32 distance maps agree
 * with the reference's captured maps to <= 1 ulp (f64) / <= 4 ulp (f32): the
 * reference normalises with numpy's exp, whose bits depend on the host CPU
 * (AVX512F Tang-style kernel in the build container, glibc elsewhere), so the
 * reference itself does not define the last bit.  This file uses one
 * deterministic table-driven exp (explicit fma, no contraction) that the HIP
 * path restates instruction for instruction, so HIP == oracle bit for bit.
 *
 * Each function cites the reference lines it follows (paths relative to
 * /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "al3d_exp_table.h"

#define AL3D_MARGIN 1e6 /* spatial_temporal_selector.py:111 */

static const uint64_t k_exp_tab[128][2] = AL3D_EXP_TABLE_INIT;

static inline double u2d(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

/* exp(x), |err| < 0.51 ulp.  x = (128 m + j) ln2/128 + r, exp(x) = 2^m * T[j] * e^r. */
double al3d_oracle_exp_f64(double x)
{
    if (x != x) return x;
    if (x > 709.782712893384) return INFINITY;
    if (x < -745.1332191019412) return 0.0;
    double kd = rint(x * AL3D_EXP_INV_LN2N);
    int64_t k = (int64_t)kd;
    double r = fma(kd, -AL3D_EXP_LN2N_HI, x);
    r = fma(kd, -AL3D_EXP_LN2N_LO, r);
    int64_t j = k & 127, m = k >> 7;
    /* e^r - 1 = r + r^2/2 + r^3/6 + r^4/24 + r^5/120, |r| <= ln2/256 */
    double r2 = r * r;
    double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r2, r);
    double th = u2d(k_exp_tab[j][0]), tl = u2d(k_exp_tab[j][1]);
    double res = th + fma(th, p, tl);
    return ldexp(res, (int)m);
}

/* f32 exp = f64 exp rounded once more (feature term, numpy f32 in the reference). */
float al3d_oracle_exp_f32(float x) { return (float)al3d_oracle_exp_f64((double)x); }

void al3d_oracle_exp_f64_array(const double* x, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = al3d_oracle_exp_f64(x[i]);
}

/* ego XY: location = -(cal[:3,3].T @ cal[:3,:3]) (spatial_temporal_selector.py:83-89).
 * numpy evaluates the 3-term dot products in index order k=0,1,2. */
void al3d_oracle_ego_xy(const double* car_from_global, int64_t n, double* xy)
{
    for (int64_t i = 0; i < n; ++i) {
        const double* c = car_from_global + 16 * i;
        for (int col = 0; col < 2; ++col) {
            double s = c[0 * 4 + 3] * c[0 * 4 + col];
            s += c[1 * 4 + 3] * c[1 * 4 + col];
            s += c[2 * 4 + 3] * c[2 * 4 + col];
            xy[2 * i + col] = -s;
        }
    }
}

/* exact k-nearest neighbours incl. self, ascending (distance, index); the
 * reference uses scipy cKDTree.query(k+1) (spatial_temporal_selector.py:97-98):
 * d = sqrt(dx*dx + dy*dy), sums in dimension order. */
void al3d_oracle_knn(const double* xy, int64_t n, int kq, double* knn_d, int64_t* knn_i)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double* bd = knn_d + (int64_t)kq * i;
        int64_t* bi = knn_i + (int64_t)kq * i;
        int cnt = 0;
        for (int64_t j = 0; j < n; ++j) {
            double dx = xy[2 * j] - xy[2 * i], dy = xy[2 * j + 1] - xy[2 * i + 1];
            double d2 = dx * dx;
            d2 += dy * dy;
            if (cnt == kq && !(d2 < bd[kq - 1])) continue;
            int pos = cnt < kq ? cnt : kq - 1;
            while (pos > 0 && bd[pos - 1] > d2) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
            bd[pos] = d2; bi[pos] = j;
            if (cnt < kq) ++cnt;
        }
        for (int t = 0; t < cnt; ++t) bd[t] = sqrt(bd[t]);
        for (int t = cnt; t < kq; ++t) { bd[t] = INFINITY; bi[t] = n; } /* scipy pads with inf / n */
    }
}

/* symmetric kNN edge list in CSR form, restating the dense assignment loop
 * spatial_temporal_selector.py:95-101: W[a][b] = W[b][a] = d for b in knn(a);
 * zero weight == no edge (csgraph drops explicit zeros of a dense input).
 * d(a,b) and d(b,a) are the same bits (dx*dx is sign-symmetric), so "later
 * assignment wins" needs no ordering.  Returns nnz; indices/weights need room
 * for 2*n*kq entries. */
typedef struct { int64_t u, v; double d; } edge_t;

static int edge_cmp(const void* pa, const void* pb)
{
    const edge_t* a = (const edge_t*)pa; const edge_t* b = (const edge_t*)pb;
    if (a->u != b->u) return a->u < b->u ? -1 : 1;
    if (a->v != b->v) return a->v < b->v ? -1 : 1;
    return 0;
}

int64_t al3d_oracle_knn_csr(const double* knn_d, const int64_t* knn_i, int64_t n, int kq,
                            int64_t* indptr, int64_t* indices, double* weights)
{
    edge_t* e = (edge_t*)malloc(sizeof(edge_t) * (size_t)(2 * n * kq + 1));
    int64_t m = 0;
    for (int64_t a = 0; a < n; ++a)
        for (int t = 0; t < kq; ++t) {
            int64_t b = knn_i[(int64_t)kq * a + t];
            double d = knn_d[(int64_t)kq * a + t];
            if (b >= n || d == 0.0 || !(d < INFINITY)) continue;
            e[m].u = a; e[m].v = b; e[m].d = d; ++m;
            e[m].u = b; e[m].v = a; e[m].d = d; ++m;
        }
    qsort(e, (size_t)m, sizeof(edge_t), edge_cmp);
    int64_t nnz = 0, u = 0;
    indptr[0] = 0;
    for (int64_t q = 0; q < m; ++q) {
        if (q > 0 && e[q].u == e[q - 1].u && e[q].v == e[q - 1].v) continue;
        while (u < e[q].u) indptr[++u] = nnz;
        indices[nnz] = e[q].v; weights[nnz] = e[q].d; ++nnz;
    }
    while (u < n) indptr[++u] = nnz;
    free(e);
    return nnz;
}

/* all-pairs Dijkstra (binary heap) over the CSR graph; the reference calls
 * scipy.sparse.csgraph.shortest_path(directed=False, method="D")
 * (spatial_temporal_selector.py:103-104).  dist[v] = dist[u] + w in f64,
 * unreachable = inf.  Rows [row0,row1) only (so callers can thread it). */
typedef struct { double d; int64_t v; } heap_item;

static void heap_push(heap_item* h, int64_t* hn, double d, int64_t v)
{
    int64_t i = (*hn)++;
    while (i > 0) {
        int64_t p = (i - 1) >> 1;
        if (h[p].d <= d) break;
        h[i] = h[p]; i = p;
    }
    h[i].d = d; h[i].v = v;
}

static heap_item heap_pop(heap_item* h, int64_t* hn)
{
    heap_item top = h[0];
    heap_item last = h[--(*hn)];
    int64_t i = 0, n = *hn;
    for (;;) {
        int64_t c = 2 * i + 1;
        if (c >= n) break;
        if (c + 1 < n && h[c + 1].d < h[c].d) ++c;
        if (last.d <= h[c].d) break;
        h[i] = h[c]; i = c;
    }
    if (n > 0) h[i] = last;
    return top;
}

void al3d_oracle_apsp(const int64_t* indptr, const int64_t* indices, const double* weights,
                      int64_t n, int64_t row0, int64_t row1, double* out)
{
    int64_t nnz = indptr[n];
#pragma omp parallel
    {
    heap_item* heap = (heap_item*)malloc(sizeof(heap_item) * (size_t)(nnz + n + 1));
#pragma omp for schedule(dynamic, 8)
    for (int64_t s = row0; s < row1; ++s) {
        double* dist = out + (s - row0) * n;
        for (int64_t i = 0; i < n; ++i) dist[i] = INFINITY;
        dist[s] = 0.0;
        int64_t hn = 0;
        heap_push(heap, &hn, 0.0, s);
        while (hn > 0) {
            heap_item it = heap_pop(heap, &hn);
            if (it.d > dist[it.v]) continue;
            for (int64_t e = indptr[it.v]; e < indptr[it.v + 1]; ++e) {
                double nd = it.d + weights[e];
                int64_t v = indices[e];
                if (nd < dist[v]) { dist[v] = nd; heap_push(heap, &hn, nd, v); }
            }
        }
    }
    free(heap);
    }
}

/* EuSpatialSelector map (euclidean_spatial_selector.py:95-106):
 * sqrt(((loc - loc[i])**2).sum(1)); frames of another map location get 1e6. */
void al3d_oracle_euclid_map(const double* xy, const int64_t* loc_id, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double dx = xy[2 * j] - xy[2 * i], dy = xy[2 * j + 1] - xy[2 * i + 1];
            double d2 = dx * dx;
            double dy2 = dy * dy;
            d2 = d2 + dy2;
            out[i * n + j] = loc_id[j] == loc_id[i] ? sqrt(d2) : AL3D_MARGIN;
        }
}

/* temporal gap between frames i and j: |i-j| inside one consecutive run of the
 * same logfile else 1e6 (spatial_temporal_selector.py:109-133).  TemporalSelector
 * (temporal_selector.py:49-63) groups by logfile *name* instead; callers pass the
 * matching id array. */
static inline double temporal_gap(const int64_t* id, int64_t i, int64_t j)
{
    if (id[i] != id[j]) return AL3D_MARGIN;
    return (double)(i > j ? i - j : j - i);
}

void al3d_oracle_temporal_map(const int64_t* id, int64_t n, double* out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < n; ++j) out[i * n + j] = temporal_gap(id, i, j);
}

/* max_temporal_distance quirk: only updated when the logfile changes, so the
 * last run is ignored (spatial_temporal_selector.py:117-129). */
int64_t al3d_oracle_max_temporal_distance(const int64_t* run_id, int64_t n)
{
    int64_t best = 0, count = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || run_id[i] == run_id[i - 1]) count++;
        else { if (count > best) best = count; count = 1; }
    }
    return best;
}

double al3d_oracle_max_finite(const double* a, int64_t n)
{
    double m = -INFINITY;
    for (int64_t i = 0; i < n; ++i) if (a[i] != INFINITY && a[i] > m) m = a[i];
    return m;
}

/* normalise + aggregate (spatial_temporal_selector.py:135-155,
 * spatial_temporal_feature_selector.py:208-219, spatial_feature_selector.py:188-197).
 *   normalize: 0 = none, 1 = exp (1-exp(-x)), 2 = linear (x / scale)
 *   aggregate: 0 = sum  (S + lambda_t*T) + (float)lambda_f*F
 *              1 = min(S, T or F), 2 = max(S, T or F)
 * spatial may be NULL (term absent), temporal id may be NULL, feat may be NULL.
 * The feature term is float32: 1.0f - expf(-F), times (float)lambda_f, then
 * widened (numpy weak-scalar promotion). */
/* Rows [row0, row0 + nrows) of the combined map: `spatial`, `feat` and `out` are [nrows, n] row
 * blocks (row i of the block is map row row0 + i).  Same arithmetic per element as the full map;
 * lets the checker compare sampled rows of a pool whose full O(N^2) oracle maps are too large. */
void al3d_oracle_combine_rows(const double* spatial, const int64_t* temporal_id, const float* feat,
                              int64_t n, int64_t row0, int64_t nrows, int normalize, int aggregate,
                              double lambda_t, double lambda_f,
                              double spatial_scale, double temporal_scale, double* out)
{
    float lf = (float)lambda_f;
#pragma omp parallel for schedule(static)
    for (int64_t ii = 0; ii < nrows; ++ii) {
        int64_t i = row0 + ii;
        for (int64_t j = 0; j < n; ++j) {
            int64_t e = ii * n + j;
            double s = 0.0, t = 0.0, f = 0.0;
            int nterm = 0;
            double terms[3];
            if (spatial) {
                s = spatial[e];
                if (normalize == 1) s = 1.0 - al3d_oracle_exp_f64(-s);
                else if (normalize == 2) s = s / spatial_scale;
                terms[nterm++] = s;
            }
            if (temporal_id) {
                t = temporal_gap(temporal_id, i, j);
                if (normalize == 1) t = 1.0 - al3d_oracle_exp_f64(-t);
                else if (normalize == 2) t = t / temporal_scale;
                terms[nterm++] = t;
            }
            if (feat) {
                float ff = feat[e];
                if (normalize == 1) ff = 1.0f - al3d_oracle_exp_f32(-ff);
                if (aggregate == 0) ff = lf * ff;
                f = (double)ff;
                terms[nterm++] = f;
            }
            double r;
            if (aggregate == 0) {
                r = spatial ? s : 0.0;
                if (temporal_id) r = spatial ? r + lambda_t * t : lambda_t * t;
                if (feat) r = (spatial || temporal_id) ? r + f : f;
            } else {
                r = terms[0];
                for (int q = 1; q < nterm; ++q)
                    r = aggregate == 1 ? (terms[q] < r ? terms[q] : r) : (terms[q] > r ? terms[q] : r);
            }
            out[e] = r;
        }
    }
}

void al3d_oracle_combine(const double* spatial, const int64_t* temporal_id, const float* feat,
                         int64_t n, int normalize, int aggregate,
                         double lambda_t, double lambda_f,
                         double spatial_scale, double temporal_scale, double* out)
{
    al3d_oracle_combine_rows(spatial, temporal_id, feat, n, 0, n, normalize, aggregate, lambda_t, lambda_f,
                             spatial_scale, temporal_scale, out);
}

/* pairwise embedding distance, float32 (feature_selector.py:96-105): both the
 * p==1 and the "p==2" branch are L1 (sqrt applied element-wise before the sum).
 * Canonical summation order of this build: c = 0..C-1 into one f32 accumulator. */
void al3d_oracle_l1_map_f32(const float* feats, int64_t n, int64_t c, int p, float* out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float* a = feats + i * c;
        for (int64_t j = 0; j < n; ++j) {
            const float* b = feats + j * c;
            float acc = 0.0f;
            for (int64_t q = 0; q < c; ++q) {
                float d = b[q] - a[q];
                acc += p == 1 ? fabsf(d) : sqrtf(d * d);
            }
            out[i * n + j] = acc;
        }
    }
}

/* greedy k-center under a cost budget (spatial_temporal_selector.py:157-193;
 * float32 torch twin feature_selector.py:142-172).
 *   D            [n,n] row-major distance map (iteration map)
 *   seed_map     map used to initialise fps from the seeded rows (== D except
 *                SpatialFeatureSelector, spatial_feature_selector.py:199-207)
 *   seeded       already-sampled frame ids; if n_seeded == 0 the first pick is
 *                ``first`` (python random.choice on the host)
 *   cost         start_cost (= get_cost_amount()); per pick  += cost_f ; += box_cost[i]
 *   stop         when cost > budget_int; the overflowing frame is not appended
 *   check_seeded also assert picks are not in the seeded list (spatial/temporal selectors)
 * returns 0, -1 when the duplicate-pick assertion would fire, -2 when out_idx
 * is full; picks made so far are out_idx[0..*out_count). */
#define GREEDY_IMPL(NAME, T)                                                                   \
int NAME(const T* D, const T* seed_map, int64_t n, const int64_t* seeded, int64_t n_seeded, \
             int64_t first, const double* box_cost, double cost_f, double start_cost,           \
             double budget_int, int check_seeded, int64_t* out_idx, int64_t cap,                \
             int64_t* out_count)                                                                \
{                                                                                               \
    T* fps = (T*)malloc(sizeof(T) * (size_t)n);                                                 \
    unsigned char* picked = (unsigned char*)calloc((size_t)n, 1);                              \
    unsigned char* in_seed = (unsigned char*)calloc((size_t)n, 1);                             \
    for (int64_t s = 0; s < n_seeded; ++s) in_seed[seeded[s]] = 1;                              \
    int64_t cnt = 0, sel;                                                                       \
    if (n_seeded > 0) {                                                                         \
        for (int64_t j = 0; j < n; ++j) fps[j] = seed_map[seeded[0] * n + j];                   \
        for (int64_t s = 1; s < n_seeded; ++s)                                                  \
            for (int64_t j = 0; j < n; ++j) {                                                   \
                T v = seed_map[seeded[s] * n + j];                                              \
                if (v < fps[j]) fps[j] = v;                                                     \
            }                                                                                   \
        sel = 0;                                                                                \
        for (int64_t j = 1; j < n; ++j) if (fps[j] > fps[sel]) sel = j;                         \
    } else {                                                                                    \
        sel = first;                                                                            \
        for (int64_t j = 0; j < n; ++j) fps[j] = seed_map[sel * n + j];                         \
    }                                                                                           \
    double cost = start_cost;                                                                   \
    cost += cost_f;                                                                             \
    cost += box_cost[sel];                                                                      \
    if (cap > 0) out_idx[cnt] = sel;                                                            \
    cnt++; picked[sel] = 1;                                                                     \
    int rc = 0;                                                                                 \
    for (;;) {                                                                                  \
        const T* row = D + sel * n;                                                             \
        for (int64_t j = 0; j < n; ++j) if (row[j] < fps[j]) fps[j] = row[j];                   \
        int64_t best = 0;                                                                       \
        for (int64_t j = 1; j < n; ++j) if (fps[j] > fps[best]) best = j;                       \
        if (picked[best] || (check_seeded && in_seed[best])) { rc = -1; break; }                \
        cost += cost_f;                                                                         \
        cost += box_cost[best];                                                                 \
        if (cost > budget_int) break;                                                           \
        if (cnt >= cap) { rc = -2; break; }                                                     \
        out_idx[cnt++] = best; picked[best] = 1; sel = best;                                    \
    }                                                                                           \
    free(fps); free(picked); free(in_seed);                                                     \
    *out_count = cnt;                                                                           \
    return rc;                                                                                  \
}

GREEDY_IMPL(al3d_oracle_greedy_f64, double)
GREEDY_IMPL(al3d_oracle_greedy_f32, float)
