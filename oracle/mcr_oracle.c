/*
 * mcr_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, fp64 CPU restatement of the statistics hot path of
 * the reference (StefanSko/mcmc-db, package `mcmc_ref` 0.1.4).  It exists only
 * so that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can
 * check / time-compare the HIP path against the reference's algorithm on a box
 * where the reference itself is not present.  Nothing under mcmc-db_amd/ may
 * import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
 * against (i) golden vectors emitted by the imported reference
 * (tests/golden/make_golden.py, run in the build container) and (ii) the
 * reference's own packaged meta.json diagnostics for the fixture models.
 *
 * Operation order deliberately mirrors CPython 3.10 semantics of the reference
 * source (left-to-right `sum`, `x ** 2` == libm pow(x, 2.0), `var ** 0.5` ==
 * libm pow(var, 0.5)) so that results are bit-identical wherever the platform
 * libm is the same.  Build with -ffp-contract=off (see oracle/Makefile).
 *
 * Citations are file:line relative to the reference repo root.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------ */
/* AS241 PPND16 (Wichura 1988), as used by statistics.NormalDist().inv_cdf,   */
/* which the reference calls at src/mcmc_ref/diagnostics.py:124,131.          */
/* ------------------------------------------------------------------------ */
ORC_API double orc_inv_cdf(double p)
{
    double q = p - 0.5, r, num, den, x;
    if (fabs(q) <= 0.425) {
        r = 0.180625 - q * q;
        num = (((((((2.5090809287301226727e+3 * r + 3.3430575583588128105e+4) * r +
                    6.7265770927008700853e+4) * r + 4.5921953931549871457e+4) * r +
                  1.3731693765509461125e+4) * r + 1.9715909503065514427e+3) * r +
                1.3314166789178437745e+2) * r + 3.3871328727963666080e+0) * q;
        den = (((((((5.2264952788528545610e+3 * r + 2.8729085735721942674e+4) * r +
                    3.9307895800092710610e+4) * r + 2.1213794301586595867e+4) * r +
                  5.3941960214247511077e+3) * r + 6.8718700749205790830e+2) * r +
                4.2313330701600911252e+1) * r + 1.0);
        x = num / den;
        return 0.0 + (x * 1.0);
    }
    r = (q <= 0.0) ? p : 1.0 - p;
    r = sqrt(-log(r));
    if (r <= 5.0) {
        r = r - 1.6;
        num = (((((((7.74545014278341407640e-4 * r + 2.27238449892691845833e-2) * r +
                    2.41780725177450611770e-1) * r + 1.27045825245236838258e+0) * r +
                  3.64784832476320460504e+0) * r + 5.76949722146069140550e+0) * r +
                4.63033784615654529590e+0) * r + 1.42343711074968357734e+0);
        den = (((((((1.05075007164441684324e-9 * r + 5.47593808499534494600e-4) * r +
                    1.51986665636164571966e-2) * r + 1.48103976427480074590e-1) * r +
                  6.89767334985100004550e-1) * r + 1.67638483018380384940e+0) * r +
                2.05319162663775882187e+0) * r + 1.0);
    } else {
        r = r - 5.0;
        num = (((((((2.01033439929228813265e-7 * r + 2.71155556874348757815e-5) * r +
                    1.24266094738807843860e-3) * r + 2.65321895265761230930e-2) * r +
                  2.96560571828504891230e-1) * r + 1.78482653991729133580e+0) * r +
                5.46378491116411436990e+0) * r + 6.65790464350110377720e+0);
        den = (((((((2.04426310338993978564e-15 * r + 1.42151175831644588870e-7) * r +
                    1.84631831751005468180e-5) * r + 7.86869131145613259100e-4) * r +
                  1.48753612908506148525e-2) * r + 1.36929880922735805310e-1) * r +
                5.99832206555887937690e-1) * r + 1.0);
    }
    x = num / den;
    if (q < 0.0) x = -x;
    return 0.0 + (x * 1.0);
}

/* ------------------------------------------------------------------------ */
/* helpers                                                                    */
/* ------------------------------------------------------------------------ */
typedef struct { double v; int64_t i; } orc_pair;

static int cmp_pair(const void* a, const void* b)
{
    const orc_pair* x = (const orc_pair*)a; const orc_pair* y = (const orc_pair*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);     /* stable, like list.sort */
}
static int cmp_dbl(const void* a, const void* b)
{
    double x = *(const double*)a, y = *(const double*)b;
    return (x > y) - (x < y);
}
/* Python `sum(values)` for floats: left-to-right double adds (CPython 3.10). */
static double py_sum(const double* v, int64_t n)
{
    double s = 0.0; for (int64_t i = 0; i < n; ++i) s += v[i]; return s;
}
/* Neumaier-compensated sum of v[i] (sq=0) or (v[i]-shift)^2 (sq=1). */
static double comp_sum(const double* v, int64_t n, double shift, int sq)
{
    double s = 0.0, c = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double x = sq ? (v[i] - shift) * (v[i] - shift) : v[i];
        double t = s + x;
        if (fabs(s) >= fabs(x)) c += (s - t) + x; else c += (x - t) + s;
        s = t;
    }
    if (isinf(s)) return s;   /* overflow: the plain sum numpy / pyarrow return (the correction term is inf - inf) */
    return s + c;
}
/* diagnostics.py:196-201 _variance (ddof=1; n<2 -> 0.0) */
static double py_variance(const double* v, int64_t n)
{
    if (n < 2) return 0.0;
    double mean = py_sum(v, n) / (double)n, s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += pow(v[i] - mean, 2.0);
    return s / (double)(n - 1);
}

/* ------------------------------------------------------------------------ */
/* diagnostics.py:101-133 _rank_normalize.                                    */
/* x: pooled values, chain c occupies [off[c], off[c+1]).  z (and optionally   */
/* the average ranks) are written in the same pooled order.                    */
/* ------------------------------------------------------------------------ */
ORC_API int orc_rank_normalize(const double* x, int64_t M, double* z, double* avg_rank)
{
    if (M <= 0) return 0;
    orc_pair* f = (orc_pair*)malloc(sizeof(orc_pair) * (size_t)M);
    if (!f) return -1;
    for (int64_t i = 0; i < M; ++i) { f[i].v = x[i]; f[i].i = i; }
    qsort(f, (size_t)M, sizeof(orc_pair), cmp_pair);
    int64_t i = 0;
    while (i < M) {                                   /* diagnostics.py:113-122 */
        int64_t j = i + 1;
        while (j < M && f[j].v == f[i].v) ++j;
        double r = (double)(i + 1 + j) / 2.0;
        for (int64_t k = i; k < j; ++k) {
            double p = (r - 0.5) / (double)M;         /* diagnostics.py:130 */
            z[f[k].i] = orc_inv_cdf(p);               /* diagnostics.py:131 */
            if (avg_rank) avg_rank[f[k].i] = r;
        }
        i = j;
    }
    free(f);
    return 0;
}

/* statistics.median over all values + |v - med|  (diagnostics.py:93-98) */
ORC_API int orc_fold(const double* x, int64_t M, double* out, double* med_out)
{
    if (M <= 0) return 0;
    double* s = (double*)malloc(sizeof(double) * (size_t)M);
    if (!s) return -1;
    memcpy(s, x, sizeof(double) * (size_t)M);
    qsort(s, (size_t)M, sizeof(double), cmp_dbl);
    double med = (M % 2 == 1) ? s[M / 2] : (s[M / 2 - 1] + s[M / 2]) / 2.0;
    free(s);
    for (int64_t i = 0; i < M; ++i) out[i] = fabs(x[i] - med);
    if (med_out) *med_out = med;
    return 0;
}

/* diagnostics.py:136-151 _rhat over m chains given as (pointer,length) pairs */
static double rhat_core(const double* const* ch, const int64_t* len, int m)
{
    if (m < 2) return NAN;
    int64_t n = len[0];
    for (int k = 1; k < m; ++k) if (len[k] < n) n = len[k];
    if (n < 2) return NAN;
    double* means = (double*)malloc(sizeof(double) * (size_t)m);
    for (int k = 0; k < m; ++k) means[k] = py_sum(ch[k], n) / (double)n;
    double mean_total = py_sum(means, m) / (double)m;
    double sb = 0.0;
    for (int k = 0; k < m; ++k) sb += pow(means[k] - mean_total, 2.0);
    double var_between = (double)n * sb / (double)(m - 1);
    double sw = 0.0;
    for (int k = 0; k < m; ++k) sw += py_variance(ch[k], n);
    double var_within = sw / (double)m;
    double var_hat = (double)(n - 1) / (double)n * var_within + var_between / (double)n;
    free(means);
    if (var_within == 0) return (var_between == 0) ? 1.0 : INFINITY;
    return sqrt(var_hat / var_within);
}

/* diagnostics.py:76-85 _split_chains followed by _rhat */
ORC_API double orc_split_rhat_of_z(const double* z, const int64_t* off, int C)
{
    const double** ch = (const double**)malloc(sizeof(double*) * (size_t)(2 * C + 1));
    int64_t* len = (int64_t*)malloc(sizeof(int64_t) * (size_t)(2 * C + 1));
    int m = 0;
    for (int c = 0; c < C; ++c) {
        int64_t n = off[c + 1] - off[c], half = n / 2;
        if (half == 0) continue;
        ch[m] = z + off[c];        len[m++] = half;
        ch[m] = z + off[c] + half; len[m++] = half;
    }
    double r = rhat_core(ch, len, m);
    free(ch); free(len);
    return r;
}

/* diagnostics.py:154-193 _ess + _autocorr.  *n_terms = number of rho terms
 * accumulated before the first negative rho (n-1 if the loop never broke). */
ORC_API double orc_ess_of_z(const double* z, const int64_t* off, int C, int64_t* n_terms)
{
    int m = C;
    if (n_terms) *n_terms = 0;
    if (m == 0) return NAN;
    int64_t n = off[1] - off[0];
    for (int k = 1; k < m; ++k) if (off[k + 1] - off[k] < n) n = off[k + 1] - off[k];
    if (n < 2) return NAN;
    double* means = (double*)malloc(sizeof(double) * (size_t)m);
    for (int k = 0; k < m; ++k) means[k] = py_sum(z + off[k], n) / (double)n;
    double mean_total = py_sum(means, m) / (double)m;
    double var_between = 0.0;
    if (m > 1) {
        double sb = 0.0;
        for (int k = 0; k < m; ++k) sb += pow(means[k] - mean_total, 2.0);
        var_between = (double)n * sb / (double)(m - 1);
    }
    double sw = 0.0;
    for (int k = 0; k < m; ++k) sw += py_variance(z + off[k], n);
    double var_within = sw / (double)m;
    double var_hat = (double)(n - 1) / (double)n * var_within + var_between / (double)n;
    if (var_hat == 0) { free(means); return (double)((int64_t)m * n); }

    double rho_sum = 0.0;
    int64_t terms = 0;
    for (int64_t lag = 1; lag < n; ++lag) {
        double cov_sum = 0.0;
        for (int k = 0; k < m; ++k) {
            const double* c = z + off[k];
            double mean = means[k], cov = 0.0;
            for (int64_t i = 0; i < n - lag; ++i) cov += (c[i] - mean) * (c[i + lag] - mean);
            cov /= (double)(n - lag);
            cov_sum += cov;
        }
        double rho = cov_sum / ((double)m * var_hat);
        if (rho < 0) break;
        rho_sum += rho;
        ++terms;
    }
    free(means);
    if (n_terms) *n_terms = terms;
    return (double)((int64_t)m * n) / (1.0 + 2.0 * rho_sum);
}

typedef struct {
    double rhat, rhat_bulk, rhat_tail, ess_bulk, ess_tail, median;
    int64_t lag_bulk, lag_tail;
} orc_diag_t;

/* split_rhat / ess_bulk / ess_tail for one parameter (diagnostics.py:13-73),
 * after the min_chains guards.  x pooled, chain c = [off[c], off[c+1]).
 * Returns 0, or -2 if C < min_chains, -3 if min_chains < 1, -1 on OOM. */
ORC_API int orc_diag(const double* x, const int64_t* off, int C, int min_chains, orc_diag_t* out)
{
    if (min_chains < 1) return -3;
    if (C < min_chains) return -2;
    out->lag_bulk = out->lag_tail = 0; out->median = NAN;
    int64_t M = C > 0 ? off[C] : 0;
    if (C < 2) {
        out->rhat = out->rhat_bulk = out->rhat_tail = out->ess_bulk = out->ess_tail = NAN;
        if (M > 0) {                       /* the median is still a pooled order statistic */
            double* t = (double*)malloc(sizeof(double) * (size_t)M);
            if (!t) return -1;
            orc_fold(x, M, t, &out->median);
            free(t);
        }
        return 0;
    }
    double* z = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    double* f = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    if (!z || !f) { free(z); free(f); return -1; }
    orc_rank_normalize(x, M, z, NULL);
    out->rhat_bulk = orc_split_rhat_of_z(z, off, C);
    out->ess_bulk = orc_ess_of_z(z, off, C, &out->lag_bulk);
    orc_fold(x, M, f, &out->median);
    orc_rank_normalize(f, M, z, NULL);
    out->rhat_tail = orc_split_rhat_of_z(z, off, C);
    out->ess_tail = orc_ess_of_z(z, off, C, &out->lag_tail);
    /* Python max(a, b): b if b > a else a   (diagnostics.py:40) */
    out->rhat = (out->rhat_tail > out->rhat_bulk) ? out->rhat_tail : out->rhat_bulk;
    free(z); free(f);
    return 0;
}

/* compare.py:58-64 compute_basic_stats */
ORC_API void orc_basic_stats(const double* v, int64_t n, double* mean_out, double* std_out)
{
    if (n == 0) { *mean_out = NAN; *std_out = NAN; return; }
    double mean = py_sum(v, n) / (double)n, s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += pow(v[i] - mean, 2.0);
    double var = s / (double)n;
    *mean_out = mean; *std_out = pow(var, 0.5);
}

/* Pooled stats as the Backend protocol defines them (backends.py:14-24):
 * mean, population std (backends_numpy.py:41-42, backends_arrow.py:38-39) and
 * linear-interpolated quantiles (backends_numpy.py:44, backends_arrow.py:40-42).
 * Quantile arithmetic restates numpy's published `linear` method:
 *   h=(n-1)q, lo=floor(h), g=h-lo, lerp(a,b,g) = g>=0.5 ? b-(b-a)(1-g) : a+(b-a)g
 * q_lo receives the integer order-statistic index (bit-exact gate). */
ORC_API int orc_stats(const double* v, int64_t n, const double* qs, int nq,
                      double* mean_out, double* std_out, double* q_out, int64_t* q_lo)
{
    if (n <= 0) return -4;
    /* numpy/arrow sum pairwise / in blocks; a compensated (Neumaier) sum is the
     * order-free restatement that agrees with both to ~1 ulp. */
    double mean = comp_sum(v, n, 0.0, 0) / (double)n;
    double s = comp_sum(v, n, mean, 1);
    *mean_out = mean; *std_out = sqrt(s / (double)n);
    if (nq > 0) {
        double* srt = (double*)malloc(sizeof(double) * (size_t)n);
        if (!srt) return -1;
        memcpy(srt, v, sizeof(double) * (size_t)n);
        qsort(srt, (size_t)n, sizeof(double), cmp_dbl);
        for (int k = 0; k < nq; ++k) {
            double h = (double)(n - 1) * qs[k];
            double fl = floor(h);
            int64_t lo = (int64_t)fl;
            if (lo < 0) lo = 0;
            if (lo > n - 1) lo = n - 1;
            int64_t hi = (lo + 1 < n) ? lo + 1 : n - 1;
            double g = h - fl, a = srt[lo], b = srt[hi], d = b - a;
            q_out[k] = (g >= 0.5) ? b - d * (1.0 - g) : a + d * g;
            if (q_lo) q_lo[k] = lo;
        }
        free(srt);
    }
    return 0;
}

/* compare.py:38-51 compare_stats inner arithmetic */
ORC_API void orc_compare(const double* ref, const double* act, int64_t n, double tol,
                         double* rel, uint8_t* pass)
{
    for (int64_t i = 0; i < n; ++i) {
        double denom = fabs(ref[i]) > 1e-12 ? fabs(ref[i]) : 1e-12;   /* max(abs(ref),1e-12) */
        if (fabs(ref[i]) != fabs(ref[i])) denom = fabs(ref[i]);        /* max(nan,1e-12) -> nan */
        double r = fabs(act[i] - ref[i]) / denom;
        rel[i] = r; pass[i] = (uint8_t)(r <= tol);
    }
}

/* Whole-tensor convenience with the same argument meaning as mcr_summarize
 * (include/mcmcref_hip.h): strides in elements, dtype 0=f64 1=f32.  Used as the
 * `cpu_baseline` ("port") leg of bench.py and as the checker in tests. */
typedef struct {
    double *mean, *std, *q, *rhat, *rhat_bulk, *rhat_tail, *ess_bulk, *ess_tail, *median;
    int64_t *lag_bulk, *lag_tail;
} orc_summary_t;

ORC_API int orc_summarize(const void* draws, int dtype, int64_t C, int64_t N, int64_t P,
                          int64_t sc, int64_t sn, int64_t sp, int min_chains,
                          const double* qs, int nq, orc_summary_t* o)
{
    if (min_chains < 1) return -3;
    if (C < min_chains) return -2;
    int64_t M = C * N;
    double* x = (double*)malloc(sizeof(double) * (size_t)(M > 0 ? M : 1));
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(C + 1));
    if (!x || !off) { free(x); free(off); return -1; }
    for (int64_t c = 0; c <= C; ++c) off[c] = c * N;
    for (int64_t p = 0; p < P; ++p) {
        for (int64_t c = 0; c < C; ++c)
            for (int64_t t = 0; t < N; ++t) {
                int64_t e = c * sc + t * sn + p * sp;
                x[c * N + t] = dtype == 0 ? ((const double*)draws)[e] : (double)((const float*)draws)[e];
            }
        if (M > 0) orc_stats(x, M, qs, nq, &o->mean[p], &o->std[p], o->q + p * nq, NULL);
        else { o->mean[p] = o->std[p] = NAN; for (int k = 0; k < nq; ++k) o->q[p * nq + k] = NAN; }
        orc_diag_t d;
        int rc = orc_diag(x, off, (int)C, min_chains, &d);
        if (rc) { free(x); free(off); return rc; }
        o->rhat[p] = d.rhat; o->rhat_bulk[p] = d.rhat_bulk; o->rhat_tail[p] = d.rhat_tail;
        o->ess_bulk[p] = d.ess_bulk; o->ess_tail[p] = d.ess_tail; o->median[p] = d.median;
        o->lag_bulk[p] = d.lag_bulk; o->lag_tail[p] = d.lag_tail;
    }
    free(x); free(off);
    return 0;
}
