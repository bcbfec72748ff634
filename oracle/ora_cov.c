/*
 * ora_cov.c -- CPU ORACLE (test infrastructure): `ska cov` / `--min-count auto` restated from the reference
 *   CoverageHistogram::new / fit_histogram / plot_hist, MixPoisson, lse, ln_dpois, a, b, log_likelihood, grad_ll,
 *   find_cutoff                                                           src/coverage.rs:70-366
 * The optimiser is argmin 0.9 (Cargo.toml:54, not under /root/reference): BFGS (inverse-Hessian update
 * H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T, rho = 1 / y.s; converged when |grad| < sqrt(eps) or
 * |cost_prev - cost| < tol_cost) with BacktrackingLineSearch (step 1, contraction 0.9) under ArmijoCondition(1e-4),
 * restated from its published algorithm.  Pinned by the reference's own known answer (coverage.rs:369-385: the 77-bin
 * example histogram -> cutoff 9); the iterates themselves are not pinned by anything in the reference.
 * Nothing here is part of the product path.
 */
#include "ora_internal.h"
#include <math.h>

#define COV_MAX_COUNT 1000u      /* coverage.rs:21 */
#define COV_MIN_FREQ 50u         /* :22 */

static double lse(double a, double b) { const double x = a > b ? a : b; return x + log(exp(a - x) + exp(b - x)); }       /* :289-292 */
static double ln_dpois(double x, double lambda) { return x * log(lambda) - lgamma(x + 1.0) - lambda; }                     /* :295-297 */
static double comp_a(double w0, double i) { return log(w0) + ln_dpois(i, 1.0); }                                           /* :300-302 */
static double comp_b(double w0, double c, double i) { return log(1.0 - w0) + ln_dpois(i, c); }                             /* :305-307 */

static double log_likelihood(const double p[2], const double *counts, size_t n)                                            /* :310-326 */
{
    const double w0 = p[0], c = p[1];
    if (!(w0 >= 0.0 && w0 <= 1.0) || c < 1.0) return -1.7976931348623157e308;      /* f64::MIN */
    double ll = 0.0;
    for (size_t i = 0; i < n; i++) { const double x = (double)i + 1.0; ll += counts[i] * lse(comp_a(w0, x), comp_b(w0, c, x)); }
    return ll;
}
static void grad_ll(const double p[2], const double *counts, size_t n, double g[2])                                        /* :329-346 */
{
    const double w0 = p[0], c = p[1];
    double g0 = 0.0, g1 = 0.0;
    for (size_t i = 0; i < n; i++) {
        const double x = (double)i + 1.0, a = comp_a(w0, x), b = comp_b(w0, c, x);
        const double dlda = 1.0 / (1.0 + exp(b - a)), dldb = 1.0 / (1.0 + exp(a - b));
        g0 += counts[i] * (dlda / w0 - dldb / (1.0 - w0));
        g1 += counts[i] * (dldb * (x / c - 1.0));
    }
    g[0] = g0; g[1] = g1;
}
static size_t find_cutoff(const double p[2], size_t max_cutoff)                                                            /* :349-363 */
{
    size_t cutoff = 1;
    while (cutoff < max_cutoff) {
        if (comp_a(p[0], (double)cutoff) - comp_b(p[0], p[1], (double)cutoff) < 0.0) break;
        cutoff++;
    }
    return cutoff;
}

/* fit_histogram (coverage.rs:151-224) on an already truncated histogram; 0 ok, -1 "did not converge" */
int ora_cov_fit(const double *counts, size_t n, double *w0_out, double *c_out, size_t *cutoff)
{
    double x[2] = {0.8, 20.0}, H[2][2] = {{1.0, 0.0}, {0.0, 1.0}}, g[2];        /* INIT_W0, INIT_C, init_hessian (:23-24,:183) */
    double f = -log_likelihood(x, counts, n);
    grad_ll(x, counts, n, g); g[0] = -g[0]; g[1] = -g[1];
    int converged = 0;
    for (int it = 0; it < 20 && !converged; it++) {                              /* max_iters(20) */
        const double p[2] = {-(H[0][0] * g[0] + H[0][1] * g[1]), -(H[1][0] * g[0] + H[1][1] * g[1])};
        const double gp = g[0] * p[0] + g[1] * p[1];
        double alpha = 1.0, xn[2], fn;
        for (;;) {                                                               /* BacktrackingLineSearch + ArmijoCondition(1e-4) */
            xn[0] = x[0] + alpha * p[0]; xn[1] = x[1] + alpha * p[1];
            fn = -log_likelihood(xn, counts, n);
            if (fn <= f + 1e-4 * alpha * gp) break;
            alpha *= 0.9;
            if (alpha == 0.0) break;
        }
        double gn[2];
        grad_ll(xn, counts, n, gn); gn[0] = -gn[0]; gn[1] = -gn[1];
        const double y[2] = {gn[0] - g[0], gn[1] - g[1]}, s[2] = {xn[0] - x[0], xn[1] - x[1]};
        const double ys = y[0] * s[0] + y[1] * s[1], prev = f;
        x[0] = xn[0]; x[1] = xn[1]; f = fn; g[0] = gn[0]; g[1] = gn[1];
        if (sqrt(g[0] * g[0] + g[1] * g[1]) < 1.4901161193847656e-8 || fabs(prev - f) < 1e-6) { converged = 1; break; }   /* tol_grad = sqrt(eps), with_tolerance_cost(1e-6) */
        const double rho = 1.0 / ys;
        double t1[2][2], t2[2][2], m[2][2], r[2][2];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) { t1[a][b] = (a == b) - rho * s[a] * y[b]; t2[a][b] = (a == b) - rho * y[a] * s[b]; }
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) m[a][b] = t1[a][0] * H[0][b] + t1[a][1] * H[1][b];
        for (int a = 0; a < 2; a++) for (int b = 0; b < 2; b++) r[a][b] = m[a][0] * t2[0][b] + m[a][1] * t2[1][b] + rho * s[a] * s[b];
        memcpy(H, r, sizeof r);
    }
    if (!converged) { ora_set_error("Optimiser did not converge: Maximum number of iterations reached"); return -1; }
    *w0_out = x[0]; *c_out = x[1]; *cutoff = find_cutoff(x, n);
    return 0;
}

/* Rust `{:e}` of an f64: shortest digits that round-trip, no exponent padding */
static void fmt_lower_exp(double v, char *out, size_t cap)
{
    char tmp[64];
    int prec = 0;
    for (; prec < 17; prec++) { snprintf(tmp, sizeof tmp, "%.*e", prec, v); if (strtod(tmp, NULL) == v) break; }
    snprintf(tmp, sizeof tmp, "%.*e", prec, v);
    char *e = strchr(tmp, 'e');
    int ex = atoi(e + 1);
    *e = 0;
    size_t L = strlen(tmp);
    if (strchr(tmp, '.')) { while (L && tmp[L - 1] == '0') tmp[--L] = 0; if (L && tmp[L - 1] == '.') tmp[--L] = 0; }
    snprintf(out, cap, "%se%d", tmp, ex);
}

static int cmp_key3(const void *a, const void *b)
{
    const ora_key *x = (const ora_key *)a, *y = (const ora_key *)b;
    return x->hi != y->hi ? (x->hi < y->hi ? -1 : 1) : (x->lo != y->lo ? (x->lo < y->lo ? -1 : 1) : 0);
}

/* CoverageHistogram::new (:70-148): split k-mer occurrence counts over both FASTQ files, quality ignored;
 * hist[c - 1] = number of split k-mers seen c times, c <= 1000 (fit_histogram :158-163) */
int ora_cov_histogram(const char *fq1, const char *fq2, int k, int rc, uint32_t hist[1000])
{
    if (k < 5 || k > 63 || !(k & 1)) { ora_set_error("Invalid k-mer length"); return -1; }
    const char *files[2] = {fq1, fq2};
    ora_fastx fx[2];
    for (int f = 0; f < 2; f++) {
        if (ora_fastx_read(files[f], &fx[f])) { if (f) ora_fastx_free(&fx[0]); ora_set_error("Invalid path/file: %s", files[f]); return -1; }
        if (!fx[f].is_fastq) {
            ora_set_error("%s appears to be FASTA.\nCoverage can only be used with FASTQ files, not FASTA.", files[f]);   /* :97-99 */
            for (int q = 0; q <= f; q++) ora_fastx_free(&fx[q]);
            return -1;
        }
    }
    size_t total = 0;
    for (int f = 0; f < 2; f++) for (size_t r = 0; r < fx[f].n; r++) total += fx[f].recs[r].len;
    ora_key *keys = (ora_key *)malloc((total ? total : 1) * sizeof(ora_key));
    size_t n = 0;
    for (int f = 0; f < 2; f++)
        for (size_t r = 0; r < fx[f].n; r++)
            n += extract_record_pos(fx[f].recs[r].seq, fx[f].recs[r].len, k, rc, keys + n, NULL, NULL, total - n);
    for (int f = 0; f < 2; f++) ora_fastx_free(&fx[f]);
    qsort(keys, n, sizeof *keys, cmp_key3);
    memset(hist, 0, 1000 * sizeof(uint32_t));
    for (size_t i = 0; i < n;) {
        size_t j = i; while (j < n && cmp_key3(&keys[j], &keys[i]) == 0) j++;
        if (j - i - 1 < COV_MAX_COUNT) hist[j - i - 1]++;
        i = j;
    }
    free(keys);
    return 0;
}

/* `ska cov` end to end: histogram, truncation (:166-173), fit, plot_hist text (:227-250) */
char *ora_cov(const char *fq1, const char *fq2, int k, int rc, size_t *len, size_t *cutoff_out)
{
    uint32_t hist[1000];
    if (ora_cov_histogram(fq1, fq2, k, rc, hist)) return NULL;
    size_t n = 1000;
    while (n && hist[n - 1] < COV_MIN_FREQ) n--;
    double *cf = (double *)malloc((n ? n : 1) * sizeof(double));
    for (size_t i = 0; i < n; i++) cf[i] = (double)hist[i];
    double w0, c; size_t cutoff;
    if (ora_cov_fit(cf, n, &w0, &c, &cutoff)) { free(cf); return NULL; }
    free(cf);
    size_t cap = 64 + n * 96, L = 0;
    char *o = (char *)malloc(cap);
    L += (size_t)snprintf(o + L, cap - L, "Count\tK_mers\tMixture_density\tComponent\n");
    for (size_t i = 0; i < n; i++) {
        char dens[64];
        fmt_lower_exp(exp(lse(comp_a(w0, (double)i + 1.0), comp_b(w0, c, (double)i + 1.0))), dens, sizeof dens);
        L += (size_t)snprintf(o + L, cap - L, "%zu\t%u\t%s\t%s\n", i + 1, hist[i], dens, (i + 1) < cutoff ? "Error" : "Coverage");
    }
    if (len) *len = L;
    if (cutoff_out) *cutoff_out = cutoff;
    return o;
}
