/* CPU ORACLE (test infrastructure only) -- effective lengths, EM, TPM.
 * Restates seekmer/mapper.py:117-141 and seekmer/infer.py:88-168, 233-252.
 * The reference computes with numpy (third-party, present in this image);
 * numpy.sum is restated as its published pairwise summation and
 * numpy.bincount as sequential accumulation; tests/test_oracle_quant.py checks
 * both against numpy itself. */
#include "skmo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* numpy.sum of a contiguous float64 array as numpy 2.2 (the version in this
 * image) computes it: the reduction is fed to the add loop in blocks of the
 * ufunc buffer size (8192 elements); inside a block numpy's published pairwise
 * summation applies (loops_utils.h.src: n < 8 sequential; n <= 128 eight
 * accumulators; else split at n/2 rounded down to a multiple of 8); the block
 * sums are accumulated left to right.  Arrays of <= 8192 elements are plain
 * pairwise sums (also what the numpy 1.15 pinned by environment.yml:30 does
 * for any length; the two differ by rounding only, ~1e-16 relative). */
static double pairwise_block(const double *a, int64_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_block(a, n2) + pairwise_block(a + n2, n - n2);
    }
}

double skmo_pairwise_sum(const double *a, int64_t n)
{
    double acc = 0.0;
    for (int64_t i = 0; i < n; i += 8192)
        acc += pairwise_block(a + i, n - i < 8192 ? n - i : 8192);
    return acc;
}

/* seekmer/mapper.py:134-141 -- p = fld / fld.sum(); eff += clip(len - i, 1) * p[i], i ascending */
void skmo_effective_lengths(const int64_t *fld, const double *lengths,
                            int64_t n_tx, double *out)
{
    int64_t total = 0;
    for (int i = 0; i < SKMO_MAX_FRAGMENT_LENGTH; ++i) total += fld[i];
    for (int64_t t = 0; t < n_tx; ++t) out[t] = 0.0;
    for (int i = 0; i < SKMO_MAX_FRAGMENT_LENGTH; ++i) {
        double p = (double)fld[i] / (double)total;   /* 0/0 = NaN as in numpy */
        for (int64_t t = 0; t < n_tx; ++t) {
            double v = lengths[t] - (double)i;
            if (v < 1.0) v = 1.0;
            out[t] += v * p;
        }
    }
}

/* seekmer/mapper.py:117-132 */
double skmo_harmonic_mean_fragment_length(const int64_t *fld)
{
    int64_t numerator = 0;
    for (int i = 0; i < SKMO_MAX_FRAGMENT_LENGTH; ++i) numerator += fld[i];
    if (numerator == 0) return 0.0;
    double terms[SKMO_MAX_FRAGMENT_LENGTH - 1];
    for (int i = 1; i < SKMO_MAX_FRAGMENT_LENGTH; ++i)
        terms[i - 1] = (double)fld[i] / (double)i;
    return (double)numerator / skmo_pairwise_sum(terms, SKMO_MAX_FRAGMENT_LENGTH - 1);
}

/* one EM step, seekmer/infer.py:154-159 (== 162-167) */
static void em_step(const double *old_x, double *new_x, const double *l, int64_t n_tx,
                    const int64_t *cls, const int64_t *tx, int64_t n_pairs,
                    const double *class_count, int64_t n_classes, double n,
                    double *w, double *class_inner)
{
    for (int64_t j = 0; j < n_pairs; ++j) w[j] = old_x[tx[j]];
    for (int64_t c = 0; c < n_classes; ++c) class_inner[c] = 0.0;
    for (int64_t j = 0; j < n_pairs; ++j) class_inner[cls[j]] += w[j];
    for (int64_t c = 0; c < n_classes; ++c) class_inner[c] = class_inner[c] / class_count[c];
    for (int64_t t = 0; t < n_tx; ++t) new_x[t] = 0.0;
    for (int64_t j = 0; j < n_pairs; ++j) new_x[tx[j]] += w[j] / class_inner[cls[j]];
    for (int64_t t = 0; t < n_tx; ++t) {
        double v = new_x[t] / l[t] / n;
        if (v != v) v = 0.0;
        new_x[t] = v;
    }
}

/* seekmer/infer.py:133-168 */
int64_t skmo_em(double *x, const double *l, int64_t n_tx,
                const int64_t *class_of_pair, const int64_t *tx_of_pair,
                int64_t n_pairs, const double *class_count, int64_t n_classes,
                int64_t max_iters, int64_t fixed_iters,
                const int64_t *trace_iters, int64_t n_trace, double *trace)
{
    double n = skmo_pairwise_sum(class_count, n_classes);
    double *old_x = (double *)malloc(sizeof(double) * (size_t)(n_tx > 0 ? n_tx : 1));
    double *w = (double *)malloc(sizeof(double) * (size_t)(n_pairs > 0 ? n_pairs : 1));
    double *inner = (double *)malloc(sizeof(double) * (size_t)(n_classes > 0 ? n_classes : 1));
    int64_t iters = 0;
    int64_t status = 0;
    for (;;) {
        memcpy(old_x, x, sizeof(double) * (size_t)n_tx);
        em_step(old_x, x, l, n_tx, class_of_pair, tx_of_pair, n_pairs,
                class_count, n_classes, n, w, inner);
        iters++;
        for (int64_t k = 0; k < n_trace; ++k)
            if (trace_iters[k] == iters)
                memcpy(trace + k * n_tx, x, sizeof(double) * (size_t)n_tx);
        if (fixed_iters > 0) {
            if (iters >= fixed_iters) break;
            continue;
        }
        /* (abs(x - old_x) / x)[x > 1e-8].max() > 0.01 ; numpy max propagates NaN */
        int any = 0, nan = 0;
        double m = 0.0;
        for (int64_t t = 0; t < n_tx; ++t) {
            if (x[t] > 1e-8) {
                double r = fabs(x[t] - old_x[t]) / x[t];
                if (r != r) nan = 1;
                if (!any || r > m) m = r;
                any = 1;
            }
        }
        if (!any) { status = -1; break; }   /* numpy raises ValueError on an empty max */
        if (nan) break;
        if (!(m > 0.01)) break;
        if (max_iters > 0 && iters >= max_iters) break;
    }
    free(old_x); free(w); free(inner);
    return status < 0 ? status : iters;
}

/* seekmer/infer.py:127-129 */
void skmo_tpm(double *x, int64_t n_tx)
{
    double d = skmo_pairwise_sum(x, n_tx) / 1000000;
    for (int64_t t = 0; t < n_tx; ++t) x[t] /= d;
    for (int64_t t = 0; t < n_tx; ++t) if (x[t] < 0.001) x[t] = 0.0;
    d = skmo_pairwise_sum(x, n_tx) / 1000000;
    for (int64_t t = 0; t < n_tx; ++t) x[t] /= d;
}

/* seekmer/infer.py:250-251 -- raw transcript length, not the effective one */
void skmo_est_counts(const double *tpm, const double *lengths, int64_t n_tx,
                     double aligned, double *out)
{
    for (int64_t t = 0; t < n_tx; ++t) out[t] = tpm[t] * lengths[t];
    double s = skmo_pairwise_sum(out, n_tx);
    double f = aligned / s;
    for (int64_t t = 0; t < n_tx; ++t) out[t] *= f;
}
