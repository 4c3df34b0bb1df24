/*
 * ORACLE -- test infrastructure, not product code.
 *
 * Plain-C restatement of the reference's hot path over flat tables, for parity
 * checks at sizes where the Python loops are too slow and as an alternative
 * single-thread CPU baseline.  Checked against oracle/build_oracle.py (bitwise)
 * and oracle/em_oracle.py (1e-13) in tests/test_oracle.py; those in turn are
 * pinned to the reference by tests/golden.
 *
 *   orc_build_em_matrix   /root/reference/mixemt/preprocess.py:177-198 (+ :69-96)
 *   orc_em_step           /root/reference/mixemt/em.py:57-91, with
 *                         scipy.special.logsumexp 1.15.3 (_logsumexp.py:192-247)
 *
 * Build:  make -C oracle      ->  oracle/_build/liboracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* M[r][h] = sum_k (E[site_k][h] == obs_k ? lhit : lmiss), k in signature order */
void orc_build_em_matrix(const uint8_t *E, int64_t lde, const double *lhit, const double *lmiss,
                         const int64_t *row_ptr, const uint16_t *site, const uint8_t *obs,
                         int64_t R, int32_t H, double *M, int64_t ldm)
{
    for (int64_t r = 0; r < R; ++r) {
        double *row = M + r * ldm;
        for (int32_t h = 0; h < H; ++h) row[h] = 0.0;
        for (int64_t j = row_ptr[r]; j < row_ptr[r + 1]; ++j) {
            const uint8_t *e = E + (int64_t)site[j] * lde;
            const double hit = lhit[site[j]], miss = lmiss[site[j]];
            const uint8_t o = obs[j];
            for (int32_t h = 0; h < H; ++h) row[h] += (e[h] == o) ? hit : miss;
        }
    }
}

/* scipy's logsumexp over n strided values with optional weights b (NULL = 1) */
static double lse(const double *a, int64_t stride, const double *b, int64_t n)
{
    double amax = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        double v = (b && b[i] == 0.0) ? -INFINITY : a[i * stride];
        if (v > amax) amax = v;
    }
    const double shift = isfinite(amax) ? amax : 0.0;
    double m = 0.0, s = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double wgt = b ? b[i] : 1.0;
        double v = (b && wgt == 0.0) ? -INFINITY : a[i * stride];
        if (v == amax) m += wgt;
        else s += wgt * exp(v - shift);
    }
    if (s != 0.0) s /= m;
    return log1p(s) + log(fabs(m)) + amax;
}

/* out = (lnp + M) - rowLSE ; new = colLSE_w(out) - LSE(colLSE_w(out)) */
void orc_em_step(const double *M, int64_t ldm, const double *w, const double *lnp, int64_t R,
                 int32_t H, double *out, int64_t ldo, double *new_props)
{
    for (int64_t r = 0; r < R; ++r) {
        double *o = out + r * ldo;
        const double *src = M + r * ldm;
        for (int32_t h = 0; h < H; ++h) o[h] = lnp[h] + src[h];
        const double l = lse(o, 1, NULL, H);
        for (int32_t h = 0; h < H; ++h) o[h] -= l;
    }
    for (int32_t h = 0; h < H; ++h) new_props[h] = lse(out + h, ldo, w, R);
    const double tot = lse(new_props, 1, NULL, H);
    for (int32_t h = 0; h < H; ++h) new_props[h] -= tot;
}
