/*
 * ORACLE (test infrastructure, never the product path).
 *
 * CPU restatement, in plain C, of the unimodal least-squares projection the reference performs per
 * factor-matrix column:
 *   /root/reference/src/matcouply/_unimodal_regression.py:27-69   prefix isotonic regression (PAVA)
 *   /root/reference/src/matcouply/_unimodal_regression.py:73-81   fit reconstruction from a prefix
 *   /root/reference/src/matcouply/_unimodal_regression.py:84-92   best split index (strict '<', t ascending)
 *   /root/reference/src/matcouply/_unimodal_regression.py:95-104  left fit ++ reversed right fit
 * (algorithm: Stout 2008, "Unimodal regression via prefix isotonic regression").
 *
 * Pinned against tests/golden/prox.npz (outputs of the reference itself) by tests/test_oracle_golden.py.
 * Built by __graft_entry__.build() into oracle/_build/libmcl_oracle.so and loaded with ctypes by
 * oracle/aoadmm_oracle.py; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 */
#include <stdint.h>
#include <stdlib.h>

/* Prefix isotonic (non-decreasing) regression of y[0..n) read with stride `st` starting at `y0`.
 * For every prefix length p (0..n) err[p] = SSE of the best non-decreasing (optionally >= 0) fit.
 * level[i], start[i] describe the LAST block of the fit of prefix i+1: it covers [start[i], i]. */
static void prefix_isotonic(const double *y0, int64_t st, int64_t n, int nonneg, double *level, int64_t *start,
                            double *err, double *sy, double *sy2, double *sw, double *cum2) {
    err[0] = 0.0;
    double run2 = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double v = y0[i * st];
        run2 += v * v;
        cum2[i] = run2;
        sy[i] = v;
        sy2[i] = v * v;
        sw[i] = 1.0;
        level[i] = v;
        start[i] = i;
        /* pool with the previous block while it is not strictly below (ties merge) */
        while (start[i] != 0 && level[i] <= level[start[i] - 1]) {
            int64_t p = start[i] - 1;
            sy[i] += sy[p];
            sy2[i] += sy2[p];
            sw[i] += sw[p];
            level[i] = sy[i] / sw[i];
            start[i] = start[p];
        }
        if (nonneg && level[i] < 0.0) {
            /* last block negative => every earlier block is negative too: whole prefix clamps to 0 */
            err[i + 1] = cum2[i];
        } else {
            err[i + 1] = (sy2[i] - sy[i] * sy[i] / sw[i]) + err[start[i]];
        }
    }
    if (nonneg)
        for (int64_t i = 0; i < n; ++i)
            if (level[i] < 0.0) level[i] = 0.0;
}

/* Project each of the `ncol` columns of the row-major (n x ld) matrix `in` onto unimodal vectors. */
void mcl_oracle_unimodal_columns(const double *in, double *out, int64_t n, int64_t ncol, int64_t ld, int nonneg) {
    if (n <= 0) return;
    double *buf = (double *)malloc(sizeof(double) * (size_t)(n * 10 + 4));
    int64_t *ibuf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n * 2));
    double *lvL = buf, *lvR = buf + n, *eL = buf + 2 * n, *eR = buf + 3 * n + 1;
    double *sy = buf + 4 * n + 2, *sy2 = buf + 5 * n + 2, *sw = buf + 6 * n + 2, *cum2 = buf + 7 * n + 2;
    int64_t *stL = ibuf, *stR = ibuf + n;
    for (int64_t c = 0; c < ncol; ++c) {
        const double *y = in + c;
        prefix_isotonic(y, ld, n, nonneg, lvL, stL, eL, sy, sy2, sw, cum2);
        prefix_isotonic(y + (n - 1) * ld, -ld, n, nonneg, lvR, stR, eR, sy, sy2, sw, cum2);
        double best = eR[n];
        int64_t t = 0;
        for (int64_t i = 0; i <= n; ++i) {
            double e = eL[i] + eR[n - i];
            if (e < best) { best = e; t = i; }
        }
        /* increasing part on [0, t) */
        for (int64_t idx = t - 1; idx >= 0;) {
            for (int64_t j = stL[idx]; j <= idx; ++j) out[j * ld + c] = lvL[idx];
            idx = stL[idx] - 1;
        }
        /* decreasing part on [t, n): increasing fit of the reversed suffix of length n - t */
        for (int64_t idx = n - t - 1; idx >= 0;) {
            for (int64_t j = stR[idx]; j <= idx; ++j) out[(n - 1 - j) * ld + c] = lvR[idx];
            idx = stR[idx] - 1;
        }
    }
    free(buf);
    free(ibuf);
}
