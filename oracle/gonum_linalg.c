/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, -ffp-contract=off) of the dense linear algebra that
 * gonum's lp.Simplex runs per pivot.  Every routine cites the reference file:line
 * it follows; paths are relative to /root/reference/vendor/gonum.org/v1/gonum/.
 * Arithmetic order (which operand is multiplied first, ascending-k accumulation,
 * multiply-by-reciprocal, zero-skips) is kept because the golden vectors of
 * ilp_test.go:143-251 are sensitive to it in the last digit.
 */
#include "gonum_linalg.h"
#include "gonum_blas.h"

#include <stdlib.h>
#include <assert.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DLAMCH_E (1.0 / 9007199254740992.0)           /* 2^-53  lapack/gonum/general.go:131 */
#define DLAMCH_P (2.0 * DLAMCH_E)                     /* general.go:137 */
#define DLAMCH_S 2.2250738585072014e-308              /* 2^-1022 general.go:142 */

static inline int64_t imin(int64_t a, int64_t b) { return a < b ? a : b; }
static inline int64_t imax(int64_t a, int64_t b) { return a > b ? a : b; }

static int g_threads = 1;
void g_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int g_get_threads(void) { return g_threads; }

/* ======================================================================
 * BLAS level 2   (blas/gonum/level2double.go)
 * ==================================================================== */

/* Dgemv, level2double.go:18-116.  trans=0: y = alpha*A*x + beta*y ; trans=1: y = alpha*A^T*x + beta*y */
static void g_dgemv(int trans, int64_t m, int64_t n, double alpha, const double *a, int64_t lda,
                    const double *x, int64_t incx, double beta, double *y, int64_t incy) {
    int64_t leny = trans ? n : m;
    if (m == 0 || n == 0 || (alpha == 0 && beta == 1)) return;
    g_dscal(leny, beta, y, incy); /* :65-69: First form y = beta*y */
    if (alpha == 0) return;
    if (!trans) {
        for (int64_t i = 0; i < m; i++) {
            double d = (incx == 1 && incy == 1) ? g_dot_unitary(n, a + lda * i, x)
                                                : g_dot_inc(n, x, incx, a + lda * i, 1);
            y[i * incy] += alpha * d;
        }
        return;
    }
    /* :99-115 row-axpy, sequential in i, zero multipliers skipped */
    for (int64_t i = 0; i < m; i++) {
        double tmp = alpha * x[i * incx];
        if (tmp != 0) g_axpy_inc(n, tmp, a + lda * i, 1, y, incy);
    }
}

/* Dger, level2double.go:118-175: A += alpha * x * y^T (no zero-skip) */
static void g_dger(int64_t m, int64_t n, double alpha, const double *x, int64_t incx,
                   const double *y, int64_t incy, double *a, int64_t lda) {
    if (m == 0 || n == 0 || alpha == 0) return;
    for (int64_t i = 0; i < m; i++) g_axpy_inc(n, alpha * x[i * incx], y, incy, a + i * lda, 1);
}

/* Dtrsv with incX == 1, level2double.go (all four uplo/trans forms; used by Dlatrs) */
static void g_dtrsv(int upper, int trans, int nonunit, int64_t n, const double *a, int64_t lda, double *x) {
    if (n == 0) return;
    if (n == 1) { if (nonunit) x[0] /= a[0]; return; }
    if (!trans) {
        if (upper) {
            for (int64_t i = n - 1; i >= 0; i--) {
                double sum = 0;
                for (int64_t j = i + 1; j < n; j++) sum += x[j] * a[i * lda + j];
                x[i] -= sum;
                if (nonunit) x[i] /= a[i * lda + i];
            }
            return;
        }
        for (int64_t i = 0; i < n; i++) {
            double sum = 0;
            for (int64_t j = 0; j < i; j++) sum += x[j] * a[i * lda + j];
            x[i] -= sum;
            if (nonunit) x[i] /= a[i * lda + i];
        }
        return;
    }
    if (upper) {
        for (int64_t i = 0; i < n; i++) {
            if (nonunit) x[i] /= a[i * lda + i];
            double xi = x[i];
            for (int64_t j = i + 1; j < n; j++) x[j] -= a[i * lda + j] * xi;
        }
        return;
    }
    for (int64_t i = n - 1; i >= 0; i--) {
        if (nonunit) x[i] /= a[i * lda + i];
        double xi = x[i];
        for (int64_t j = 0; j < i; j++) x[j] -= a[i * lda + j] * xi;
    }
}

/* Dtrmv(Upper, NoTrans, NonUnit) with a strided x — the only form Dlarft(Forward) uses */
static void g_dtrmv_upper_notrans_nonunit(int64_t n, const double *a, int64_t lda, double *x, int64_t incx) {
    if (n == 0) return;
    if (n == 1) { x[0] *= a[0]; return; }
    for (int64_t i = 0; i < n; i++) {
        double tmp = a[i * lda + i] * x[i * incx];
        double d = (incx == 1) ? g_dot_unitary(n - i - 1, a + i * lda + i + 1, x + i + 1)
                               : g_dot_inc(n - i - 1, x + (i + 1) * incx, incx, a + i * lda + i + 1, 1);
        x[i * incx] = tmp + d;
    }
}

/* ======================================================================
 * BLAS level 3   (blas/gonum/level3double.go, dgemm.go)
 * ==================================================================== */

/* Dtrsm(Left, uplo, NoTrans, diag), level3double.go:75-118: ascending-k row axpys with
 * zero-skip, diagonal applied as a multiplication by the reciprocal. */
static void g_dtrsm_left_notrans(int upper, int nonunit, int64_t m, int64_t n, double alpha,
                                 const double *a, int64_t lda, double *b, int64_t ldb) {
    if (m == 0 || n == 0) return;
    if (alpha == 0) { for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) b[i * ldb + j] = 0; return; }
    if (upper) {
        for (int64_t i = m - 1; i >= 0; i--) {
            double *bt = b + i * ldb;
            if (alpha != 1) for (int64_t j = 0; j < n; j++) bt[j] *= alpha;
            for (int64_t k = i + 1; k < m; k++) {
                double va = a[i * lda + k];
                if (va != 0) g_axpy_to(n, bt, -va, b + k * ldb, bt);
            }
            if (nonunit) { double t = 1 / a[i * lda + i]; for (int64_t j = 0; j < n; j++) bt[j] *= t; }
        }
        return;
    }
    for (int64_t i = 0; i < m; i++) {
        double *bt = b + i * ldb;
        if (alpha != 1) for (int64_t j = 0; j < n; j++) bt[j] *= alpha;
        for (int64_t k = 0; k < i; k++) {
            double va = a[i * lda + k];
            if (va != 0) g_axpy_to(n, bt, -va, b + k * ldb, bt);
        }
        if (nonunit) { double t = 1 / a[i * lda + i]; for (int64_t j = 0; j < n; j++) bt[j] *= t; }
    }
}

/* Dtrmm(Right, ...), level3double.go — the three forms Dlarfb(Left,Trans,Forward,ColumnWise) uses */
static void g_dtrmm_right(int upper, int trans, int nonunit, int64_t m, int64_t n, double alpha,
                          const double *a, int64_t lda, double *b, int64_t ldb) {
    if (m == 0 || n == 0) return;
    if (alpha == 0) { for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) b[i * ldb + j] = 0; return; }
    if (!trans) {
        if (upper) {
            for (int64_t i = 0; i < m; i++) {
                double *bt = b + i * ldb;
                for (int64_t k = n - 1; k >= 0; k--) {
                    double tmp = alpha * bt[k];
                    if (tmp != 0) {
                        bt[k] = tmp;
                        if (nonunit) bt[k] *= a[k * lda + k];
                        for (int64_t j = k + 1; j < n; j++) bt[j] += tmp * a[k * lda + j];
                    }
                }
            }
            return;
        }
        for (int64_t i = 0; i < m; i++) {
            double *bt = b + i * ldb;
            for (int64_t k = 0; k < n; k++) {
                double tmp = alpha * bt[k];
                if (tmp != 0) {
                    bt[k] = tmp;
                    if (nonunit) bt[k] *= a[k * lda + k];
                    g_axpy_to(k, bt, tmp, a + k * lda, bt);
                }
            }
        }
        return;
    }
    if (upper) {
        for (int64_t i = 0; i < m; i++) {
            double *bt = b + i * ldb;
            for (int64_t j = 0; j < n; j++) {
                double tmp = bt[j];
                if (nonunit) tmp *= a[j * lda + j];
                tmp += g_dot_unitary(n - j - 1, a + j * lda + j + 1, bt + j + 1);
                bt[j] = alpha * tmp;
            }
        }
        return;
    }
    for (int64_t i = 0; i < m; i++) {
        double *bt = b + i * ldb;
        for (int64_t j = n - 1; j >= 0; j--) {
            double tmp = bt[j];
            if (nonunit) tmp *= a[j * lda + j];
            tmp += g_dot_unitary(j, a + j * lda, bt);
            bt[j] = alpha * tmp;
        }
    }
}

/* dgemmSerial*, dgemm.go:176-250 */
static void gemm_serial(int ta, int tb, int64_t m, int64_t n, int64_t k, const double *a, int64_t lda,
                        const double *b, int64_t ldb, double *c, int64_t ldc, double alpha) {
    if (!ta && !tb) {
        for (int64_t i = 0; i < m; i++) {
            double *ct = c + i * ldc;
            for (int64_t l = 0; l < k; l++) {
                double tmp = alpha * a[i * lda + l];
                if (tmp != 0) g_axpy_to(n, ct, tmp, b + l * ldb, ct);
            }
        }
    } else if (ta && !tb) {
        for (int64_t l = 0; l < k; l++) {
            const double *bt = b + l * ldb;
            for (int64_t i = 0; i < m; i++) {
                double tmp = alpha * a[l * lda + i];
                if (tmp != 0) { double *ct = c + i * ldc; g_axpy_to(n, ct, tmp, bt, ct); }
            }
        }
    } else if (!ta && tb) {
        for (int64_t i = 0; i < m; i++) {
            const double *at = a + i * lda;
            double *ct = c + i * ldc;
            for (int64_t j = 0; j < n; j++) ct[j] += alpha * g_dot_unitary(k, at, b + j * ldb);
        }
    } else {
        for (int64_t l = 0; l < k; l++)
            for (int64_t i = 0; i < m; i++) {
                double tmp = alpha * a[l * lda + i];
                if (tmp != 0) g_axpy_inc(n, tmp, b + l, ldb, c + i * ldc, 1);
            }
    }
}

/* Dgemm, dgemm.go:14-174.  Below 4 blocks of 64x64 the serial kernel runs on the whole
 * operand; otherwise every (i,j) block walks its k-blocks in ascending order
 * (dgemm.go:131-152), which fixes the per-element summation order independently of the
 * number of workers.  The block loop is spread over host threads like gonum's goroutines. */
void g_dgemm(int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double *a, int64_t lda,
             const double *b, int64_t ldb, double beta, double *c, int64_t ldc) {
    const int64_t BS = 64; /* blas/gonum/gonum.go:43 */
    if (beta != 1) {
        if (beta == 0) { for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) c[i * ldc + j] = 0; }
        else { for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) c[i * ldc + j] *= beta; }
    }
    int64_t bm = (m + BS - 1) / BS, bn = (n + BS - 1) / BS;
    if (bm * bn < 4) { gemm_serial(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, alpha); return; }
    int64_t nblk = bm * bn;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads) if (g_threads > 1)
#endif
    for (int64_t blk = 0; blk < nblk; blk++) {
        int64_t i = (blk / bn) * BS, j = (blk % bn) * BS;
        int64_t leni = imin(BS, m - i), lenj = imin(BS, n - j);
        double *cs = c + i * ldc + j;
        for (int64_t kk = 0; kk < k; kk += BS) {
            int64_t lenk = imin(BS, k - kk);
            const double *as = ta ? a + kk * lda + i : a + i * lda + kk;
            const double *bs = tb ? b + j * ldb + kk : b + kk * ldb + j;
            gemm_serial(ta, tb, leni, lenj, lenk, as, lda, bs, ldb, cs, ldc, alpha);
        }
    }
}

/* ======================================================================
 * LAPACK: LU   (lapack/gonum/dgetf2.go, dlaswp.go, dgetrf.go, dgetrs.go, dlange.go)
 * ==================================================================== */

/* Dgetf2, dgetf2.go:30-69.  Returns ok (0 when an exact zero pivot was met). */
int g_dgetf2(int64_t m, int64_t n, double *a, int64_t lda, int64_t *ipiv) {
    int64_t mn = imin(m, n);
    int ok = 1;
    if (m == 0 || n == 0) return 1;
    for (int64_t j = 0; j < mn; j++) {
        int64_t jp = j + g_idamax(m - j, a + j * lda + j, lda);
        ipiv[j] = jp;
        if (a[jp * lda + j] == 0) {
            ok = 0;
        } else {
            if (jp != j) g_dswap(n, a + j * lda, 1, a + jp * lda, 1);
            if (j < m - 1) {
                double aj = a[j * lda + j];
                if (fabs(aj) >= DLAMCH_S) {
                    g_dscal(m - j - 1, 1 / aj, a + (j + 1) * lda + j, lda); /* :56 multiply by reciprocal */
                } else {
                    /* :58-60 — the reference divides the same element m-j-1 times; kept as is */
                    for (int64_t i = 0; i < m - j - 1; i++) a[(j + 1) * lda + j] = a[(j + 1) * lda + j] / a[lda * j + j];
                }
            }
        }
        if (j < mn - 1)
            g_dger(m - j - 1, n - j - 1, -1, a + (j + 1) * lda + j, lda, a + j * lda + j + 1, 1,
                   a + (j + 1) * lda + j + 1, lda);
    }
    return ok;
}

/* Dlaswp with incX = +1, dlaswp.go:16-47 (ipiv addressed absolutely) */
static void g_dlaswp_fwd(int64_t n, double *a, int64_t lda, int64_t k1, int64_t k2, const int64_t *ipiv) {
    if (n == 0) return;
    for (int64_t k = k1; k <= k2; k++) g_dswap(n, a + k * lda, 1, a + ipiv[k] * lda, 1);
}

/* Dgetrf, dgetrf.go:29-70: nb = 64 (ilaenv.go:43-51); unblocked when nb >= min(m,n) */
int g_dgetrf(int64_t m, int64_t n, double *a, int64_t lda, int64_t *ipiv) {
    int64_t mn = imin(m, n);
    if (m == 0 || n == 0) return 0;
    const int64_t nb = 64;
    if (nb >= mn) return g_dgetf2(m, n, a, lda, ipiv);
    int ok = 1;
    for (int64_t j = 0; j < mn; j += nb) {
        int64_t jb = imin(mn - j, nb);
        if (!g_dgetf2(m - j, jb, a + j * lda + j, lda, ipiv + j)) ok = 0;
        for (int64_t i = j; i <= imin(m - 1, j + jb - 1); i++) ipiv[i] = j + ipiv[i];
        g_dlaswp_fwd(j, a, lda, j, j + jb - 1, ipiv);
        if (j + jb < n) {
            g_dlaswp_fwd(n - j - jb, a + j + jb, lda, j, j + jb - 1, ipiv);
            g_dtrsm_left_notrans(0, 0, jb, n - j - jb, 1, a + j * lda + j, lda, a + j * lda + j + jb, lda);
            if (j + jb < m)
                g_dgemm(0, 0, m - j - jb, n - j - jb, jb, -1, a + (j + jb) * lda + j, lda, a + j * lda + j + jb, lda, 1,
                        a + (j + jb) * lda + j + jb, lda);
        }
    }
    return ok;
}

/* Dgetrs(NoTrans) with one right-hand side, dgetrs.go:37-45 */
static void g_dgetrs_vec(int64_t n, const double *a, int64_t lda, const int64_t *ipiv, double *b) {
    if (n == 0) return;
    for (int64_t k = 0; k < n; k++) { double t = b[k]; b[k] = b[ipiv[k]]; b[ipiv[k]] = t; }
    g_dtrsm_left_notrans(0, 0, n, 1, 1, a, lda, b, 1);
    g_dtrsm_left_notrans(1, 1, n, 1, 1, a, lda, b, 1);
}

/* Dlange for MaxRowSum / MaxColumnSum, dlange.go:40-71 */
static double g_dlange(int norm, int64_t m, int64_t n, const double *a, int64_t lda) {
    if (m == 0 && n == 0) return 0;
    double value = 0;
    if (norm == G_NORM_MAXCOLSUM) {
        double *work = (double *)calloc((size_t)imax(n, 1), sizeof(double));
        for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) work[j] += fabs(a[i * lda + j]);
        for (int64_t i = 0; i < n; i++) value = go_max(value, work[i]);
        free(work);
        return value;
    }
    for (int64_t i = 0; i < m; i++) {
        double sum = 0;
        for (int64_t j = 0; j < n; j++) sum += fabs(a[i * lda + j]);
        value = go_max(value, sum);
    }
    return value;
}

/* Dlantr(norm, Upper, NonUnit, n, n), dlantr.go (the R factor of mat/qr.go:23-39) */
static double g_dlantr_upper_nonunit(int norm, int64_t n, const double *a, int64_t lda) {
    if (n == 0) return 0;
    if (norm == G_NORM_MAXCOLSUM) {
        double *work = (double *)calloc((size_t)n, sizeof(double));
        for (int64_t i = 0; i < n; i++) for (int64_t j = i; j < n; j++) work[j] += fabs(a[i * lda + j]);
        double mx = 0;
        for (int64_t j = 0; j < n; j++) {
            if (isnan(work[j])) { free(work); return NAN; }
            if (work[j] > mx) mx = work[j];
        }
        free(work);
        return mx;
    }
    double maxsum = 0;
    for (int64_t i = 0; i < n; i++) {
        double sum = 0;
        for (int64_t j = i; j < n; j++) sum += fabs(a[i * lda + j]);
        if (isnan(sum)) return sum;
        if (sum > maxsum) maxsum = sum;
    }
    return maxsum;
}

/* ======================================================================
 * LAPACK: condition estimation  (dlacn2.go, dlatrs.go, drscl.go, dgecon.go, dtrcon.go)
 * ==================================================================== */

/* Dlacn2, dlacn2.go:24-136 (Hager/Higham reverse-communication 1-norm estimator) */
static double g_dlacn2(int64_t n, double *v, double *x, int64_t *isgn, double est, int *kase, int64_t isave[3]) {
    const int itmax = 5;
    if (*kase == 0) {
        for (int64_t i = 0; i < n; i++) x[i] = 1 / (double)n;
        *kase = 1; isave[0] = 1;
        return est;
    }
    switch (isave[0]) {
    case 1:
        if (n == 1) { v[0] = x[0]; est = fabs(v[0]); *kase = 0; return est; }
        est = g_dasum(n, x, 1);
        for (int64_t i = 0; i < n; i++) { x[i] = copysign(1, x[i]); isgn[i] = (int64_t)x[i]; }
        *kase = 2; isave[0] = 2;
        return est;
    case 2:
        isave[1] = g_idamax(n, x, 1);
        isave[2] = 2;
        for (int64_t i = 0; i < n; i++) x[i] = 0;
        x[isave[1]] = 1;
        *kase = 1; isave[0] = 3;
        return est;
    case 3: {
        g_dcopy(n, x, 1, v, 1);
        double estold = est;
        est = g_dasum(n, v, 1);
        int same = 1;
        for (int64_t i = 0; i < n; i++)
            if ((int64_t)copysign(1, x[i]) != isgn[i]) { same = 0; break; }
        if (!same && est > estold) {
            for (int64_t i = 0; i < n; i++) { x[i] = copysign(1, x[i]); isgn[i] = (int64_t)x[i]; }
            *kase = 2; isave[0] = 4;
            return est;
        }
        break;
    }
    case 4: {
        int64_t jlast = isave[1];
        isave[1] = g_idamax(n, x, 1);
        if (x[jlast] != fabs(x[isave[1]]) && isave[2] < itmax) {
            isave[2] += 1;
            for (int64_t i = 0; i < n; i++) x[i] = 0;
            x[isave[1]] = 1;
            *kase = 1; isave[0] = 3;
            return est;
        }
        break;
    }
    case 5: {
        double tmp = 2 * (g_dasum(n, x, 1)) / (double)(3 * n);
        if (tmp > est) { g_dcopy(n, x, 1, v, 1); est = tmp; }
        *kase = 0;
        return est;
    }
    default:
        assert(0 && "dlacn2: bad isave");
    }
    /* :126-135 iteration complete, final stage: alternating-sign test vector */
    double altsgn = 1;
    for (int64_t i = 0; i < n; i++) {
        x[i] = altsgn * (1 + (double)i / (double)(n - 1));
        altsgn *= -1;
    }
    *kase = 1; isave[0] = 5;
    return est;
}

/* Drscl, drscl.go:15-46: x *= 1/a without over/underflow */
static void g_drscl(int64_t n, double a, double *x, int64_t incx) {
    double cden = a, cnum = 1.0;
    const double smlnum = DLAMCH_S, bignum = 1 / smlnum;
    for (;;) {
        double cden1 = cden * smlnum, cnum1 = cnum / bignum, mul;
        int done;
        if (cnum != 0 && fabs(cden1) > fabs(cnum)) { mul = smlnum; done = 0; cden = cden1; }
        else if (fabs(cnum1) > fabs(cden)) { mul = bignum; done = 0; cnum = cnum1; }
        else { mul = cnum / cden; done = 1; }
        g_dscal(n, mul, x, incx);
        if (done) break;
    }
}

/* Dlatrs, dlatrs.go:24-359: triangular solve with scaling against overflow.
 * Returns scale; x overwritten; cnorm in/out (computed when !normin). */
static double g_dlatrs(int upper, int trans, int nonunit, int normin, int64_t n, const double *a, int64_t lda,
                       double *x, double *cnorm) {
    if (n == 0) return 0;
    const double smlnum = DLAMCH_S / DLAMCH_P, bignum = 1 / smlnum;
    double scale = 1;
    int notrans = !trans;
    if (!normin) {
        if (upper) {
            cnorm[0] = 0;
            for (int64_t j = 1; j < n; j++) cnorm[j] = g_dasum(j, a + j, lda);
        } else {
            for (int64_t j = 0; j < n - 1; j++) cnorm[j] = g_dasum(n - j - 1, a + (j + 1) * lda + j, lda);
            cnorm[n - 1] = 0;
        }
    }
    int64_t imx = g_idamax(n, cnorm, 1);
    double tmax = cnorm[imx], tscal;
    if (tmax <= bignum) tscal = 1;
    else { tscal = 1 / (smlnum * tmax); g_dscal(n, tscal, cnorm, 1); }
    int64_t jm = g_idamax(n, x, 1);
    double xmax = fabs(x[jm]), xbnd = xmax, grow;
    int64_t jfirst, jlast, jinc;
    if (notrans) {
        if (upper) { jfirst = n - 1; jlast = -1; jinc = -1; } else { jfirst = 0; jlast = n; jinc = 1; }
        if (tscal != 1) { grow = 0; goto Solve; }
        if (nonunit) {
            grow = 1 / go_max(xbnd, smlnum);
            xbnd = grow;
            for (int64_t j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) goto Solve;
                double tjj = fabs(a[j * lda + j]);
                xbnd = go_min(xbnd, go_min(1, tjj) * grow);
                if (tjj + cnorm[j] >= smlnum) grow *= tjj / (tjj + cnorm[j]);
                else grow = 0;
            }
            grow = xbnd;
        } else {
            grow = go_min(1, 1 / go_max(xbnd, smlnum));
            for (int64_t j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) goto Solve;
                grow *= 1 / (1 + cnorm[j]);
            }
        }
    } else {
        if (upper) { jfirst = 0; jlast = n; jinc = 1; } else { jfirst = n - 1; jlast = -1; jinc = -1; }
        if (tscal != 1) { grow = 0; goto Solve; }
        if (nonunit) {
            grow = 1 / go_max(xbnd, smlnum);
            xbnd = grow;
            for (int64_t j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) goto Solve;
                double xj = 1 + cnorm[j];
                grow = go_min(grow, xbnd / xj);
                double tjj = fabs(a[j * lda + j]);
                if (xj > tjj) xbnd *= tjj / xj;
            }
            grow = go_min(grow, xbnd);
        } else {
            grow = go_min(1, 1 / go_max(xbnd, smlnum));
            for (int64_t j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) goto Solve;
                double xj = 1 + cnorm[j];
                grow /= xj;
            }
        }
    }
Solve:
    if (grow * tscal > smlnum) {
        /* the bound on the growth is fine: plain Dtrsv (:165-171) */
        g_dtrsv(upper, trans, nonunit, n, a, lda, x);
        if (tscal != 1) g_dscal(n, 1 / tscal, cnorm, 1);
        return scale;
    }
    /* careful solve (:173-352) */
    if (xmax > bignum) { scale = bignum / xmax; g_dscal(n, scale, x, 1); xmax = bignum; }
    if (notrans) {
        for (int64_t j = jfirst; j != jlast; j += jinc) {
            double xj = fabs(x[j]), tjj, tjjs;
            int skip = 0;
            if (nonunit) tjjs = a[j * lda + j] * tscal;
            else { tjjs = tscal; if (tscal == 1) skip = 1; }
            if (!skip) {
                tjj = fabs(tjjs);
                if (tjj > smlnum) {
                    if (tjj < 1) {
                        if (xj > tjj * bignum) { double rec = 1 / xj; g_dscal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                    }
                    x[j] /= tjjs; xj = fabs(x[j]);
                } else if (tjj > 0) {
                    if (xj > tjj * bignum) {
                        double rec = (tjj * bignum) / xj;
                        if (cnorm[j] > 1) rec /= cnorm[j];
                        g_dscal(n, rec, x, 1); scale *= rec; xmax *= rec;
                    }
                    x[j] /= tjjs; xj = fabs(x[j]);
                } else {
                    for (int64_t i = 0; i < n; i++) x[i] = 0;
                    x[j] = 1; xj = 1; scale = 0; xmax = 0;
                }
            }
            /* Skip1 */
            if (xj > 1) {
                double rec = 1 / xj;
                if (cnorm[j] > (bignum - xmax) * rec) { rec *= 0.5; g_dscal(n, rec, x, 1); scale *= rec; }
            } else if (xj * cnorm[j] > bignum - xmax) {
                g_dscal(n, 0.5, x, 1); scale *= 0.5;
            }
            if (upper) {
                if (j > 0) {
                    g_daxpy(j, -x[j] * tscal, a + j, lda, x, 1);
                    int64_t i = g_idamax(j, x, 1);
                    xmax = fabs(x[i]);
                }
            } else {
                if (j < n - 1) {
                    g_daxpy(n - j - 1, -x[j] * tscal, a + (j + 1) * lda + j, lda, x + j + 1, 1);
                    int64_t i = j + g_idamax(n - j - 1, x + j + 1, 1);
                    xmax = fabs(x[i]);
                }
            }
        }
    } else {
        for (int64_t j = jfirst; j != jlast; j += jinc) {
            double xj = fabs(x[j]);
            double uscal = tscal;
            double rec = 1 / go_max(xmax, 1);
            double tjjs = 0;
            if (cnorm[j] > (bignum - xj) * rec) {
                rec *= 0.5;
                if (nonunit) tjjs = a[j * lda + j] * tscal; else tjjs = tscal;
                double tjj = fabs(tjjs);
                if (tjj > 1) { rec = go_min(1, rec * tjj); uscal /= tjjs; }
                if (rec < 1) { g_dscal(n, rec, x, 1); scale *= rec; xmax *= rec; }
            }
            double sumj = 0;
            if (uscal == 1) {
                if (upper) sumj = g_ddot(j, a + j, lda, x, 1);
                else if (j < n - 1) sumj = g_ddot(n - j - 1, a + (j + 1) * lda + j, lda, x + j + 1, 1);
            } else {
                if (upper) { for (int64_t i = 0; i < j; i++) sumj += (a[i * lda + j] * uscal) * x[i]; }
                else if (j < n) { for (int64_t i = j + 1; i < n; i++) sumj += (a[i * lda + j] * uscal) * x[i]; }
            }
            if (uscal == tscal) {
                x[j] -= sumj;
                double xj2 = fabs(x[j]), tjjs2;
                int skip = 0;
                if (nonunit) tjjs2 = a[j * lda + j] * tscal;
                else { tjjs2 = tscal; if (tscal == 1) skip = 1; }
                if (!skip) {
                    double tjj = fabs(tjjs2);
                    if (tjj > smlnum) {
                        if (tjj < 1) {
                            if (xj2 > tjj * bignum) { rec = 1 / xj2; g_dscal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                        }
                        x[j] /= tjjs2;
                    } else if (tjj > 0) {
                        if (xj2 > tjj * bignum) { rec = (tjj * bignum) / xj2; g_dscal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                        x[j] /= tjjs2;
                    } else {
                        for (int64_t i = 0; i < n; i++) x[i] = 0;
                        x[j] = 1; scale = 0; xmax = 0;
                    }
                }
            } else {
                x[j] = x[j] / tjjs - sumj;
            }
            /* Skip2 */
            xmax = go_max(xmax, fabs(x[j]));
        }
    }
    scale /= tscal;
    if (tscal != 1) g_dscal(n, 1 / tscal, cnorm, 1);
    return scale;
}

/* Dgecon, dgecon.go:26-81: reciprocal condition number of an LU-factored matrix */
double g_dgecon(int norm, int64_t n, const double *a, int64_t lda, double anorm) {
    if (n == 0) return 1;
    if (anorm == 0) return 0;
    double *work = (double *)calloc((size_t)(4 * n), sizeof(double));
    int64_t *iwork = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    double rcond = 0, ainvnm = 0;
    int kase = 0, normin = 0;
    int64_t isave[3] = {0, 0, 0};
    int kase1 = (norm == G_NORM_MAXCOLSUM) ? 1 : 2;
    const double smlnum = DLAMCH_S;
    for (;;) {
        ainvnm = g_dlacn2(n, work + n, work, iwork, ainvnm, &kase, isave);
        if (kase == 0) {
            if (ainvnm != 0) rcond = (1 / ainvnm) / anorm;
            break;
        }
        double sl, su;
        if (kase == kase1) {
            sl = g_dlatrs(0, 0, 0, normin, n, a, lda, work, work + 2 * n);
            su = g_dlatrs(1, 0, 1, normin, n, a, lda, work, work + 3 * n);
        } else {
            su = g_dlatrs(1, 1, 1, normin, n, a, lda, work, work + 3 * n);
            sl = g_dlatrs(0, 1, 0, normin, n, a, lda, work, work + 2 * n);
        }
        double scale = sl * su;
        normin = 1;
        if (scale != 1) {
            int64_t ix = g_idamax(n, work, 1);
            if (scale == 0 || scale < fabs(work[ix]) * smlnum) break;
            g_drscl(n, scale, work, 1);
        }
    }
    free(work); free(iwork);
    return rcond;
}

/* Dtrcon(norm, Upper, NonUnit), dtrcon.go:20-85 */
double g_dtrcon_upper_nonunit(int norm, int64_t n, const double *a, int64_t lda) {
    if (n == 0) return 1;
    double rcond = 0;
    const double smlnum = DLAMCH_S * (double)n;
    double anorm = g_dlantr_upper_nonunit(norm, n, a, lda);
    if (anorm <= 0) return rcond; /* :49 */
    double *work = (double *)calloc((size_t)(3 * n), sizeof(double));
    int64_t *iwork = (int64_t *)calloc((size_t)n, sizeof(int64_t));
    double ainvnm = 0, scale;
    int kase = 0, normin = 0;
    int64_t isave[3] = {0, 0, 0};
    int kase1 = (norm == G_NORM_MAXCOLSUM) ? 1 : 2;
    for (;;) {
        ainvnm = g_dlacn2(n, work + n, work, iwork, ainvnm, &kase, isave);
        if (kase == 0) {
            if (ainvnm != 0) rcond = (1 / anorm) / ainvnm;
            break;
        }
        if (kase == kase1) scale = g_dlatrs(1, 0, 1, normin, n, a, lda, work, work + 2 * n);
        else scale = g_dlatrs(1, 1, 1, normin, n, a, lda, work, work + 2 * n);
        normin = 1;
        if (scale != 1) {
            int64_t ix = g_idamax(n, work, 1);
            double xnorm = fabs(work[ix]);
            if (scale == 0 || scale < xnorm * smlnum) break;
            g_drscl(n, scale, work, 1);
        }
    }
    free(work); free(iwork);
    return rcond;
}

/* ======================================================================
 * LAPACK: Householder QR  (dlarfg.go, dlarf.go, dgeqr2.go, dlarft.go, dlarfb.go, dgeqrf.go)
 * ==================================================================== */

/* Dlarfg, dlarfg.go:26-62; returns beta, writes *tau, scales x */
static double g_dlarfg(int64_t n, double alpha, double *x, int64_t incx, double *tau) {
    if (n <= 1) { *tau = 0; return alpha; }
    double xnorm = g_dnrm2(n - 1, x, incx);
    if (xnorm == 0) { *tau = 0; return alpha; }
    double beta = -copysign(go_hypot(alpha, xnorm), alpha);
    const double safmin = DLAMCH_S / DLAMCH_E;
    int knt = 0;
    if (fabs(beta) < safmin) {
        double rsafmn = 1 / safmin;
        for (;;) {
            knt++;
            g_dscal(n - 1, rsafmn, x, incx);
            beta *= rsafmn;
            alpha *= rsafmn;
            if (fabs(beta) >= safmin) break;
        }
        xnorm = g_dnrm2(n - 1, x, incx);
        beta = -copysign(go_hypot(alpha, xnorm), alpha);
    }
    *tau = (beta - alpha) / beta;
    g_dscal(n - 1, 1 / (alpha - beta), x, incx);
    for (int j = 0; j < knt; j++) beta *= safmin;
    return beta;
}

/* Iladlc, iladlc.go:11-31: last non-zero column */
static int64_t g_iladlc(int64_t m, int64_t n, const double *a, int64_t lda) {
    if (n == 0 || m == 0) return n - 1;
    if (a[n - 1] != 0 || a[(m - 1) * lda + (n - 1)] != 0) return n - 1;
    int64_t highest = -1;
    for (int64_t i = 0; i < m; i++)
        for (int64_t j = n - 1; j >= 0; j--)
            if (a[i * lda + j] != 0) { highest = imax(highest, j); break; }
    return highest;
}

/* Dlarf(Left), dlarf.go:25-77: C = (I - tau v v^T) C */
static void g_dlarf_left(int64_t m, int64_t n, const double *v, int64_t incv, double tau, double *c, int64_t ldc,
                         double *work) {
    int64_t lastv = 0, lastc = 0;
    if (tau != 0) {
        lastv = m - 1;
        int64_t i = lastv * incv;
        while (lastv >= 0 && v[i] == 0) { lastv--; i -= incv; }
        lastc = g_iladlc(lastv + 1, n, c, ldc);
    }
    if (lastv == -1 || lastc == -1) return;
    g_dgemv(1, lastv + 1, lastc + 1, 1, c, ldc, v, incv, 0, work, 1);
    g_dger(lastv + 1, lastc + 1, -tau, v, incv, work, 1, c, ldc);
}

/* Dgeqr2, dgeqr2.go:28-51 */
void g_dgeqr2(int64_t m, int64_t n, double *a, int64_t lda, double *tau, double *work) {
    int64_t k = imin(m, n);
    for (int64_t i = 0; i < k; i++) {
        a[i * lda + i] = g_dlarfg(m - i, a[i * lda + i], a + imin(i + 1, m - 1) * lda + i, lda, &tau[i]);
        if (i < n - 1) {
            double aii = a[i * lda + i];
            a[i * lda + i] = 1;
            g_dlarf_left(m - i, n - i - 1, a + i * lda + i, lda, tau[i], a + i * lda + i + 1, lda, work);
            a[i * lda + i] = aii;
        }
    }
}

/* Dlarft(Forward, ColumnWise), dlarft.go:55-110: T (k×k upper) of the block reflector */
static void g_dlarft_fwd_col(int64_t n, int64_t k, const double *v, int64_t ldv, const double *tau, double *t,
                             int64_t ldt) {
    if (n == 0) return;
    int64_t prevlastv = n - 1;
    for (int64_t i = 0; i < k; i++) {
        prevlastv = imax(i, prevlastv);
        if (tau[i] == 0) {
            for (int64_t j = 0; j <= i; j++) t[j * ldt + i] = 0;
            continue;
        }
        int64_t lastv;
        for (lastv = n - 1; lastv >= i + 1; lastv--)
            if (v[lastv * ldv + i] != 0) break;
        for (int64_t j = 0; j < i; j++) t[j * ldt + i] = -tau[i] * v[i * ldv + j];
        int64_t j = imin(lastv, prevlastv);
        g_dgemv(1, j - i, i, -tau[i], v + (i + 1) * ldv, ldv, v + (i + 1) * ldv + i, ldv, 1, t + i, ldt);
        g_dtrmv_upper_notrans_nonunit(i, t, ldt, t + i, ldt);
        t[i * ldt + i] = tau[i];
        if (i > 1) prevlastv = imax(prevlastv, lastv);
        else prevlastv = lastv;
    }
}

/* Dlarfb(Left, Trans, Forward, ColumnWise), dlarfb.go:75-107: C = H^T C with H = I - V T V^T */
static void g_dlarfb_left_trans_fwd_col(int64_t m, int64_t n, int64_t k, const double *v, int64_t ldv,
                                        const double *t, int64_t ldt, double *c, int64_t ldc, double *work,
                                        int64_t ldwork) {
    if (m == 0 || n == 0) return;
    for (int64_t j = 0; j < k; j++) g_dcopy(n, c + j * ldc, 1, work + j, ldwork);
    g_dtrmm_right(0, 0, 0, n, k, 1, v, ldv, work, ldwork);
    if (m > k) g_dgemm(1, 0, n, k, m - k, 1, c + k * ldc, ldc, v + k * ldv, ldv, 1, work, ldwork);
    g_dtrmm_right(1, 0, 1, n, k, 1, t, ldt, work, ldwork); /* transt = NoTrans for trans = Trans */
    if (m > k) g_dgemm(0, 1, m - k, n, k, -1, v + k * ldv, ldv, work, ldwork, 1, c + k * ldc, ldc);
    g_dtrmm_right(0, 1, 0, n, k, 1, v, ldv, work, ldwork);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < k; j++) c[j * ldc + i] -= work[i * ldwork + j];
}

/* Dgeqrf, dgeqrf.go:31-103 with the workspace mat/qr.go:60-67 gives it (lwork = n*nb, so the
 * nb = 32 blocking is never reduced); blocked only when min(m,n) > nx = 128 (ilaenv.go ispec 3). */
void g_dgeqrf(int64_t m, int64_t n, double *a, int64_t lda, double *tau) {
    int64_t k = imin(m, n);
    if (k == 0) return;
    const int64_t nb = 32;
    int64_t nbmin = 2, nx = 0, ldwork = nb;
    int64_t lwork = imax(n, n * nb);
    double *work = (double *)calloc((size_t)lwork, sizeof(double));
    if (1 < nb && nb < k) nx = 128;
    int64_t i = 0;
    if (nbmin <= nb && nb < k && nx < k) {
        for (i = 0; i < k - nx; i += nb) {
            int64_t ib = imin(k - i, nb);
            g_dgeqr2(m - i, ib, a + i * lda + i, lda, tau + i, work);
            if (i + ib < n) {
                g_dlarft_fwd_col(m - i, ib, a + i * lda + i, lda, tau + i, work, ldwork);
                g_dlarfb_left_trans_fwd_col(m - i, n - i - ib, ib, a + i * lda + i, lda, work, ldwork,
                                            a + i * lda + i + ib, lda, work + ib * ldwork, ldwork);
            }
        }
    }
    if (i < k) g_dgeqr2(m - i, n - i, a + i * lda + i, lda, tau + i, work);
    free(work);
}

/* ======================================================================
 * gonum/mat: LU, Solve, Cond
 * ==================================================================== */

void g_lu_init(g_lu *f) { f->n = 0; f->lu = 0; f->piv = 0; f->cond = 0; }
void g_lu_free(g_lu *f) { free(f->lu); free(f->piv); g_lu_init(f); }

static void lu_finish(g_lu *f, int norm) {
    int64_t n = f->n;
    double anorm = g_dlange(norm, n, n, f->lu, n);  /* mat/lu.go:80 */
    g_dgetrf(n, n, f->lu, n, f->piv);                /* :82 */
    double v = g_dgecon(norm, n, f->lu, n, anorm);   /* updateCond :28-50 (anorm >= 0 branch) */
    f->cond = 1 / v;
}
static void lu_alloc(g_lu *f, int64_t n) {
    if (f->n != n || !f->lu) {
        free(f->lu); free(f->piv);
        f->lu = (double *)malloc(sizeof(double) * (size_t)imax(n * n, 1));
        f->piv = (int64_t *)malloc(sizeof(int64_t) * (size_t)imax(n, 1));
        f->n = n;
    }
}
void g_lu_factorize(g_lu *f, int64_t n, const double *a, int64_t lda, int norm) {
    lu_alloc(f, n);
    for (int64_t i = 0; i < n; i++) memcpy(f->lu + i * n, a + i * lda, sizeof(double) * (size_t)n);
    lu_finish(f, norm);
}
void g_lu_factorize_trans(g_lu *f, int64_t n, const double *a, int64_t lda, int norm) {
    lu_alloc(f, n);
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n; j++) f->lu[i * n + j] = a[j * lda + i]; /* mat/dense.go:423-431 */
    lu_finish(f, norm);
}

/* LU.Det() == 0 test of mat/lu.go:301 via LogDet (:118-135) and floats.Sum */
static int lu_det_is_zero(const g_lu *f) {
    double s = 0;
    for (int64_t i = 0; i < f->n; i++) s += log(fabs(f->lu[i * f->n + i]));
    /* sign is ±1, so Det()==0 iff exp(s)==0 (or s is NaN -> exp NaN != 0) */
    return exp(s) == 0;
}

int g_lu_solve_vec(const g_lu *f, double *x) {
    if (lu_det_is_zero(f)) return 2;                    /* :301-303 */
    g_dgetrs_vec(f->n, f->lu, f->n, f->piv, x);         /* :319 */
    if (f->cond > 1e16) return 1;                       /* :321-323, mat/errors.go:33 */
    return 0;
}

int g_solve_vec(int64_t n, const double *a, int64_t lda, double *x) {
    g_lu f; g_lu_init(&f);
    g_lu_factorize(&f, n, a, lda, G_NORM_MAXROWSUM);
    int rc = g_lu_solve_vec(&f, x);
    g_lu_free(&f);
    return rc;
}
int g_solve_vec_trans(int64_t n, const double *a, int64_t lda, double *x) {
    g_lu f; g_lu_init(&f);
    g_lu_factorize_trans(&f, n, a, lda, G_NORM_MAXROWSUM);
    int rc = g_lu_solve_vec(&f, x);
    g_lu_free(&f);
    return rc;
}

/* mat.Cond(a, 1), mat/matrix.go:284-322 for r >= c */
double g_cond1(int64_t r, int64_t c, const double *a, int64_t lda) {
    assert(r >= c && c > 0);
    if (r == c) {
        g_lu f; g_lu_init(&f);
        g_lu_factorize(&f, r, a, lda, G_NORM_MAXCOLSUM);
        double cond = f.cond;
        g_lu_free(&f);
        return cond;
    }
    /* QR.factorize + updateCond, mat/qr.go:23-69 */
    double *qr = (double *)malloc(sizeof(double) * (size_t)(r * c));
    double *tau = (double *)calloc((size_t)c, sizeof(double));
    for (int64_t i = 0; i < r; i++) memcpy(qr + i * c, a + i * lda, sizeof(double) * (size_t)c);
    g_dgeqrf(r, c, qr, c, tau);
    double v = g_dtrcon_upper_nonunit(G_NORM_MAXCOLSUM, c, qr, c);
    free(qr); free(tau);
    return 1 / v;
}

/* what mat.LU holds after Factorize(a) (trans: Factorize(a.T())) with mat.CondNorm: its cond, and whether LU.Solve's Det() == 0 test
 * fires (mat/lu.go:70-84, 301) — one call for tests that compare a product-side computation of the same two things */
double g_lu_cond_rowsum(int64_t n, const double *a, int64_t lda, int trans, int *det_zero) {
    g_lu f; g_lu_init(&f);
    if (trans) g_lu_factorize_trans(&f, n, a, lda, G_NORM_MAXROWSUM);
    else g_lu_factorize(&f, n, a, lda, G_NORM_MAXROWSUM);
    const double cond = f.cond;
    if (det_zero) *det_zero = lu_det_is_zero(&f);
    g_lu_free(&f);
    return cond;
}
