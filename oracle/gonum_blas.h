/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the gonum BLAS / asm kernels that sit under
 * lp.Simplex (reference: vendor/gonum.org/v1/gonum, rev 6b03bc22e15a...,
 * Gopkg.lock:28-46).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use anything under oracle/.
 *
 * Bit-level rules restated here:
 *   - no FMA anywhere: every a*x+y is a rounded multiply followed by a rounded
 *     add (gonum/internal/asm/f64/axpyunitaryto_amd64.s:95-103 MULPD+ADDPD);
 *     compile with -ffp-contract=off.
 *   - DotUnitary keeps 4 interleaved partial sums, tail into lane 0, reduced
 *     as (s0+s2)+(s1+s3)   (gonum/internal/asm/f64/dot_amd64.s:43-92)
 *   - DotInc keeps 2 lanes (dot_amd64.s:94-140)
 *
 * All matrices are row-major with an explicit leading dimension, as in gonum.
 */
#ifndef GOMILP_ORACLE_GONUM_BLAS_H
#define GOMILP_ORACLE_GONUM_BLAS_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* ---- Go math package semantics --------------------------------------- */

/* math.Max: +Inf wins, NaN propagates, Max(+0,-0)=+0 */
static inline double go_max(double x, double y) {
    if (isinf(x) && x > 0) return x;
    if (isinf(y) && y > 0) return y;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? y : x;
    return x > y ? x : y;
}
/* math.Min: -Inf wins, NaN propagates, Min(-0,+0)=-0 */
static inline double go_min(double x, double y) {
    if (isinf(x) && x < 0) return x;
    if (isinf(y) && y < 0) return y;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0 && x == y) return signbit(x) ? x : y;
    return x < y ? x : y;
}
/* math.Hypot (Go's algorithm, also what hypot_amd64.s computes): p*sqrt(1+(q/p)^2) */
static inline double go_hypot(double p, double q) {
    if (isinf(p) || isinf(q)) return INFINITY;
    if (isnan(p) || isnan(q)) return NAN;
    p = fabs(p); q = fabs(q);
    if (p < q) { double t = p; p = q; q = t; }
    if (p == 0) return 0;
    q = q / p;
    return p * sqrt(1 + q * q);
}

/* ---- gonum/internal/asm/f64 ------------------------------------------ */

/* dot_amd64.s:43-92 */
static inline double g_dot_unitary(int64_t n, const double *x, const double *y) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    for (; i < n; i++) s0 += x[i] * y[i];
    return (s0 + s2) + (s1 + s3);
}
/* dot_amd64.s:94-140: two lanes; element 0 goes to the HIGH lane */
static inline double g_dot_inc(int64_t n, const double *x, int64_t incx, const double *y, int64_t incy) {
    double hi = 0, lo = 0;
    int64_t i = 0;
    for (; i + 2 <= n; i += 2) {
        hi += x[i * incx] * y[i * incy];
        lo += x[(i + 1) * incx] * y[(i + 1) * incy];
    }
    if (i < n) lo += x[i * incx] * y[i * incy];
    return hi + lo;
}
/* axpyunitaryto_amd64.s: dst = alpha*x + y  (dst may alias y) */
static inline void g_axpy_to(int64_t n, double *dst, double alpha, const double *x, const double *y) {
    for (int64_t i = 0; i < n; i++) dst[i] = alpha * x[i] + y[i];
}
/* axpyinc_amd64.s: y += alpha*x, strided */
static inline void g_axpy_inc(int64_t n, double alpha, const double *x, int64_t incx, double *y, int64_t incy) {
    for (int64_t i = 0; i < n; i++) y[i * incy] = alpha * x[i * incx] + y[i * incy];
}

/* ---- gonum/blas/gonum level 1 (level1double.go) ---------------------- */

/* Dasum, level1double.go */
static inline double g_dasum(int64_t n, const double *x, int64_t inc) {
    double s = 0;
    for (int64_t i = 0; i < n; i++) s += fabs(x[i * inc]);
    return s;
}
/* Idamax, level1double.go:121-165 — first index of max |x|, NaN never wins; n==0 -> -1 */
static inline int64_t g_idamax(int64_t n, const double *x, int64_t inc) {
    if (n < 1) return -1;
    int64_t idx = 0;
    double mx = fabs(x[0]);
    for (int64_t i = 1; i < n; i++) {
        double a = fabs(x[i * inc]);
        if (a > mx) { mx = a; idx = i; }
    }
    return idx;
}
static inline void g_dswap(int64_t n, double *x, int64_t incx, double *y, int64_t incy) {
    for (int64_t i = 0; i < n; i++) { double t = x[i * incx]; x[i * incx] = y[i * incy]; y[i * incy] = t; }
}
static inline void g_dcopy(int64_t n, const double *x, int64_t incx, double *y, int64_t incy) {
    for (int64_t i = 0; i < n; i++) y[i * incy] = x[i * incx];
}
/* Dscal: alpha == 0 writes exact zeros */
static inline void g_dscal(int64_t n, double alpha, double *x, int64_t inc) {
    if (n < 1) return;
    if (alpha == 0) { for (int64_t i = 0; i < n; i++) x[i * inc] = 0; return; }
    for (int64_t i = 0; i < n; i++) x[i * inc] *= alpha;
}
/* Daxpy: alpha == 0 is a no-op */
static inline void g_daxpy(int64_t n, double alpha, const double *x, int64_t incx, double *y, int64_t incy) {
    if (n < 1 || alpha == 0) return;
    g_axpy_inc(n, alpha, x, incx, y, incy);
}
static inline double g_ddot(int64_t n, const double *x, int64_t incx, const double *y, int64_t incy) {
    if (n <= 0) return 0;
    if (incx == 1 && incy == 1) return g_dot_unitary(n, x, y);
    return g_dot_inc(n, x, incx, y, incy);
}
/* Dnrm2, level1double.go (scaled sum of squares) */
static inline double g_dnrm2(int64_t n, const double *x, int64_t inc) {
    if (n < 1) return 0;
    if (n == 1) return fabs(x[0]);
    double scale = 0, ssq = 1;
    for (int64_t i = 0; i < n; i++) {
        double v = x[i * inc];
        if (v == 0) continue;
        double a = fabs(v);
        if (isnan(a)) return NAN;
        if (scale < a) { ssq = 1 + ssq * (scale / a) * (scale / a); scale = a; }
        else { ssq = ssq + (a / scale) * (a / scale); }
    }
    if (isinf(scale)) return INFINITY;
    return scale * sqrt(ssq);
}

#endif
