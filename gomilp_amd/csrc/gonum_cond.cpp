// The reference's own verdict on a basis solve, for bases in the range its LU does not block (n <= 64).
//
// gonum's LU.Solve returns mat.Condition when LU.Det() == 0 or when the condition number it ESTIMATED at factorization time exceeds 1e16
// (mat/lu.go:29-50, 70-84, 301, 321); lp.Simplex leaves its loop on that error (simplex.go:236-239, 289-292, 316-318).  The estimate is
// 1 / Dgecon(MaxRowSum) on the factors Dgetrf left behind: Higham's 1-norm estimator (Dlacn2) driven by scaled triangular solves
// (Dlatrs) — a LOWER bound of the condition number of the FACTORS, which for kappa near 1e16 can sit on either side of the exact
// kappa_inf of the matrix (the factors carry rounding errors of relative size eps * kappa ~ 1 there).  An engine that wants the
// reference's statuses on such inputs has to compute the same quantity, not a better one: until round 5 the host replay of small
// bases used exact condition numbers and differed from the reference on one of 1500 badly scaled LPs (seed 1079).
//
// This file restates, operation by operation and without fused multiply-adds, what the vendored gonum computes (paths relative to
// /root/reference/vendor/gonum.org/v1/gonum): lapack/gonum/dlange.go (MaxRowSum), dgetf2.go:30-69 (Dgetrf calls it as is up to n = 64:
// dgetrf.go:43-47), dgecon.go:26-81, dlacn2.go, dlatrs.go:24-359, drscl.go, blas/gonum level1double.go (Dasum, Idamax, Dscal, Daxpy,
// Ddot with the 4-lane / 2-lane partial sums of internal/asm/f64/dot_amd64.s) and level2double.go Dtrsv.  Host code; compiled with
// -ffp-contract=off like the rest of the library.  tests/test_c_abi.py compares it with the checker's independent restatement
// (test infrastructure, outside this package) bit for bit on thousands of matrices, near-singular ones included.
#include <cmath>
#include <cstdint>
#include <vector>

#include "engine.hpp"

namespace gomilp {

namespace {

constexpr double kEps = 1.0 / 9007199254740992.0;   // 2^-53, lapack/gonum/general.go:131 (dlamchE)
constexpr double kPrec = 2.0 * kEps;                 // dlamchP
constexpr double kSafe = 2.2250738585072014e-308;    // dlamchS

// math.Max / math.Min (NaN propagates, +Inf / -Inf win)
inline double gmax(double x, double y) {
    if (std::isinf(x) && x > 0) return x;
    if (std::isinf(y) && y > 0) return y;
    if (std::isnan(x) || std::isnan(y)) return NAN;
    if (x == 0 && x == y) return std::signbit(x) ? y : x;
    return x > y ? x : y;
}
inline double gmin(double x, double y) {
    if (std::isinf(x) && x < 0) return x;
    if (std::isinf(y) && y < 0) return y;
    if (std::isnan(x) || std::isnan(y)) return NAN;
    if (x == 0 && x == y) return std::signbit(x) ? x : y;
    return x < y ? x : y;
}
inline double asum(int n, const double *x, int inc) {
    double s = 0;
    for (int i = 0; i < n; i++) s += std::fabs(x[(size_t)i * inc]);
    return s;
}
inline int iamax(int n, const double *x, int inc) {   // first index of max |x|, NaN never wins; n < 1: -1
    if (n < 1) return -1;
    int idx = 0;
    double mx = std::fabs(x[0]);
    for (int i = 1; i < n; i++) {
        const double a = std::fabs(x[(size_t)i * inc]);
        if (a > mx) { mx = a; idx = i; }
    }
    return idx;
}
inline void scal(int n, double alpha, double *x, int inc) {
    if (n < 1) return;
    if (alpha == 0) { for (int i = 0; i < n; i++) x[(size_t)i * inc] = 0; return; }
    for (int i = 0; i < n; i++) x[(size_t)i * inc] *= alpha;
}
inline void axpy(int n, double alpha, const double *x, int incx, double *y, int incy) {
    if (n < 1 || alpha == 0) return;
    for (int i = 0; i < n; i++) y[(size_t)i * incy] = alpha * x[(size_t)i * incx] + y[(size_t)i * incy];
}
inline double dot(int n, const double *x, int incx, const double *y, int incy) {
    if (n <= 0) return 0;
    if (incx == 1 && incy == 1) {   // dot_amd64.s:43-92: four interleaved partial sums, tail into lane 0
        double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        int i = 0;
        for (; i + 4 <= n; i += 4) { s0 += x[i] * y[i]; s1 += x[i + 1] * y[i + 1]; s2 += x[i + 2] * y[i + 2]; s3 += x[i + 3] * y[i + 3]; }
        for (; i < n; i++) s0 += x[i] * y[i];
        return (s0 + s2) + (s1 + s3);
    }
    double hi = 0, lo = 0;          // dot_amd64.s:94-140: two lanes, element 0 in the high one
    int i = 0;
    for (; i + 2 <= n; i += 2) { hi += x[(size_t)i * incx] * y[(size_t)i * incy]; lo += x[(size_t)(i + 1) * incx] * y[(size_t)(i + 1) * incy]; }
    if (i < n) lo += x[(size_t)i * incx] * y[(size_t)i * incy];
    return hi + lo;
}

// Dtrsv, incX = 1 (level2double.go)
void trsv(bool upper, bool trans, bool nonunit, int n, const double *a, int lda, double *x) {
    if (n == 0) return;
    if (n == 1) { if (nonunit) x[0] /= a[0]; return; }
    if (!trans) {
        if (upper) {
            for (int i = n - 1; i >= 0; i--) {
                double sum = 0;
                for (int j = i + 1; j < n; j++) sum += x[j] * a[(size_t)i * lda + j];
                x[i] -= sum;
                if (nonunit) x[i] /= a[(size_t)i * lda + i];
            }
            return;
        }
        for (int i = 0; i < n; i++) {
            double sum = 0;
            for (int j = 0; j < i; j++) sum += x[j] * a[(size_t)i * lda + j];
            x[i] -= sum;
            if (nonunit) x[i] /= a[(size_t)i * lda + i];
        }
        return;
    }
    if (upper) {
        for (int i = 0; i < n; i++) {
            if (nonunit) x[i] /= a[(size_t)i * lda + i];
            const double xi = x[i];
            for (int j = i + 1; j < n; j++) x[j] -= a[(size_t)i * lda + j] * xi;
        }
        return;
    }
    for (int i = n - 1; i >= 0; i--) {
        if (nonunit) x[i] /= a[(size_t)i * lda + i];
        const double xi = x[i];
        for (int j = 0; j < i; j++) x[j] -= a[(size_t)i * lda + j] * xi;
    }
}

// Dgetf2 (dgetf2.go:30-69): row-major, partial pivoting, the column below the pivot scaled by the reciprocal, Dger with alpha = -1
void getf2(int n, double *a, int lda, int *ipiv = nullptr) {
    for (int j = 0; j < n; j++) {
        const int jp = j + iamax(n - j, a + (size_t)j * lda + j, lda);
        if (ipiv) ipiv[j] = jp;
        if (a[(size_t)jp * lda + j] != 0) {
            if (jp != j) for (int c = 0; c < n; c++) { const double t = a[(size_t)j * lda + c]; a[(size_t)j * lda + c] = a[(size_t)jp * lda + c]; a[(size_t)jp * lda + c] = t; }
            if (j < n - 1) {
                const double aj = a[(size_t)j * lda + j];
                if (std::fabs(aj) >= kSafe) scal(n - j - 1, 1 / aj, a + (size_t)(j + 1) * lda + j, lda);
                else for (int i = 0; i < n - j - 1; i++) a[(size_t)(j + 1) * lda + j] = a[(size_t)(j + 1) * lda + j] / a[(size_t)j * lda + j];   // (:58-60 as written: the same element every time)
            }
        }
        if (j < n - 1) {   // Dger with alpha = -1 (level2double.go): row i of the trailing block += (-1 * l_i) * u^T, every row (no zero shortcut on this path)
            for (int i = j + 1; i < n; i++) {
                const double t = -1 * a[(size_t)i * lda + j];
                double *row = a + (size_t)i * lda + j + 1;
                const double *u = a + (size_t)j * lda + j + 1;
                for (int c = 0; c < n - j - 1; c++) row[c] = t * u[c] + row[c];
            }
        }
    }
}

// Dlacn2 (dlacn2.go): one step of the estimator's reverse-communication loop
double lacn2(int n, double *v, double *x, int *isgn, double est, int *kase, int (&isave)[3]) {
    const int itmax = 5;
    if (*kase == 0) {
        for (int i = 0; i < n; i++) x[i] = 1 / (double)n;
        *kase = 1; isave[0] = 1;
        return est;
    }
    switch (isave[0]) {
    case 1:
        if (n == 1) { v[0] = x[0]; est = std::fabs(v[0]); *kase = 0; return est; }
        est = asum(n, x, 1);
        for (int i = 0; i < n; i++) { x[i] = std::copysign(1.0, x[i]); isgn[i] = (int)x[i]; }
        *kase = 2; isave[0] = 2;
        return est;
    case 2:
        isave[1] = iamax(n, x, 1);
        isave[2] = 2;
        for (int i = 0; i < n; i++) x[i] = 0;
        x[isave[1]] = 1;
        *kase = 1; isave[0] = 3;
        return est;
    case 3: {
        for (int i = 0; i < n; i++) v[i] = x[i];
        const double estold = est;
        est = asum(n, v, 1);
        bool same = true;
        for (int i = 0; i < n; i++) if ((int)std::copysign(1.0, x[i]) != isgn[i]) { same = false; break; }
        if (!same && est > estold) {
            for (int i = 0; i < n; i++) { x[i] = std::copysign(1.0, x[i]); isgn[i] = (int)x[i]; }
            *kase = 2; isave[0] = 4;
            return est;
        }
        break;
    }
    case 4: {
        const int jlast = isave[1];
        isave[1] = iamax(n, x, 1);
        if (x[jlast] != std::fabs(x[isave[1]]) && isave[2] < itmax) {
            isave[2] += 1;
            for (int i = 0; i < n; i++) x[i] = 0;
            x[isave[1]] = 1;
            *kase = 1; isave[0] = 3;
            return est;
        }
        break;
    }
    case 5: {
        const double tmp = 2 * (asum(n, x, 1)) / (double)(3 * n);
        if (tmp > est) { for (int i = 0; i < n; i++) v[i] = x[i]; est = tmp; }
        *kase = 0;
        return est;
    }
    default:
        break;
    }
    double altsgn = 1;   // the alternating-sign probe (:126-135)
    for (int i = 0; i < n; i++) {
        x[i] = altsgn * (1 + (double)i / (double)(n - 1));
        altsgn *= -1;
    }
    *kase = 1; isave[0] = 5;
    return est;
}

void rscl(int n, double a, double *x) {   // drscl.go:15-46
    double cden = a, cnum = 1.0;
    const double smlnum = kSafe, bignum = 1 / smlnum;
    for (;;) {
        const double cden1 = cden * smlnum, cnum1 = cnum / bignum;
        double mul;
        bool done;
        if (cnum != 0 && std::fabs(cden1) > std::fabs(cnum)) { mul = smlnum; done = false; cden = cden1; }
        else if (std::fabs(cnum1) > std::fabs(cden)) { mul = bignum; done = false; cnum = cnum1; }
        else { mul = cnum / cden; done = true; }
        scal(n, mul, x, 1);
        if (done) break;
    }
}

// Dlatrs (dlatrs.go:24-359): triangular solve that scales against overflow; returns the scale, x overwritten, cnorm in / out
double latrs(bool upper, bool trans, bool nonunit, bool normin, int n, const double *a, int lda, double *x, double *cnorm) {
    if (n == 0) return 0;
    const double smlnum = kSafe / kPrec, bignum = 1 / smlnum;
    double scale = 1;
    if (!normin) {
        if (upper) {
            cnorm[0] = 0;
            for (int j = 1; j < n; j++) cnorm[j] = asum(j, a + j, lda);
        } else {
            for (int j = 0; j < n - 1; j++) cnorm[j] = asum(n - j - 1, a + (size_t)(j + 1) * lda + j, lda);
            cnorm[n - 1] = 0;
        }
    }
    const int imx = iamax(n, cnorm, 1);
    const double tmax = cnorm[imx];
    double tscal;
    if (tmax <= bignum) tscal = 1;
    else { tscal = 1 / (smlnum * tmax); scal(n, tscal, cnorm, 1); }
    const int jm = iamax(n, x, 1);
    double xmax = std::fabs(x[jm]), xbnd = xmax, grow = 0;
    int jfirst, jlast, jinc;
    bool bounded = false;   // (the Go code's `goto Solve` out of the bound loops)
    if (!trans) {
        if (upper) { jfirst = n - 1; jlast = -1; jinc = -1; } else { jfirst = 0; jlast = n; jinc = 1; }
        if (tscal != 1) { grow = 0; bounded = true; }
        else if (nonunit) {
            grow = 1 / gmax(xbnd, smlnum);
            xbnd = grow;
            for (int j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) { bounded = true; break; }
                const double tjj = std::fabs(a[(size_t)j * lda + j]);
                xbnd = gmin(xbnd, gmin(1, tjj) * grow);
                if (tjj + cnorm[j] >= smlnum) grow *= tjj / (tjj + cnorm[j]);
                else grow = 0;
            }
            if (!bounded) grow = xbnd;
        } else {
            grow = gmin(1, 1 / gmax(xbnd, smlnum));
            for (int j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) { bounded = true; break; }
                grow *= 1 / (1 + cnorm[j]);
            }
        }
    } else {
        if (upper) { jfirst = 0; jlast = n; jinc = 1; } else { jfirst = n - 1; jlast = -1; jinc = -1; }
        if (tscal != 1) { grow = 0; bounded = true; }
        else if (nonunit) {
            grow = 1 / gmax(xbnd, smlnum);
            xbnd = grow;
            for (int j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) { bounded = true; break; }
                const double xj = 1 + cnorm[j];
                grow = gmin(grow, xbnd / xj);
                const double tjj = std::fabs(a[(size_t)j * lda + j]);
                if (xj > tjj) xbnd *= tjj / xj;
            }
            if (!bounded) grow = gmin(grow, xbnd);
        } else {
            grow = gmin(1, 1 / gmax(xbnd, smlnum));
            for (int j = jfirst; j != jlast; j += jinc) {
                if (grow <= smlnum) { bounded = true; break; }
                const double xj = 1 + cnorm[j];
                grow /= xj;
            }
        }
    }
    if (grow * tscal > smlnum) {   // the growth bound is fine: plain Dtrsv (:165-171)
        trsv(upper, trans, nonunit, n, a, lda, x);
        if (tscal != 1) scal(n, 1 / tscal, cnorm, 1);
        return scale;
    }
    // the careful solve (:173-352)
    if (xmax > bignum) { scale = bignum / xmax; scal(n, scale, x, 1); xmax = bignum; }
    if (!trans) {
        for (int j = jfirst; j != jlast; j += jinc) {
            double xj = std::fabs(x[j]), tjjs;
            bool skip = false;
            if (nonunit) tjjs = a[(size_t)j * lda + j] * tscal;
            else { tjjs = tscal; if (tscal == 1) skip = true; }
            if (!skip) {
                const double tjj = std::fabs(tjjs);
                if (tjj > smlnum) {
                    if (tjj < 1) {
                        if (xj > tjj * bignum) { const double rec = 1 / xj; scal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                    }
                    x[j] /= tjjs; xj = std::fabs(x[j]);
                } else if (tjj > 0) {
                    if (xj > tjj * bignum) {
                        double rec = (tjj * bignum) / xj;
                        if (cnorm[j] > 1) rec /= cnorm[j];
                        scal(n, rec, x, 1); scale *= rec; xmax *= rec;
                    }
                    x[j] /= tjjs; xj = std::fabs(x[j]);
                } else {
                    for (int i = 0; i < n; i++) x[i] = 0;
                    x[j] = 1; xj = 1; scale = 0; xmax = 0;
                }
            }
            if (xj > 1) {
                double rec = 1 / xj;
                if (cnorm[j] > (bignum - xmax) * rec) { rec *= 0.5; scal(n, rec, x, 1); scale *= rec; }
            } else if (xj * cnorm[j] > bignum - xmax) {
                scal(n, 0.5, x, 1); scale *= 0.5;
            }
            if (upper) {
                if (j > 0) {
                    axpy(j, -x[j] * tscal, a + j, lda, x, 1);
                    const int i = iamax(j, x, 1);
                    xmax = std::fabs(x[i]);
                }
            } else {
                if (j < n - 1) {
                    axpy(n - j - 1, -x[j] * tscal, a + (size_t)(j + 1) * lda + j, lda, x + j + 1, 1);
                    const int i = j + iamax(n - j - 1, x + j + 1, 1);
                    xmax = std::fabs(x[i]);
                }
            }
        }
    } else {
        for (int j = jfirst; j != jlast; j += jinc) {
            const double xj = std::fabs(x[j]);
            double uscal = tscal;
            double rec = 1 / gmax(xmax, 1);
            double tjjs = 0;
            if (cnorm[j] > (bignum - xj) * rec) {
                rec *= 0.5;
                if (nonunit) tjjs = a[(size_t)j * lda + j] * tscal; else tjjs = tscal;
                const double tjj = std::fabs(tjjs);
                if (tjj > 1) { rec = gmin(1, rec * tjj); uscal /= tjjs; }
                if (rec < 1) { scal(n, rec, x, 1); scale *= rec; xmax *= rec; }
            }
            double sumj = 0;
            if (uscal == 1) {
                if (upper) sumj = dot(j, a + j, lda, x, 1);
                else if (j < n - 1) sumj = dot(n - j - 1, a + (size_t)(j + 1) * lda + j, lda, x + j + 1, 1);
            } else {
                if (upper) { for (int i = 0; i < j; i++) sumj += (a[(size_t)i * lda + j] * uscal) * x[i]; }
                else if (j < n) { for (int i = j + 1; i < n; i++) sumj += (a[(size_t)i * lda + j] * uscal) * x[i]; }
            }
            if (uscal == tscal) {
                x[j] -= sumj;
                const double xj2 = std::fabs(x[j]);
                double tjjs2;
                bool skip = false;
                if (nonunit) tjjs2 = a[(size_t)j * lda + j] * tscal;
                else { tjjs2 = tscal; if (tscal == 1) skip = true; }
                if (!skip) {
                    const double tjj = std::fabs(tjjs2);
                    if (tjj > smlnum) {
                        if (tjj < 1) {
                            if (xj2 > tjj * bignum) { rec = 1 / xj2; scal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                        }
                        x[j] /= tjjs2;
                    } else if (tjj > 0) {
                        if (xj2 > tjj * bignum) { rec = (tjj * bignum) / xj2; scal(n, rec, x, 1); scale *= rec; xmax *= rec; }
                        x[j] /= tjjs2;
                    } else {
                        for (int i = 0; i < n; i++) x[i] = 0;
                        x[j] = 1; scale = 0; xmax = 0;
                    }
                }
            } else {
                x[j] = x[j] / tjjs - sumj;
            }
            xmax = gmax(xmax, std::fabs(x[j]));
        }
    }
    scale /= tscal;
    if (tscal != 1) scal(n, 1 / tscal, cnorm, 1);
    return scale;
}

// Dgecon(MaxRowSum) (dgecon.go:26-81): reciprocal condition number in the infinity norm from the LU factors and |A|_inf
double gecon_inf(int n, const double *a, int lda, double anorm) {
    if (n == 0) return 1;
    if (anorm == 0) return 0;
    std::vector<double> work((size_t)4 * n, 0.0);
    std::vector<int> iwork((size_t)n, 0);
    double rcond = 0, ainvnm = 0;
    int kase = 0;
    bool normin = false;
    int isave[3] = {0, 0, 0};
    const int kase1 = 2;   // MaxRowSum
    const double smlnum = kSafe;
    for (;;) {
        ainvnm = lacn2(n, work.data() + n, work.data(), iwork.data(), ainvnm, &kase, isave);
        if (kase == 0) {
            if (ainvnm != 0) rcond = (1 / ainvnm) / anorm;
            break;
        }
        double sl, su;
        if (kase == kase1) {   // inv(L) then inv(U)
            sl = latrs(false, false, false, normin, n, a, lda, work.data(), work.data() + 2 * n);
            su = latrs(true, false, true, normin, n, a, lda, work.data(), work.data() + 3 * n);
        } else {               // inv(U^T) then inv(L^T)
            su = latrs(true, true, true, normin, n, a, lda, work.data(), work.data() + 3 * n);
            sl = latrs(false, true, false, normin, n, a, lda, work.data(), work.data() + 2 * n);
        }
        const double scale = sl * su;
        normin = true;
        if (scale != 1) {
            const int ix = iamax(n, work.data(), 1);
            if (scale == 0 || scale < std::fabs(work[ix]) * smlnum) break;
            rscl(n, scale, work.data());
        }
    }
    return rcond;
}

}  // namespace

bool gonum_lu_cond(const double *M, int n, int ldm, bool transposed, double *cond, bool *det_zero) {
    if (n < 1 || n > kGonumCondMax) return false;
    std::vector<double> lu((size_t)n * n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) lu[(size_t)i * n + j] = transposed ? M[(size_t)j * ldm + i] : M[(size_t)i * ldm + j];
    // Dlange(MaxRowSum) on the matrix as copied (mat/lu.go:80)
    double anorm = 0;
    for (int i = 0; i < n; i++) {
        double sum = 0;
        for (int j = 0; j < n; j++) sum += std::fabs(lu[(size_t)i * n + j]);
        anorm = gmax(anorm, sum);
    }
    getf2(n, lu.data(), n);
    const double v = gecon_inf(n, lu.data(), n, anorm);
    *cond = 1 / v;
    // LU.Det() == 0 (mat/lu.go:118-135, 301): exp(floats.Sum(log |u_ii|)) * sign, the sign being +-1
    double s = 0;
    for (int i = 0; i < n; i++) s += std::log(std::fabs(lu[(size_t)i * n + i]));
    *det_zero = std::exp(s) == 0;
    return true;
}

// x = M^-1 b the way LU.SolveVec computes it (mat/lu.go:305-320): Dgetf2, then Dgetrs(NoTrans) = Dlaswp forward, Dtrsm(Left, Lower, NoTrans, Unit),
// Dtrsm(Left, Upper, NoTrans, NonUnit) on the one right-hand side (blas/gonum/level3double.go: ascending k, zero entries of the triangle
// skipped, a rounded multiply and a rounded add each, the diagonal applied as a product with its reciprocal) — the point the reference returns
// WITH a mat.Condition error is this solve's result
bool gonum_lu_solve(const double *M, int n, int ldm, const double *b, double *x) {
    if (n < 1 || n > kGonumCondMax) return false;
    std::vector<double> lu((size_t)n * n);
    std::vector<int> piv((size_t)n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) lu[(size_t)i * n + j] = M[(size_t)i * ldm + j];
    getf2(n, lu.data(), n, piv.data());
    for (int i = 0; i < n; i++) if (lu[(size_t)i * n + i] == 0) return false;   // (exactly singular: LU.Solve returns before it solves, mat/lu.go:301)
    for (int i = 0; i < n; i++) x[i] = b[i];
    for (int i = 0; i < n; i++) if (piv[i] != i) { const double t = x[i]; x[i] = x[piv[i]]; x[piv[i]] = t; }
    for (int i = 0; i < n; i++)
        for (int k = 0; k < i; k++) { const double va = lu[(size_t)i * n + k]; if (va != 0) x[i] = -va * x[k] + x[i]; }
    for (int i = n - 1; i >= 0; i--) {
        for (int k = i + 1; k < n; k++) { const double va = lu[(size_t)i * n + k]; if (va != 0) x[i] = -va * x[k] + x[i]; }
        x[i] *= 1 / lu[(size_t)i * n + i];
    }
    return true;
}

}  // namespace gomilp
