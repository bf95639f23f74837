/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Restatement of the gonum/mat + gonum/lapack/gonum + gonum/blas/gonum routines
 * reachable from lp.Simplex.  Row-major, explicit
 * leading dimensions, gonum operation order.  See gonum_linalg.c for citations.
 */
#ifndef GOMILP_ORACLE_GONUM_LINALG_H
#define GOMILP_ORACLE_GONUM_LINALG_H

#include <stdint.h>

enum { G_NORM_MAXROWSUM = 0 /* lapack.MaxRowSum, mat.CondNorm */, G_NORM_MAXCOLSUM = 1 /* lapack.MaxColumnSum */ };

/* mat.LU (mat/lu.go:21-25) */
typedef struct {
    int64_t n;
    double *lu;   /* n*n row-major, stride n */
    int64_t *piv; /* n */
    double cond;
} g_lu;

void g_lu_init(g_lu *f);
void g_lu_free(g_lu *f);
/* LU.factorize (mat/lu.go:63-84): copy a (n×n, lda), Dlange, Dgetrf, Dgecon */
void g_lu_factorize(g_lu *f, int64_t n, const double *a, int64_t lda, int norm);
/* LU.factorize of aᵀ (the `ab.T()` argument at simplex.go:236: Dense.Copy of a Transpose) */
void g_lu_factorize_trans(g_lu *f, int64_t n, const double *a, int64_t lda, int norm);
/* LU.Solve for one right-hand side (mat/lu.go:293-325). x holds b on entry.
 * returns 0 = ok; 1 = Condition error with the solve performed (cond > 1e16);
 * 2 = Condition(+Inf) because Det()==0 — x is left untouched. */
int g_lu_solve_vec(const g_lu *f, double *x);

/* (*VecDense).SolveVec(a, b) with a square (mat/solve.go:110-140 -> :78-94). x := b solved in place. */
int g_solve_vec(int64_t n, const double *a, int64_t lda, double *x);
int g_solve_vec_trans(int64_t n, const double *a, int64_t lda, double *x);

/* mat.Cond(a, 1) for r >= c (mat/matrix.go:284-322): LU path when square, QR path when tall */
double g_cond1(int64_t r, int64_t c, const double *a, int64_t lda);
double g_lu_cond_rowsum(int64_t n, const double *a, int64_t lda, int trans, int *det_zero);

/* exposed for unit tests */
int g_dgetrf(int64_t m, int64_t n, double *a, int64_t lda, int64_t *ipiv);
int g_dgetf2(int64_t m, int64_t n, double *a, int64_t lda, int64_t *ipiv);
void g_dgeqrf(int64_t m, int64_t n, double *a, int64_t lda, double *tau);
void g_dgeqr2(int64_t m, int64_t n, double *a, int64_t lda, double *tau, double *work);
double g_dgecon(int norm, int64_t n, const double *a, int64_t lda, double anorm);
double g_dtrcon_upper_nonunit(int norm, int64_t n, const double *a, int64_t lda);
void g_dgemm(int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double *a, int64_t lda,
             const double *b, int64_t ldb, double beta, double *c, int64_t ldc);

/* number of host threads the Dgemm block loop may use (mirrors GOMAXPROCS for
 * blas/gonum/dgemm.go:100-174); results do not depend on it. */
void g_set_threads(int n);
int g_get_threads(void);

#endif
