/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of gonum's dense two-phase primal simplex exactly as GoMILP drives it
 * (subproblem.go:154,172 -> lp.Simplex(c, A, b, 0, nil)).  Every function cites the lines of
 * /root/reference/vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go it follows.
 * The algorithm is the reference's: three fresh LU factorizations per pivot
 * (simplex.go:236, :315, :289), Dantzig pricing with first-index argmin, Bland fallback on a
 * degenerate step, single-artificial Phase I solved by a recursive call.
 */
#include "gonum_lp.h"
#include "gonum_linalg.h"
#include "gonum_blas.h"

#include <stdlib.h>
#include <time.h>

/* simplex.go:42-58 */
#define INIT_POS_TOL 1e-13
#define BLAND_NEG_TOL 1e-14
#define R_ROUND_TOL 1e-13
#define D_ROUND_TOL 1e-13
#define PHASE_I_ZERO_TOL 1e-12
#define BLAND_ZERO_TOL 1e-12

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void oracle_ctx_init(oracle_ctx *ctx) {
    memset(ctx, 0, sizeof(*ctx));
    ctx->stop_after_pivots = -1;
}

/* floats.MinIdx, floats/floats.go:458-474: first index of the minimum, NaNs skipped */
static int64_t min_idx(const double *s, int64_t n) {
    double mn = NAN;
    int64_t ind = 0;
    for (int64_t i = 0; i < n; i++) {
        double v = s[i];
        if (isnan(v)) continue;
        if (v < mn || isnan(mn)) { mn = v; ind = i; }
    }
    return ind;
}

/* mat.Col (mat/matrix.go:210-239) on a row-major Dense */
static void get_col(double *dst, int64_t j, const double *A, int64_t lda, int64_t m) {
    for (int64_t i = 0; i < m; i++) dst[i] = A[i * lda + j];
}
/* extractColumns, simplex.go:474-488 (dst is m×ncols, stride ncols) */
static void extract_columns(double *dst, int64_t ncols, const double *A, int64_t lda, int64_t m, const int64_t *cols) {
    for (int64_t j = 0; j < ncols; j++)
        for (int64_t i = 0; i < m; i++) dst[i * ncols + j] = A[i * lda + cols[j]];
}

/* SolveVec wrappers that count factorizations */
static int solve_vec(oracle_ctx *ctx, int64_t n, const double *a, int64_t lda, double *x) {
    ctx->lu_factorizations++;
    return g_solve_vec(n, a, lda, x);
}
static int solve_vec_trans(oracle_ctx *ctx, int64_t n, const double *a, int64_t lda, double *x) {
    ctx->lu_factorizations++;
    return g_solve_vec_trans(n, a, lda, x);
}

/* verifyInputs, simplex.go:385-439 (length panics are checked by the caller) */
static int verify_inputs(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n) {
    for (int64_t i = 0; i < m; i++) {
        int is_zero = 1;
        for (int64_t j = 0; j < n; j++)
            if (A[i * lda + j] != 0) { is_zero = 0; break; }
        if (is_zero && b[i] != 0) return ORACLE_ERR_INFEASIBLE;
        else if (is_zero) return ORACLE_ERR_ZERO_ROW;
    }
    for (int64_t j = 0; j < n; j++) {
        int is_zero = 1;
        for (int64_t i = 0; i < m; i++)
            if (A[i * lda + j] != 0) { is_zero = 0; break; }
        if (is_zero && c[j] < 0) return ORACLE_ERR_UNBOUNDED;
        else if (is_zero) return ORACLE_ERR_ZERO_COLUMN;
    }
    return ORACLE_OK;
}

/* initializeFromBasic, simplex.go:447-471.  0 = feasible; 1 = "singular"; 2 = "not feasible" */
static int initialize_from_basic(oracle_ctx *ctx, double *xb, const double *ab, const double *b, int64_t m) {
    /* xbMat.SolveVec(ab, b): the receiver only takes b's values once LU.Solve copies them
     * (mat/lu.go:318), i.e. not when Det()==0 */
    double *tmp = (double *)malloc(sizeof(double) * (size_t)m);
    memcpy(tmp, b, sizeof(double) * (size_t)m);
    int rc = solve_vec(ctx, m, ab, m, tmp);
    if (rc != 2) memcpy(xb, tmp, sizeof(double) * (size_t)m);
    free(tmp);
    if (rc != 0) return 1;
    for (int64_t i = 0; i < m; i++)
        if (xb[i] < -INIT_POS_TOL) return 2;
    return 0;
}

/* does column j of A equal a unit vector? returns the row of the 1, or -1 */
static int64_t unit_row(const double *A, int64_t lda, int64_t m, int64_t j) {
    int64_t row = -1;
    for (int64_t i = 0; i < m; i++) {
        double v = A[i * lda + j];
        if (v == 0) continue;
        if (v != 1 || row != -1) return -1;
        row = i;
    }
    return row;
}

/* findLinearlyIndependent, simplex.go:611-637 */
int64_t oracle_find_linearly_independent(const double *A, int64_t lda, int64_t m, int64_t n, int64_t *idxs,
                                         oracle_ctx *ctx) {
    int64_t cnt = 0;
    double *columns = (double *)calloc((size_t)(m * m), sizeof(double));
    char *row_used = (char *)calloc((size_t)m, 1);
    int all_unit = ctx->fast_initial_basis; /* stays 1 while every accepted column is a distinct unit vector */
    for (int64_t i = n - 1; i >= 0; i--) {
        if (cnt == m) break;
        for (int64_t r = 0; r < m; r++) columns[r * m + cnt] = A[r * lda + i];
        if (cnt == 0) {
            if (all_unit) {
                int64_t ur = unit_row(A, lda, m, i);
                if (ur >= 0) row_used[ur] = 1; else all_unit = 0;
            }
            idxs[cnt++] = i;
            continue;
        }
        if (all_unit) {
            /* fast path: a unit column on a fresh row next to distinct unit columns has cond == 1 */
            int64_t ur = unit_row(A, lda, m, i);
            if (ur >= 0 && !row_used[ur]) { row_used[ur] = 1; idxs[cnt++] = i; continue; }
            all_unit = 0;
        }
        ctx->cond_evaluations++;
        if (g_cond1(m, cnt + 1, columns, m) > 1e12) continue; /* :630 not linearly independent */
        idxs[cnt++] = i;
    }
    free(columns); free(row_used);
    return cnt;
}

/* computeMove, simplex.go:306-342 */
static int compute_move(oracle_ctx *ctx, double *move, int64_t minidx, const double *A, int64_t lda, int64_t m,
                        const double *ab, const double *xb, const int64_t *nonbasic, double *d) {
    get_col(d, nonbasic[minidx], A, lda, m);       /* :308 */
    int rc = solve_vec(ctx, m, ab, m, d);          /* :315 d = ab^-1 a_e */
    if (rc != 0) return ORACLE_ERR_LINSOLVE;       /* :316-318 */
    for (int64_t i = 0; i < m; i++) d[i] *= -1;    /* :319 floats.Scale(-1, d) */
    for (int64_t i = 0; i < m; i++)
        if (fabs(d[i]) < D_ROUND_TOL) d[i] = 0;    /* :321-325 */
    if (d[min_idx(d, m)] >= 0) return ORACLE_ERR_UNBOUNDED; /* :328 */
    for (int64_t i = 0; i < m; i++) {              /* :334-340 */
        if (d[i] >= 0) move[i] = INFINITY;
        else move[i] = xb[i] / fabs(d[i]);
    }
    return ORACLE_OK;
}

/* replaceBland, simplex.go:347-383 */
static int replace_bland(oracle_ctx *ctx, const double *A, int64_t lda, int64_t m, int64_t nn /* n-m */,
                         const double *ab, const double *xb, const int64_t *basic, const int64_t *nonbasic,
                         const double *r, double *move, double *d, int64_t *replace_out, int64_t *minidx_out) {
    int64_t *bicopy = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    double *abtmp = (double *)malloc(sizeof(double) * (size_t)(m * m));
    int rc = ORACLE_ERR_BLAND;
    for (int64_t i = 0; i < nn; i++) {
        if (r[i] > -BLAND_NEG_TOL) continue;
        int err = compute_move(ctx, move, i, A, lda, m, ab, xb, nonbasic, d);
        if (err != ORACLE_OK) { rc = err; goto done; }
        int64_t replace = min_idx(move, m);
        if (fabs(move[replace]) > BLAND_ZERO_TOL) { *replace_out = replace; *minidx_out = i; rc = ORACLE_OK; goto done; }
        for (int64_t rp = 0; rp < m; rp++) {
            if (move[rp] > BLAND_ZERO_TOL) continue;
            memcpy(bicopy, basic, sizeof(int64_t) * (size_t)m);
            bicopy[rp] = nonbasic[i];
            extract_columns(abtmp, m, A, lda, m, bicopy);
            ctx->cond_evaluations++;
            if (g_cond1(m, m, abtmp, m) < 1e16) { *replace_out = rp; *minidx_out = i; rc = ORACLE_OK; goto done; }
        }
    }
done:
    free(bicopy); free(abtmp);
    return rc;
}

static int simplex(oracle_ctx *ctx, int depth, const int64_t *initial_basic, const double *c, const double *A,
                   int64_t lda, const double *b, int64_t m, int64_t n, double tol, double *opt_f, double *opt_x,
                   int32_t *has_x, int64_t *basis_out);

/* findInitialBasic, simplex.go:492-607.  On success fills basic (m), ab (m×m), xb (m). */
static int find_initial_basic(oracle_ctx *ctx, int depth, const double *A, int64_t lda, const double *b, int64_t m,
                              int64_t n, int64_t *basic, double *ab, double *xb) {
    int64_t cnt = oracle_find_linearly_independent(A, lda, m, n, basic, ctx);
    if (cnt != m) return ORACLE_ERR_SINGULAR; /* :495-497 */
    extract_columns(ab, m, A, lda, m, basic);
    for (int64_t i = 0; i < m; i++) xb[i] = 0;
    if (initialize_from_basic(ctx, xb, ab, b, m) == 0) return ORACLE_OK; /* :504-507 */

    /* Phase I, :529-556 */
    ctx->phase1_used = 1;
    int64_t minidx = min_idx(xb, m);
    double *ax1 = (double *)malloc(sizeof(double) * (size_t)m);
    double *col = (double *)malloc(sizeof(double) * (size_t)m);
    memcpy(ax1, b, sizeof(double) * (size_t)m);
    for (int64_t i = 0; i < m; i++) {
        if (i == minidx) continue;
        get_col(col, basic[i], A, lda, m);
        for (int64_t k = 0; k < m; k++) ax1[k] = -1 * col[k] + ax1[k]; /* floats.Sub, floats.go:880-885 */
    }
    int64_t n1 = n + 1;
    double *anew = (double *)malloc(sizeof(double) * (size_t)(m * n1));
    for (int64_t i = 0; i < m; i++) {
        memcpy(anew + i * n1, A + i * lda, sizeof(double) * (size_t)n);
        anew[i * n1 + n] = ax1[i];
    }
    basic[minidx] = n;
    double *c1 = (double *)calloc((size_t)n1, sizeof(double));
    c1[n] = 1;
    double *xopt = (double *)calloc((size_t)n1, sizeof(double));
    int64_t *newbasic = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    double f1;
    int32_t hx = 0;
    int rc = simplex(ctx, depth + 1, basic, c1, anew, n1, b, m, n1, 1e-10, &f1, xopt, &hx, newbasic); /* :556 */
    int ret;
    if (rc != ORACLE_OK) {
        if (rc == ORACLE_ERR_BAD_SHAPE || rc == ORACLE_ERR_PANIC) ret = rc;
        else { ctx->wrapped_code = rc; ret = ORACLE_ERR_PHASE1_WRAPPED; } /* :557-559 */
        goto out;
    }
    if (fabs(xopt[n]) > PHASE_I_ZERO_TOL) { ret = ORACLE_ERR_INFEASIBLE; goto out; } /* :563-565 */
    {
        int64_t added = -1;
        for (int64_t i = 0; i < m; i++) {
            if (newbasic[i] == n) added = i;
            xb[i] = xopt[newbasic[i]];
        }
        if (added == -1) { /* :576-579 */
            memcpy(basic, newbasic, sizeof(int64_t) * (size_t)m);
            extract_columns(ab, m, A, lda, m, basic);
            ret = ORACLE_OK;
            goto out;
        }
        /* :581-606 the artificial stayed basic at zero: try to exchange it */
        ctx->art_exchanges++;
        char *inbasic = (char *)calloc((size_t)n1, 1);
        for (int64_t i = 0; i < m; i++) inbasic[newbasic[i]] = 1;
        ret = ORACLE_ERR_INFEASIBLE;
        for (int64_t i = 0; i < n1; i++) {
            if (inbasic[i]) continue;
            newbasic[added] = i;
            extract_columns(ab, m, A, lda, m, newbasic); /* (:594-600 sets one column; same matrix) */
            if (initialize_from_basic(ctx, xb, ab, b, m) == 0) {
                memcpy(basic, newbasic, sizeof(int64_t) * (size_t)m);
                ret = ORACLE_OK;
                break;
            }
        }
        free(inbasic);
    }
out:
    free(ax1); free(col); free(anew); free(c1); free(xopt); free(newbasic);
    return ret;
}

/* simplex, simplex.go:93-302 */
static int simplex(oracle_ctx *ctx, int depth, const int64_t *initial_basic, const double *c, const double *A,
                   int64_t lda, const double *b, int64_t m, int64_t n, double tol, double *opt_f, double *opt_x,
                   int32_t *has_x, int64_t *basis_out) {
    *has_x = 0;
    int err = verify_inputs(c, A, lda, b, m, n);
    if (err != ORACLE_OK) {
        *opt_f = (err == ORACLE_ERR_UNBOUNDED) ? -INFINITY : NAN; /* :95-100 */
        return err;
    }
    if (m == n) { /* :103-119 exactly constrained */
        double *x = (double *)malloc(sizeof(double) * (size_t)n);
        memcpy(x, b, sizeof(double) * (size_t)n);
        int rc = solve_vec(ctx, n, A, lda, x);
        if (rc != 0) { free(x); *opt_f = NAN; return ORACLE_ERR_SINGULAR; }
        for (int64_t i = 0; i < n; i++)
            if (x[i] < 0) { free(x); *opt_f = NAN; return ORACLE_ERR_INFEASIBLE; }
        *opt_f = g_dot_unitary(n, x, c);
        memcpy(opt_x, x, sizeof(double) * (size_t)n);
        *has_x = 1;
        free(x);
        return ORACLE_OK;
    }

    int64_t nn = n - m;
    int64_t *basic = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    int64_t *nonbasic = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nn > 0 ? nn : 1));
    double *ab = (double *)malloc(sizeof(double) * (size_t)(m * m));
    double *an = (double *)malloc(sizeof(double) * (size_t)(m * (nn > 0 ? nn : 1)));
    double *xb = (double *)calloc((size_t)m, sizeof(double));
    double *cb = (double *)malloc(sizeof(double) * (size_t)m);
    double *cn = (double *)malloc(sizeof(double) * (size_t)(nn > 0 ? nn : 1));
    double *r = (double *)malloc(sizeof(double) * (size_t)(nn > 0 ? nn : 1));
    double *data = (double *)malloc(sizeof(double) * (size_t)(nn > 0 ? nn : 1));
    double *move = (double *)malloc(sizeof(double) * (size_t)m);
    double *d = (double *)malloc(sizeof(double) * (size_t)m);
    double *tmp = (double *)malloc(sizeof(double) * (size_t)m);
    char *inbasic = (char *)calloc((size_t)n, 1);
    int ret = ORACLE_OK;

    if (initial_basic) { /* :147-160 */
        extract_columns(ab, m, A, lda, m, initial_basic);
        if (initialize_from_basic(ctx, xb, ab, b, m) != 0) { ret = ORACLE_ERR_PANIC; *opt_f = NAN; goto cleanup; }
        memcpy(basic, initial_basic, sizeof(int64_t) * (size_t)m);
    } else { /* :161-167 */
        ret = find_initial_basic(ctx, depth, A, lda, b, m, n, basic, ab, xb);
        if (ret != ORACLE_OK) { *opt_f = NAN; goto cleanup; }
    }

    /* :173-197 */
    for (int64_t i = 0; i < m; i++) inbasic[basic[i]] = 1;
    {
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++)
            if (!inbasic[i]) nonbasic[k++] = i;
    }
    for (int64_t i = 0; i < m; i++) cb[i] = c[basic[i]];
    for (int64_t i = 0; i < nn; i++) cn[i] = c[nonbasic[i]];
    extract_columns(an, nn, A, lda, m, nonbasic);

    double t0 = now_s();
    int64_t my_pivots = 0;
    err = ORACLE_OK;
    for (;;) { /* :233-293 */
        if (depth == 0 && ctx->stop_after_pivots >= 0 && my_pivots >= ctx->stop_after_pivots) { ctx->truncated = 1; break; }
        /* :236 tmp = solve(ab^T, cb) */
        memcpy(tmp, cb, sizeof(double) * (size_t)m);
        {
            /* the receiver `tmp` is a fresh VecDense: on Det()==0 it stays zero-valued; either way we break */
            int rc = solve_vec_trans(ctx, m, ab, m, tmp);
            if (rc != 0) { err = ORACLE_ERR_CONDITION; break; }
        }
        /* :240-243 data = an^T tmp (Dgemv Trans: row-axpy over i with zero-skip), r = cn - data */
        for (int64_t j = 0; j < nn; j++) data[j] = 0;
        for (int64_t i = 0; i < m; i++) {
            double t = 1 * tmp[i];
            if (t != 0) g_axpy_to(nn, data, t, an + i * nn, data);
        }
        for (int64_t j = 0; j < nn; j++) r[j] = -1 * data[j] + cn[j]; /* floats.SubTo, floats.go:889-898 */

        int64_t minidx = min_idx(r, nn); /* :247 */
        if (r[minidx] >= -tol) break;    /* :248 */
        for (int64_t j = 0; j < nn; j++)
            if (fabs(r[j]) < R_ROUND_TOL) r[j] = 0; /* :252-256 */

        err = compute_move(ctx, move, minidx, A, lda, m, ab, xb, nonbasic, d); /* :259 */
        if (err != ORACLE_OK) {
            if (err == ORACLE_ERR_UNBOUNDED) { *opt_f = -INFINITY; ret = err; goto cleanup; } /* :261-263 */
            break;
        }
        int64_t replace = min_idx(move, m); /* :268 */
        int bland = 0;
        if (move[replace] <= 0) { /* :269 */
            bland = 1;
            ctx->bland_steps++;
            err = replace_bland(ctx, A, lda, m, nn, ab, xb, basic, nonbasic, r, move, d, &replace, &minidx);
            if (err != ORACLE_OK) {
                if (err == ORACLE_ERR_UNBOUNDED) { *opt_f = -INFINITY; ret = err; goto cleanup; }
                break;
            }
        }
        /* record */
        if (ctx->trace && ctx->trace_len < ctx->trace_cap) {
            oracle_pivot *p = &ctx->trace[ctx->trace_len];
            p->phase = depth == 0 ? 2 : 1; p->bland = bland; p->min_idx = minidx; p->replace = replace;
            p->entering = nonbasic[minidx]; p->leaving = basic[replace];
        }
        ctx->trace_len++;
        if (depth == 0) ctx->pivots_phase2++; else ctx->pivots_phase1++;
        my_pivots++;

        /* :280-285 swap */
        { int64_t t = basic[replace]; basic[replace] = nonbasic[minidx]; nonbasic[minidx] = t; }
        { double t = cb[replace]; cb[replace] = cn[minidx]; cn[minidx] = t; }
        for (int64_t i = 0; i < m; i++) {
            double t = ab[i * m + replace];
            ab[i * m + replace] = an[i * nn + minidx];
            an[i * nn + minidx] = t;
        }
        /* :288-292 xb = solve(ab, b); receiver keeps its old values when Det()==0 */
        memcpy(tmp, b, sizeof(double) * (size_t)m);
        {
            int rc = solve_vec(ctx, m, ab, m, tmp);
            if (rc != 2) memcpy(xb, tmp, sizeof(double) * (size_t)m);
            if (rc != 0) { err = ORACLE_ERR_CONDITION; break; }
        }
    }
    if (depth == 0) ctx->seconds_loop = now_s() - t0;
    /* :296-301 */
    *opt_f = g_dot_unitary(m, cb, xb);
    for (int64_t i = 0; i < n; i++) opt_x[i] = 0;
    for (int64_t i = 0; i < m; i++) opt_x[basic[i]] = xb[i];
    *has_x = 1;
    if (basis_out) memcpy(basis_out, basic, sizeof(int64_t) * (size_t)m);
    ret = err;

cleanup:
    free(basic); free(nonbasic); free(ab); free(an); free(xb); free(cb); free(cn); free(r); free(data);
    free(move); free(d); free(tmp); free(inbasic);
    return ret;
}

int oracle_lp_simplex(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n,
                      double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                      int64_t *basis_out, oracle_ctx *ctx) {
    oracle_ctx local;
    if (!ctx) { oracle_ctx_init(&local); ctx = &local; }
    ctx->trace_len = 0;
    ctx->pivots_phase1 = ctx->pivots_phase2 = ctx->bland_steps = 0;
    ctx->lu_factorizations = ctx->cond_evaluations = 0;
    ctx->phase1_used = ctx->truncated = ctx->wrapped_code = 0;
    ctx->seconds_loop = 0;
    ctx->art_exchanges = 0;
    *has_x = 0;
    if (m <= 0 || n <= 0 || lda < n) { *opt_f = NAN; return ORACLE_ERR_BAD_SHAPE; }
    return simplex(ctx, 0, initial_basic, c, A, lda, b, m, n, tol, opt_f, opt_x, has_x, basis_out);
}
