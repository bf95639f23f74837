/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of gonum's lp.Simplex as GoMILP calls it
 * (reference: subproblem.go:154,172 -> vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:88).
 * Pinned against the reference's own golden vectors K1-K9 by
 * tests/test_oracle_golden.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may call this.
 */
#ifndef GOMILP_ORACLE_GONUM_LP_H
#define GOMILP_ORACLE_GONUM_LP_H

#include <stdint.h>

/* Status codes: numerically identical to include/gomilp_lp.h so tests compare them directly.
 * They name the lp.Err* sentinels of simplex.go:26-34. */
enum {
    ORACLE_OK = 0,
    ORACLE_ERR_BLAND = 1,
    ORACLE_ERR_INFEASIBLE = 2,
    ORACLE_ERR_LINSOLVE = 3,
    ORACLE_ERR_UNBOUNDED = 4,
    ORACLE_ERR_SINGULAR = 5,
    ORACLE_ERR_ZERO_COLUMN = 6,
    ORACLE_ERR_ZERO_ROW = 7,
    ORACLE_ERR_CONDITION = 8,      /* mat.Condition from a mid-loop SolveVec (simplex.go:236-239,289-292) */
    ORACLE_ERR_PHASE1_WRAPPED = 9, /* fmt.Errorf("lp: error finding feasible basis: %s") simplex.go:558 */
    ORACLE_ERR_BAD_SHAPE = 10,     /* the reference panics (simplex.go:387-398) */
    ORACLE_ERR_PANIC = 11          /* any other reference panic (simplex.go:150,157) */
};

/* one record per pivot, in execution order (Phase I pivots of the recursive call first) */
typedef struct {
    int32_t phase;      /* 1 = inside the Phase-I recursive simplex, 2 = Phase II */
    int32_t bland;      /* 1 when (replace, minIdx) came from replaceBland */
    int64_t min_idx;    /* position in nonBasicIdx (simplex.go:247) */
    int64_t replace;    /* position in basicIdxs   (simplex.go:268) */
    int64_t entering;   /* variable id nonBasicIdx[minIdx] before the swap */
    int64_t leaving;    /* variable id basicIdxs[replace] before the swap */
} oracle_pivot;

typedef struct {
    /* ---- options ---- */
    int32_t fast_initial_basis; /* 1: when the columns met by the descending scan of simplex.go:618-635 are
                                   distinct unit vectors, accept them without evaluating mat.Cond (cond == 1
                                   exactly there; tests/test_oracle.py checks the equivalence) */
    int64_t stop_after_pivots;  /* >=0: leave the outermost Phase-II loop after this many pivots (CPU baseline
                                   on a bounded sample); <0: run to completion */
    /* ---- trace (caller-owned buffer, may be NULL) ---- */
    oracle_pivot *trace;
    int64_t trace_cap;
    int64_t trace_len;          /* out: number of pivots performed (may exceed trace_cap) */
    /* ---- counters (out) ---- */
    int64_t pivots_phase1, pivots_phase2, bland_steps, lu_factorizations, cond_evaluations;
    int32_t phase1_used;        /* the slack / independent basis was infeasible and Phase I ran */
    int32_t truncated;          /* stop_after_pivots was hit */
    int32_t wrapped_code;       /* ORACLE_ERR_* inside a PHASE1_WRAPPED error */
    double seconds_loop;        /* wall time of the outermost Phase-II loop */
    int64_t art_exchanges;      /* the Phase-I artificial stayed basic at level zero and was exchanged (simplex.go:581-606) */
} oracle_ctx;

void oracle_ctx_init(oracle_ctx *ctx);

/*
 * lp.Simplex(c, A, b, tol, initialBasic) — simplex.go:88-91.
 *   A: m×n row-major with leading dimension lda (GoMILP always passes *mat.Dense).
 *   initial_basic: NULL or m column indices.
 *   opt_f: objective (NaN / -Inf exactly as the reference returns them).
 *   opt_x: caller-owned, length n; written only when *has_x = 1 (the reference returns nil otherwise).
 *   basis_out: NULL or length m: final basicIdxs in positional order.
 * Returns ORACLE_*.  A mid-loop failure returns the error code AND the current point (has_x = 1).
 */
int oracle_lp_simplex(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n,
                      double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                      int64_t *basis_out, oracle_ctx *ctx);

/* findLinearlyIndependent (simplex.go:611-637); idxs has room for m entries; returns count */
int64_t oracle_find_linearly_independent(const double *A, int64_t lda, int64_t m, int64_t n, int64_t *idxs,
                                         oracle_ctx *ctx);

#endif
