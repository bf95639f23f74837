/*
 * gomilp_lp.h — C-ABI of the MI355X-native LP-relaxation engine.
 *
 * Drop-in boundary for GoMILP's one hot path: the call
 *     z, x, err = lp.Simplex(c, A, b, 0, nil)
 * at /root/reference/subproblem.go:154 (branch-and-bound child) and :172 (root), i.e.
 * /root/reference/vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:88.
 * The Go side binds these entry points through cgo (INTEGRATION.md shows the stub); all
 * arguments are plain pointers and sizes, the caller owns every host buffer, nothing is
 * retained after return (cgo pointer rules), and every entry point is thread-safe.
 *
 * There is no CPU fallback: without a usable HIP device every solve returns
 * GOMILP_ERR_DEVICE.
 */
#ifndef GOMILP_LP_H
#define GOMILP_LP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  1..7 are the lp.Err* sentinels of simplex.go:26-34 that the Go wrapper maps
 * back to the SAME sentinel values (GoMILP tests them by identity: ilp.go:37-40, tree.go:77,266-273). */
typedef enum {
    GOMILP_OK = 0,
    GOMILP_ERR_BLAND = 1,          /* lp.ErrBland       simplex.go:27 */
    GOMILP_ERR_INFEASIBLE = 2,     /* lp.ErrInfeasible  simplex.go:28 */
    GOMILP_ERR_LINSOLVE = 3,       /* lp.ErrLinSolve    simplex.go:29 */
    GOMILP_ERR_UNBOUNDED = 4,      /* lp.ErrUnbounded   simplex.go:30 — opt_f = -Inf, no x */
    GOMILP_ERR_SINGULAR = 5,       /* lp.ErrSingular    simplex.go:31 */
    GOMILP_ERR_ZERO_COLUMN = 6,    /* lp.ErrZeroColumn  simplex.go:32 */
    GOMILP_ERR_ZERO_ROW = 7,       /* lp.ErrZeroRow     simplex.go:33 */
    GOMILP_ERR_CONDITION = 8,      /* mat.Condition out of a mid-loop solve, simplex.go:236-239,289-292 */
    GOMILP_ERR_PHASE1_WRAPPED = 9, /* fmt.Errorf("lp: error finding feasible basis: %s"), simplex.go:558 */
    GOMILP_ERR_BAD_SHAPE = 10,     /* the reference panics: simplex.go:387-398 */
    GOMILP_ERR_PANIC = 11,         /* any other reference panic (simplex.go:150,157) */
    GOMILP_ERR_DEVICE = 12,        /* HIP runtime failure / no device / extension not built for it */
    GOMILP_ERR_UNSUPPORTED = 13    /* input outside what this round's device path covers */
} gomilp_status;

/* Per-solve statistics (all optional output). */
typedef struct gomilp_lp_stats {
    int64_t pivots_phase1;      /* pivots of the Phase-I recursive simplex (simplex.go:556) */
    int64_t pivots_phase2;      /* pivots of the Phase-II loop (simplex.go:233-293) */
    int64_t bland_steps;        /* degenerate steps routed through replaceBland (simplex.go:269-277) */
    int64_t refreshes;          /* times x_B / y were recomputed from B^-1 */
    int64_t kernel_launches;    /* pivot-loop kernel launches enqueued */
    int32_t phase1_used;        /* 1 when the initial basis was infeasible and Phase I ran */
    int32_t device_id;
    int32_t wrapped_status;     /* inner status when the result is GOMILP_ERR_PHASE1_WRAPPED */
    int32_t pipeline;           /* pivot pipeline that ran: 0 three-kernel, 1 fused two-kernel, 2 single-kernel tableau,
                                   3 blocked tableau (deferred rank-K updates) */
    double seconds_total;       /* host wall clock of the call (upload included for the flat call) */
    double seconds_upload;      /* host->device copies + layout conversion */
    double seconds_pivot_loop;  /* HIP-event time of all pivot-loop kernels (Phase I + II) */
    double seconds_final_solve; /* gonum-order LU on device + host triangular solves */
    double drift_xb;            /* max |x_B(updated) - x_B(fresh LU)| at termination: accuracy of the B^-1 updates */
    double pivot_kernel_seconds[4]; /* HIP-event time of sampled pivots: [0] pricing kernel, [1] ftran kernel (0 when fused),
                                       [2] update(+ftran) kernel, [3] number of sampled pivots */
    double seconds_final_device; /* part of seconds_final_solve: gather + LU kernels + device->host copy */
    double seconds_final_host;   /* part of seconds_final_solve: the two triangular solves on the host */
    int64_t lu_dense_steps;      /* elimination steps of the final LU that did arithmetic (the rest hit the unit-column fast path) */
    int64_t lu_rounds;           /* panel rounds of the compressed LU schedule (0 for the other schedules) */
    int64_t art_exchanges;       /* 1 when a zero-level artificial was exchanged out of the basis after Phase I (simplex.go:581-606) */
    int64_t cond_fallbacks;      /* exact condition-number evaluations made because a cheap guard was near a threshold */
    int64_t device_retries;      /* 1 when a transient device condition (a workgroup of the multi-workgroup block kernel, or of a look-ahead
                                    launch of the final solve's LU, was not resident in time) made the engine repeat the solve on the
                                    single-workgroup kernels, resp. the factorization with the plain LU schedule */
    double cond1_final;          /* exact kappa_1 / kappa_inf of the basis the Phase-II loop ended with, from the resident tableau (slack-basis */
    double condinf_final;        /* starts beyond 64 rows; 0: not evaluated): kappa_1 > 1e16 -> GOMILP_ERR_CONDITION like mat/lu.go:321 */
} gomilp_lp_stats;

/* One record per pivot, execution order (Phase I first).  Same fields as the oracle's trace. */
typedef struct gomilp_pivot {
    int32_t phase;    /* 1 = Phase I, 2 = Phase II */
    int32_t bland;    /* 1 when chosen by the Bland rule */
    int64_t min_idx;  /* position in nonBasicIdx (simplex.go:247) */
    int64_t replace;  /* position in basicIdxs   (simplex.go:268) */
    int64_t entering; /* variable id entering */
    int64_t leaving;  /* variable id leaving */
} gomilp_pivot;

/* ------------------------------------------------------------------------------------------
 * Flat drop-in: signature-isomorphic to
 *   func Simplex(c []float64, A mat.Matrix, b []float64, tol float64, initialBasic []int)
 *        (optF float64, optX []float64, err error)                          — simplex.go:88
 * A is the RawMatrix() of the *mat.Dense GoMILP always passes: row-major m x n, stride lda.
 * opt_x (length n, caller-owned) is written only when *has_x = 1 (the reference returns nil x on
 * most errors); a mid-loop failure returns the error AND the current point, like the reference.
 * basis_out (nullable, length m) receives the final basicIdxs in positional order.
 * initial_basic (nullable, exactly m entries; GoMILP passes nil): a supplied feasible basis skips Phase I
 * (simplex.go:147-160); an index out of range, a singular or an infeasible set return GOMILP_ERR_PANIC (the
 * reference panics); like any non-slack starting basis it needs the host copy of A, kept for m * n <= 2^25
 * (GOMILP_ERR_UNSUPPORTED above): the column search runs on the device from 224 rows on.
 * ---------------------------------------------------------------------------------------- */
int gomilp_lp_simplex(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n,
                      double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                      int64_t *basis_out, gomilp_lp_stats *stats);

/* ------------------------------------------------------------------------------------------
 * Handle API: one context per (thread, GPU); problems stay resident in HBM so that a B&B
 * frontier uploads the root (c, A0, b0) once (subproblem.go:20-29 shares them by pointer).
 * ---------------------------------------------------------------------------------------- */
typedef struct gomilp_ctx gomilp_ctx;

/* device < 0: current HIP device.  Returns NULL (and *status) on failure. */
gomilp_ctx *gomilp_ctx_create(int device, int *status);
void gomilp_ctx_destroy(gomilp_ctx *ctx);
int gomilp_ctx_device(const gomilp_ctx *ctx);
/* knobs: "chunk" (pivots enqueued between host checks), "refresh" (pivots between x_B/y recomputations, 0 = never),
 * "trace" (1 = record pivots), "max_pivots" (safety cap, 0 = none).  Developer knobs of the pivot pipelines (tests force
 * kernel instances with them; results do not depend on them): "tableau", "blocked", "block_k", "bt_nt" (threads of the
 * single-workgroup block kernel), "bt_groups" (-1: single-workgroup block kernels only, 0: by shape, 2 / 4 / 8: that many
 * workgroups), "bt_old", "bt_stamps", "sample_events"; of the persistent loop kernel: "bt_lag" (0: the launch pairs of round 2),
 * "loop_chunk" (pivots per launch), "loop_g" (8 / 16 pivot workgroups), "loop_k" (8 / 12 / 16 pivots per block), "loop_upd"
 * (update workgroups that take part), "loop_grid", "poll_delay", "loop_rep" (default 0, opt-in: the pivot role with replicated reduced costs, ONE exchange
 * per pivot — bit-identical pivots, measured slower: DESIGN.md section 2.1d); of the bit-exact final solve: "lu_blocked" (3 default: compressed rounds
 * in the look-ahead schedule — one launch per round, the panel beside the previous round's update — for bases beyond 768 rows while the
 * engine holds the device's loop slots, 2: compressed rounds with the whole update behind each panel, 1: blocked panels, 0: one launch per
 * column — all bit-identical), "lu_look" (0: never the look-ahead schedule; a pool sets it on its workers).  The
 * diagnostic flavour of the library (libgomilp_hip_debug.so, GOMILP_DEBUG_BUILD=1) adds "bt_fault" and the GOMILP_DEBUG_* / GOMILP_LUC_*
 * environment hooks; the product library has none of them.  Knobs that DO change what is decided, and how faithfully:
 * "exact_degenerate" (0 never, 1 default: bases of up to 256 rows, non-slack starts and badly scaled inputs, 2 always — degenerate,
 * tied and tiny pivots are decided on fresh gonum-order solves; 3 strict: EVERY pivot and the stop test are the
 * reference's iteration on fresh gonum-order solves with its condition guard, simplex.go:233-293 — milliseconds per pivot, the mode that
 * follows the reference wherever the rounding noise of its solves leads), "cond_guard" (1 default: gonum's
 * mat.Condition guard, from a pivot-by-pivot replay up to 64 rows and from the tableau's exact condition numbers beyond — at every exact
 * step (Phase I too), on the final basis, and in front of any pivot whose element is of rounding-noise size).
 * Returns GOMILP_OK or GOMILP_ERR_BAD_SHAPE. */
int gomilp_ctx_set(gomilp_ctx *ctx, const char *key, int64_t value);

/* Upload a standard-form LP (row-major A, stride lda) and keep it resident.  Returns a problem id >= 0, or
 * -(gomilp_status) on failure.  Replaces the per-call operands of simplex.go:88. */
int64_t gomilp_lp_upload(gomilp_ctx *ctx, const double *c, const double *A, int64_t lda, const double *b, int64_t m,
                         int64_t n);
int gomilp_lp_free(gomilp_ctx *ctx, int64_t problem);

/* Solve a resident problem (inputs already in HBM: the timed region of bench.py).  Same outputs as the flat call. */
int gomilp_lp_solve_resident(gomilp_ctx *ctx, int64_t problem, double tol, const int64_t *initial_basic,
                             double *opt_f, double *opt_x, int32_t *has_x, int64_t *basis_out,
                             gomilp_lp_stats *stats);

/* Pivot trace of the last solve on this context (needs "trace" = 1).  Copies up to cap records, returns the
 * total number of pivots performed. */
int64_t gomilp_lp_last_trace(gomilp_ctx *ctx, gomilp_pivot *out, int64_t cap);

/* Child of a resident root problem, assembled on the device: the K branch-and-bound rows
 * G#_k = sign_k * e_{var_k}, h#_k = rhs_k of /root/reference/subproblem.go:36-44,245-255 are appended exactly like
 * convertToEqualities (subproblem.go:81-139) does: A' = [[A0, 0], [G#, I_K]], c' = [c0, 0], b' = [b0; h#].
 * Returns a problem id (solve it with gomilp_lp_solve_resident; x has n0 + K entries, the caller keeps the
 * first n0 like subproblem.go:157-159) or -(gomilp_status). */
int64_t gomilp_lp_upload_child(gomilp_ctx *ctx, int64_t root_problem, int32_t K, const int32_t *var, const double *sign,
                               const double *rhs);

/* ------------------------------------------------------------------------------------------
 * Frontier API: one FIFO level of the enumeration tree (independent relaxations sharing the root data,
 * tree.go:98-100,196-205) solved by a pool of `workers` contexts on ONE GPU, each on its own HIP stream.
 * Multi-GPU runs use one pool per process/GPU and shard the children by index (bench.py).
 * ---------------------------------------------------------------------------------------- */
typedef struct gomilp_pool gomilp_pool;

typedef struct gomilp_frontier_stats {
    int64_t relaxations;
    int64_t pivots_phase1, pivots_phase2, bland_steps, phase1_runs;
    int64_t kernel_launches;
    int32_t workers, device_id;
    double seconds_total;       /* host wall clock of the call */
    double seconds_busy_sum;    /* sum over workers of time spent inside solves */
    int64_t batched_relaxations; /* relaxations whose pivot loops ran in the device-batched schedule (one launch per kernel type
                                    for the whole wave) */
    int64_t host_fallbacks;      /* relaxations the batched schedule handed to a worker's single-relaxation engine */
    int64_t supersteps;          /* host round trips of the batched schedule for the WHOLE wave */
    double seconds_batch;        /* wall clock of the batched schedule */
    int64_t blocks;              /* block steps of the batched schedule (one batched inner launch + one batched update launch each) */
    int64_t blocks_sampled;      /* of which timed with HIP events (pool knob "sample_batch") */
    double seconds_inner_kernels;  /* HIP-event time of the sampled batched inner launches (k_bt_inner2_batch) */
    double seconds_update_kernels; /* ... of the sampled batched update launches (k_bt_update_tiled_batch) */
    /* warm start (gomilp_frontier_solve_warm) */
    int64_t warm_started;        /* relaxations that started from their parent's basis */
    int64_t warm_fallbacks;      /* of which handed back to the cold path (dual-pivot budget spent) */
    int64_t warm_kept;           /* final states kept for children */
    int64_t pivots_dual;         /* dual-simplex pivots of the warm starts (not counted in pivots_phase1 / 2) */
} gomilp_frontier_stats;

gomilp_pool *gomilp_pool_create(int device, int workers, int *status);
void gomilp_pool_destroy(gomilp_pool *pool);
/* knobs: "split_phase" (default 1: in a wave of >= 16 the relaxations that start feasible — the long Phase-II chains — and those that need
 * Phase I run as two schedules side by side, the long chains on the higher-priority stream), "batch_loop" (default 1: whenever the active
 * relaxations of a schedule fit one launch their block steps run in the persistent kernel k_b_loop — per relaxation one pivot workgroup and
 * seven update workgroups, the rank-4 update of block t beside block t + 1), "batch_virt" (default 1: a WIDE wave of slack-start
 * relaxations runs its set-up pivot and its first block of 8 pivots on computed tableau entries and writes out only the tableaus that are
 * still alive behind that block — bit-identical), "batch_res" (default 0, opt-in: narrow waves in the
 * register-resident kernel k_b_res instead of k_b_loop — bit-identical, slower: DESIGN.md section 2.5d), "split_large" (default 1: a wave of >= 4 relaxations beyond 1024 rows runs as two interleaved schedules), "sample_batch" (1: time every batched block launch with HIP events), "batched" (default 1: the pivot loops of a wave run device-batched — grid.x = relaxation, O(1) host round trips per
 * superstep for the whole wave; 0: one host thread + stream per relaxation); any gomilp_ctx_set key is forwarded to the
 * worker contexts. */
int gomilp_pool_set(gomilp_pool *pool, const char *key, int64_t value);
/* Upload the root standard form (row-major A0, stride lda) to every worker context of the pool. */
int gomilp_pool_set_root(gomilp_pool *pool, const double *c0, const double *A0, int64_t lda, const double *b0, int64_t m0,
                         int64_t n0);
/* Solve `count` children.  Child i owns the triples [koff[i], koff[i+1]) of (var, sign, rhs) (koff has count+1 entries).
 * Outputs, all caller-owned: z_out[count]; x_out[count*n0] (root width, subproblem.go:157-159); status_out[count]
 * (gomilp_status); has_x_out[count].  Returns GOMILP_OK unless the call itself could not run. */
int gomilp_frontier_solve(gomilp_pool *pool, int64_t count, const int64_t *koff, const int32_t *var, const double *sign,
                          const double *rhs, double tol, double *z_out, double *x_out, int32_t *status_out,
                          int32_t *has_x_out, gomilp_frontier_stats *stats);

/* ------------------------------------------------------------------------------------------
 * Incumbent exchange of a frontier sharded over GPUs, one process per GPU: the only state the
 * reference's workers share is the incumbent (/root/reference/tree.go:207-263; pruning at :228-230).  One RCCL
 * all-reduce(min) of 2 * world doubles per wave over xGMI.  The caller ships the 128-byte id from rank 0 to the
 * other ranks by whatever transport it has (the Go host: its own RPC; bench.py: torch.distributed).
 * ---------------------------------------------------------------------------------------- */
typedef struct gomilp_comm gomilp_comm;
#define GOMILP_COMM_ID_BYTES 128
#define GOMILP_NO_INCUMBENT ((int64_t)1 << 52)   /* index meaning "this rank has no integer-feasible candidate" */
/* rank 0: fill id_out[GOMILP_COMM_ID_BYTES] (ncclGetUniqueId). */
int gomilp_comm_unique_id(char *id_out);
/* every rank, with rank 0's id; collective (ncclCommInitRank).  device < 0: current HIP device. */
gomilp_comm *gomilp_comm_create(int rank, int world, const char *id, int device, int *status);
void gomilp_comm_destroy(gomilp_comm *comm);
int gomilp_comm_rank(const gomilp_comm *comm);
int gomilp_comm_world(const gomilp_comm *comm);
/* global_z = min over ranks of local_z, global_index = the smallest child index among the ranks that attain it
 * (ties resolved like the reference's FIFO order would); a rank without a candidate passes (+Inf, GOMILP_NO_INCUMBENT).
 * Collective: every rank of the communicator calls it once per wave. */
int gomilp_incumbent_allreduce(gomilp_comm *comm, double local_z, int64_t local_index, double *global_z,
                               int64_t *global_index);
/* the host logic of the exchange (no GPU): lexicographic minimum of a table of `world` (z, index) pairs, +Inf = none */
void gomilp_incumbent_pick(const double *table, int world, double *global_z, int64_t *global_index);

/* The root relaxation (subproblem.go:172) on the pool's first worker. */
int gomilp_pool_solve_root(gomilp_pool *pool, double tol, double *opt_f, double *opt_x, int32_t *has_x, gomilp_lp_stats *stats);

/* Warm start from the parent's basis — OPT-IN (/root/reference/README.md TODO "initiate the simplex at solution of
 * parent?"; the reference's own hook would be initialBasic, simplex.go:147-161).  As gomilp_frontier_solve, plus per relaxation:
 *   tag[i]     caller-chosen id (>= 0; the B&B node id) under which the relaxation's final state stays resident when keep[i] != 0 and it
 *              ends GOMILP_OK inside the device-batched schedule (2-3 MB of HBM per 520-row relaxation; gomilp_pool_release_warm drops it,
 *              gomilp_pool_set_root drops all);
 *   parent[i]  tag of a kept relaxation that is THIS relaxation minus its last branch row, or < 0.  With a parent the relaxation starts
 *              from the parent's final basis + the slack of the new row: dual feasible, primal infeasible at most in that row — a dual-
 *              simplex loop on the device (k_bt_inner2_dual_batch) repairs it or proves the row infeasible; a new row the parent's
 *              optimum already satisfies costs no pivot.  After `dual_budget` dual pivots (0: default 64) the relaxation is handed to the
 *              cold path of this same call (the reference's Phase I), so the worst case is cold + budget.
 * Parity in this mode (it does not follow the reference's pivot path): status, branching / pruning decisions and |z - z_ref| <= 1e-9
 * max(1, |z_ref|); x is the gonum-order solve of the final basis, as always — the same vertex reached through another basis order can
 * differ in the last bits.  Without parents (parent == NULL or all < 0) the call is gomilp_frontier_solve + keeping. */
int gomilp_frontier_solve_warm(gomilp_pool *pool, int64_t count, const int64_t *koff, const int32_t *var, const double *sign,
                               const double *rhs, const int64_t *parent, const int64_t *tag, const int32_t *keep, int32_t dual_budget,
                               double tol, double *z_out, double *x_out, int32_t *status_out, int32_t *has_x_out,
                               gomilp_frontier_stats *stats);
int gomilp_pool_release_warm(gomilp_pool *pool, int64_t tag);   /* tag < 0: all */

/* Several roots in one pool: relaxation i of a wave is a child (K_i >= 0 rows) of root root_of[i]; index 0 is the root of
 * gomilp_pool_set_root, gomilp_pool_add_root returns 1, 2, ... (or -(gomilp_status)).  Independent LPs of similar shape
 * are children with K = 0 of different roots: they advance together through the device-batched schedule (one launch per
 * kernel type for all of them).  x_out has row stride ldx >= the widest root; gomilp_pool_set_root drops the added roots. */
int gomilp_pool_add_root(gomilp_pool *pool, const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n);
int gomilp_frontier_solve_roots(gomilp_pool *pool, int64_t count, const int32_t *root_of, const int64_t *koff, const int32_t *var,
                                const double *sign, const double *rhs, double tol, double *z_out, double *x_out, int64_t ldx,
                                int32_t *status_out, int32_t *has_x_out, gomilp_frontier_stats *stats);

/* Diagnostic, host only: the column search of findLinearlyIndependent (simplex.go:611-637) as the engine performs it for
 * non-slack starting bases (exact kappa_1 instead of the Hager estimate).  fast = 1: one Householder QR carried along,
 * O(m^2 n); fast = 0: a fresh factorisation per candidate, O(m^4).  idx_out has room for m entries; returns their count. */
int64_t gomilp_debug_find_independent(const double *A, int64_t lda, int64_t m, int64_t n, int64_t *idx_out, int fast);
/* the same search with the column scan on the device, on a resident problem (tests compare the two) */
int64_t gomilp_debug_find_independent_device(gomilp_ctx *ctx, int64_t problem, int64_t *idx_out, int64_t cap);
/* Diagnostic (host only): the condition-number estimate behind the engine's mat.Condition verdicts (mat/lu.go:321 with
 * lapack/gonum/dgecon.go:26-81, dlacn2.go:24-136) for a row-major n x n matrix: 1-norm (inf = 0) or infinity norm (inf = 1);
 * -1 when B is singular to working precision.  Tests compare it with the checker's Dgecon. */
double gomilp_debug_cond_estimate(const double *B, int64_t n, int inf);
/* Diagnostic (host only): what gonum's mat.LU holds after Factorize(M) (transposed = 1: Factorize(M.T())) for a row-major n x n matrix with
 * n <= 64 — *cond = 1 / Dgecon(MaxRowSum) on the factors of Dgetf2 (mat/lu.go:29-50,70-84), bit for bit; returns 1 when LU.Solve's
 * Det() == 0 test fires (mat/lu.go:301), 0 when not, -1 for n out of range.  The host replay of the reference's guards for bases of up
 * to 64 rows uses exactly this (engine_general.cpp general_condition_replay); tests compare it with the checker's restatement. */
int gomilp_debug_gonum_lu_cond(const double *M, int64_t n, int transposed, double *cond);

/* Library / device probes (no compute): used by the loader checks and by __graft_entry__. */
const char *gomilp_version(void);
int gomilp_device_count(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char *gomilp_compiled_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* GOMILP_LP_H */
