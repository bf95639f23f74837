// Shared host/device plain structs for the gfx950 simplex engine.
#pragma once
#include <stdint.h>
#include <stdlib.h>

// Diagnostic hooks exist only in the -DGOMILP_DEBUG flavour of the library (gomilp_amd/build.py, GOMILP_DEBUG_BUILD=1): the product
// build reads no environment variable on its paths and carries no fault-injection branch in its kernels.
#ifdef GOMILP_DEBUG
#define GOMILP_DBG_ENV(name) getenv(name)
#else
#define GOMILP_DBG_ENV(name) (static_cast<const char *>(nullptr))
#endif

namespace gomilp {

// loop status written by the kernels (DevState::status)
enum : int32_t {
    ST_RUNNING = 0,
    ST_OPTIMAL = 1,     // min r >= -tol                      (simplex.go:248)
    ST_UNBOUNDED = 2,   // no d_i < 0                         (simplex.go:328-330)
    ST_NEED_BLAND = 3,  // move[replace] <= 0: degenerate     (simplex.go:269)
    ST_MAX_PIVOTS = 4,  // safety cap hit
    ST_LU_SINGULAR = 5,  // exact zero pivot in the final gonum-order LU
    ST_BLAND_FAILED = 6,  // replaceBland exhausted its candidates (lp.ErrBland, simplex.go:382)
    ST_FORCED_DONE = 7,   // a set-up pivot ordered with forced_nocommit = 3 has run: later launches of the superstep are no-ops
    ST_XCHG_TIMEOUT = 9,    // multi-workgroup block kernel (btg_kernels.hip): an exchange saw no progress (a workgroup never ran)
    ST_DUAL_INFEASIBLE = 11,   // dual simplex (warm start): a row with x_B < 0 has no entry that can restore it — the relaxation is infeasible
    ST_NEED_EXACT = 10      // the winning ratio is within BTArgs::guard of zero: the host refreshes x_B with a gonum-order solve of the
                            // current basis before the decision is taken (simplex.go:268-277 sees a fresh x_B every pivot)
};

constexpr int kMaxPartials = 1024;  // per-workgroup partial arg-reductions (grid <= 1024 workgroups)
constexpr int kBlock = 256;         // threads per workgroup (4 waves of 64)
constexpr int kWavesPerBlock = 4;

// One per solve, lives in device memory; a pinned host mirror is copied after every chunk.
struct DevState {
    int32_t done;      // 0 while pivoting; kernels become no-ops once set
    int32_t status;    // ST_*
    int64_t pivots;    // pivots performed since the phase started
    int32_t q;         // entering position (index into nonbasic[])
    int32_t p;         // leaving position  (index into basic[])
    double rq;         // reduced cost of the entering column
    double dp;         // pivot element (B^-1 a_q)[p]
    double mv;         // winning ratio x_B[p] / |d_p|
    int64_t trace_len; // pivots appended to the device trace since the solve started
    int64_t max_pivots;
    int32_t lu_singular;
    int32_t pad;
    // fused two-kernel pipeline: pivot t is committed by the pricing kernel of pivot t+1
    double theta;      // x_B[p] / d_p of the pivot being committed
    int32_t ent_cur;   // variable entering in the pivot whose FTRAN ran last
    int32_t ent_prev;  // variable entering in the pivot being committed
    int32_t lea;       // variable leaving in the pivot being committed
    int32_t pad2;
    // single-kernel tableau pipeline: kernel t reads slot (t & 1) and writes slot ((t + 1) & 1), so a workgroup
    // that starts late never sees values produced by its own launch
    int32_t nq[2];     // entering position chosen for the next pivot
    int32_t nent[2];   // its variable id
    double nrq[2];     // its reduced cost
    int64_t stop_at;   // launches with index >= stop_at are no-ops (set by launch t to t + 1)
    // blocked tableau pipeline: pivots of the last inner-kernel block whose rank-1 terms the update kernel must apply
    int32_t kdone;
    int32_t bland_steps;  // degenerate steps resolved inside the inner kernel (cumulative over the loop)
    // persistent loop kernel (BTArgs::loop): pivots of the block of each parity (the update of block t runs beside block
    // t+1), which of BTArgs::Tbuf holds the current tableau between launches, blocks run by the last launch
    int32_t kdone2[2];
    int32_t tsel2[2], loop_blocks, dead1;  // launch i reads tsel2[i & 1] and writes tsel2[(i & 1) ^ 1]: a workgroup that starts late still sees its input
                                           // dead1 (virtual tableau): Phase I ended inside the first block and nobody will read this relaxation's tableau (it is never written)
};

struct DevPivot {  // mirrors gomilp_pivot
    int32_t phase, bland;
    int64_t min_idx, replace, entering, leaving;
};

// Kernel argument block (passed by value).
struct LPArgs {
    int32_t m;         // rows
    int32_t ld;        // padded row length (doubles) of At rows and Binv rows; multiple of 2, zero padded
    int32_t nn;        // number of nonbasic positions in this phase
    int32_t phase;     // 1 / 2 (trace only)
    double tol;
    const double *At;      // (ncols) x ld, row j = column j of A   (row n = artificial column in Phase I)
    const double *cost;    // cost vector of the current phase
    const double *b;       // right-hand side, padded to ld
    const double *binv_cur;  // m x ld
    double *binv_next;       // m x ld (ping-pong target of the rank-1 update)
    double *xb;            // m
    double *y;             // ld (padded with zeros)
    double *dvec;          // m   d' = B^-1 a_q (unrounded)
    double *move;          // m   ratio vector (simplex.go:334-340)
    double *rvec;          // nn  reduced costs (simplex.go:243)
    int32_t *basic;        // m   variable id per basis position
    int32_t *nonbasic;     // nn  variable id per nonbasic position
    unsigned long long *pk_price;  // kMaxPartials
    unsigned int *pi_price;
    unsigned long long *pk_ratio;
    unsigned int *pi_ratio;
    unsigned int *pv_price;        // fused pipeline payloads: variable id of the best column of each workgroup
    double *pd_ratio;              //   d'_i of the best row of each workgroup
    unsigned int *pb_ratio;        //   basic[i] of the best row of each workgroup
    DevState *st;
    DevPivot *trace;
    int64_t trace_cap;
};

// Arguments of the single-kernel tableau pivot (tableau_kernels.hip).
struct TabArgs {
    int32_t m, nn;     // rows, nonbasic positions (tableau columns)
    int32_t ldt;       // padded row length of T in doubles (multiple of 128), zero padded
    int32_t phase;
    double tol;
    const double *T_cur;   // m x ldt : B^-1 A_N, column j = nonbasic position j
    double *T_next;
    const double *r_in;    // ldt : reduced costs by position (padding = 0, ignored)
    double *r_out;
    double *xb, *dvec, *move;
    int32_t *basic, *nonbasic;
    unsigned long long *pk_ratio;
    unsigned int *pi_ratio, *pb_ratio;
    double *pd_ratio, *px_ratio;
    DevState *st;
    DevPivot *trace;
    int64_t trace_cap;
};

// Arguments of the blocked tableau kernels (bt_kernels.hip).
struct BTArgs {
    int32_t m, nn;      // rows, nonbasic positions
    int32_t ldt, ldu;   // padded row length of T / V (multiple of 512 doubles) and of U (multiple of 2)
    int32_t phase, kmax;
    double tol;
    double *T;          // m x ldt, STALE by the rank-1 terms of the running block; updated in place by k_bt_update
    double *U;          // kmax x ldu : u_k
    double *V;          // kmax x ldt : v_k'
    double *r, *xb;
    int32_t *basic, *nonbasic;
    DevState *st;
    DevPivot *trace;
    int64_t trace_cap;
    int32_t forced_q, forced_p, forced_nocommit;  // first pivot of the block chosen by the host (set-up pivots)
    int32_t nt_force;     // context knob "bt_nt": 0 = pick the thread count by shape, else 256 / 512 / 1024
    int32_t tiled, old_only;  // T is in the 4x4-tile layout of the register-resident inner kernel; knob "bt_old"
    unsigned long long *stamps;   // diagnostic build only (knob "bt_stamps"): per-wave cycle sums per pivot segment
    double *xbuf;                 // multi-workgroup block kernel: exchange records in HBM (btg_kernels.hip)
    int32_t groups, group_ri;     // its workgroup count (0: single-workgroup kernels) and rows / columns per thread
    int32_t fault, upd_cap;         // knob "bt_fault" (tests): workgroup 1 of the multi-workgroup block kernel leaves at once
    int32_t group_nt, upd_valu;   // threads per workgroup; knob "bt_upd_valu": rank-16 update on the VALU instead of the matrix cores
    // Persistent loop kernel (btg_kernels.hip k_bt_loop; engine_tableau.cpp run_loop_bt): ONE launch runs up to `nblocks` blocks of
    // 8 pivots on the G workgroups of one XCD while the other workgroups of the same launch apply the rank-8 update of block
    // t beside block t+1 (ping-pong between Tbuf[0] and Tbuf[1]; DevState::tsel says which one holds the tableau when a
    // launch starts / ends).  Block t+1 therefore reads a tableau WITHOUT the terms of block t and corrects columns / rows
    // with 8 lagging + up to 8 current terms.  U / V rows [8 (t & 1), 8 (t & 1) + 8) take the terms of block t.
    double *Tbuf[2];
    int32_t loop, nblocks;
    int32_t par, xcd;     // launch parity; XCD of the pivot workgroups (blocks xcd, xcd + 8, ...: concurrent loop kernels take different ones)  // (DevState::tsel2, and the counter bases in the exchange buffer's header)
    // Degenerate vertices: the reference recomputes x_B from a fresh LU every pivot (simplex.go:289), so basic variables at level
    // zero carry that solve's rounding noise, and `move[replace] <= 0` (:269) as well as the argmin among several zero-level rows
    // are decided by it.  guard > 0: a block stops (ST_NEED_EXACT) in front of a pivot whose winning ratio is <= guard; the host
    // uploads the gonum-order x_B of the current basis and restarts with exact_once = 1 (the first pivot then decides as is).
    double guard;
    // Pivot-element floor at ANY size (knob cond_guard): gonum leaves its loop with mat.Condition when the condition estimate of a solve
    // exceeds 1e16 (mat/lu.go:321) — on the tableau a basis on its way there shows as a Dantzig pivot element of rounding-noise size.
    // cguard > 0: a block stops (ST_NEED_EXACT) in front of a pivot with |d_p| <= cguard; the host measures the exact condition numbers
    // and repeats the reference's iteration on fresh solves (Engine::exact_step); a batched relaxation goes to the worker path
    double cguard;
    int32_t exact_once, poll_delay;   // (poll_delay: units of 64 cycles a wave waits between its post and its first poll, knob "poll_delay")
};

// shape of the multi-workgroup block kernel for a tableau (btg_kernels.hip): groups == 0 -> single-workgroup kernels
struct BtGroupCfg { int groups, nt, ri; };

// ---- device-batched relaxations (batch_kernels.hip, engine_batch.cpp) ------------------------------------------------
// One BatchLP per relaxation of a wave, resident in HBM.  `bt` is the argument block of the block kernels for the
// relaxation's current phase; the control kernel (k_b_ctrl) rewrites it at the phase changes, so the host enqueues a fixed
// schedule of launches with grid.x / grid.z = relaxation and never waits for a single relaxation.
enum : int32_t {
    BS_FORCED = 0,    // Phase I needed: the set-up pivot (artificial enters at argmin x_B) is queued as a forced pivot
    BS_P2_START = 1,  // slack basis feasible: Phase II starts after the reduced costs are built
    BS_P1 = 2,        // Phase-I loop running
    BS_P2 = 3,        // Phase-II loop running
    BS_EXCH = 6,      // the zero-level artificial is being exchanged out of the basis (one forced pivot, simplex.go:581-606)
    BS_DONE = 4,      // terminal: `status` holds the outcome (GOMILP_OK = basis + x_B ready for the final gonum-order solve)
    BS_HOST = 5,      // terminal: a path the device schedule does not cover (artificial exchange, guard band, ...): the
                      // host solves this relaxation through the single-relaxation engine
    // warm start (opt-in): the relaxation starts from its parent's final tableau + the slack of its one new branch row
    BS_DUAL_START = 7,  // reduced costs of the parent's basis are being rebuilt; then BS_DUAL (new row violated) or BS_P2 (still optimal)
    BS_DUAL = 8,        // dual-simplex loop running (k_bt_inner2_dual_batch)
    BS_COLD = 9         // terminal: the dual simplex spent its pivot budget — the caller solves this relaxation cold
};

struct BatchLP {
    BTArgs bt;
    // root data (shared by the children of a frontier) and this child's branch-and-bound rows (subproblem.go:36-44)
    const double *At0;       // (n0 + 1) x ld0 : row j = column j of the root A
    const double *c0, *b0;   // root cost (n0), right-hand side (m0)
    const int32_t *rho0;     // m0 : row of the 1 in root column n0 - 1 - pos (the unit columns the descending scan meets)
    const int32_t *var;      // K
    const double *sign, *rhs;
    int32_t ld0, m0, n0, K;
    int32_t m, n;            // child: m0 + K, n0 + K
    int32_t ldu, cap_ldt;    // padded row count (U rows), row length the T buffers were sized for
    double *T[2];            // tableau buffers (4x4 tiles); bt.T is the current one
    double *R, *xb, *U, *V, *scratch, *art;
    int32_t *basic, *nonbasic, *srcpos;
    DevState *st;
    // root WITHOUT a slack basis (equality rows): the root's initial basis from the column search (simplex.go:611-637) and its
    // tableau, computed once per root (Engine::root_general); a child's basis is that one + its K branch slacks
    const double *gT0;       // m0 x gldt row-major: B0^-1 A_N0, column jp = root nonbasic position jp (ascending variable id)
    const double *gxb0;      // m0: x_B of the root's initial basis (gonum-order solve)
    const int32_t *gbasic0, *gnonbasic0;   // its positional lists
    const int32_t *gposvar0; // n0: variable id -> basis position (>= 0) or -1 - nonbasic position
    int32_t gldt, gen;       // gen = 1: this relaxation starts from that basis
    // warm start: the parent's final state (WarmStore entry, engine_batch.cpp): tableau in 4x4 tiles (wm rows, the child's own ldt),
    // updated x_B, positional lists, variable -> position map; the child's one new branch row is the LAST of var / sign / rhs
    const double *wT, *wxb;
    const int32_t *wbasic, *wnonbasic, *wposvar;
    int32_t warm, wm;        // warm = 1: start from that state (m = wm + 1)
    int32_t dual_budget, art_pos;   // dual pivots before the relaxation is handed back (BS_COLD); basis position the Phase-I artificial enters at (k_b_setup)
    int64_t pivd;            // dual pivots performed
    // Virtual tableau (wide waves of slack-start relaxations, engine_batch.cpp): virt = 2 while the host-chosen Phase-I pivot runs, 1 while the
    // first block of 8 pivots runs, 0 from then on (k_b_ctrl counts it down).  While virt > 0 NO tableau exists in HBM: the block kernel
    // (k_bt_inner2_virt_batch) and the reduced-cost kernel compute the entries they read — T0 = rows / columns of the ROOT's A (b_entry) plus
    // the set-up pivot's rank-1 term (virt_t0: U / V row 8) — and only what survives the first block is written out, once, with the
    // block's terms applied in the update kernel's arithmetic (k_b_gather mode 3).  On a B&B frontier most relaxations are proved
    // infeasible inside that block: their 2.4 MB tableaus (4.9 of the 7.2 GB a 2048-wide wave moved) are never written.
    const double *A0r;       // the root's A row-major (m0 x lda0r): a row of T0 as one contiguous read (At0 serves the columns)
    int32_t lda0r, virt, virt_t0, pad2;
    double tol_user;         // Phase-II tolerance of the call (GoMILP: 0)
    int32_t stage;           // BS_*
    int32_t tcur;            // index of the current T buffer
    int32_t do_permute, do_r, r_phase;   // work orders for the set-up kernels of this superstep (written by k_b_ctrl)
    int32_t perm_ld_in, perm_nn_out;     // k_b_permute: row length of the source, columns of the target
    int32_t kblock;          // pivots per block (8)
    // outcome
    int32_t status;          // gomilp_status
    int32_t wrapped;         // inner status of GOMILP_ERR_PHASE1_WRAPPED
    int32_t phase1_used, pad0;
    int64_t piv1, piv2, bland;
};

// What the host needs to see of a relaxation after every control step (one small D2H copy per superstep for the whole wave)
struct BatchOut {
    int32_t stage, status, wrapped, phase1_used;
    int64_t piv1, piv2, bland, pivd;
    int32_t tcur, pad;   // which tableau buffer holds the final state (warm store)
};

// State of the device column search (general_kernels.hip)
struct GsState {
    int32_t k;         // columns accepted so far
    int32_t accept;    // decision about the candidate in flight
    int32_t kacc;      // its position when accepted
    int32_t done;      // m - 1 columns accepted: the host takes the last (square) step
    int32_t stop_col;  // first column the host still has to examine
    int32_t scanned;   // candidates examined
    double nR, nRinv;  // |R|_1, |R^-1|_1
    double vv;         // v^T v of the reflector in flight
    double beta_last;  // diagonal entry of R produced by the square step (0: the candidate is dependent)
};

// Record of one block of the blocked column search (general_block.hip): written by the decide kernel, read by the apply kernel
struct GsBlock {
    int32_t nacc;      // reflectors the block accepted
    int32_t k0;        // columns accepted before the block (reflector i sits at position k0 + i)
    int32_t pad0, pad1;
    double vv[16];     // v^T v per reflector (0: H = I)
};

// Control block of the compressed LU schedule (lu_compressed.hip): written by the panel kernel of a round, read by the
// U-solve / trailing kernels of the same round and by the host between batches of rounds.
struct LUCtl {
    int32_t k_next;     // first elimination step not performed yet
    int32_t k0, k1;     // steps [k0, k1) were performed by the last round
    int32_t nsteps;     // how many of them did arithmetic ("dense" steps)
    int32_t ndrop;      // register columns the panel dropped un-eliminated (a newly dense column took their slot)
    int32_t rounds;     // rounds that did work
    int32_t nnext;      // look-ahead schedule: the columns the NEXT round's panel loads (the first dense columns >= k1, as the panel's own scan finds them)
    int32_t ksync;      // cross-workgroup panel (lu_cross.hip): its index maps hold the row interchanges of the steps < ksync (the rest waits in pivrow: replayed when a pivot search ties)
    int32_t steps[32];  // step (= column) index of each dense step, ascending
    int32_t prow[32];   // its pivot row
    // dropped column d was in the register list for the rows that left the active set at steps [dropin, dropout):
    // those rows hold final values in it (the panel wrote them), every other row still holds the original
    int32_t dropcol[32], dropin[32], dropout[32];
    int32_t next[32];
    // look-ahead schedule, control block 0 only: arrival counters of the workgroups that run beside the panel (monotonic over the rounds of
    // a factorization: the update of the next panel's columns / the U-solve tiles), and the flag a wait that ran out of patience raises
    uint32_t cnt_x, cnt_u, cnt_s;   // (cnt_x: the U-solve of the next panel's columns, one workgroup per round)
    int32_t fault;
};

// Arguments of the gonum-order LU kernels (final basis solve).
struct LUArgs {
    double *W;          // m x ldw working copy of ab (simplex.go:144), overwritten by L\U in place (rows never move)
    int32_t ldw, m;
    int32_t *lpos;      // logical (LAPACK) position of each physical row
    int32_t *rowstep;   // -1 while active, else the step at which the row became the pivot row
    unsigned long long *pk[2];  // partial argmax keys, double buffered by step parity
    unsigned int *pl[2];        // logical position of the candidate
    unsigned int *pr[2];        // physical row of the candidate
    DevState *st;
    const int32_t *unit_row;    // per column: row of the 1 when the column of ab is a unit vector, else -1 (nullable)
    int32_t *dense_flag;        // per step: 1 when the step did arithmetic (0 = unit-column fast path); nullable
    LUCtl *ctl;                 // compressed schedule only
    double *Lp, *Up;            // compressed schedule only: compact panels of the running round (32 x ldw each)
    int32_t slots, look;        // compressed schedule: the slot form of the panel (lu_compressed.hip k_luc_panel_slots); look: the look-ahead schedule (knob lu_blocked = 3, default; 2: the trailing update of a round in one launch behind it)
    // look-ahead schedule: control block / panels / row snapshot of the round before (by round parity), read by the update workgroups that
    // run beside this round's panel; ctl_prev == ctl, ... without look-ahead
    const LUCtl *ctl_prev;
    const double *Lp_prev, *Up_prev;
    int32_t *rowsnap;           // rowstep as this round leaves it (the panel of the next round changes rowstep while this round's update still reads it)
    const int32_t *rowsnap_prev;
    LUCtl *ctl_base;            // control block 0 (counters, fault flag)
    int32_t round, pad3;        // rounds launched for this factorization before this one; pad3: diagnostic flavour, fault injection (the U-solve workgroup never arrives)
};

}  // namespace gomilp
