// Host-side engine of the gfx950 LP-relaxation path: owns the HBM-resident problem and work buffers,
// enqueues the pivot kernels in chunks on one HIP stream, and reproduces the control flow of
// gonum's simplex() (vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:93-302) around them.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gomilp_lp.h"
#include "device_types.h"

namespace gomilp {

// A standard-form LP resident in HBM.
struct Problem {
    int m = 0, n = 0, ld = 0;
    double *dAt = nullptr;    // (n+1) x ld : row j = column j of A; row n is reserved for the Phase-I artificial column
    double *dc = nullptr;     // n+1 : c, then 0
    double *dc1 = nullptr;    // n+1 : Phase-I cost e_n
    double *db = nullptr;     // ld  : b, zero padded
    std::vector<double> hb, hc;
    mutable std::vector<double> hA;   // host copy of A (row-major m x n), kept for small problems: general initial basis (S7);
                                      // a child builds it from its root only when a solve needs it (ensure_host_A)
    std::vector<int32_t> nnz, lastrow, allone;  // per column of A
    int verify_status = GOMILP_OK;              // verifyInputs, simplex.go:385-439
    double seconds_upload = 0;
    uint64_t serial = 0;      // unique per upload (device buffers are recycled: a pointer does not identify a problem)
    // children are recycled: their device buffers go back to a per-engine pool instead of hipFree (which synchronises
    // the whole device and would serialise the worker streams of a frontier pool)
    bool lazy_host = false;   // no host copy of A was kept at upload: ensure_host_A downloads it (flat call only)
    double scale_span = 1.0;  // max |a_ij| / min nonzero |a_ij| of A: beyond 1e9 the solve keeps the degenerate-pivot guard on at every size
    bool is_child = false;
    size_t cap_at = 0, cap_c = 0, cap_b = 0;    // capacities in doubles
    int32_t *dvar = nullptr;                    // child: branched variables / signs on the device
    double *dsign = nullptr;
    int cap_k = 0;
    int64_t root = -1;                          // child: the problem it was assembled from, its branching rows
    const Problem *root_ptr = nullptr;          //   (the same as a pointer: also set when the root lives in another engine)
    std::vector<int32_t> kvar;
    std::vector<double> ksign;
};

class Engine {
   public:
    explicit Engine(int device);
    ~Engine();
    int device() const { return device_; }
    static int loop_acquire(int dev, int weight);          // slots for persistent loop kernels on a device (engine.cpp): returns the slot
    static void loop_release(int dev, int weight, int slot);
    static bool loop_try_acquire_all(int dev);             // the whole device, if nobody holds a slot right now (no waiting); release: loop_release(dev, 4, 0)
    int set(const std::string &key, int64_t v);

    // lazy_host: do not keep a host copy of a large A at upload (the flat call: the context is the caller's alone, and the copy is
    // only read on rare paths — ensure_host_A fetches it from the device then)
    int64_t upload(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n, bool lazy_host = false);
    int free_problem(int64_t id);
    // child of a resident root: K branch-and-bound rows (var, sign, rhs) appended on the device (subproblem.go:141-159)
    int64_t upload_child(int64_t root, int K, const int32_t *var, const double *sign, const double *rhs);
    // the same with the root resident in ANOTHER engine of the same device (read in place: device buffers are immutable
    // after upload); the owner must not free the root before the child is freed
    int64_t upload_child_of(Engine &owner, int64_t root, int K, const int32_t *var, const double *sign, const double *rhs);
    int solve(int64_t id, double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
              int64_t *basis_out, gomilp_lp_stats *stats);
   private:
    int solve_locked(int64_t id, double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                     int64_t *basis_out, gomilp_lp_stats *stats);
   public:
    int64_t last_trace(gomilp_pivot *out, int64_t cap);

    // ---- hooks of the device-batched frontier (engine_batch.cpp) ----
    // device pointers / statistics of a resident problem: the batch engine reads the root's columns where they lie
    struct RootView {
        int m = 0, n = 0, ld = 0;
        const double *dAt = nullptr, *dc = nullptr, *db = nullptr;
        double scale_span = 1.0;          // Problem::scale_span of the root
        bool unit_basis = false;          // the descending scan of simplex.go:618-635 meets m distinct unit columns
        std::vector<int32_t> rho0;        // their rows, by basis position
        int verify_status = GOMILP_OK;
        uint64_t serial = 0;              // identifies the upload (device buffers are recycled)
        std::vector<double> hb, hc;       // host copies of b and c (final solves on another engine's behalf)
        // no slack basis (equality rows): the result of the column search (simplex.go:611-637) and the tableau of that basis,
        // computed ONCE per root (root_general) — a child's starting basis is this one + its K branch slacks, so the batched
        // schedule takes such children too instead of repeating the search per node
        struct General {
            int m = 0, nn = 0, ldt = 0;
            double *dT0 = nullptr, *dxb0 = nullptr;                              // m x ldt row-major B0^-1 A_N0; x_B of the basis
            int32_t *dbasic0 = nullptr, *dnonbasic0 = nullptr, *dposvar0 = nullptr;
            ~General();
        };
        std::shared_ptr<General> gen;     // null: slack basis, or no usable basis (the single-relaxation engine reports why)
        // the root's A row-major on the device (one transposing copy per root): rows of the virtual tableau of a wide wave (BatchLP::A0r)
        struct RowMajor { double *dA = nullptr; int lda = 0; ~RowMajor(); };
        std::shared_ptr<RowMajor> rm;     // null: not made (large roots)
    };
    bool root_view(int64_t id, RootView *out);
    // fills out->gen for a root without a slack basis (false: not possible — too large for the host copy of A, singular, ...)
    bool root_general(int64_t id, RootView *out);
    // epilogue of simplex() (simplex.go:296-301) for a relaxation whose pivot loop ran elsewhere: final basis positions
    // `basic` (m entries) and updated x_B in, gonum-order solve of that basis, z, x out; `loop_rc` as Engine::solve
    int finish_from_basis(int64_t id, const int32_t *basic, const double *xb_updated, int loop_rc, double *opt_f, double *opt_x,
                          int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *stats);
    // diagnostic: the device column search on a resident problem (tests compare it with the host forms)
    int debug_find_independent(int64_t id, std::vector<int32_t> &idxs);

   private:
    struct Work;  // device work buffers, sized for the largest problem seen
    int ensure_work(int m, int ncols);
    const Problem *problem_ptr(int64_t id);
    int64_t upload_child_impl(const Problem &R, int64_t root_id, int K, const int32_t *var, const double *sign, const double *rhs);
    LPArgs make_args(const Problem &P, int phase, double tol, int nn, const double *cost);
    int run_loop(const Problem &P, int phase, double tol, int nn, const double *cost, gomilp_lp_stats *st);
    int run_loop_fused(const Problem &P, int phase, double tol, int nn, const double *cost, gomilp_lp_stats *st);
    int run_loop_tab(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st);
    int host_bland_tab(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st, int *par_out);
    TabArgs make_tab_args(const Problem &P, int phase, double tol, int nn);
    int tab_forced_pivot(const Problem &P, int phase, double tol, int nn, int q, int ent, double rq, int p, double dp, double xp,
                         int lea, int flags, long long t);
    int solve_tableau(const Problem &P, double tol, std::vector<int32_t> &basic, const std::vector<int32_t> &rho,
                      std::vector<double> &xb, bool feasible, gomilp_lp_stats *st, int *loop_rc,
                      const std::vector<double> *binv_host);
    void bt_layout(const Problem &P, bool tiled);
    BTArgs make_bt_args(const Problem &P, int phase, double tol, int nn, int kmax);
    // block size and tableau layout of the blocked pipeline for the current ldt_ (knobs block_k / bt_nt / bt_old / bt_groups)
    void bt_plan(const Problem &P, int *K, bool *tiled, bool *lag = nullptr) const;
    int bt_forced_pivot(const Problem &P, int phase, double tol, int nn, int q, int p, int nocommit);
    int run_loop_bt(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st);
    void account_samples(gomilp_lp_stats *st, const std::vector<int64_t> &sample_t, int64_t executed, int nk);
    int host_bland(const Problem &P, LPArgs &a, gomilp_lp_stats *st);
    int refresh_xb_y(const Problem &P, const double *cost);
    int epilogue(const Problem &P, std::vector<int32_t> &basic, std::vector<double> &xb, int loop_rc, double *opt_f, double *opt_x,
                 int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *st);
    int final_solve(const Problem &P, int ncols_rows, std::vector<double> &xb_exact, bool *singular, const int32_t *basic_host = nullptr,
                    bool transpose = false, const double *rhs_host = nullptr);
    // its two halves (engine.cpp): one factorization, any number of right-hand sides
    int lu_factor(const Problem &P, bool *singular, const int32_t *basic_host = nullptr, bool transpose = false);
    int lu_solve(const Problem &P, std::vector<double> &x, const double *rhs_host = nullptr);
    struct LuCache {
        bool valid = false, split = false, singular = false;
        int m = 0, nd = 0;
        std::vector<int32_t> phys, dl;   // physical row at a logical position; the positions whose step did arithmetic
        std::vector<double> diag;        // u_ii by physical row
        LUArgs args;
    } lu_cache_;
    // one iteration of the reference on fresh solves (engine_tableau.cpp): the decision a degenerate or tied pivot needs
    int exact_step(const Problem &P, int phase, double tol, int nn, int *q_out, int *p_out, gomilp_lp_stats *st);
    int cond_check(const Problem &P, int nn, double *k1, double *kinf);
    int cond_fresh(const Problem &P, const int32_t *basic_host, double *k1, double *kinf);   // from a fresh host inverse of the basis
    int groups_knob(const Problem &P) const;
    bool ensure_host_A(const Problem &P);
    // findLinearlyIndependent with the scan on the device (general_kernels.hip) and the last, square step on the host
    int find_independent_device(const Problem &P, std::vector<int32_t> &basic, std::vector<double> *binv_out, bool *binv_on_device = nullptr);
    int stage_upload(void *dst, const void *src, size_t bytes);
    hipError_t sync_stream();
    int upload_index_lists(const std::vector<int32_t> &basic, const std::vector<int32_t> &nonbasic);
    void sync_state_to_device();

    int device_;
    hipStream_t stream_ = nullptr;
    int ncu_ = 256;   // compute units of the device (grid of the persistent loop kernel)
    std::mutex mu_;
    std::vector<std::unique_ptr<Problem>> problems_;
    std::vector<std::unique_ptr<Problem>> child_pool_;  // released children, buffers kept
    // released root problems and the staging buffers of upload(), kept for the next upload of a fitting shape: the flat
    // lp.Simplex drop-in uploads and frees once per call, and hipFree synchronises the whole device (it would serialise
    // the concurrent callers of the drop-in)
    std::vector<std::unique_ptr<Problem>> root_pool_;
    double *up_dA_ = nullptr; size_t up_dA_cap_ = 0;
    int32_t *up_stats_ = nullptr; size_t up_stats_cap_ = 0;
    std::unique_ptr<Work> w_;
    // knobs
    int64_t lu_look_faults_ = 0;   // final solves repeated with the plain schedule after a look-ahead launch gave up a wait
    bool lu_look_fault_ = false;   // ... in the running solve (stats.device_retries)
    int64_t lu_look_ = 1;          // knob lu_look: 0 = never the look-ahead schedule (the workers of a pool)
    int64_t chunk_ = 32, refresh_ = 0, trace_on_ = 0, max_pivots_ = 0, sample_events_ = 0, fused_ = 1, lu_blocked_ = 3, tableau_ = 1, blocked_ = 1, block_k_ = 0,  // block_k_ 0 = auto
            bt_nt_ = 0, bt_old_ = 0, bt_stamps_ = 0, bt_upd_valu_ = 0, bt_fault_ = 0, general_device_ = 1, general_block_ = 1, general_min_rows_ = 96, lu_cross_ = 0, bt_groups_ = 0,   // bt_groups_: -1 never, 0 by shape, 2 / 4 / 8 forced
            bt_lag_ = 1,       // persistent loop kernel where the multi-workgroup block kernel runs (0: block kernel + update launches)
            loop_chunk_ = 512, // pivots per launch of the persistent loop kernel
            loop_grid_ = 0,    // its workgroups (0: one per CU)
            poll_delay_ = 0,   // loop kernel: 64-cycle units between a wave's post and its first poll of an exchange
            loop_upd_ = 0,     // update workgroups of the loop kernel that take part (0: all of the grid's)
            loop_rep_ = 0,     // 1: pivot role with replicated reduced costs (btr_kernels.hip k_bt_loopR: ONE exchange per pivot; bit-identical pivots, measured slower — DESIGN 2.1d) where the shape fits and the engine gets the whole device
            loop_g_ = 0,       // its pivot workgroups (0 / 16: 16 x 128 threads up to 2048 rows, 16 x 256 beyond; 8: 8 x 256 / 8 x 512)
            loop_k_ = 0,       // its pivots per block (0: 8 up to 2048 rows, 16 beyond; 8 / 16 forced where instantiated)
            exact_degenerate_ = 1,   // 0 never, 1 bases of up to 256 rows and every non-slack start, 2 always: pivots whose winning ratio is (nearly) zero are decided on a fresh gonum-order x_B; 3 strict: EVERY pivot and the stop test are decided on fresh gonum-order solves
            cond_guard_ = 1;   // replay the condition guards of the reference on the host for bases of up to 64 rows
    bool badly_scaled_ = false;   // the current problem's entries span more than nine decades (Problem::scale_span): guard on, tableau checked
    bool gen_binv_dev_ = false;   // the searched basis' B^-1 = R^-1 Q^T is resident in the first B^-1 buffer (the device judged the square step): no upload
    bool gen_start_ = false;      // the current solve starts from a searched (non-slack) basis: the degenerate-pivot guard stays on
    bool xchg_timeout_ = false;   // the last pivot loop ended in ST_XCHG_TIMEOUT (Engine::solve repeats the solve once)
    bool shadow_trace_ = false;   // this solve records its pivots for the replay even when the caller did not ask for a trace   // developer knobs of the block kernels (per context: tests force the 1024-thread instance)
    // per-solve state
    int cur_ = 0;   // which Binv buffer is current
    int ycur_ = 0;  // which y buffer is current
    int grid_ratio_ = 1;
    int tcur_ = 0, rcur_ = 0, ldt_ = 0;
    bool t_tiled_ = false;   // layout of T[tcur_]: 4x4 tiles (blocked pipeline, register-resident kernel) or row-major
    bool use_bt_ = false;  // blocked tableau (deferred rank-K updates) instead of one launch per pivot  // tableau pipeline: current T / r buffer, row length of T  // workgroups of the last ratio-test kernel (partials to reduce)
    int64_t launches_ = 0;
    double fs_device_ = 0, fs_host_ = 0;
    int64_t lu_dense_ = 0, lu_rounds_ = 0;
    std::vector<gomilp_pivot> last_trace_;
    int64_t last_trace_total_ = 0;
};

// engine_general.cpp
int general_find_linearly_independent(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs, std::vector<double> *binv_out = nullptr);
int general_finish_last_column(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs, int start_col, std::vector<double> *binv_out);
int general_find_linearly_independent_slow(const std::vector<double> &A, int m, int n, std::vector<int32_t> &idxs);
bool general_basis_inverse(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, int ncols_with_art,
                           const std::vector<double> &art, std::vector<double> &binv);

int general_condition_replay(const std::vector<double> &A, int m, int n, std::vector<int32_t> &basic, const std::vector<std::pair<int, int>> &pivots,
                             bool ended_in_compute_move, int *status_out, int64_t *evaluations);
double general_cond_inf(const std::vector<double> &A, int n);
bool general_invert(const std::vector<double> &B, int m, std::vector<double> &inv);
double inverse_norm1_estimate(const std::vector<double> &M, int n, bool transposed);
// gonum_cond.cpp: what gonum's mat.LU reports for a row-major n x n matrix (transposed: for its transpose) — cond = 1 / Dgecon(MaxRowSum) of the
// factors, and the Det() == 0 test of LU.Solve — bit for bit, for n <= kGonumCondMax (the range Dgetrf does not block); false beyond it
constexpr int kGonumCondMax = 64;
bool gonum_lu_cond(const double *M, int n, int ldm, bool transposed, double *cond, bool *det_zero);
bool gonum_lu_solve(const double *M, int n, int ldm, const double *b, double *x);   // LU.SolveVec's result (the point returned with a mat.Condition error)
double general_basis_cond1(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, const std::vector<double> &art);
bool general_solve_basis(const std::vector<double> &A, int m, int n, const std::vector<int32_t> &basic, const std::vector<double> &b, std::vector<double> &x);

int device_count();
const char *compiled_arch();

// launch wrappers implemented in simplex_kernels.hip
int launch_price(const LPArgs &a, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
int launch_ftran(const LPArgs &a, int nparts_price, int forced_pos, int forced_var, hipStream_t s, hipEvent_t e0 = nullptr,
                 hipEvent_t e1 = nullptr);
void launch_update(const LPArgs &a, int nparts_ratio, int forced_p, int no_swap, int bland, hipStream_t s,
                   hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// fused_kernels.hip
bool fused_supported(int ld);
int launch_price_fused(const LPArgs &a, const double *y_in, double *y_out, int pending, int nparts_ratio, hipStream_t s,
                       hipEvent_t e0, hipEvent_t e1);
int launch_update_ftran_fused(const LPArgs &a, int pending, int nparts_price, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// tableau_kernels.hip
int tab_ld(int nn);
int launch_tableau_pivot(const TabArgs &a, int flags, int nparts, long long t, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
void launch_tab_gather(const double *At, int ld, int m, int nn, const int32_t *nonbasic, const int32_t *rho, double *T, int ldt,
                       bool tiled, hipStream_t s);
void launch_tab_gemm(const double *Binv, int ldb, const double *At, int ld, int m, int nn, const int32_t *nonbasic, double *T, int ldt, bool tiled, hipStream_t s);
void launch_tab_permute_cols(const double *Tin, int ld_in, double *Tout, int ld_out, int m, int nn_out, const int32_t *srcpos,
                             bool tiled, hipStream_t s);
int tab_r_chunks(int m);
void launch_tab_r(const double *T, int ldt, int m, int nn, const double *cost, const int32_t *basic, const int32_t *nonbasic,
                  double *scratch, double *r, bool tiled, hipStream_t s);
void launch_tab_row_colmax(const double *T, int ldt, int m, int nn, int row, double *out, bool tiled, hipStream_t s);
void launch_tab_column(const double *T, int ldt, int m, int jp, const double *xb, double *dvec, double *move, bool tiled, hipStream_t s);
void launch_cond_check(const double *T, int ldt, int m, int nn, const int32_t *nonbasic, const int32_t *basic, const double *At, int ld, int slack0, int nvar,
                       double *scratch, bool tiled, hipStream_t s);
void launch_exact_r(const double *At, int ld, int m, int nn, const int32_t *nonbasic, const double *y, const double *cost, double *r, int ldt, hipStream_t s);
// general_kernels.hip
void launch_gs_init(double *QT, int ldq, int m, GsState *st, hipStream_t s);
void launch_gs_init_perm(double *QT, double *Rinv, int ldq, int m, const int32_t *perm, const double *sgn, const double *beta, int s0, GsState *st, hipStream_t s);
void launch_gs_candidate(const double *acol, double *QT, double *Rinv, int ldq, int m, double *w, double *t, double *ypart, int cand, int32_t *idxs, GsState *st, hipStream_t s,
                         int last = 0);
int launch_luc_rounds_cross(const LUArgs &base, int32_t *pivrow, int nrounds, double *xrec, int G, hipStream_t s);
size_t luc_cross_doubles();
void launch_luc_lpos_final(const LUArgs &a, hipStream_t s);
int luc_cross_groups(int m, int want);
int gs_scratch_rows();
int gs_block_width(int m);
int gs_block_scratch_rows();
void launch_gs_block(const double *At, int ld, int col0, int ncand, double *QT, double *Rinv, int ldq, int m, double *scratch, int32_t *idxs, GsState *st, hipStream_t s);
void launch_gs_binv(const double *Rinv, const double *QT, int ldq, int m, double *C, hipStream_t s);
void launch_gs_art(const double *At, int ld, int m, const int32_t *basic, int minidx, const double *b, double *art, hipStream_t s);
void launch_gs_norms(const double *C, int ldq, int m, double *out, const double *At, int ld, const int32_t *idxs, int cand, double *colsum, hipStream_t s);
// bt_kernels.hip
bool bt_supported(int m, int nn);
int bt_max_k();
int bt_reg_k(int m, int ldt, int nt_force);
bool bt_tiled(int m, int ldt, int kmax, int nt_force, bool old_only);
// btg_kernels.hip: the block kernel over G workgroups of one XCD
BtGroupCfg bt_group_cfg(int m, int ldt, int knob);
size_t bt_xbuf_doubles();
void launch_bt_inner_groups(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
const char *bt_group_kernel_name(int G, int ri);
constexpr int kBtStampSegs = 16;   // cycle sums per wave written by the diagnostic build of k_bt_inner2
void launch_bt_tile(const double *src, double *dst, int m, int ldt, bool to_tiles, hipStream_t s);
void launch_bt_inner(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
void launch_bt_update(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
bool bt_loop_supported(const BtGroupCfg &c);
void launch_bt_loop(const BTArgs &a, int ncu, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
// btr_kernels.hip
bool bt_loop_rep_supported(int m, int ldt);
int bt_loop_rep_threads(int m, int ldt);
long long bt_loop_rep_launches();
void launch_bt_loop_rep(const BTArgs &a, int ncu, hipStream_t s, hipEvent_t e0, hipEvent_t e1);
bool bt_batch_supported(int m_max, int ldt_max);
int bt_batch_k(int m_max, int ldt_max);
void launch_bt_inner_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, int xcd_off = 0);
void launch_bt_update_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
const char *bt_batch_kernel_name(int m_max, int ldt_max);
// the batched block kernel on a virtual tableau (bt_kernels.hip k_bt_inner2_virt_batch; BatchLP::virt)
bool bt_virt_batch_supported(int m_max, int ldt_max);
void launch_bt_inner_virt_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// batched persistent loop kernel (bt_kernels.hip k_b_loop): relaxations per launch on a device with ncu CUs; shapes it takes
int b_loop_slots(int ncu);
bool b_loop_supported(int m_max, int ldt_max);
void launch_b_loop(const BatchLP *lps, const int *ids, const int *count, int nlp, int nblocks, int par, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// res_kernels.hip: the register-resident tableau kernel for the long chains of a narrow wave (G workgroups per relaxation on one XCD)
size_t b_res_slot_bytes();
int b_res_max_slots();
int b_res_groups(int m_max, int nn_max, int ldt_max);
void launch_b_res(const BatchLP *lps, const int *ids, const int *count, int nlp, int G, int nb, double seq0, void *xbase, hipStream_t s, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// batch_kernels.hip
int batch_ldt(int nn);
void launch_b_setup(BatchLP *lps, int nlp, hipStream_t s);
void launch_b_setup_warm(BatchLP *lps, int nlp, hipStream_t s);
void launch_b_posvar(const int32_t *basic, int m, const int32_t *nonbasic, int nn, int32_t *posvar, hipStream_t s);
bool bt_dual_batch_supported(int m_max, int ldt_max);
void launch_bt_inner_dual_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s);
void launch_b_gather(const BatchLP *lps, int nlp, int m_max, int ldt_max, int mode, hipStream_t s);   // mode: batch_kernels.hip k_b_gather
void launch_b_write_virt(const BatchLP *lps, const int *ids, const int *count, int *list, int *nalive, int bound, int m_max, int ldt_max, int ncu, hipStream_t s);          // the survivors of a virtual first block (= mode 3, one tile per thread)
void launch_b_ctrl(BatchLP *lps, const int *ids_in, const int *count_in, int bound, int n_max, BatchOut *outs, int *ids_out, int *count_out, int loop_par, hipStream_t s);
void launch_b_init_ids(int *ids, int *count, int nlp, hipStream_t s);
void launch_b_permute(const BatchLP *lps, const int *ids, const int *count, int bound, int m_max, int ldt_max, hipStream_t s);
void launch_b_tab_r(const BatchLP *lps, const int *ids, const int *count, int bound, int m_max, int ldt_max, hipStream_t s);
void launch_transpose_in(const double *A, int64_t lda, int m, int n, double *At, int ld, hipStream_t s);
void launch_col_stats(const double *At, int ld, int m, int n, int32_t *nnz, int32_t *lastrow, int32_t *allone, int32_t *rowflag, unsigned long long *range,
                      hipStream_t s);
void launch_set_binv_perm(double *binv, int ld, int m, const int32_t *rho, hipStream_t s);
void launch_child_assemble(const double *At0, int ld0, int m0, int n0, double *At1, int ld1, int K, const int32_t *var,
                           const double *sign, hipStream_t s);
void launch_matvec_rows(const double *M, int ld, int m, const double *vec, double *out, hipStream_t s);
int y_chunks(int m);
void launch_y_from_binv(const double *binv, int ld, int m, const double *cost, const int32_t *basic, double *scratch,
                        double *y, hipStream_t s);
void launch_gather_w(const double *At, int ld, int m, const int32_t *basic, double *W, int ldw, hipStream_t s);
int lu_grid(int m);
void launch_lu(const LUArgs &a, hipStream_t s);
// lu_kernels.hip
bool lu_blocked_supported(int m);
int launch_lu_blocked(const LUArgs &a, int32_t *pivrow, hipStream_t s);
// lu_compressed.hip
bool lu_compressed_supported(int m);
int lu_compressed_nb(int m, bool slots);
void launch_luc_init(const LUArgs &a, hipStream_t s);
int launch_luc_rounds(const LUArgs &a, int32_t *pivrow, int nrounds, int round_base, hipStream_t s);
void launch_luc_gather(const double *At, int ld, int m, const int32_t *basic, double *W, int ldw, hipStream_t s);
void launch_luc_pack_dense(const LUArgs &a, const int32_t *dlist, int nd, const int32_t *pivrow, double *Wdd, double *diag, hipStream_t s);
void launch_luc_solve_rows(const LUArgs &a, const int32_t *dlist, int nd, const double *b, const double *xdL, const double *xdU,
                           double *x, hipStream_t s);
void launch_luc_pack(const LUArgs &a, const int32_t *dlist, int nd, double *Wd, double *diag, hipStream_t s);
void launch_luc_pack_small(const LUArgs &a, double *out, hipStream_t s);   // small bases: factors, diagonal, row positions, flags, control blocks in one block
size_t luc_pack_small_bytes(int m);
void launch_lu_pack(const LUArgs &a, const int32_t *dlist, int nd, double *Wd, double *diag, hipStream_t s);

}  // namespace gomilp
