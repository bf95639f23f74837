// Host control of the single-kernel tableau pipeline (tableau_kernels.hip).  Same reference control flow as
// engine.cpp (simplex.go:93-302); only the per-pivot device work differs: T = B^-1 A_N is kept explicitly and
// one launch applies a whole pivot.
#include <stdio.h>
#include <stdlib.h>

#include "engine_work.hpp"

namespace gomilp {

TabArgs Engine::make_tab_args(const Problem &P, int phase, double tol, int nn) {
    Work &w = *w_;
    TabArgs a;
    memset(&a, 0, sizeof(a));
    a.m = P.m; a.nn = nn; a.ldt = ldt_; a.phase = phase; a.tol = tol;
    bt_layout(P, false);
    a.T_cur = w.T[tcur_]; a.T_next = w.T[tcur_ ^ 1];
    a.r_in = w.R[rcur_]; a.r_out = w.R[rcur_ ^ 1];
    a.xb = w.xb; a.dvec = w.dvec; a.move = w.move;
    a.basic = w.basic; a.nonbasic = w.nonbasic;
    a.pk_ratio = w.pk_ratio; a.pi_ratio = w.pi_ratio; a.pb_ratio = w.pb_ratio; a.pd_ratio = w.pd_ratio; a.px_ratio = w.px_ratio;
    a.st = w.st;
    a.trace = (trace_on_ || shadow_trace_) ? w.trace : nullptr;
    a.trace_cap = w.trace_cap;
    return a;
}

// One pivot chosen by the host (Phase-I set-up, artificial exchange, Bland step): entering position q (variable ent,
// reduced cost rq), leaving position p with pivot element dp, x_B[p] = xp, leaving variable lea.  dvec must already hold
// column q of the current T (launch_tab_column).  The launch is kernel `t` of a fresh segment.
int Engine::tab_forced_pivot(const Problem &P, int phase, double tol, int nn, int q, int ent, double rq, int p, double dp,
                             double xp, int lea, int flags, long long t) {
    Work &w = *w_;
    DevState &hs = *w.st_host;
    const int par = (int)(t & 1);
    hs.done = 0; hs.status = ST_RUNNING; hs.stop_at = std::numeric_limits<int64_t>::max();
    hs.nq[par] = q; hs.nent[par] = ent; hs.nrq[par] = rq;
    sync_state_to_device();
    // fabricate the single ratio partial the kernel will reduce
    struct { unsigned long long k; unsigned int i, b; double d, x; } part;
    part.k = 0xBFF0000000000000ull;  // ordkey(1.0): any finite positive ratio
    part.i = (unsigned int)p; part.b = (unsigned int)lea; part.d = dp; part.x = xp;
    HIP_TRY(hipMemcpyAsync(w.pk_ratio, &part.k, sizeof(part.k), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(w.pi_ratio, &part.i, sizeof(part.i), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(w.pb_ratio, &part.b, sizeof(part.b), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(w.pd_ratio, &part.d, sizeof(part.d), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(w.px_ratio, &part.x, sizeof(part.x), hipMemcpyHostToDevice, stream_));
    HIP_TRY(sync_stream());  // `part` lives on this stack frame
    TabArgs a = make_tab_args(P, phase, tol, nn);
    grid_ratio_ = launch_tableau_pivot(a, flags | 1 | 2, 1, t, stream_, nullptr, nullptr);
    launches_++;
    tcur_ ^= 1;
    rcur_ ^= 1;
    return GOMILP_OK;
}

// replaceBland (simplex.go:347-383) on the tableau: a candidate's FTRAN is just its column of T.
// On success the Bland pivot has been enqueued as kernel 0 of a new segment.
int Engine::host_bland_tab(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st, int *) {
    Work &w = *w_;
    const int m = P.m;
    std::vector<double> r(nn), move(m), xb(m), dv(m);
    std::vector<int32_t> bas(m);
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.R[rcur_], (size_t)nn * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    std::vector<double> rraw(w.h_vec, w.h_vec + nn);
    for (int j = 0; j < nn; j++) { r[j] = rraw[j]; if (fabs(r[j]) < 1e-13) r[j] = 0; }  // rRoundTol, :252-256
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    for (int i = 0; i < m; i++) { xb[i] = w.h_vec[i]; bas[i] = w.h_idx[i]; }
    for (int i = 0; i < nn; i++) {
        if (r[i] > -1e-14) continue;  // blandNegTol, :352
        bt_layout(P, false);
        launch_tab_column(w.T[tcur_], ldt_, m, i, w.xb, w.dvec, w.move, false, stream_);
        launches_++;
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.move, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int k = 0; k < m; k++) move[k] = w.h_vec[k];
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.dvec, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int k = 0; k < m; k++) dv[k] = w.h_vec[k];
        int64_t replace = min_idx(move.data(), m);
        if (move[replace] == std::numeric_limits<double>::infinity()) return GOMILP_ERR_UNBOUNDED;  // computeMove :328
        if (!(fabs(move[replace]) > 1e-12)) {  // blandZeroTol, :362
            replace = -1;
            for (int rp = 0; rp < m; rp++)
                if (!(move[rp] > 1e-12)) { replace = rp; break; }  // :368-379 (cond guard: DESIGN.md §3)
            if (replace < 0) continue;
        }
        int32_t ent = 0;
        HIP_TRY(hipMemcpy(&ent, w.nonbasic + i, sizeof(int32_t), hipMemcpyDeviceToHost));
        int rc = tab_forced_pivot(P, phase, tol, nn, i, ent, rraw[i], (int)replace, dv[replace], xb[replace], bas[replace], 8, 0);
        if (rc != GOMILP_OK) return rc;
        if (st) st->bland_steps++;
        return GOMILP_OK;
    }
    return GOMILP_ERR_BLAND;
}

// Pivot loop, one launch per pivot.  Kernel index t inside a segment: slot parity t & 1; a launch that applies a
// pivot reads T[(c0+a)&1], r[(r0+a)&1] and writes the other copies, a = pivots applied so far in the segment.
int Engine::run_loop_tab(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st) {
    Work &w = *w_;
    DevState &hs = *w.st_host;
    hs.done = 0; hs.status = ST_RUNNING; hs.pivots = 0; hs.q = hs.p = -1; hs.rq = hs.dp = hs.mv = 0;
    hs.max_pivots = 0; hs.lu_singular = 0;
    hs.stop_at = std::numeric_limits<int64_t>::max();
    sync_state_to_device();
    HIP_TRY(hipEventRecord(w.ev[0], stream_));
    int ret = GOMILP_OK;
    const bool sampling = sample_events_ > 0;
    int c0 = tcur_, r0 = rcur_;
    long long tseg = 0;          // next kernel index in the segment
    int64_t applied = 0;         // pending launches enqueued in the segment
    int64_t seg_start = 0;       // pivots committed when the segment started
    bool first_pending = false;  // the segment started with a forced (already enqueued) pivot
    for (;;) {
        const int64_t before = hs.pivots;
        std::vector<int64_t> sample_t;
        int64_t nlaunch = chunk_;
        if (max_pivots_ > 0) nlaunch = std::min<int64_t>(nlaunch, std::max<int64_t>(1, max_pivots_ - hs.pivots + 1));
        const int64_t applied_before = applied;
        for (int64_t t = 0; t < nlaunch; t++, tseg++) {
            const int pending = (tseg > 0 || first_pending) ? 1 : 0;
            TabArgs a = make_tab_args(P, phase, tol, nn);
            a.T_cur = w.T[(c0 + applied) & 1]; a.T_next = w.T[(c0 + applied + 1) & 1];
            a.r_in = w.R[(r0 + applied) & 1]; a.r_out = w.R[(r0 + applied + 1) & 1];
            const bool sample = sampling && pending && ((before + t) % sample_events_ == 0);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (sample) {
                const size_t b0 = sample_t.size() * 6;
                while (w.sample_ev.size() < b0 + 6) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); w.sample_ev.push_back(ev); }
                e0 = w.sample_ev[b0]; e1 = w.sample_ev[b0 + 1];
                sample_t.push_back(applied - applied_before);
            }
            grid_ratio_ = launch_tableau_pivot(a, pending, grid_ratio_, tseg, stream_, e0, e1);
            launches_++;
            if (pending) applied++;
        }
        HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        const int64_t executed = hs.pivots - before;
        if (st) {
            for (size_t s = 0; s < sample_t.size(); s++) {
                if (sample_t[s] >= executed) break;
                float ms = 0;
                if (hipEventElapsedTime(&ms, w.sample_ev[s * 6], w.sample_ev[s * 6 + 1]) != hipSuccess) continue;
                st->pivot_kernel_seconds[2] += ms * 1e-3;
                st->pivot_kernel_seconds[3] += 1.0;
            }
        }
        const int64_t T = hs.pivots - seg_start;  // pivots applied and committed in this segment
        if (!hs.done) {
            if (max_pivots_ > 0 && hs.pivots >= max_pivots_) { tcur_ = (c0 + T) & 1; rcur_ = (r0 + T) & 1; ret = GOMILP_ERR_UNSUPPORTED; break; }
            continue;
        }
        tcur_ = (int)((c0 + T) & 1);
        rcur_ = (int)((r0 + T) & 1);
        if (hs.status == ST_OPTIMAL) break;
        if (hs.status == ST_UNBOUNDED) { ret = GOMILP_ERR_UNBOUNDED; break; }
        if (hs.status == ST_NEED_BLAND) {
            int rc = host_bland_tab(P, phase, tol, nn, st, nullptr);  // enqueues the Bland pivot as kernel 0 of a new segment
            if (rc != GOMILP_OK) { ret = rc; break; }
            // tab_forced_pivot flipped tcur_/rcur_ for the pivot it enqueued; the new segment starts BEFORE that pivot
            c0 = tcur_ ^ 1; r0 = rcur_ ^ 1;
            seg_start = hs.pivots; tseg = 1; applied = 1; first_pending = true;
            continue;
        }
        ret = GOMILP_ERR_DEVICE;
        break;
    }
    HIP_TRY(hipEventRecord(w.ev[1], stream_));
    HIP_TRY(hipEventSynchronize(w.ev[1]));
    float ms = 0;
    hipEventElapsedTime(&ms, w.ev[0], w.ev[1]);
    if (st) {
        st->seconds_pivot_loop += ms * 1e-3;
        if (phase == 1) st->pivots_phase1 += hs.pivots; else st->pivots_phase2 += hs.pivots;
    }
    return ret;
}

// ---- blocked tableau (bt_kernels.hip) -----------------------------------------------------------------------

// T[tcur_] lives in 4x4 tiles while the register-resident block kernels run (bt_kernels.hip) and in row-major
// order for everything else (set-up, column extraction, reduced-cost rebuild, the single-kernel pipeline).  One streaming
// conversion into the other T buffer per change of layout.
void Engine::bt_layout(const Problem &P, bool tiled) {
    if (tiled == t_tiled_) return;
    Work &w = *w_;
    launch_bt_tile(w.T[tcur_], w.T[tcur_ ^ 1], P.m, ldt_, tiled, stream_);
    launches_++;
    tcur_ ^= 1;
    t_tiled_ = tiled;
}

// knob "bt_groups" as the shape functions see it: by default (0) relaxations of 600..1024 rows also take the loop-kernel shapes
// (value 16) — the persistent kernel's 3.75 us per pivot with the update hidden beat the single-workgroup block kernel's 3.1 us +
// update + two launch boundaries per 8 pivots (loop of C2: 5.27 -> 4.74 ms; 600 rows: 1.24 -> 1.13; 700: 2.56 -> 2.35; 900: 2.73 ->
// 2.48); at 512 rows the two are level (1.21 ms both) and the 520-row children of the 512-row trees keep the single-workgroup kernel
int Engine::groups_knob(const Problem &P) const {
    if (bt_groups_ != 0 || !bt_lag_ || bt_stamps_ || block_k_ != 0 || max_pivots_ != 0) return (int)bt_groups_;
    const int need = std::max(P.m, ldt_);
    return (P.m >= 600 && need <= 1024) ? 16 : 0;
}

void Engine::bt_plan(const Problem &P, int *K, bool *tiled, bool *lag) const {
    const BtGroupCfg gc = bt_old_ ? BtGroupCfg{0, 0, 0} : bt_group_cfg(P.m, ldt_, groups_knob(P));
    if (lag) *lag = false;
    if (gc.groups) {   // multi-workgroup block kernel: 16 terms per row / column in registers, tiled layout
        *K = block_k_ > 0 ? (int)std::min<int64_t>(block_k_, 16) : 16;
        *tiled = true;
        // persistent loop kernel (default, btg_kernels.hip k_bt_loop): blocks of 8 pivots, 8 lagging + 8 current terms, the rank-8
        // update of block t applied by the other workgroups of the same launch beside block t+1
        // (512-thread shapes — beyond 2048 rows — run blocks of 16: there the rank-8 update of a 134 MB tableau takes longer than 8
        // pivots, and 16 + 16 terms still fit the 256 registers of a 512-thread workgroup)
        if (lag && bt_lag_ && block_k_ == 0 && max_pivots_ == 0 && (!bt_stamps_ || gc.nt == 256) && bt_loop_supported(gc)) {
            *lag = true;
            *K = (gc.nt == 512 && gc.groups == 8 && loop_k_ != 8) ? (loop_k_ == 12 && loop_g_ != 8 ? 12 : 16) : 8;
            // knob loop_k = 16 on the 1025..2048-row class: blocks of 16 on 16 x 256 threads (the 4096-row instance) — half the update
            // traffic per pivot, which is what several such loop kernels side by side run out of (gomilp_frontier_solve_roots, large_loop)
            if (gc.nt == 256 && gc.groups == 8 && loop_k_ == 16 && loop_g_ != 8 && !bt_stamps_) *K = 16;
        }
        return;
    }
    // block size: 8 when the block's rank-1 terms fit in registers (bt_kernels.hip), else 16
    *K = block_k_ > 0 ? (int)block_k_ : (bt_reg_k(P.m, ldt_, (int)bt_nt_) > 0 ? 8 : 16);
    *tiled = bt_tiled(P.m, ldt_, *K, (int)bt_nt_, bt_old_ != 0);
}

BTArgs Engine::make_bt_args(const Problem &P, int phase, double tol, int nn, int kmax) {
    Work &w = *w_;
    BTArgs a;
    memset(&a, 0, sizeof(a));
    a.tiled = t_tiled_ ? 1 : 0;
    a.m = P.m; a.nn = nn; a.ldt = ldt_; a.ldu = P.ld; a.phase = phase; a.kmax = kmax; a.tol = tol;
    a.T = w.T[tcur_]; a.U = w.btU; a.V = w.btV; a.r = w.R[rcur_]; a.xb = w.xb;
    a.basic = w.basic; a.nonbasic = w.nonbasic; a.st = w.st;
    a.trace = (trace_on_ || shadow_trace_) ? w.trace : nullptr; a.trace_cap = w.trace_cap;
    a.forced_q = a.forced_p = -1; a.forced_nocommit = 0;
    a.nt_force = (int)bt_nt_; a.old_only = bt_old_ ? 1 : 0;
    a.stamps = bt_stamps_ ? w.stamps : nullptr;
    a.upd_valu = bt_upd_valu_ ? 1 : 0;
    a.fault = bt_fault_ == 1 ? 1 : 0;
    // degenerate vertices decided on a fresh gonum-order x_B: by default for bases of up to 256 rows, and for
    // every start that is not a slack basis (equality rows: a tree's repeated branch rows make nearly dependent tableau rows
    // there, and a pivot on their 1e-12 drift walks into a singular basis)
    // (and for inputs whose entries span more than nine decades: their updated tableau loses digits, the exact steps check and rebuild it)
    // (3, strict: an infinite guard — the block kernel stops in front of EVERY decision, stop test included, and the host repeats the
    // reference's iteration on fresh solves: slow by design, the mode that follows the reference wherever its rounding noise leads)
    a.guard = exact_degenerate_ == 3 ? std::numeric_limits<double>::infinity()
                                     : (exact_degenerate_ == 2 || (exact_degenerate_ == 1 && (P.m <= 256 || gen_start_ || badly_scaled_))) ? 1e-9 : 0.0;
    a.cguard = (cond_guard_ && !gen_start_ && P.m > 64) ? 1e-9 : 0.0;   // (<= 64 rows: the per-pivot replay of Engine::solve; general starts: the guard above is on)
    if (a.tiled && !bt_old_) {
        const BtGroupCfg gc = bt_group_cfg(P.m, ldt_, groups_knob(P));
        a.groups = gc.groups; a.group_ri = gc.ri; a.group_nt = gc.nt; a.xbuf = w.xbuf;
    }
    return a;
}

// one host-chosen pivot (Phase-I set-up / artificial exchange): a block of one forced pivot + its update
int Engine::bt_forced_pivot(const Problem &P, int phase, double tol, int nn, int q, int p, int nocommit) {
    Work &w = *w_;
    DevState &hs = *w.st_host;
    hs.done = 0; hs.status = ST_RUNNING; hs.kdone = 0;
    sync_state_to_device();
    { int K1; bool tiled1; bt_plan(P, &K1, &tiled1); bt_layout(P, tiled1 || bt_tiled(P.m, ldt_, 1, (int)bt_nt_, bt_old_ != 0)); }
    BTArgs a = make_bt_args(P, phase, tol, nn, 1);
    a.forced_q = q; a.forced_p = p; a.forced_nocommit = nocommit;
    launch_bt_inner(a, stream_, nullptr, nullptr);
    launch_bt_update(a, stream_, nullptr, nullptr);
    launches_ += 2;
    return GOMILP_OK;
}

// One iteration of the reference on FRESH solves (simplex.go:233-277), for a pivot the block kernel would not decide on its
// updated quantities (ST_NEED_EXACT: reduced cost at the stop threshold, tied reduced costs, winning ratio (nearly) zero, tied
// ratios — the places where the rounding noise of the reference's three LU solves per pivot takes the decision).  Same
// arithmetic: y from a gonum-order LU of ab^T with right-hand side c_B, r = c_N - an^T y in Dgemv / SubTo order on the device,
// d and x_B from gonum-order LUs of ab.  Afterwards the fresh r and x_B are resident; returns
//   0 pivot (q, p) decided — the caller enqueues it as a forced pivot; 1 optimal; 2 unbounded; 3 move[replace] <= 0: the Bland
//   rule, which the block kernel runs on the fresh r / x_B (exact_once); < 0: -status of a failure.
int Engine::exact_step(const Problem &P, int phase, double tol, int nn, int *q_out, int *p_out, gomilp_lp_stats *st) {
    Work &w = *w_;
    const int m = P.m, n = P.n;
    std::vector<int32_t> basic(m), nonbasic(nn);
    {
        const size_t gap = (size_t)(w.nonbasic - w.basic);   // the two lists share one device block
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (gap + (size_t)nn) * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int i = 0; i < m; i++) basic[i] = w.h_idx[i];
        for (int j = 0; j < nn; j++) nonbasic[j] = w.h_idx[gap + j];
    }
    // gonum's guards on the solves of this iteration (mat/lu.go:321: cond > 1e16 -> mat.Condition out of the duals' solve,
    // simplex.go:236-239; lp.ErrLinSolve out of computeMove, :316-318): the tableau of a slack-basis start holds B^-1, so the exact
    // kappa_1 / kappa_inf of the current basis cost three small launches (cond_check) — every exact step measures them
    if (cond_guard_ && !gen_start_ && m > 64) {   // (Phase I too: the artificial column is a column of the basis like any other)
        double k1 = 0, kinf = 0;
        int rc0 = cond_check(P, nn, &k1, &kinf);
        if (rc0 != GOMILP_OK) return -rc0;
        // (kappa_inf belongs to the x_B solve that closed the PREVIOUS iteration, simplex.go:289-292: the reference would have left
        // its loop there with mat.Condition, before this iteration's computeMove could report lp.ErrLinSolve)
        if (k1 > 1e16 || k1 != k1 || kinf > 1e16 || kinf != kinf) return -GOMILP_ERR_CONDITION;
    } else if (cond_guard_ && exact_degenerate_ == 3) {
        // strict mode on a searched start basis (or a small one): no B^-1 inside the tableau — the same verdict from a fresh host inverse,
        // every pivot (the mode is slow by design: it exists to follow the reference wherever its solves' rounding noise leads)
        double k1 = 0, kinf = 0;
        int rc0 = cond_fresh(P, basic.data(), &k1, &kinf);
        if (rc0 != GOMILP_OK) return -rc0;
        if (k1 > 1e16 || k1 != k1 || kinf > 1e16 || kinf != kinf) return -GOMILP_ERR_CONDITION;
    }
    auto cost = [&](int var) -> double { return phase == 1 ? (var == n ? 1.0 : 0.0) : (var < n ? P.hc[var] : 0.0); };
    std::vector<double> cb(m), y, xb, dsol, col(m);
    for (int i = 0; i < m; i++) cb[i] = cost(basic[i]);
    bool sing = false;
    int rc;
    if ((rc = final_solve(P, n, y, &sing, basic.data(), true, cb.data())) != GOMILP_OK) return -rc;
    if (sing) { if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "exact_step: ab^T singular (phase %d, m %d)\n", phase, m); return -GOMILP_ERR_LINSOLVE; }
    {
        std::vector<double> ypad(P.ld, 0.0);
        std::copy(y.begin(), y.begin() + m, ypad.begin());
        if ((rc = stage_upload(w.dvec, ypad.data(), (size_t)P.ld * sizeof(double))) != GOMILP_OK) return -rc;
    }
    launch_exact_r(P.dAt, P.ld, m, nn, w.nonbasic, w.dvec, phase == 1 ? P.dc1 : P.dc, w.R[rcur_], ldt_, stream_);
    launches_++;
    std::vector<double> r(nn);
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.R[rcur_], (size_t)nn * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    for (int j = 0; j < nn; j++) r[j] = w.h_vec[j];
    // x_B of this iteration (simplex.go:289 of the previous one): resident from here on
    // (ONE factorization of ab serves x_B, the entering column and every Bland candidate below: the reference factors the same matrix
    // again each time, simplex.go:289 / :315 / :356 — same bits; Engine::lu_factor / lu_solve)
    if ((rc = lu_factor(P, &sing, basic.data())) != GOMILP_OK) return -rc;
    if (sing) { if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "exact_step: ab singular for x_B (phase %d, m %d)\n", phase, m); return -GOMILP_ERR_LINSOLVE; }
    if ((rc = lu_solve(P, xb)) != GOMILP_OK) return -rc;
    {
        std::vector<double> xpad(P.ld, 0.0);
        std::copy(xb.begin(), xb.begin() + m, xpad.begin());
        if ((rc = stage_upload(w.xb, xpad.data(), (size_t)P.ld * sizeof(double))) != GOMILP_OK) return -rc;
    }
    if (st) st->cond_fallbacks++;
    const int64_t q = min_idx(r.data(), nn);   // simplex.go:247
    if (r[q] >= -tol) return 1;                // :248
    // computeMove (:306-342): d = -solve(ab, A[:, entering])
    if (!P.hA.empty() && nonbasic[q] < n) {   // (a host copy of A is there: no round trip for the entering column)
        for (int i = 0; i < m; i++) col[i] = P.hA[(size_t)i * n + nonbasic[q]];
    } else {
        HIP_TRY(hipMemcpyAsync(w.h_vec, P.dAt + (size_t)nonbasic[q] * P.ld, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int i = 0; i < m; i++) col[i] = w.h_vec[i];
    }
    if ((rc = lu_solve(P, dsol, col.data())) != GOMILP_OK) return -rc;
    // Accuracy of the resident tableau: its column q against the fresh one.  On badly scaled LPs (entries over 1e19) the updated
    // tableau loses digits within a few hundred pivots, its ratio tests then leave the reference's path and may never end; beyond
    // 1e-6 of the column's size the tableau is rebuilt from a fresh inverse of the basis (B^-1 on the host, T = B^-1 A_N as one
    // device GEMM: the set-up of a general start), up to 1024 rows
    if (m <= 1024) {
        launch_tab_column(w.T[tcur_], ldt_, m, (int)q, w.xb, w.dvec, w.move, t_tiled_, stream_);
        launches_++;
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.dvec, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        double scale = 0, err = 0;
        for (int i = 0; i < m; i++) { scale = std::max(scale, fabs(dsol[i])); err = std::max(err, fabs(w.h_vec[i] - dsol[i])); }
        if (!(err <= 1e-6 * scale)) {
            std::vector<double> cols((size_t)m * P.ld), B((size_t)m * m), inv;
            for (int p = 0; p < m; p++)
                HIP_TRY(hipMemcpyAsync(&cols[(size_t)p * P.ld], P.dAt + (size_t)basic[p] * P.ld, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(sync_stream());
            for (int i = 0; i < m; i++) for (int p = 0; p < m; p++) B[(size_t)i * m + p] = cols[(size_t)p * P.ld + i];
            if (!general_invert(B, m, inv)) return -GOMILP_ERR_LINSOLVE;
            HIP_TRY(hipMemcpy2DAsync(w.binv[0], (size_t)P.ld * sizeof(double), inv.data(), (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                     hipMemcpyHostToDevice, stream_));
            HIP_TRY(sync_stream());   // (pageable source)
            launch_tab_gemm(w.binv[0], P.ld, P.dAt, P.ld, m, nn, w.nonbasic, w.T[tcur_], ldt_, t_tiled_, stream_);
            launches_++;
            if (st) st->refreshes++;
            if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "exact_step: tableau rebuilt (column error %.3g of %.3g, m %d)\n", err, scale, m);
        }
    }
    std::vector<double> move(m);
    bool anyneg = false;
    for (int i = 0; i < m; i++) {
        double d = -dsol[i];
        if (fabs(d) < 1e-13) d = 0;            // dRoundTol, :321-325
        if (d < 0) anyneg = true;
        move[i] = d >= 0 ? std::numeric_limits<double>::infinity() : xb[i] / fabs(d);   // :334-340
    }
    if (!anyneg) return 2;                      // :328-330
    const int64_t p = min_idx(move.data(), m);  // :268
    if (!(move[p] <= 0)) {
        *q_out = (int)q; *p_out = (int)p; return 0;
    }
    // ---- :269 -> replaceBland (:347-383) on the same fresh quantities: candidates in position order with r <= -1e-14 after the
    // rounding of :252-256, each with its own computeMove; a zero-level row is only taken when the basis it gives is not
    // singular (mat.Cond(abTmp, 1) < 1e16 — here the exact kappa_1 on the host copy of A: the device Bland rule has no such test
    // and has walked into singular bases on equality-constrained LPs)
    if (!ensure_host_A(P)) return 3;
    if (st) st->bland_steps++;
    std::vector<double> art(m, 0.0);
    if (phase == 1) {
        HIP_TRY(hipMemcpyAsync(w.h_vec, P.dAt + (size_t)n * P.ld, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int i = 0; i < m; i++) art[i] = w.h_vec[i];
    }
    for (int j = 0; j < nn; j++) if (fabs(r[j]) < 1e-13) r[j] = 0;   // rRoundTol
    for (int i = 0; i < nn; i++) {
        if (r[i] > -1e-14) continue;            // blandNegTol, :352
        const int var = nonbasic[i];
        for (int t = 0; t < m; t++) col[t] = var < n ? P.hA[(size_t)t * n + var] : art[t];
        if ((rc = lu_solve(P, dsol, col.data())) != GOMILP_OK) return -rc;
        bool neg = false;
        for (int t = 0; t < m; t++) {
            double d = -dsol[t];
            if (fabs(d) < 1e-13) d = 0;
            if (d < 0) neg = true;
            move[t] = d >= 0 ? std::numeric_limits<double>::infinity() : xb[t] / fabs(d);
        }
        if (!neg) return 2;                     // computeMove inside Bland: ErrUnbounded, :356-360
        const int64_t rp0 = min_idx(move.data(), m);
        if (fabs(move[rp0]) > 1e-12) { *q_out = i; *p_out = (int)rp0; return 0; }   // blandZeroTol, :362
        for (int rp = 0; rp < m; rp++) {
            if (move[rp] > 1e-12) continue;
            std::vector<int32_t> trial = basic;
            trial[rp] = var;
            const double kappa = general_basis_cond1(P.hA, m, n, trial, art);
            if (kappa < 1e16) { *q_out = i; *p_out = rp; return 0; }           // :377
        }
    }
    return -GOMILP_ERR_BLAND;                   // :382
}

// Pivot loop: blocks of block_k_ pivots (one single-workgroup launch) followed by one rank-K update launch.
int Engine::run_loop_bt(const Problem &P, int phase, double tol, int nn, gomilp_lp_stats *st) {
    Work &w = *w_;
    DevState &hs = *w.st_host;
    hs.done = 0; hs.status = ST_RUNNING; hs.pivots = 0; hs.kdone = 0; hs.bland_steps = 0; hs.lu_singular = 0;
    int K; bool tiled_plan, lag;
    bt_plan(P, &K, &tiled_plan, &lag);
    // The persistent loop kernel needs all its workgroups resident (they wait for each other) and takes one slot on every CU:
    // one at a time per device.  Concurrent solves of large LPs queue up here for the duration of their pivot loops (set-up and
    // final solve still overlap); many large LPs at once belong in the device-batched schedule (gomilp_frontier_solve_roots).
    struct LoopSlot {
        int dev, weight, slot = -1;
        LoopSlot(int d, int w2, bool on) : dev(d), weight(w2) { if (on) slot = Engine::loop_acquire(d, w2); }
        ~LoopSlot() { if (slot >= 0) Engine::loop_release(dev, weight, slot); }
    };
    // (weight: the 128-thread shape shares the device with up to three others, every other shape runs alone — engine.cpp)
    // (the 2048-row class in blocks of 16 — knob loop_k — shares the device too: 256-thread workgroups, two per CU, grid of half the CUs)
    const bool small_loop = lag && (K == 8 || K == 16) && loop_g_ != 8 && !bt_stamps_ && bt_group_cfg(P.m, ldt_, groups_knob(P)).nt == 256;
    // The pivot role with replicated reduced costs (btr_kernels.hip): 512-thread workgroups that fill a CU each, so the launch must own
    // the device — taken only if nobody else holds a loop slot right now (no waiting: busy means the shared 16 x 128 shape as before)
    const BtGroupCfg gc0 = bt_group_cfg(P.m, ldt_, groups_knob(P));
    const bool rep_ok = lag && K == 8 && loop_rep_ && loop_g_ != 8 && gc0.groups == 8 && gc0.nt == 256 && gc0.ri == 1 && bt_loop_rep_supported(P.m, ldt_);
    struct RepHold {
        int dev; bool held;
        RepHold(int d, bool want) : dev(d), held(want && Engine::loop_try_acquire_all(d)) {}
        ~RepHold() { if (held) Engine::loop_release(dev, 4, 0); }
    } rep_hold(device_, rep_ok);
    LoopSlot loop_slot(device_, small_loop ? 1 : 4, lag && !rep_hold.held);
    const int loop_xcd = (small_loop && loop_slot.slot > 0) ? 2 * loop_slot.slot : 0;
    bt_layout(P, tiled_plan);
    // persistent loop kernel: DevState::tsel2 hands the buffer that holds the tableau from launch to launch
    hs.tsel2[0] = hs.tsel2[1] = tcur_; hs.kdone2[0] = hs.kdone2[1] = 0; hs.loop_blocks = 0;
    sync_state_to_device();
    if (bt_stamps_) {   // diagnostic build of the block kernel: cycle sums per wave and pivot segment
        const size_t nb = (size_t)(16 * kBtStampSegs + 8) * sizeof(unsigned long long);
        if (!w.stamps) {
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&w.stamps), nb));
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.stamps_host), nb, hipHostMallocDefault));
        }
        HIP_TRY(hipMemsetAsync(w.stamps, 0, nb, stream_));
    }
    const double t_loop0 = now_s();
    int ret = GOMILP_OK;
    // persistent loop kernel: one launch per chunk of loop_chunk_ pivots (the host learns the buffer choice from the state it
    // reads anyway)
    const int64_t blocks_per_chunk = lag ? std::max<int64_t>(4, loop_chunk_ / K) : std::max<int64_t>(1, chunk_ / K);
    const bool sampling = sample_events_ > 0;
    int64_t block_no = 0;
    // launch parity: a launch reads the counter bases / buffer choice the PREVIOUS launch of this context wrote (whichever loop
    // that was in), so the count runs over the life of the context
    int64_t &launch_no = w.loop_launches;
    // Chunks are pipelined: chunk c+1 is enqueued BEFORE the host waits for the state of chunk c, so the GPU never idles
    // for a host round trip; the price is one chunk of no-op launches after the device has set `done`.
    if (!w.pipe_state[0]) {
        for (int t = 0; t < 2; t++) {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.pipe_state[t]), sizeof(DevState), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&w.pipe_ev[t], hipEventDisableTiming));
        }
    }
    struct ChunkInfo { int64_t before_pred; size_t samp0, nsamp; int64_t nblocks; int par; };
    int last_slot_enqueued = 0;
    bool exact_pending = false;   // the next launch starts with a pivot that decides on the r / x_B just refreshed (ST_NEED_EXACT)
    int forced_q_pending = -1, forced_p_pending = -1;   // ... or with the pivot the host decided on the fresh solves (exact_step)
    int64_t seg0 = 0;             // pivots when the running segment started
    ChunkInfo info[2];
    size_t samp_total = 0;
    auto enqueue_chunk = [&](int slot, int64_t before_pred, int64_t nblocks) -> int {
        info[slot].before_pred = before_pred;
        info[slot].samp0 = samp_total;
        info[slot].nsamp = 0;
        info[slot].nblocks = nblocks;
        info[slot].par = (int)(launch_no & 1);
        last_slot_enqueued = slot;
        if (lag) {   // the whole chunk is one launch
            BTArgs ai = make_bt_args(P, phase, tol, nn, K);
            ai.loop = 1; ai.nblocks = (int)nblocks; ai.par = (int)(launch_no & 1);
            // 2049..4096 rows: 16 pivot workgroups of 256 threads (one wave per SIMD: the per-pivot chain of the 2048-row shape)
            // instead of 8 of 512 — inside the loop kernel only; set-up pivots and the batched schedule keep bt_group_cfg's shape
            if (ai.groups == 8 && ai.group_nt == 512 && ai.group_ri == 1 && loop_g_ != 8 && (K == 16 || K == 12)) { ai.groups = 16; ai.group_nt = 256; }
            const bool k16_small = ai.groups == 8 && ai.group_nt == 256 && ai.group_ri == 1 && loop_g_ != 8 && K == 16;
            if (k16_small) ai.groups = 16;   // (16 x 256 threads, blocks of 16: the instance of the 4096-row class)
            // 1025..2048 rows: 16 x 128 threads (two waves per workgroup: a cheaper workgroup stage in front of every exchange; measured
            // 11.47 ms against 11.85 ms per solve of the metric LP for 8 x 256)
            const bool rep = rep_hold.held && ai.groups == 8 && ai.group_nt == 256 && ai.group_ri == 1;
            if (rep) { ai.groups = 16; ai.group_nt = bt_loop_rep_threads(P.m, ldt_); }
            else if (ai.groups == 8 && ai.group_nt == 256 && ai.group_ri == 1 && loop_g_ != 8 && K == 8 && !bt_stamps_) {
                ai.groups = 16; ai.group_nt = 128;   // (8 x 128 threads measured the same at 1024 rows: 4.74 against 4.75 ms per loop of C2)
            }
            ai.Tbuf[0] = w.T[0]; ai.Tbuf[1] = w.T[1];
            ai.xcd = loop_xcd; ai.upd_cap = (int)loop_upd_; ai.poll_delay = (int)poll_delay_;
            // beyond 2048 rows the tableau pair (268 MB at 4096 x 4096) no longer fits the Infinity Cache: all 240 update workgroups
            // stream a block at the full HBM rate in 51 of its 84 us and the pivot workgroups' dependent tableau reads queue behind
            // them; 112 spread the same bytes over ~60 us (measured per solve of C4: 240: 50.3 ms, 170: 49.0, 140: 47.8, 110: 47.0, 90: 48.9,
            // 70: 57.9 — update-bound from there)
            if (ai.upd_cap == 0 && ai.groups == 16 && ai.group_nt == 256 && !k16_small) ai.upd_cap = 112;
            ai.exact_once = exact_pending ? 1 : 0; exact_pending = false;
            ai.forced_q = forced_q_pending; ai.forced_p = forced_p_pending; forced_q_pending = forced_p_pending = -1;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (sampling) {
                while (w.sample_ev.size() < (samp_total + 1) * 6) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); w.sample_ev.push_back(ev); }
                e0 = w.sample_ev[samp_total * 6]; e1 = w.sample_ev[samp_total * 6 + 1];
                samp_total++; info[slot].nsamp++;
            }
            // workgroups: the 8 G pivot blocks + update workgroups for 256 KB of tableau each, at most one per CU (measured at
            // 2048 x 2048: 128 workgroups 12.0 ms per solve, 256: 12.4, 512: 13.6; at 4096 x 4096: 256 64.3 ms, 128 68.4 — and a
            // grid beyond one 512-thread workgroup per CU is not resident as a whole: the launch gives up after its bounded waits)
            int grid = (int)std::min<int64_t>(ncu_, std::max<int64_t>(64, (int64_t)P.m * ldt_ * 8 / (256 << 10)));
            if (loop_grid_ > 0) grid = (int)loop_grid_;
            if (rep) launch_bt_loop_rep(ai, grid, stream_, e0, e1);
            else launch_bt_loop(ai, grid, stream_, e0, e1);
            launch_no++; block_no += nblocks; launches_++;
            HIP_TRY(hipMemcpyAsync(w.pipe_state[slot], w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(hipEventRecord(w.pipe_ev[slot], stream_));
            return GOMILP_OK;
        }
        for (int64_t bkk = 0; bkk < nblocks; bkk++, block_no++) {
            int kmax = K;
            if (max_pivots_ > 0) kmax = (int)std::max<int64_t>(1, std::min<int64_t>(K, max_pivots_ - before_pred - bkk * K));
            BTArgs a = make_bt_args(P, phase, tol, nn, kmax);
            a.kmax = K;  // the update kernel is instantiated for the configured block size
            BTArgs ai = a; ai.kmax = kmax;
            ai.exact_once = exact_pending ? 1 : 0; exact_pending = false;
            ai.forced_q = forced_q_pending; ai.forced_p = forced_p_pending; forced_q_pending = forced_p_pending = -1;
            const bool sample = sampling && (block_no % std::max<int64_t>(1, sample_events_ / K) == 0);
            hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
            if (sample) {
                while (w.sample_ev.size() < (samp_total + 1) * 6) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); w.sample_ev.push_back(ev); }
                for (int k = 0; k < 4; k++) e[k] = w.sample_ev[samp_total * 6 + k];
                samp_total++; info[slot].nsamp++;
            }
            launch_bt_inner(ai, stream_, e[0], e[1]);
            launch_bt_update(a, stream_, e[2], e[3]);
            launches_ += 2;
        }
        HIP_TRY(hipMemcpyAsync(w.pipe_state[slot], w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipEventRecord(w.pipe_ev[slot], stream_));
        return GOMILP_OK;
    };
    int cur = 0, last_par = 0;
    int64_t pred = 0;   // pivots if every enqueued block ran in full
  restart_segment:
    // small relaxations (B&B children) usually finish Phase I within a handful of pivots: their first chunk is one block
    const int64_t first_blocks = (P.m <= 1024 && !lag) ? 1 : blocks_per_chunk;
    cur = 0;
    { int rc0 = enqueue_chunk(0, seg0, first_blocks); if (rc0 != GOMILP_OK) return rc0; }
    pred = seg0 + first_blocks * K;
    int64_t seen = seg0;   // pivots at the end of the previous inspected chunk
    for (;;) {
        // keep one chunk in flight behind the one whose state is awaited (not past a pivot budget)
        // (not behind the very first chunk: short loops — B&B children — usually end inside it, and the speculative
        // chunk would be pure no-op launches)
        const bool more = !(max_pivots_ > 0 && pred >= max_pivots_);
        const bool speculate = more && seen > seg0;
        if (speculate) { int rc1 = enqueue_chunk(cur ^ 1, pred, blocks_per_chunk); if (rc1 != GOMILP_OK) return rc1; pred += blocks_per_chunk * K; }
        HIP_TRY(hipEventSynchronize(w.pipe_ev[cur]));
        HIP_TRY(hipGetLastError());
        hs = *w.pipe_state[cur];
        const int64_t executed = hs.pivots - seen;
        const ChunkInfo &ci = info[cur];
        if (st && ci.nsamp && executed == ci.nblocks * K) {  // only chunks made of full blocks are samples
            for (size_t s2 = ci.samp0; s2 < ci.samp0 + ci.nsamp; s2++) {
                float ms0 = 0, ms1 = 0;
                if (hipEventElapsedTime(&ms0, w.sample_ev[s2 * 6], w.sample_ev[s2 * 6 + 1]) != hipSuccess) continue;
                if (lag) {   // one launch = the block kernels and (beside them) the updates of ci.nblocks blocks
                    st->pivot_kernel_seconds[0] += ms0 * 1e-3;
                    st->pivot_kernel_seconds[1] += (double)ci.nblocks;
                    st->pivot_kernel_seconds[3] += (double)(K * ci.nblocks);
                    continue;
                }
                if (hipEventElapsedTime(&ms1, w.sample_ev[s2 * 6 + 2], w.sample_ev[s2 * 6 + 3]) != hipSuccess) continue;
                st->pivot_kernel_seconds[0] += ms0 * 1e-3;  // inner kernel: K pivots
                st->pivot_kernel_seconds[2] += ms1 * 1e-3;  // rank-K update
                st->pivot_kernel_seconds[1] += 1.0;         // sampled blocks
                st->pivot_kernel_seconds[3] += (double)K;   // sampled pivots
            }
        }
        seen = hs.pivots;
        last_par = ci.par;
        if (!hs.done) {
            if (max_pivots_ > 0 && hs.pivots >= max_pivots_) { ret = GOMILP_ERR_UNSUPPORTED; break; }
            if (!more) { ret = GOMILP_ERR_UNSUPPORTED; break; }
            if (!speculate) { int rc1 = enqueue_chunk(cur ^ 1, pred, blocks_per_chunk); if (rc1 != GOMILP_OK) return rc1; pred += blocks_per_chunk * K; }
            cur ^= 1;
            continue;
        }
        if (hs.status == ST_OPTIMAL) break;
        if (hs.status == ST_UNBOUNDED) { ret = GOMILP_ERR_UNBOUNDED; break; }
        if (hs.status == ST_BLAND_FAILED) { ret = GOMILP_ERR_BLAND; break; }
        if (hs.status == ST_NEED_EXACT) {
            // The next pivot is degenerate or nearly so.  The reference decides it on the x_B of THIS iteration's fresh LU solve
            // (simplex.go:289 -> :268-277): the rounding noise of that solve picks the leaving row among the zero-level ones and
            // says whether the Bland rule takes over.  Same arithmetic here: gonum-order LU of the current basis, its x_B
            // uploaded, the block kernel decides the first pivot of the next launch on it.
            if (lag) {
                tcur_ = hs.tsel2[(last_par ^ 1) & 1];
                if (launch_no > 0) hipEventSynchronize(w.pipe_ev[(int)(last_slot_enqueued)]);
            } else HIP_TRY(hipEventSynchronize(w.pipe_ev[last_slot_enqueued]));   // launches enqueued ahead of the news (no-ops)
            hs.done = 0; hs.status = ST_RUNNING;
            const DevState keep = hs;   // (final_solve keeps its singular flag in the same block and reads it back)
            int fq = -1, fp = -1;
            const int verdict = exact_step(P, phase, tol, nn, &fq, &fp, st);
            hs = keep;
            hs.done = 0; hs.status = ST_RUNNING; hs.lu_singular = 0;
            hs.tsel2[0] = hs.tsel2[1] = tcur_; hs.kdone2[0] = hs.kdone2[1] = 0; hs.loop_blocks = 0;
            if (verdict < 0) { ret = -verdict; break; }
            if (verdict == 1) { hs.done = 1; hs.status = ST_OPTIMAL; break; }
            if (verdict == 2) { ret = GOMILP_ERR_UNBOUNDED; break; }
            sync_state_to_device();
            forced_q_pending = fq; forced_p_pending = fp;   // (-1, -1: the Bland rule, run by the block kernel on the fresh r / x_B)
            exact_pending = true;
            seg0 = hs.pivots;
            goto restart_segment;
        }
        if (hs.status == ST_XCHG_TIMEOUT && w.xbuf) {   // records of the abandoned exchange must not meet a later launch
            hipStreamSynchronize(stream_);   // (every workgroup of the abandoned launches has left before the records are cleared)
            hipMemsetAsync(w.xbuf, 0, bt_xbuf_doubles() * sizeof(double), stream_);
            w.loop_launches = 0;
            xchg_timeout_ = true;
        }
        if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "run_loop_bt: device status %d after %lld pivots (lag %d, loop_blocks %d)\n", hs.status, (long long)hs.pivots, (int)lag, hs.loop_blocks);
        ret = GOMILP_ERR_DEVICE;
        break;
    }
    if (lag) {
        // the launch whose state was inspected last left the tableau in the buffer it wrote into tsel2 (launches enqueued
        // behind it are no-ops that pass the choice on)
        tcur_ = hs.tsel2[(last_par ^ 1) & 1];
        // a launch enqueued ahead of the news (a no-op, but one that wants all its workgroups resident) has to be through
        // before the next loop kernel of this device may start
        if (launch_no > 0) hipEventSynchronize(w.pipe_ev[(int)(last_slot_enqueued)]);
    }
    // host clock from the first launch to the arrival of the final state: no extra event / sync per loop (the loop is
    // GPU-bound: the host only waits for chunk states)
    if (bt_stamps_ && w.stamps) {
        const size_t nb = (size_t)(16 * kBtStampSegs + 8) * sizeof(unsigned long long);
        HIP_TRY(hipMemcpyAsync(w.stamps_host, w.stamps, nb, hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        const unsigned long long np = w.stamps_host[16 * kBtStampSegs];
        fprintf(stderr, "{\"bt_stamps\": {\"m\": %d, \"nn\": %d, \"phase\": %d, \"pivots\": %llu, \"cycles_per_pivot_by_wave\": [", P.m, nn, phase, np);
        for (int wv = 0; wv < 16; wv++) {
            fprintf(stderr, "%s[", wv ? ", " : "");
            for (int sg = 0; sg < 13; sg++) fprintf(stderr, "%s%.1f", sg ? ", " : "", np ? (double)w.stamps_host[wv * kBtStampSegs + sg] / (double)np : 0.0);
            fprintf(stderr, "]");
        }
        fprintf(stderr, "]}}\n");
    }
    if (st) {
        st->seconds_pivot_loop += now_s() - t_loop0;
        st->bland_steps += hs.bland_steps;
        if (phase == 1) st->pivots_phase1 += hs.pivots; else st->pivots_phase2 += hs.pivots;
    }
    return ret;
}

// Phase I / Phase II on the tableau pipeline.  Called by Engine::solve after the initial (slack) basis is known.
// On return `basic` / `xb` hold the final basis positions and updated x_B; *loop_rc is the Phase-II loop result.
int Engine::solve_tableau(const Problem &P, double tol, std::vector<int32_t> &basic, const std::vector<int32_t> &rho,
                          std::vector<double> &xb, bool feasible, gomilp_lp_stats *st, int *loop_rc,
                          const std::vector<double> *binv_host) {
    Work &w = *w_;
    const int m = P.m, n = P.n;
    int rc;
    std::vector<int32_t> nonbasic;
    auto build_nonbasic = [&](int ncols) {
        std::vector<char> inb(ncols, 0);
        for (int i = 0; i < m; i++) inb[basic[i]] = 1;
        nonbasic.clear();
        for (int j = 0; j < ncols; j++) if (!inb[j]) nonbasic.push_back(j);
    };
    tcur_ = 0; rcur_ = 0; t_tiled_ = false;
    if (!binv_host && (rc = stage_upload(w.rho, rho.data(), (size_t)m * sizeof(int32_t))) != GOMILP_OK) return rc;
    // (x_B itself was uploaded by Engine::solve)
    std::vector<double> art(P.ld, 0.0);  // Phase-I artificial column (simplex.go:533-542)
    bool binv_on_device = gen_binv_dev_;   // (the device search left B^-1 = R^-1 Q^T in place: engine.cpp find_independent_device)
    auto set_up_T = [&](int nn) -> int {  // T = B^-1 A_N
        ldt_ = tab_ld(nn);
        if (!binv_host) {  // slack basis: B^-1 is the permutation rho
            // straight into the layout the block kernels want: no conversion pass before the first pivot
            { int K0; bool tl0; bt_plan(P, &K0, &tl0); t_tiled_ = use_bt_ && tl0; }
            launch_tab_gather(P.dAt, P.ld, m, nn, w.nonbasic, w.rho, w.T[0], ldt_, t_tiled_, stream_);
            launches_++;
            return GOMILP_OK;
        }
        // general basis: B^-1 from the host (engine_general.cpp), T = B^-1 A_N as a device GEMM over the resident columns
        // (At row n holds the Phase-I artificial column); straight into the layout the block kernels want
        { int K0; bool tl0; bt_plan(P, &K0, &tl0); t_tiled_ = use_bt_ && tl0; }
        if (!binv_on_device) {
            HIP_TRY(hipMemcpy2DAsync(w.binv[0], (size_t)P.ld * sizeof(double), binv_host->data(), (size_t)m * sizeof(double), (size_t)m * sizeof(double), m,
                                     hipMemcpyHostToDevice, stream_));
            HIP_TRY(sync_stream());   // binv_host is pageable
            binv_on_device = true;
        }
        launch_tab_gemm(w.binv[0], P.ld, P.dAt, P.ld, m, nn, w.nonbasic, w.T[0], ldt_, t_tiled_, stream_);
        launches_++;
        return GOMILP_OK;
    };
    int nn;
    const double t_st0 = now_s();
    if (!feasible) {
        // ---- Phase I (simplex.go:529-606) ----
        st->phase1_used = 1;
        const int64_t minidx = min_idx(xb.data(), m);
        const bool art_on_device = binv_host && m >= general_min_rows_;   // (large general starts: the same subtractions, element by element, by k_gs_art)
        for (int k = 0; k < m; k++) art[k] = P.hb[k];
        if (!binv_host) {
            for (int i = 0; i < m; i++) { if (i == minidx) continue; art[rho[i]] = -1 * 1.0 + art[rho[i]]; }  // floats.Sub, :536-542
        } else if (!art_on_device) {
            for (int i = 0; i < m; i++) {  // same loop over full columns
                if (i == minidx) continue;
                for (int k = 0; k < m; k++) art[k] = -1 * P.hA[(size_t)k * n + basic[i]] + art[k];
            }
        }
        // tableau over the n+1-m columns that are nonbasic w.r.t. the slack basis (the artificial is the last one),
        // then one forced pivot brings the artificial into position minidx (the basis of simplex.go:551)
        build_nonbasic(n + 1);
        nn = (int)nonbasic.size();
        if ((rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return rc;
        if (art_on_device) {
            launch_gs_art(P.dAt, P.ld, m, w.basic, (int)minidx, P.db, P.dAt + (size_t)n * P.ld, stream_);
            launches_++;
            HIP_TRY(hipMemcpyAsync(art.data(), P.dAt + (size_t)n * P.ld, (size_t)P.ld * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(sync_stream());
        }
        bool art_zero = true;
        for (int k = 0; k < m; k++) if (art[k] != 0) { art_zero = false; break; }
        if (art_zero) { st->wrapped_status = GOMILP_ERR_ZERO_COLUMN; return GOMILP_ERR_PHASE1_WRAPPED; }
        if (!art_on_device && (rc = stage_upload(P.dAt + (size_t)n * P.ld, art.data(), (size_t)P.ld * sizeof(double))) != GOMILP_OK) return rc;
        const double t_st1 = now_s();
        if ((rc = set_up_T(nn)) != GOMILP_OK) return rc;
        if (binv_host && GOMILP_DBG_ENV("GOMILP_DEBUG_GS")) { (void)sync_stream(); fprintf(stderr, "phase I set-up: artificial column + lists %.2f ms, B^-1 upload + T = B^-1 A_N %.2f ms\n", 1e3 * (t_st1 - t_st0), 1e3 * (now_s() - t_st1)); }
        const int qa = nn - 1;  // position of the artificial
        if (!use_bt_) {   // the single-kernel pipeline's forced pivot takes the column from dvec; the block kernel reads T itself
            launch_tab_column(w.T[0], ldt_, m, qa, w.xb, w.dvec, w.move, t_tiled_, stream_);
            launches_++;
        }
        double dp = 0;  // pivot element of the forced pivot: (B^-1 a_art)[minidx]
        if (!binv_host) dp = art[rho[minidx]];
        else if (!use_bt_) {   // (only the single-kernel pipeline's forced pivot takes it from the host)
            std::vector<double> brow(m);
            if (gen_binv_dev_) { HIP_TRY(hipMemcpyAsync(brow.data(), w.binv[0] + (size_t)minidx * P.ld, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_)); HIP_TRY(sync_stream()); }
            else for (int i = 0; i < m; i++) brow[i] = (*binv_host)[(size_t)minidx * m + i];
            for (int i = 0; i < m; i++) dp += brow[i] * art[i];
        }
        const int slack = basic[minidx];
        // the register-resident block kernel exchanges the two list entries itself (uncounted), which saves the second
        // upload of the lists when no re-sort follows
        const bool swap_on_device = use_bt_ && t_tiled_;
        if (use_bt_) rc = bt_forced_pivot(P, 1, 1e-10, nn, qa, (int)minidx, swap_on_device ? 2 : 1);
        else rc = tab_forced_pivot(P, 1, 1e-10, nn, qa, n, 0.0, (int)minidx, dp, xb[minidx], slack, 4, 0);
        if (rc != GOMILP_OK) return rc;
        basic[minidx] = n;
        nonbasic[qa] = slack;  // slack basis: ascending order is preserved (every structural id < slack id < n)
        bool lists_current = swap_on_device;
        {
            // general basis: the replaced variable may belong earlier in the ascending list of simplex.go:174-184
            std::vector<int32_t> asc = nonbasic;
            std::sort(asc.begin(), asc.end());
            if (asc != nonbasic) {
                std::vector<int32_t> pos_of(n + 1, -1), srcpos(nn);
                for (int jp = 0; jp < nn; jp++) pos_of[nonbasic[jp]] = jp;
                for (int jp = 0; jp < nn; jp++) srcpos[jp] = pos_of[asc[jp]];
                if ((rc = stage_upload(w.srcpos, srcpos.data(), (size_t)nn * sizeof(int32_t))) != GOMILP_OK) return rc;
                launch_tab_permute_cols(w.T[tcur_], ldt_, w.T[tcur_ ^ 1], ldt_, m, nn, w.srcpos, t_tiled_, stream_);
                launches_++;
                tcur_ ^= 1;
                nonbasic = asc;
                lists_current = false;
            }
        }
        if (!lists_current && (rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return rc;
        launch_tab_r(w.T[tcur_], ldt_, m, nn, P.dc1, w.basic, w.nonbasic, w.tscratch, w.R[rcur_], t_tiled_, stream_);
        launches_ += 2;
        // the Phase-I starting vertex must be feasible (initializeFromBasic inside the recursive call panics otherwise,
        // simplex.go:155-158): its x_B is copied out here and inspected after the loop's first host wait — a violation
        // discards whatever the loop did
        HIP_TRY(hipMemcpyAsync(w.h_chk, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        rc = use_bt_ ? run_loop_bt(P, 1, 1e-10, nn, st) : run_loop_tab(P, 1, 1e-10, nn, st);
        if (rc == GOMILP_ERR_DEVICE) return rc;
        for (int i = 0; i < m; i++) if (w.h_chk[i] < -1e-13) return GOMILP_ERR_PANIC;
        if (rc != GOMILP_OK) { st->wrapped_status = rc; return GOMILP_ERR_PHASE1_WRAPPED; }  // :557-559
        const size_t gap = (size_t)(w.nonbasic - w.basic);   // the two lists share one device block: one copy
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (gap + (size_t)nn) * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        int added = -1;
        for (int i = 0; i < m; i++) { basic[i] = w.h_idx[i]; xb[i] = w.h_vec[i]; if (basic[i] == n) added = i; }
        for (int jp = 0; jp < nn; jp++) nonbasic[jp] = w.h_idx[gap + jp];
        double xart = added >= 0 ? xb[added] : 0.0;
        if (added >= 0 && fabs(xart) > 1e-13 && fabs(xart) < 1e-11) {
            std::vector<double> xe;
            bool sing = false;
            if ((rc = final_solve(P, n + 1, xe, &sing)) != GOMILP_OK) return rc;
            if (!sing) xart = xe[added];
        }
        if (fabs(xart) > 1e-12) return GOMILP_ERR_INFEASIBLE;  // phaseIZeroTol, :563-565
        if (added >= 0) {
            // :581-606 exchange the zero-level artificial for the first nonbasic variable (ascending id) that works
            std::vector<std::pair<int32_t, int>> cand;  // (variable id, position)
            for (int jp = 0; jp < nn; jp++) if (nonbasic[jp] < n) cand.emplace_back(nonbasic[jp], jp);
            std::sort(cand.begin(), cand.end());
            bool exchanged = false;
            // pivot elements T[added][jp] and column maxima of every candidate in one pass (tscratch holds 64 rows of ldt)
            launch_tab_row_colmax(w.T[tcur_], ldt_, m, nn, added, w.tscratch, t_tiled_, stream_);
            launches_++;
            std::vector<double> rowmax((size_t)2 * ldt_);
            HIP_TRY(hipMemcpyAsync(rowmax.data(), w.tscratch, rowmax.size() * sizeof(double), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(sync_stream());
            for (auto &cv : cand) {
                const int jp = cv.second;
                if (!(fabs(rowmax[jp]) > 1e-9 * std::max(1.0, rowmax[(size_t)ldt_ + jp]))) continue;   // same test as below
                launch_tab_column(w.T[tcur_], ldt_, m, jp, w.xb, w.dvec, w.move, t_tiled_, stream_);
                launches_++;
                HIP_TRY(hipMemcpyAsync(w.h_vec, w.dvec, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(sync_stream());
                double dmax = 0;
                for (int i = 0; i < m; i++) dmax = std::max(dmax, fabs(w.h_vec[i]));
                const double dpv = w.h_vec[added];
                if (!(fabs(dpv) > 1e-9 * std::max(1.0, dmax))) continue;
                const double theta = xb[added] / dpv;
                bool feas = true;
                for (int i = 0; i < m && feas; i++) {
                    const double v = (i == added) ? theta : xb[i] - theta * w.h_vec[i];
                    if (v < -1e-13) feas = false;
                }
                if (!feas) continue;
                if (use_bt_) rc = bt_forced_pivot(P, 1, 1e-10, nn, jp, added, 1);
                else rc = tab_forced_pivot(P, 1, 1e-10, nn, jp, cv.first, 0.0, added, dpv, xb[added], n, 4, 0);
                if (rc != GOMILP_OK) return rc;
                basic[added] = cv.first;
                nonbasic[jp] = n;
                exchanged = true;
                st->art_exchanges++;
                break;
            }
            if (!exchanged) return GOMILP_ERR_INFEASIBLE;  // :606
            // (the host copy of x_B is refreshed by the epilogue of Engine::solve; nothing reads it before)
        }
        // ---- Phase I -> Phase II: nonbasic list rebuilt in ascending order (simplex.go:174-184), T columns follow
        std::vector<int32_t> old_nb = nonbasic;
        build_nonbasic(n);
        const int nn2 = (int)nonbasic.size();
        std::vector<int32_t> pos_of(n + 1, -1), srcpos(nn2);
        for (int jp = 0; jp < nn; jp++) pos_of[old_nb[jp]] = jp;
        for (int jp = 0; jp < nn2; jp++) srcpos[jp] = pos_of[nonbasic[jp]];
        if ((rc = stage_upload(w.srcpos, srcpos.data(), (size_t)nn2 * sizeof(int32_t))) != GOMILP_OK) return rc;
        const int ldt2 = tab_ld(nn2);
        launch_tab_permute_cols(w.T[tcur_], ldt_, w.T[tcur_ ^ 1], ldt2, m, nn2, w.srcpos, t_tiled_, stream_);
        launches_++;
        tcur_ ^= 1;
        ldt_ = ldt2;
        nn = nn2;
        if ((rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return rc;
    } else {
        build_nonbasic(n);
        nn = (int)nonbasic.size();
        if ((rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return rc;
        if ((rc = set_up_T(nn)) != GOMILP_OK) return rc;
    }
    // ---- Phase II ----
    launch_tab_r(w.T[tcur_], ldt_, m, nn, P.dc, w.basic, w.nonbasic, w.tscratch, w.R[rcur_], t_tiled_, stream_);
    launches_ += 2;
    *loop_rc = use_bt_ ? run_loop_bt(P, 2, tol, nn, st) : run_loop_tab(P, 2, tol, nn, st);
    // gonum's condition guard on the basis the loop ends with (mat/lu.go:321 through simplex.go:236-239: the duals' solve of the
    // iteration that finds the optimum reports mat.Condition when cond > 1e16 and the loop leaves with the current point): exact
    // kappa_1 from the tableau for slack-basis starts of any size (bases of up to 64 rows have the pivot-by-pivot replay instead)
    if (*loop_rc == GOMILP_OK && cond_guard_ && !binv_host && m > 64) {
        double k1 = 0, kinf = 0;
        if ((rc = cond_check(P, nn, &k1, &kinf)) != GOMILP_OK) return rc;
        st->cond1_final = k1; st->condinf_final = kinf;
        // kappa_inf: the x_B solve behind the last pivot (simplex.go:289-292, CondNorm = MaxRowSum) ends the loop with mat.Condition too
        if (k1 > 1e16 || k1 != k1 || kinf > 1e16 || kinf != kinf) *loop_rc = GOMILP_ERR_CONDITION;
    }
    return GOMILP_OK;
}

// exact kappa_1 / kappa_inf of the current basis from the resident tableau (tableau_kernels.hip: launch_cond_check); 0 when the work
// buffers are too small for the sums (never for the shapes the tableau pipelines take)
int Engine::cond_check(const Problem &P, int nn, double *k1, double *kinf) {
    Work &w = *w_;
    *k1 = *kinf = 0.0;
    const size_t need = (size_t)ldt_ + 3 * (size_t)P.ld + 4;
    if (need > (size_t)64 * w.cap_ldt) return GOMILP_OK;
    launch_cond_check(w.T[tcur_], ldt_, P.m, nn, w.nonbasic, w.basic, P.dAt, P.ld, P.n - P.m, P.n, w.tscratch, t_tiled_, stream_);
    launches_ += 3;
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.tscratch + (size_t)ldt_ + 3 * (size_t)P.ld, 4 * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    const double b1 = w.h_vec[0], binf = w.h_vec[2];
    *k1 = b1 * w.h_vec[1];
    *kinf = binf * w.h_vec[3];
    if (!(*k1 > 1e16) && !(*kinf > 1e16)) return GOMILP_OK;   // (NaN: stays NaN — mat.Cond propagates it)
    // Beyond the threshold (rare) the verdict is taken on a FRESH inverse of the basis on the host — on badly scaled LPs the updated
    // tableau itself has lost digits (seen: 8.6e16 from the tableau where the basis has 1.9e13) — and with the reference's own
    // number: gonum measures cond with Dgecon's Hager / Higham estimate of |B^-1| (lapack/gonum/dgecon.go:26-81,
    // dlacn2.go:24-136), a lower bound of the exact norm within a small factor.
    const int m = P.m;
    const double tk1 = *k1, tkinf = *kinf;
    { int rc = cond_fresh(P, nullptr, k1, kinf); if (rc != GOMILP_OK) return rc; }
    if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "cond_check: m %d tableau kappa_1 %.6g kappa_inf %.6g -> fresh %.6g %.6g\n", m, tk1, tkinf, *k1, *kinf);
    return GOMILP_OK;
}


// kappa_1 / kappa_inf of the current basis (positions in basic_host, or the device list) from a FRESH inverse on the host: exact norms,
// and beyond 1e16 the Hager / Higham estimate gonum's Dgecon computes (a lower bound of the exact norm within a small factor); a singular
// basis: +Inf.  The columns come from the resident At (row n = the Phase-I artificial column)
int Engine::cond_fresh(const Problem &P, const int32_t *basic_host, double *k1, double *kinf) {
    Work &w = *w_;
    const int m = P.m;
    std::vector<int32_t> basic(m);
    if (basic_host) for (int i = 0; i < m; i++) basic[i] = basic_host[i];
    else {
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int i = 0; i < m; i++) basic[i] = w.h_idx[i];
    }
    std::vector<double> cols((size_t)m * P.ld), B((size_t)m * m), inv;
    for (int p = 0; p < m; p++)
        HIP_TRY(hipMemcpyAsync(&cols[(size_t)p * P.ld], P.dAt + (size_t)basic[p] * P.ld, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    for (int i = 0; i < m; i++) for (int p = 0; p < m; p++) B[(size_t)i * m + p] = cols[(size_t)p * P.ld + i];
    if (m <= kGonumCondMax) {   // the reference's own estimates, bit for bit (gonum_cond.cpp): cond of ab^T for the duals' solve, of ab for x_B / computeMove
        bool dz = false;
        gonum_lu_cond(B.data(), m, m, true, k1, &dz);
        if (dz) *k1 = std::numeric_limits<double>::infinity();
        gonum_lu_cond(B.data(), m, m, false, kinf, &dz);
        if (dz) *kinf = std::numeric_limits<double>::infinity();
        return GOMILP_OK;
    }
    if (!general_invert(B, m, inv)) { *k1 = *kinf = std::numeric_limits<double>::infinity(); return GOMILP_OK; }
    double n1 = 0, ninf = 0, i1 = 0, iinf = 0;
    {
        std::vector<double> cb(m, 0.0), ci(m, 0.0);
        for (int i = 0; i < m; i++) {
            double rb = 0, ri = 0;
            for (int p = 0; p < m; p++) { const double vb = fabs(B[(size_t)i * m + p]), vi = fabs(inv[(size_t)i * m + p]); rb += vb; ri += vi; cb[p] += vb; ci[p] += vi; }
            ninf = std::max(ninf, rb); iinf = std::max(iinf, ri);
        }
        for (int p = 0; p < m; p++) { n1 = std::max(n1, cb[p]); i1 = std::max(i1, ci[p]); }
    }
    *k1 = n1 * i1; *kinf = ninf * iinf;
    if (*k1 > 1e16 || *kinf > 1e16) {
        *k1 = n1 * inverse_norm1_estimate(inv, m, false);
        *kinf = ninf * inverse_norm1_estimate(inv, m, true);
    }
    return GOMILP_OK;
}

}  // namespace gomilp
