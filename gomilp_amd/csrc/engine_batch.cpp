// Host side of the device-batched wave (engine_batch.hpp): buffers for `count` relaxations, the fixed launch schedule,
// and the hand-over of finished relaxations while the others keep pivoting.
#include "engine_batch.hpp"

#include <string.h>

#include <algorithm>
#include <limits>
#include <chrono>

namespace gomilp {

namespace {
double bnow() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}
#define B_TRY(expr)                                     \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return GOMILP_ERR_DEVICE; \
    } while (0)
constexpr int kRing = 2;          // snapshots in flight
constexpr int kMaxSteps = 4096;   // supersteps per wave (slots of the active counter; the last slot serves the prologue)
constexpr int kBlockK = 16;       // rows of U / V per relaxation: pivots per block are 8 (k_bt_inner2_batch) or 16 (k_bt_innerG_batch), bt_batch_k()
template <typename T>
hipError_t bmalloc(T **p, size_t count) { return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T)); }
template <typename T>
hipError_t bhost(T **p, size_t count) { return hipHostMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T), hipHostMallocDefault); }
}  // namespace

WarmEntry::~WarmEntry() {
    for (void *p : {(void *)T, (void *)xb, (void *)basic, (void *)nonbasic, (void *)posvar}) if (p) hipFree(p);
}
std::shared_ptr<WarmEntry> WarmStore::find(int64_t tag) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = by_tag_.find(tag);
    return it == by_tag_.end() ? nullptr : it->second;
}
std::shared_ptr<WarmEntry> WarmStore::acquire(int m4, int ldt, int m, int nn, int n) {
    const size_t need_t = (size_t)m4 * ldt;
    {
        std::lock_guard<std::mutex> g(mu_);
        for (size_t i = 0; i < free_.size(); i++) {
            WarmEntry &e = *free_[i];
            if (e.cap_t >= need_t && e.cap_m >= (size_t)m && e.cap_nn >= (size_t)nn && e.cap_n >= (size_t)n) {
                auto r = free_[i];
                free_[i] = free_.back();
                free_.pop_back();
                return r;
            }
        }
    }
    std::shared_ptr<WarmEntry> e(new WarmEntry);
    // head-room: the children of this node are one row taller, theirs two, ...: a recycled entry should fit a few levels deeper
    e->cap_t = (size_t)(m4 + 32) * ldt; e->cap_m = (size_t)m + 32; e->cap_nn = (size_t)nn + 8; e->cap_n = (size_t)n + 32;
    if (bmalloc(&e->T, e->cap_t) != hipSuccess || bmalloc(&e->xb, e->cap_m) != hipSuccess || bmalloc(&e->basic, e->cap_m) != hipSuccess ||
        bmalloc(&e->nonbasic, e->cap_nn) != hipSuccess || bmalloc(&e->posvar, e->cap_n) != hipSuccess) return nullptr;
    return e;
}
void WarmStore::put(int64_t tag, std::shared_ptr<WarmEntry> e) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = by_tag_.find(tag);
    if (it != by_tag_.end()) { free_.push_back(it->second); by_tag_.erase(it); }
    by_tag_[tag] = std::move(e);
}
void WarmStore::release(int64_t tag) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = by_tag_.find(tag);
    if (it == by_tag_.end()) return;
    if (free_.size() < 4096) free_.push_back(it->second);
    by_tag_.erase(it);
}
void WarmStore::clear() {
    std::lock_guard<std::mutex> g(mu_);
    by_tag_.clear();
    free_.clear();
}
size_t WarmStore::size() {
    std::lock_guard<std::mutex> g(mu_);
    return by_tag_.size();
}

struct BatchEngine::Buf {
    int cap_lp = 0, cap_m4 = 0, cap_ldt = 0, cap_ldu = 0, cap_n = 0;
    int64_t cap_k = 0;
    BatchLP *d_lps = nullptr, *h_lps = nullptr;            // device array, pinned build area
    BatchOut *d_out = nullptr;                             // per relaxation: what the host needs to see (written by the control kernel)
    BatchOut *h_snap[kRing] = {nullptr, nullptr};          // pinned snapshots of d_out, one per superstep in flight
    int *h_active[kRing] = {nullptr, nullptr};
    hipEvent_t ev[kRing] = {nullptr, nullptr};
    double *d_T = nullptr, *d_R = nullptr, *d_xb = nullptr, *d_U = nullptr, *d_V = nullptr, *d_scratch = nullptr, *d_art = nullptr;
    double *d_xbuf = nullptr;   // exchange records of the multi-workgroup block kernel, one set per relaxation slot
    void *d_res = nullptr;      // register-resident kernel (res_kernels.hip): records + candidate rows per launch SLOT, zeroed once: every exchange of
    uint64_t res_launches = 0;  // every launch carries its own sequence numbers (launch number * 2^20 + exchange)
    int32_t *d_basic = nullptr, *d_nonbasic = nullptr, *d_srcpos = nullptr;
    DevState *d_st = nullptr;
    int32_t *d_var = nullptr, *h_var = nullptr;
    double *d_sr = nullptr, *h_sr = nullptr;               // sign | rhs
    struct Rho { uint64_t serial; int32_t *d; };
    std::vector<Rho> rho;   // unit-column rows of the roots seen (device), keyed by the upload's serial number
    int *d_active = nullptr;
    int *d_ids[2] = {nullptr, nullptr};   // active lists, double buffered by superstep parity
    int32_t *h_basic = nullptr;                            // pinned result arenas
    double *h_xb = nullptr;
    std::vector<hipEvent_t> lp_ev;
    std::vector<hipEvent_t> samp_ev;   // 4 per sampled block: inner start / stop, update start / stop

    void free_lp_buffers() {
        for (void *p : {(void *)d_lps, (void *)d_T, (void *)d_R, (void *)d_xb, (void *)d_U, (void *)d_V, (void *)d_scratch, (void *)d_art,
                        (void *)d_basic, (void *)d_nonbasic, (void *)d_srcpos, (void *)d_st, (void *)d_ids[0], (void *)d_ids[1], (void *)d_out, (void *)d_xbuf})
            if (p) hipFree(p);
        d_lps = nullptr; d_T = d_R = d_xb = d_U = d_V = d_scratch = d_art = d_xbuf = nullptr; d_basic = d_nonbasic = d_srcpos = nullptr; d_st = nullptr; d_ids[0] = d_ids[1] = nullptr; d_out = nullptr;
        for (void *p : {(void *)h_lps, (void *)h_snap[0], (void *)h_snap[1], (void *)h_basic, (void *)h_xb})
            if (p) hipHostFree(p);
        h_lps = nullptr; h_snap[0] = h_snap[1] = nullptr; h_basic = nullptr; h_xb = nullptr;
        cap_lp = cap_m4 = cap_ldt = cap_ldu = cap_n = 0;
    }
    void free_all() {
        free_lp_buffers();
        for (void *p : {(void *)d_var, (void *)d_sr, (void *)d_active, d_res}) if (p) hipFree(p);
        d_res = nullptr;
        for (auto &r : rho) hipFree(r.d);
        rho.clear();
        for (void *p : {(void *)h_var, (void *)h_sr, (void *)h_active[0], (void *)h_active[1]}) if (p) hipHostFree(p);
        d_var = nullptr; d_sr = nullptr; d_active = nullptr; h_var = nullptr; h_sr = nullptr; h_active[0] = h_active[1] = nullptr;
        for (auto &e : ev) { if (e) hipEventDestroy(e); e = nullptr; }
        for (auto &e : lp_ev) hipEventDestroy(e);
        lp_ev.clear();
        for (auto &e : samp_ev) hipEventDestroy(e);
        samp_ev.clear();
        cap_k = 0;
    }
};

BatchEngine::BatchEngine(int device) : device_(device), b_(new Buf) {}

BatchEngine::~BatchEngine() {
    hipSetDevice(device_);
    if (stream_hi_) hipStreamSynchronize(stream_hi_);
    if (stream_lo_) hipStreamSynchronize(stream_lo_);
    if (copy_stream_) hipStreamSynchronize(copy_stream_);
    b_->free_all();
    delete b_;
    if (stream_hi_) hipStreamDestroy(stream_hi_);
    if (stream_lo_) hipStreamDestroy(stream_lo_);
    if (copy_stream_) hipStreamDestroy(copy_stream_);
}

bool BatchEngine::eligible(const Engine::RootView &R, int K_max, bool phase1) const {
    if (R.verify_status != GOMILP_OK || !(R.unit_basis || R.gen)) return false;
    const int m = R.m + K_max, n = R.n + K_max;
    if (m >= n || !((n - m) < 2 * m)) return false;              // the tableau formulation (engine.cpp: use_tab)
    if (n + 2 > 7700) return false;                              // k_b_ctrl keeps two int lists of n in LDS next to ~3 KB of static arrays (64 KB limit)
    if (cond_guard_ && m <= 64) return false;                    // bases of up to 64 rows: the pivot-by-pivot replay of gonum's condition guards runs in Engine::solve only
    const int ldt1 = batch_ldt(n - m + (phase1 ? 1 : 0));
    return bt_batch_supported(m, ldt1);
}

int BatchEngine::ensure(int nlp, int m_max, int n_max, int ldt1, int64_t ktot) {
    B_TRY(hipSetDevice(device_));
    if (!stream_) {
        // a hardware queue of its own: streams of one priority share a small round-robin pool of queues (4 per process), and a
        // worker's final-solve launches that land in the schedule's queue would wait behind a whole superstep every time
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo) {
            B_TRY(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi));
            B_TRY(hipStreamCreateWithPriority(&stream_lo_, hipStreamNonBlocking, lo));
        } else {
            B_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
            B_TRY(hipStreamCreateWithFlags(&stream_lo_, hipStreamNonBlocking));
        }
        stream_hi_ = stream_;
    }
    stream_ = low_priority_ ? stream_lo_ : stream_hi_;   // (both idle between runs: every run ends with a synchronize)
    if (!copy_stream_) B_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
    Buf &b = *b_;
    if (!b.d_active) {
        B_TRY(bmalloc(&b.d_active, kMaxSteps));
        for (int r = 0; r < kRing; r++) {
            B_TRY(bhost(&b.h_active[r], 1));
            B_TRY(hipEventCreateWithFlags(&b.ev[r], hipEventDisableTiming));
        }
    }
    const int m4 = (m_max + 3) & ~3, ldu = (m_max + 1) & ~1;
    if (nlp > b.cap_lp || m4 > b.cap_m4 || ldt1 > b.cap_ldt || ldu > b.cap_ldu || n_max > b.cap_n) {
        B_TRY(hipStreamSynchronize(stream_));
        B_TRY(hipStreamSynchronize(copy_stream_));
        const int clp = std::max(nlp, b.cap_lp), cm4 = std::max(m4 + 64, b.cap_m4), cldt = std::max(ldt1, b.cap_ldt),
                  cldu = std::max(ldu + 64, b.cap_ldu), cn = std::max(n_max + 64, b.cap_n);   // head-room: deeper children add a row each
        b.free_lp_buffers();
        const size_t L = (size_t)clp;
        B_TRY(bmalloc(&b.d_lps, L));
        B_TRY(bmalloc(&b.d_out, L));
        B_TRY(bhost(&b.h_lps, L));
        for (int r = 0; r < kRing; r++) B_TRY(bhost(&b.h_snap[r], L));
        B_TRY(bmalloc(&b.d_T, L * 2 * (size_t)cm4 * cldt));
        B_TRY(bmalloc(&b.d_R, L * cldt)); B_TRY(bmalloc(&b.d_xb, L * cldu)); B_TRY(bmalloc(&b.d_art, L * cldu));
        B_TRY(bmalloc(&b.d_U, L * kBlockK * cldu)); B_TRY(bmalloc(&b.d_V, L * kBlockK * cldt));
        B_TRY(bmalloc(&b.d_scratch, L * 64 * cldt));
        B_TRY(bmalloc(&b.d_xbuf, L * bt_xbuf_doubles()));
        B_TRY(bmalloc(&b.d_basic, L * cldu)); B_TRY(bmalloc(&b.d_nonbasic, L * cldt)); B_TRY(bmalloc(&b.d_srcpos, L * cldt));
        B_TRY(bmalloc(&b.d_st, L));
        B_TRY(bmalloc(&b.d_ids[0], L)); B_TRY(bmalloc(&b.d_ids[1], L));
        B_TRY(bhost(&b.h_basic, L * cldu)); B_TRY(bhost(&b.h_xb, L * cldu));
        b.cap_lp = clp; b.cap_m4 = cm4; b.cap_ldt = cldt; b.cap_ldu = cldu; b.cap_n = cn;
        while ((int)b.lp_ev.size() < clp) { hipEvent_t e; B_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); b.lp_ev.push_back(e); }
    }
    if (ktot > b.cap_k) {
        B_TRY(hipStreamSynchronize(stream_));
        for (void *p : {(void *)b.d_var, (void *)b.d_sr}) if (p) hipFree(p);
        for (void *p : {(void *)b.h_var, (void *)b.h_sr}) if (p) hipHostFree(p);
        b.d_var = nullptr; b.d_sr = nullptr; b.h_var = nullptr; b.h_sr = nullptr;
        const int64_t ck = std::max<int64_t>(2 * ktot, 1024);
        B_TRY(bmalloc(&b.d_var, (size_t)ck)); B_TRY(bmalloc(&b.d_sr, (size_t)2 * ck));
        B_TRY(bhost(&b.h_var, (size_t)ck)); B_TRY(bhost(&b.h_sr, (size_t)2 * ck));
        b.cap_k = ck;
    }
    return GOMILP_OK;
}

int BatchEngine::run_roots(const Engine::RootView *const *roots, int nroots, const int32_t *root_of, int64_t count, const int64_t *koff,
                           const int32_t *var, const double *sign, const double *rhs, double tol, const DoneFn &on_done, Stats *stats,
                           const WarmSpec *warm) {
    const double t0 = bnow();
    Stats local;
    Stats &S = stats ? *stats : local;
    S = Stats();
    if (count <= 0) return GOMILP_OK;
    if (count > 65535 || nroots < 1) return GOMILP_ERR_UNSUPPORTED;
    const int nlp = (int)count;
    int m_max = 0, n_max = 0, ldt1 = 0, nn_max = 0;   // (nn_max: nonbasic columns of the widest tableau, the Phase-I artificial included)
    for (int i = 0; i < nlp; i++) {
        const int ri = root_of ? root_of[i] : 0;
        if (ri < 0 || ri >= nroots) return GOMILP_ERR_BAD_SHAPE;
        const Engine::RootView &Ri = *roots[ri];
        const int K = (int)(koff[i + 1] - koff[i]);
        m_max = std::max(m_max, Ri.m + K); n_max = std::max(n_max, Ri.n + K);
        bool p1 = false;   // does this relaxation start infeasible?  (initPosTol; the set-up kernel decides the same way)
        if (Ri.gen && !Ri.unit_basis) p1 = true;   // (a searched basis: feasibility is only known on the device — keep room for the artificial)
        for (double v : Ri.hb) if (v < -1e-13) { p1 = true; break; }
        for (int64_t k = koff[i]; k < koff[i + 1] && !p1; k++) if (rhs[k] < -1e-13) p1 = true;
        ldt1 = std::max(ldt1, batch_ldt(Ri.n - Ri.m + (p1 ? 1 : 0)));   // Phase-I tableau: one column more (the artificial)
        nn_max = std::max(nn_max, Ri.n - Ri.m + (p1 ? 1 : 0));
    }
    const int64_t ktot = koff[nlp] - koff[0];
    int rc = ensure(nlp, m_max, n_max, ldt1, ktot);
    if (rc != GOMILP_OK) return rc;
    Buf &b = *b_;
    // pivots per block and block kernel of this wave's shape class: large relaxations run the 8-workgroup kernel (one XCD each)
    const int kb = bt_batch_k(m_max, ldt1);
    const BtGroupCfg grp = bt_group_cfg(m_max, ldt1, 0);
    B_TRY(hipMemsetAsync(b.d_xbuf, 0, (size_t)nlp * bt_xbuf_doubles() * sizeof(double), stream_));   // no exchange has happened, every counter at 0
    // the persistent loop kernel takes the block steps whenever the active relaxations fit one launch (one workgroup per CU at most)
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_) != hipSuccess || ncu <= 0) ncu = 64;
    const int loop_slots = (loop_ && b_loop_supported(m_max, ldt1)) ? std::max(8, b_loop_slots(ncu) / loop_share_ / 4 * 4) : 0;   // (loop_share_: schedules side by side)
    int loop_launches = 0;
    // the register-resident kernel (res_kernels.hip) takes them when they are few: G workgroups per relaxation on one XCD, at most 8
    // relaxations per launch — half the CUs of every XCD, so that the two schedules of a split wave can both be resident
    const int res_G = (res_ && ncu >= 256 && exact_degenerate_ != 3) ? b_res_groups(m_max, nn_max, ldt1) : 0;
    const int res_slots = res_G ? b_res_max_slots() : 0;
    if (res_slots && !b.d_res) {
        const size_t bytes = (size_t)b_res_max_slots() * b_res_slot_bytes();
        B_TRY(hipMalloc(&b.d_res, bytes));
        B_TRY(hipMemsetAsync(b.d_res, 0, bytes, stream_));
    }
    // ---- root data the kernels read in place + the unit-column rows of each root's slack basis
    std::vector<const int32_t *> rho_of(nroots, nullptr);
    for (int r = 0; r < nroots; r++) {
        const Engine::RootView &R = *roots[r];
        for (auto &e : b.rho) if (e.serial == R.serial) rho_of[r] = e.d;
        if (rho_of[r]) continue;
        if (b.rho.size() >= 64) {   // roots come and go (B&B restarts): forget the old ones
            B_TRY(hipStreamSynchronize(stream_));
            for (auto &e : b.rho) hipFree(e.d);
            b.rho.clear();
            for (int q = 0; q < r; q++) rho_of[q] = nullptr;
            r = -1;
            continue;
        }
        Buf::Rho e{R.serial, nullptr};
        B_TRY(bmalloc(&e.d, (size_t)R.m));
        B_TRY(hipMemcpy(e.d, R.rho0.data(), (size_t)R.m * sizeof(int32_t), hipMemcpyHostToDevice));
        b.rho.push_back(e);
        rho_of[r] = e.d;
    }
    // ---- per-relaxation argument blocks
    std::vector<std::shared_ptr<WarmEntry>> warm_used;
    int nwarm = 0;
    bool slack_only = true;   // every relaxation starts from a slack basis of a root whose A is resident row-major too (virtual tableau below)
    for (int r = 0; r < nroots; r++) if (!roots[r]->unit_basis || roots[r]->gen || !roots[r]->rm) slack_only = false;
    const size_t sT = (size_t)b.cap_m4 * b.cap_ldt;
    for (int64_t k = 0; k < ktot; k++) { b.h_var[k] = var[koff[0] + k]; b.h_sr[k] = sign[koff[0] + k]; b.h_sr[b.cap_k + k] = rhs[koff[0] + k]; }
    // what all relaxations of a root share is filled once per root (an 8192-wide wave: 8192 argument blocks of 0.7 KB — cleared and
    // filled field by field they were ~1 ms of host time in front of the first kernel)
    std::vector<BatchLP> tmpl((size_t)nroots);
    for (int r = 0; r < nroots; r++) {
        BatchLP &lp = tmpl[(size_t)r];
        memset(&lp, 0, sizeof(lp));
        const Engine::RootView &R = *roots[r];
        lp.At0 = R.dAt; lp.c0 = R.dc; lp.b0 = R.db; lp.rho0 = rho_of[r];
        if (R.rm) { lp.A0r = R.rm->dA; lp.lda0r = R.rm->lda; }
        if (!R.unit_basis && R.gen) {   // equality rows: start from the root's searched basis + the branch slacks
            lp.gen = 1; lp.gT0 = R.gen->dT0; lp.gxb0 = R.gen->dxb0; lp.gbasic0 = R.gen->dbasic0; lp.gnonbasic0 = R.gen->dnonbasic0;
            lp.gposvar0 = R.gen->dposvar0; lp.gldt = R.gen->ldt;
        }
        lp.ld0 = R.ld; lp.m0 = R.m; lp.n0 = R.n; lp.cap_ldt = b.cap_ldt;
    }
    for (int i = 0; i < nlp; i++) {
        BatchLP &lp = b.h_lps[i];
        const int K = (int)(koff[i + 1] - koff[i]);
        const int64_t k0 = koff[i] - koff[0];
        const int ri = root_of ? root_of[i] : 0;
        const Engine::RootView &R = *roots[ri];
        lp = tmpl[(size_t)ri];
        lp.var = b.d_var + k0; lp.sign = b.d_sr + k0; lp.rhs = b.d_sr + b.cap_k + k0;
        lp.K = K; lp.m = R.m + K; lp.n = R.n + K;
        lp.ldu = (lp.m + 1) & ~1;
        lp.T[0] = b.d_T + (size_t)(2 * i) * sT; lp.T[1] = b.d_T + (size_t)(2 * i + 1) * sT;
        lp.R = b.d_R + (size_t)i * b.cap_ldt; lp.xb = b.d_xb + (size_t)i * b.cap_ldu; lp.art = b.d_art + (size_t)i * b.cap_ldu;
        lp.U = b.d_U + (size_t)i * kBlockK * b.cap_ldu; lp.V = b.d_V + (size_t)i * kBlockK * b.cap_ldt;
        lp.scratch = b.d_scratch + (size_t)i * 64 * b.cap_ldt;
        lp.basic = b.d_basic + (size_t)i * b.cap_ldu; lp.nonbasic = b.d_nonbasic + (size_t)i * b.cap_ldt; lp.srcpos = b.d_srcpos + (size_t)i * b.cap_ldt;
        lp.st = b.d_st + i;
        lp.tol_user = tol; lp.kblock = kb; lp.stage = BS_HOST;
        if (warm && warm->store && warm->start_warm && warm->parent && warm->parent[i] >= 0 && K >= 1 && R.unit_basis && !lp.gen && kb == 8) {
            std::shared_ptr<WarmEntry> e = warm->store->find(warm->parent[i]);
            // the parent must be this relaxation minus its last branch row, solved on the same root data
            if (e && e->m + 1 == lp.m && e->n + 1 == lp.n && e->K + 1 == K && e->root_serial == R.serial && e->ldt == batch_ldt(lp.n - lp.m)) {
                lp.warm = 1; lp.wm = e->m;
                lp.wT = e->T; lp.wxb = e->xb; lp.wbasic = e->basic; lp.wnonbasic = e->nonbasic; lp.wposvar = e->posvar;
                lp.dual_budget = warm->dual_budget > 0 ? warm->dual_budget : 64;
                warm_used.push_back(e);   // (alive until the wave is through, whatever the caller releases meanwhile)
                nwarm++;
            }
        }
        // degenerate pivots are decided on a fresh gonum-order x_B: such a relaxation is handed to the worker path (ST_NEED_EXACT -> BS_HOST)
        // (3, strict: every relaxation stops in front of its first decision and goes to a worker, whose engine runs the strict mode)
        lp.bt.guard = exact_degenerate_ == 3 ? std::numeric_limits<double>::infinity()
                                              : (exact_degenerate_ == 2 || (exact_degenerate_ == 1 && (lp.m <= 256 || lp.gen || R.scale_span > 1e9))) ? 1e-9 : 0.0;
        lp.bt.fault = fault_;
        lp.bt.cguard = (cond_guard_ && !lp.gen) ? 1e-9 : 0.0;   // a pivot element of rounding-noise size: ST_NEED_EXACT -> BS_HOST, the worker path measures the condition numbers
        lp.bt.xbuf = b.d_xbuf + (size_t)i * bt_xbuf_doubles();
        if (kb == 16) { lp.bt.groups = grp.groups; lp.bt.group_ri = grp.ri; lp.bt.group_nt = grp.nt; }
    }
    // Virtual tableau (device_types.h BatchLP::virt): a WIDE wave of slack-start relaxations runs its set-up pivot and its first block of 8
    // pivots on computed tableau entries; only what is alive behind that block is written out.  Narrow waves go to the persistent kernels
    // with the first superstep and need their tableaus at once.
    const bool virt_on = virt_ && slack_only && nwarm == 0 && kb == 8 && bt_virt_batch_supported(m_max, ldt1) && nlp > std::max(std::max(loop_slots, res_slots), 32);
    if (virt_on) for (int i = 0; i < nlp; i++) b.h_lps[i].virt = 2;
    B_TRY(hipMemsetAsync(b.d_active, 0, kMaxSteps * sizeof(int), stream_));
    B_TRY(hipMemsetAsync(b.d_out, 0, (size_t)nlp * sizeof(BatchOut), stream_));   // stage 0 = not terminal
    B_TRY(hipMemcpyAsync(b.d_lps, b.h_lps, (size_t)nlp * sizeof(BatchLP), hipMemcpyHostToDevice, stream_));
    if (ktot) {
        B_TRY(hipMemcpyAsync(b.d_var, b.h_var, (size_t)ktot * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        B_TRY(hipMemcpyAsync(b.d_sr, b.h_sr, (size_t)ktot * sizeof(double), hipMemcpyHostToDevice, stream_));
        B_TRY(hipMemcpyAsync(b.d_sr + b.cap_k, b.h_sr + b.cap_k, (size_t)ktot * sizeof(double), hipMemcpyHostToDevice, stream_));
    }
    int step = 0;
    int nsamp = 0;
    S.inner_kernel = bt_batch_kernel_name(m_max, ldt1);
    int bound = nlp;   // upper bound of the active relaxations the host knows (from the last snapshot it has seen)
    launch_b_init_ids(b.d_ids[1], b.d_active + (kMaxSteps - 1), nlp, stream_);   // list of "superstep -1": everybody
    auto snapshot = [&](int slot) -> int {
        B_TRY(hipMemcpyAsync(b.h_snap[slot], b.d_out, (size_t)nlp * sizeof(BatchOut), hipMemcpyDeviceToHost, stream_));
        B_TRY(hipMemcpyAsync(b.h_active[slot], b.d_active + step, sizeof(int), hipMemcpyDeviceToHost, stream_));
        B_TRY(hipEventRecord(b.ev[slot], stream_));
        return GOMILP_OK;
    };
    // the blocks of superstep `step` work on the active list the control step of superstep step - 1 left behind
    int last_loop_par = -1;   // parity of the loop launch that ran the blocks of the superstep being enqueued (-1: launch pairs)
    // (fused_setup: the set-up block — its update is folded into the gather of the tableau, batch_kernels.hip k_b_gather mode 2)
    auto blocks = [&](int nb, bool allow_loop, bool fused_setup = false) {
        const int *ids = b.d_ids[(step + 1) & 1];
        const int *cnt = step == 0 ? b.d_active + (kMaxSteps - 1) : b.d_active + (step - 1);
        last_loop_par = -1;
        if (virt_on && step <= 1) {
            // step 0: the set-up pivot (term -> U / V row 8); step 1: the first block, then the tableaus of the survivors — written once, both
            // applied (k_b_gather mode 3) — in front of the control step that may read them
            hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
            if (sampling_) {
                while (b.samp_ev.size() < (size_t)(nsamp + 1) * 4) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; b.samp_ev.push_back(ev); }
                if (b.samp_ev.size() >= (size_t)(nsamp + 1) * 4) { for (int q = 0; q < 4; q++) e[q] = b.samp_ev[(size_t)nsamp * 4 + q]; nsamp++; }
            }
            launch_bt_inner_virt_batch(b.d_lps, ids, cnt, bound, stream_, e[0], e[1]);
            if (e[2]) hipEventRecord(e[2], stream_);
            if (step == 1) launch_b_write_virt(b.d_lps, ids, cnt, b.d_ids[step & 1], b.d_active + step, bound, m_max, ldt1, ncu, stream_);   // (scratch: the list / count this step's control kernel writes afterwards; k_b_gather mode 3 is the same arithmetic through 32 x 32 LDS blocks: 0.8 TB/s)
            if (e[3]) hipEventRecord(e[3], stream_);
            S.launches += step == 1 ? 3 : 1; S.blocks += 1; S.virt_blocks += 1;
            return;
        }
        if (allow_loop && res_slots > 0 && bound <= res_slots && nwarm == 0) {
            hipEvent_t e[2] = {nullptr, nullptr};
            if (sampling_) {
                while (b.samp_ev.size() < (size_t)(nsamp + 1) * 4) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; b.samp_ev.push_back(ev); }
                if (b.samp_ev.size() >= (size_t)(nsamp + 1) * 4) {
                    e[0] = b.samp_ev[(size_t)nsamp * 4]; e[1] = b.samp_ev[(size_t)nsamp * 4 + 1];
                    hipEventRecord(b.samp_ev[(size_t)nsamp * 4 + 2], stream_); hipEventRecord(b.samp_ev[(size_t)nsamp * 4 + 3], stream_);   // (no update launch)
                    nsamp++;
                }
            }
            b.res_launches++;
            launch_b_res(b.d_lps, ids, cnt, bound, res_G, nb, (double)(b.res_launches << 20), b.d_res, stream_, e[0], e[1]);
            S.launches += 1; S.blocks += nb; S.loop_launches += 1; S.res_launches += 1;
            return;   // (the tableau stays in the relaxation's current buffer: the control step needs no buffer choice, last_loop_par = -1)
        }
        if (allow_loop && loop_slots > 0 && bound <= loop_slots && nwarm == 0) {
            hipEvent_t e[2] = {nullptr, nullptr};
            if (sampling_) {
                while (b.samp_ev.size() < (size_t)(nsamp + 1) * 4) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; b.samp_ev.push_back(ev); }
                if (b.samp_ev.size() >= (size_t)(nsamp + 1) * 4) {
                    e[0] = b.samp_ev[(size_t)nsamp * 4]; e[1] = b.samp_ev[(size_t)nsamp * 4 + 1];
                    hipEventRecord(b.samp_ev[(size_t)nsamp * 4 + 2], stream_); hipEventRecord(b.samp_ev[(size_t)nsamp * 4 + 3], stream_);   // (no separate update launch)
                    nsamp++;
                }
            }
            last_loop_par = loop_launches & 1;
            launch_b_loop(b.d_lps, ids, cnt, bound, nb, last_loop_par, stream_, e[0], e[1]);
            loop_launches++;
            S.launches += 1; S.blocks += nb; S.loop_launches += 1;
            return;
        }
        for (int t = 0; t < nb; t++) {
            hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
            if (sampling_) {
                while (b.samp_ev.size() < (size_t)(nsamp + 1) * 4) { hipEvent_t ev; if (hipEventCreate(&ev) != hipSuccess) break; b.samp_ev.push_back(ev); }
                if (b.samp_ev.size() >= (size_t)(nsamp + 1) * 4) { for (int q = 0; q < 4; q++) e[q] = b.samp_ev[(size_t)nsamp * 4 + q]; nsamp++; }
            }
            launch_bt_inner_batch(b.d_lps, ids, cnt, bound, m_max, ldt1, stream_, e[0], e[1], xcd_off_);
            if (nwarm) { launch_bt_inner_dual_batch(b.d_lps, ids, cnt, bound, m_max, ldt1, stream_); S.launches += 1; }   // (stage BS_DUAL only)
            if (fused_setup) {
                if (e[2]) hipEventRecord(e[2], stream_);
                launch_b_gather(b.d_lps, nlp, m_max, ldt1, 2, stream_);
                if (e[3]) hipEventRecord(e[3], stream_);
            }
            else launch_bt_update_batch(b.d_lps, ids, cnt, bound, m_max, ldt1, stream_, e[2], e[3]);
        }
        S.launches += 2 * nb; S.blocks += nb;
    };
    // the control step of superstep `step` visits the relaxations of the previous active list and leaves the next one
    auto control = [&](bool permute) {
        const int *ids = b.d_ids[(step + 1) & 1];
        const int *cnt = step == 0 ? b.d_active + (kMaxSteps - 1) : b.d_active + (step - 1);
        launch_b_ctrl(b.d_lps, ids, cnt, bound, n_max, b.d_out, b.d_ids[step & 1], b.d_active + step, last_loop_par, stream_);
        if (permute) launch_b_permute(b.d_lps, ids, cnt, bound, m_max, ldt1, stream_);
        launch_b_tab_r(b.d_lps, ids, cnt, bound, m_max, ldt1, stream_);
        S.launches += permute ? 5 : 4;
    };
    // ---- prologue: set-up, T, the forced Phase-I pivots, first reduced costs
    launch_b_setup(b.d_lps, nlp, stream_);
    if (nwarm) { launch_b_setup_warm(b.d_lps, nlp, stream_); S.launches += 1; }
    // the tiled rank-8 update (blocks of 8 pivots) has its arithmetic in the gather too: relaxations that start with the forced Phase-I
    // pivot get the pivot's row and column first, the whole tableau once, with the pivot applied, behind the set-up block
    const bool fuse_setup = kb == 8;
    if (!virt_on) launch_b_gather(b.d_lps, nlp, m_max, ldt1, fuse_setup ? 1 : 0, stream_);
    S.launches += virt_on ? 1 : 2;
    S.warm_started = nwarm;
    blocks(1, false, fuse_setup);   // (set-up pivots: one block each, launch pair)
    control(false);
    if ((rc = snapshot(0)) != GOMILP_OK) return rc;
    S.seconds_setup = bnow() - t0;
    // ---- supersteps; finished relaxations are handed over while the others keep pivoting
    std::vector<char> reported(nlp, 0);
    struct Pending { int i; Outcome o; };
    std::vector<Pending> pending;
    auto flush_pending = [&](bool wait) {
        for (size_t k = 0; k < pending.size();) {
            const int i = pending[k].i;
            hipError_t q = wait ? hipEventSynchronize(b.lp_ev[i]) : hipEventQuery(b.lp_ev[i]);
            if (q == hipSuccess) {
                on_done(i, pending[k].o, b.h_basic + (size_t)i * b.cap_ldu, b.h_xb + (size_t)i * b.cap_ldu);
                pending[k] = pending.back();
                pending.pop_back();
            } else k++;
        }
    };
    auto harvest = [&](int slot) -> int {   // look at a completed snapshot
        for (int i = 0; i < nlp; i++) {
            if (reported[i]) continue;
            const BatchOut &lp = b.h_snap[slot][i];
            if (lp.stage != BS_DONE && lp.stage != BS_HOST && lp.stage != BS_COLD) continue;
            const int m_i = b.h_lps[i].m;
            reported[i] = 1;
            Outcome o;
            o.stage = lp.stage; o.status = lp.status; o.wrapped = lp.wrapped; o.phase1_used = lp.phase1_used;
            o.piv1 = lp.piv1; o.piv2 = lp.piv2; o.bland = lp.bland; o.pivd = lp.pivd; o.warm = b.h_lps[i].warm;
            if (lp.stage == BS_DONE && (lp.status == GOMILP_OK || lp.status == GOMILP_ERR_BLAND)) {
                if (lp.status == GOMILP_OK && warm && warm->store && warm->keep && warm->keep[i] && warm->tag && !b.h_lps[i].gen && kb == 8) {
                    // keep the final state for this relaxation's children: device-to-device, behind nothing (the relaxation is terminal)
                    const BatchLP &hl = b.h_lps[i];
                    const int nn_i = hl.n - hl.m, ldt_i = batch_ldt(nn_i), m4_i = (hl.m + 3) & ~3;
                    std::shared_ptr<WarmEntry> e = warm->store->acquire(m4_i, ldt_i, hl.m, nn_i, hl.n);
                    if (e) {
                        e->m = hl.m; e->n = hl.n; e->nn = nn_i; e->ldt = ldt_i; e->K = hl.K;
                        e->root_serial = roots[root_of ? root_of[i] : 0]->serial;
                        const double *Tsrc = (lp.tcur & 1) ? hl.T[1] : hl.T[0];
                        B_TRY(hipMemcpyAsync(e->T, Tsrc, (size_t)m4_i * ldt_i * sizeof(double), hipMemcpyDeviceToDevice, copy_stream_));
                        B_TRY(hipMemcpyAsync(e->xb, hl.xb, (size_t)hl.m * sizeof(double), hipMemcpyDeviceToDevice, copy_stream_));
                        B_TRY(hipMemcpyAsync(e->basic, hl.basic, (size_t)hl.m * sizeof(int32_t), hipMemcpyDeviceToDevice, copy_stream_));
                        B_TRY(hipMemcpyAsync(e->nonbasic, hl.nonbasic, (size_t)nn_i * sizeof(int32_t), hipMemcpyDeviceToDevice, copy_stream_));
                        launch_b_posvar(e->basic, hl.m, e->nonbasic, nn_i, e->posvar, copy_stream_);
                        warm->store->put(warm->tag[i], e);
                        S.warm_kept++;
                    }
                }
                // terminal relaxations are never written again: their basis / x_B can leave on the second stream at once
                B_TRY(hipMemcpyAsync(b.h_basic + (size_t)i * b.cap_ldu, b.d_basic + (size_t)i * b.cap_ldu, (size_t)m_i * sizeof(int32_t), hipMemcpyDeviceToHost, copy_stream_));
                B_TRY(hipMemcpyAsync(b.h_xb + (size_t)i * b.cap_ldu, b.d_xb + (size_t)i * b.cap_ldu, (size_t)m_i * sizeof(double), hipMemcpyDeviceToHost, copy_stream_));
                B_TRY(hipEventRecord(b.lp_ev[i], copy_stream_));
                pending.push_back({i, o});
            } else {
                on_done(i, o, nullptr, nullptr);
            }
        }
        return GOMILP_OK;
    };
    int active = nlp;
    for (;;) {
        const int prev_slot = step % kRing;
        const bool last_possible = step + 1 >= kMaxSteps - 2;
        // enqueue superstep step + 1 before waiting for the snapshot of superstep `step`: the GPU never idles for the host
        // wide waves shed their quick relaxations early (short supersteps first: the batched update of hundreds of tableaus is
        // what a block step costs there); a narrow wave is a few single-workgroup chains: longer supersteps, fewer round trips
        int nb = nlp <= 16 ? (step < 2 ? 4 : 8) : (step < 2 ? 1 : (step < 4 ? 2 : (step < 8 ? 4 : 8)));
        // a loop launch has no boundary between its blocks: longer supersteps (fewer control steps on the chain) once the wave is narrow
        if (loop_slots > 0 && bound <= loop_slots && step >= 2 && nwarm == 0) nb = bound <= 8 ? (step >= 3 ? 32 : 16) : 8;   // (long supersteps from the FIRST loop launch were tried on the tree waves: 2.77 against 2.58 ms — a twin that leaves Phase I after five pivots then waits for its control step behind 128 pivots of the other)
        if (res_slots > 0 && bound <= res_slots && nwarm == 0) nb = step >= 3 ? 32 : (step >= 1 ? 16 : 8);   // one launch whatever the length: longer supersteps from the start
        if (virt_on && step == 0) nb = 1;   // (the first block of a wave with virtual tableaus: one block step)
        step++;
        blocks(nb, true);
        control(true);
        if ((rc = snapshot(step % kRing)) != GOMILP_OK) return rc;
        S.supersteps++;
        B_TRY(hipEventSynchronize(b.ev[prev_slot]));
        B_TRY(hipGetLastError());
        active = *b.h_active[prev_slot];
        bound = std::max(active, 1);   // the count only falls: a safe grid size for everything enqueued from here on
        if ((rc = harvest(prev_slot)) != GOMILP_OK) return rc;
        // narrow waves are a few long chains: whoever finished is handed to its final solve at once (the copy of its basis takes ~10 us;
        // polling would leave it waiting for the next superstep — and the last one for the superstep enqueued behind the end)
        flush_pending(active == 0 || nlp <= 16);
        if (active == 0 || last_possible) break;
    }
    // the superstep enqueued behind the deciding snapshot: no-op launches when every relaxation was terminal
    B_TRY(hipEventSynchronize(b.ev[step % kRing]));
    if ((rc = harvest(step % kRing)) != GOMILP_OK) return rc;
    for (int i = 0; i < nlp; i++)
        if (!reported[i]) {   // superstep budget spent (never seen): the single-relaxation engine takes over
            reported[i] = 1;
            Outcome o;
            o.stage = BS_HOST;
            on_done(i, o, nullptr, nullptr);
        }
    flush_pending(true);
    B_TRY(hipStreamSynchronize(stream_));
    for (int k = 0; k < nsamp; k++) {
        float a = 0, u = 0;
        if (hipEventElapsedTime(&a, b.samp_ev[(size_t)k * 4], b.samp_ev[(size_t)k * 4 + 1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&u, b.samp_ev[(size_t)k * 4 + 2], b.samp_ev[(size_t)k * 4 + 3]) != hipSuccess) continue;
        S.seconds_inner += a * 1e-3; S.seconds_update += u * 1e-3; S.blocks_sampled++;
    }
    S.seconds_total = bnow() - t0;
    return GOMILP_OK;
}

}  // namespace gomilp
