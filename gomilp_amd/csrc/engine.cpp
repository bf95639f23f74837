// Host-side engine: see engine.hpp.  Control flow mirrors gonum's simplex()
// (vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:93-302); all O(m^2)/O(m*n) work is
// enqueued as gfx950 kernels (simplex_kernels.hip).  No CPU fallback exists: any HIP failure
// surfaces as GOMILP_ERR_DEVICE.
#include "engine_work.hpp"

#include <condition_variable>
#include <atomic>

namespace gomilp {

int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *compiled_arch() { return "gfx950"; }

Engine::Engine(int device) : device_(device), w_(new Work) {}

Engine::~Engine() {
    hipSetDevice(device_);
    if (up_dA_) hipFree(up_dA_);
    if (up_stats_) hipFree(up_stats_);
    for (auto *vec : {&problems_, &child_pool_, &root_pool_})
        for (auto &p : *vec) {
            if (!p) continue;
            hipFree(p->dAt); hipFree(p->db);
            if (!p->is_child) {   // a child's c / c1 / sign / var live inside its b block
                hipFree(p->dc); hipFree(p->dc1);
                if (p->dvar) hipFree(p->dvar);
                if (p->dsign) hipFree(p->dsign);
            }
        }
    w_->release_all();
    if (stream_) hipStreamDestroy(stream_);
}

// Persistent loop kernels wait for their own workgroups, so every workgroup of a launch must be resident.  The 128-thread shape
// (up to 2048 rows: 136 workgroups of two waves, four fit a CU) leaves room for four launches at once, each with its pivot
// workgroups on an XCD of its own; the 256- / 512-thread shapes fill the chip (one workgroup per CU) and run alone.
namespace {
struct LoopSlots { std::mutex mu; std::condition_variable cv; int used = 0, big_waiting = 0; bool busy[4] = {false, false, false, false}; };
LoopSlots &loop_slots(int dev) { static LoopSlots s[64]; return s[dev & 63]; }
}  // namespace
int Engine::loop_acquire(int dev, int weight) {
    LoopSlots &L = loop_slots(dev);
    std::unique_lock<std::mutex> lk(L.mu);
    if (weight >= 4) {
        L.big_waiting++;   // from here on no further small launch is admitted: a stream of them would never let `used` reach 0
        L.cv.wait(lk, [&] { return L.used == 0; });
        L.big_waiting--;
        L.used = 4;
        for (bool &b2 : L.busy) b2 = true;
        return 0;
    }
    L.cv.wait(lk, [&] { return L.used < 4 && L.big_waiting == 0; });
    L.used++;
    for (int i = 0; i < 4; i++) if (!L.busy[i]) { L.busy[i] = true; return i; }
    return 0;
}
bool Engine::loop_try_acquire_all(int dev) {
    LoopSlots &L = loop_slots(dev);
    std::lock_guard<std::mutex> lk(L.mu);
    if (L.used != 0 || L.big_waiting != 0) return false;
    L.used = 4;
    for (bool &b2 : L.busy) b2 = true;
    return true;
}
void Engine::loop_release(int dev, int weight, int slot) {
    LoopSlots &L = loop_slots(dev);
    {
        std::lock_guard<std::mutex> lk(L.mu);
        if (weight >= 4) { L.used = 0; for (bool &b2 : L.busy) b2 = false; }
        else { L.used--; L.busy[slot & 3] = false; }
    }
    L.cv.notify_all();
}

int Engine::set(const std::string &key, int64_t v) {
    std::lock_guard<std::mutex> g(mu_);
    if (key == "chunk") { if (v < 1) return GOMILP_ERR_BAD_SHAPE; chunk_ = v; }
    else if (key == "refresh") { if (v < 0) return GOMILP_ERR_BAD_SHAPE; refresh_ = v; }
    else if (key == "trace") trace_on_ = v ? 1 : 0;
    else if (key == "max_pivots") max_pivots_ = v < 0 ? 0 : v;
    else if (key == "sample_events") sample_events_ = v < 0 ? 0 : v;
    else if (key == "fused") fused_ = v ? 1 : 0;
    else if (key == "lu_look") lu_look_ = v ? 1 : 0;
    else if (key == "lu_blocked") lu_blocked_ = v < 0 ? 0 : (v > 2 ? 3 : v);  // 0 per column, 1 blocked panels, 2 compressed rounds (slot panel), 3 the same with the look-ahead schedule (default; lu_compressed.hip)
    else if (key == "tableau") tableau_ = v ? 1 : 0;
    else if (key == "blocked") blocked_ = v ? 1 : 0;
    else if (key == "bt_nt") { if (v != 0 && v != 256 && v != 512 && v != 1024) return GOMILP_ERR_BAD_SHAPE; bt_nt_ = v; }
    else if (key == "bt_old") bt_old_ = v ? 1 : 0;
    else if (key == "bt_upd_valu") bt_upd_valu_ = v ? 1 : 0;
#ifdef GOMILP_DEBUG
    else if (key == "bt_fault") bt_fault_ = v;   // fault injection, diagnostic flavour only: 1 a workgroup of the block / loop kernels, 2 the U-solve workgroup of the look-ahead LU
#endif
    else if (key == "general_device") general_device_ = v ? 1 : 0;
    else if (key == "lu_cross") lu_cross_ = v ? 1 : 0;
    else if (key == "general_block") general_block_ = v ? 1 : 0;
    else if (key == "general_min_rows") general_min_rows_ = v < 2 ? 2 : v;
    else if (key == "bt_groups") { if (v != -1 && v != 0 && v != 2 && v != 4 && v != 8 && v != 16) return GOMILP_ERR_BAD_SHAPE; bt_groups_ = v; }
    else if (key == "bt_stamps") bt_stamps_ = v ? 1 : 0;
    else if (key == "bt_lag") bt_lag_ = v ? 1 : 0;
    else if (key == "exact_degenerate") { if (v < 0 || v > 3) return GOMILP_ERR_BAD_SHAPE; exact_degenerate_ = v; }   // 3: strict — EVERY pivot decided on fresh gonum-order solves (engine_tableau.cpp exact_step)
    else if (key == "loop_grid") { if (v < 0 || v > 4096) return GOMILP_ERR_BAD_SHAPE; loop_grid_ = v; }
    else if (key == "loop_chunk") { if (v < 32) return GOMILP_ERR_BAD_SHAPE; loop_chunk_ = v; }
    else if (key == "cond_guard") cond_guard_ = v ? 1 : 0;
    else if (key == "poll_delay") { if (v < 0 || v > 64) return GOMILP_ERR_BAD_SHAPE; poll_delay_ = v; }
    else if (key == "loop_upd") { if (v < 0) return GOMILP_ERR_BAD_SHAPE; loop_upd_ = v; }
    else if (key == "loop_rep") loop_rep_ = v ? 1 : 0;
    else if (key == "loop_g") { if (v != 0 && v != 8 && v != 16) return GOMILP_ERR_BAD_SHAPE; loop_g_ = v; }
    else if (key == "loop_k") { if (v != 0 && v != 8 && v != 12 && v != 16) return GOMILP_ERR_BAD_SHAPE; loop_k_ = v; }
    else if (key == "block_k") { if (v < 0 || v > bt_max_k()) return GOMILP_ERR_BAD_SHAPE; block_k_ = v; }
    else return GOMILP_ERR_BAD_SHAPE;
    return GOMILP_OK;
}

int Engine::ensure_work(int m, int ncols) {
    HIP_TRY(hipSetDevice(device_));
    if (!stream_) HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    Work &w = *w_;
    if (!w.st) {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && ncu > 0) ncu_ = ncu;
        HIP_TRY(dmalloc(&w.pk_price, kMaxPartials)); HIP_TRY(dmalloc(&w.pk_ratio, kMaxPartials));
        HIP_TRY(dmalloc(&w.pi_price, kMaxPartials)); HIP_TRY(dmalloc(&w.pi_ratio, kMaxPartials));
        HIP_TRY(dmalloc(&w.pv_price, kMaxPartials)); HIP_TRY(dmalloc(&w.pb_ratio, kMaxPartials)); HIP_TRY(dmalloc(&w.pd_ratio, kMaxPartials)); HIP_TRY(dmalloc(&w.px_ratio, kMaxPartials));
        for (int t = 0; t < 2; t++) {
            HIP_TRY(dmalloc(&w.lpk[t], kMaxPartials)); HIP_TRY(dmalloc(&w.lpl[t], kMaxPartials)); HIP_TRY(dmalloc(&w.lpr[t], kMaxPartials));
        }
        HIP_TRY(dmalloc(&w.st, 1));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.st_host), sizeof(DevState), hipHostMallocDefault));
        HIP_TRY(dmalloc(&w.luctl, 2));   // (by round parity: lu_compressed.hip, look-ahead schedule)
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.luctl_host), 2 * sizeof(LUCtl), hipHostMallocDefault));
        w.trace_cap = 1 << 18;
        HIP_TRY(dmalloc(&w.trace, (size_t)w.trace_cap));
        HIP_TRY(hipEventCreate(&w.ev[0])); HIP_TRY(hipEventCreate(&w.ev[1]));
    }
    const int ld = (m + 1) & ~1;
    if (m <= w.cap_m && ncols <= w.cap_cols) return GOMILP_OK;
    // head-room: B&B children grow by one row and one column per level; without it every level would release and
    // re-allocate the whole work set (hipFree synchronises the device, pinned allocations are slow)
    const int nm = std::max(m + 32 + m / 16, w.cap_m), nc = std::max(ncols + 64 + ncols / 16, w.cap_cols);
    const int nld = std::max((nm + 1) & ~1, w.cap_ld);
    w.release();
    for (auto &p : w.binv) HIP_TRY(dmalloc(&p, (size_t)nm * nld));
    HIP_TRY(dmalloc(&w.W, (size_t)nm * nld));
    HIP_TRY(dmalloc(&w.xb, (size_t)nld)); HIP_TRY(dmalloc(&w.yb[0], (size_t)nld)); HIP_TRY(dmalloc(&w.yb[1], (size_t)nld));
    HIP_TRY(dmalloc(&w.dvec, (size_t)nld));
    HIP_TRY(dmalloc(&w.move, (size_t)nld)); HIP_TRY(dmalloc(&w.rvec, (size_t)nc));
    HIP_TRY(dmalloc(&w.yscratch, (size_t)64 * nld));
    HIP_TRY(dmalloc(&w.basic, (size_t)nm + (size_t)nc)); w.nonbasic = w.basic + nm;   // one block: both lists go up in one copy
    // (rowstep: + two snapshots, by round parity, for the look-ahead schedule of the compressed LU)
    HIP_TRY(dmalloc(&w.lpos, (size_t)nm)); HIP_TRY(dmalloc(&w.rowstep, (size_t)3 * nm)); HIP_TRY(dmalloc(&w.rho, (size_t)nm));
    HIP_TRY(dmalloc(&w.unitrow, (size_t)nm)); HIP_TRY(dmalloc(&w.denseflag, (size_t)nm)); HIP_TRY(dmalloc(&w.dlist, (size_t)nm));
    HIP_TRY(dmalloc(&w.ludiag, (size_t)nm)); HIP_TRY(dmalloc(&w.Wd, (size_t)nm * nld + 512));   // (+ the tail of k_luc_pack_small: row positions, flags, two control blocks)
    HIP_TRY(dmalloc(&w.luLp, (size_t)64 * nld)); HIP_TRY(dmalloc(&w.luUp, (size_t)64 * nld));   // (2 x 32 rows: by round parity)
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.h_W), ((size_t)nm * nld + 512) * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.h_vec), (size_t)std::max(nld, nc) * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.h_chk), (size_t)nld * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.h_idx), ((size_t)nm + (size_t)nc) * sizeof(int32_t), hipHostMallocDefault));   // mirrors the basic | nonbasic device block
    {
        const size_t cap = std::max<size_t>(4096, sizeof(double) * ((size_t)std::max(nld, nc) + 64));
        for (int t = 0; t < Work::kStageSlots; t++) {
            if (w.stage_cap[t] >= cap) continue;
            if (w.stage_inflight) { hipStreamSynchronize(stream_); w.stage_inflight = 0; }
            if (w.stage_buf[t]) hipHostFree(w.stage_buf[t]);
            w.stage_buf[t] = nullptr; w.stage_cap[t] = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.stage_buf[t]), cap, hipHostMallocDefault));
            w.stage_cap[t] = cap;
        }
    }
    w.cap_m = nm; w.cap_ld = nld; w.cap_cols = nc;
    return GOMILP_OK;
}

// ------------------------------------------------------------------------------------------------
// upload: A (row-major m x n) -> At ((n+1) x ld) in HBM, column statistics for verifyInputs and for
// the unit-column fast path of findLinearlyIndependent (simplex.go:385-439, :611-637)
// ------------------------------------------------------------------------------------------------
int64_t Engine::upload(const double *c, const double *A, int64_t lda, const double *b, int64_t m64, int64_t n64, bool lazy_host) {
    std::lock_guard<std::mutex> g(mu_);
    if (!c || !A || !b || m64 <= 0 || n64 <= 0 || lda < n64 || m64 > (1 << 20) || n64 > (1 << 22)) return -GOMILP_ERR_BAD_SHAPE;
    const int m = (int)m64, n = (int)n64;
    const int ld = (m + 1) & ~1;
    if ((size_t)ld * sizeof(double) > 64 * 1024) return -GOMILP_ERR_UNSUPPORTED;  // LDS staging of one row
    if (hipSetDevice(device_) != hipSuccess) return -GOMILP_ERR_DEVICE;
    if (!stream_ && hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return -GOMILP_ERR_DEVICE;
    const double t0 = now_s();
    const size_t need_at = (size_t)(n + 1) * ld, need_c = (size_t)n + 1, need_b = (size_t)ld;
    std::unique_ptr<Problem> P;
    for (size_t i = 0; i < root_pool_.size(); i++) {   // a released root whose buffers fit
        Problem &q = *root_pool_[i];
        if (q.cap_at >= need_at && q.cap_c >= need_c && q.cap_b >= need_b) {
            P = std::move(root_pool_[i]);
            root_pool_.erase(root_pool_.begin() + i);
            break;
        }
    }
    const bool recycled = (bool)P;
    if (!P) P.reset(new Problem);
    P->m = m; P->n = n; P->ld = ld;
    double *dA = nullptr;
    int32_t *dstats = nullptr;
    auto fail = [&](int code) -> int64_t {
        if (P->dAt) hipFree(P->dAt);
        if (P->dc) hipFree(P->dc);
        if (P->dc1) hipFree(P->dc1);
        if (P->db) hipFree(P->db);
        return -code;
    };
#define UP_TRY(expr) do { if ((expr) != hipSuccess) return fail(GOMILP_ERR_DEVICE); } while (0)
    if (up_dA_cap_ < (size_t)m * n) {
        if (up_dA_) hipFree(up_dA_);
        up_dA_ = nullptr; up_dA_cap_ = 0;
        UP_TRY(dmalloc(&up_dA_, (size_t)m * n));
        up_dA_cap_ = (size_t)m * n;
    }
    const size_t range_off = ((size_t)3 * n + m + 1) & ~(size_t)1;   // two 64-bit words behind the lists: max / min non-zero |a_ij|
    if (up_stats_cap_ < range_off + 4) {
        if (up_stats_) hipFree(up_stats_);
        up_stats_ = nullptr; up_stats_cap_ = 0;
        UP_TRY(dmalloc(&up_stats_, range_off + 4));
        up_stats_cap_ = range_off + 4;
    }
    dA = up_dA_; dstats = up_stats_;
    if (!recycled) {
        UP_TRY(dmalloc(&P->dAt, need_at));
        UP_TRY(dmalloc(&P->dc, need_c));
        UP_TRY(dmalloc(&P->dc1, need_c));
        UP_TRY(dmalloc(&P->db, need_b));
        P->cap_at = need_at; P->cap_c = need_c; P->cap_b = need_b;
    }
    UP_TRY(hipMemcpy2DAsync(dA, (size_t)n * sizeof(double), A, (size_t)lda * sizeof(double), (size_t)n * sizeof(double), m,
                            hipMemcpyHostToDevice, stream_));
    UP_TRY(hipMemsetAsync(P->dAt, 0, (size_t)(n + 1) * ld * sizeof(double), stream_));
    UP_TRY(hipMemsetAsync(P->dc, 0, ((size_t)n + 1) * sizeof(double), stream_));
    UP_TRY(hipMemsetAsync(P->dc1, 0, ((size_t)n + 1) * sizeof(double), stream_));
    UP_TRY(hipMemsetAsync(P->db, 0, (size_t)ld * sizeof(double), stream_));
    UP_TRY(hipMemsetAsync(dstats, 0, (range_off + 2) * sizeof(int32_t), stream_));
    UP_TRY(hipMemsetAsync(dstats + range_off + 2, 0xFF, 2 * sizeof(int32_t), stream_));
    UP_TRY(hipMemcpyAsync(P->dc, c, (size_t)n * sizeof(double), hipMemcpyHostToDevice, stream_));
    UP_TRY(hipMemcpyAsync(P->db, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, stream_));
    const double one = 1.0;
    UP_TRY(hipMemcpyAsync(P->dc1 + n, &one, sizeof(double), hipMemcpyHostToDevice, stream_));
    launch_transpose_in(dA, n, m, n, P->dAt, ld, stream_);
    launch_col_stats(P->dAt, ld, m, n, dstats, dstats + n, dstats + 2 * n, dstats + 3 * n, reinterpret_cast<unsigned long long *>(dstats + range_off), stream_);
    std::vector<int32_t> hs(range_off + 4);
    UP_TRY(hipMemcpyAsync(hs.data(), dstats, hs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    UP_TRY(sync_stream());
    UP_TRY(hipGetLastError());
#undef UP_TRY
    P->nnz.assign(hs.begin(), hs.begin() + n);
    P->lastrow.assign(hs.begin() + n, hs.begin() + 2 * n);
    P->allone.assign(hs.begin() + 2 * n, hs.begin() + 3 * n);
    P->hb.assign(b, b + m);
    P->hc.assign(c, c + n);
    {   // spread of the entries (k_col_stats): badly scaled inputs take the careful path (Engine::make_bt_args)
        unsigned long long bits[2];
        memcpy(bits, &hs[range_off], sizeof(bits));
        double amax, amin;
        memcpy(&amax, &bits[0], 8); memcpy(&amin, &bits[1], 8);
        P->scale_span = (bits[0] != 0 && bits[1] != ~0ull && amin > 0 && amin <= amax) ? amax / amin : 1.0;
    }
    P->hA.clear();
    P->lazy_host = false;
    if ((size_t)m * n <= ((size_t)1 << 25)) {  // up to 256 MB: keep A for the general initial-basis path (equality rows, supplied basis)
        // (the flat call skips the copy of a large A — 67 MB, ~5 ms at the metric size — and fetches it from the device if a rare path asks)
        if (lazy_host && (size_t)m * n > ((size_t)1 << 20)) P->lazy_host = true;
        else {
            P->hA.resize((size_t)m * n);
            for (int i = 0; i < m; i++) memcpy(&P->hA[(size_t)i * n], A + (size_t)i * lda, sizeof(double) * (size_t)n);
        }
    }
    // verifyInputs (simplex.go:404-438): rows first, then columns, first offender decides
    P->verify_status = GOMILP_OK;
    for (int i = 0; i < m && P->verify_status == GOMILP_OK; i++)
        if (!hs[(size_t)3 * n + i]) P->verify_status = (b[i] != 0) ? GOMILP_ERR_INFEASIBLE : GOMILP_ERR_ZERO_ROW;
    for (int j = 0; j < n && P->verify_status == GOMILP_OK; j++)
        if (P->nnz[j] == 0) P->verify_status = (c[j] < 0) ? GOMILP_ERR_UNBOUNDED : GOMILP_ERR_ZERO_COLUMN;
    P->seconds_upload = now_s() - t0;
    { static std::atomic<uint64_t> next_serial(1); P->serial = next_serial.fetch_add(1); }
    for (size_t i = 0; i < problems_.size(); i++)
        if (!problems_[i]) { problems_[i] = std::move(P); return (int64_t)i; }
    problems_.push_back(std::move(P));
    return (int64_t)problems_.size() - 1;
}

const Problem *Engine::problem_ptr(int64_t id) {
    std::lock_guard<std::mutex> g(mu_);
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id]) return nullptr;
    return problems_[id].get();
}

int64_t Engine::upload_child(int64_t root, int K, const int32_t *var, const double *sign, const double *rhs) {
    std::lock_guard<std::mutex> g(mu_);
    if (root < 0 || (size_t)root >= problems_.size() || !problems_[root]) return -GOMILP_ERR_BAD_SHAPE;
    return upload_child_impl(*problems_[root], root, K, var, sign, rhs);
}

int64_t Engine::upload_child_of(Engine &owner, int64_t root, int K, const int32_t *var, const double *sign, const double *rhs) {
    if (&owner == this) return upload_child(root, K, var, sign, rhs);
    if (owner.device_ != device_) return -GOMILP_ERR_BAD_SHAPE;
    const Problem *R = owner.problem_ptr(root);
    if (!R) return -GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(mu_);
    return upload_child_impl(*R, -1, K, var, sign, rhs);   // no root link: the host copy of A (general-basis path) is not reachable
}

int64_t Engine::upload_child_impl(const Problem &R, int64_t root, int K, const int32_t *var, const double *sign, const double *rhs) {
    if (K < 0 || (K > 0 && (!var || !sign || !rhs))) return -GOMILP_ERR_BAD_SHAPE;
    const int m0 = R.m, n0 = R.n, m = m0 + K, n = n0 + K;
    const int ld = (m + 1) & ~1;
    for (int k = 0; k < K; k++) if (var[k] < 0 || var[k] >= n0) return -GOMILP_ERR_BAD_SHAPE;
    if ((size_t)ld * sizeof(double) > 64 * 1024) return -GOMILP_ERR_UNSUPPORTED;
    if (hipSetDevice(device_) != hipSuccess) return -GOMILP_ERR_DEVICE;
    if (!stream_ && hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return -GOMILP_ERR_DEVICE;
    const double t0 = now_s();
    const size_t need_at = (size_t)(n + 1) * ld, need_c = (size_t)n + 1, need_b = (size_t)ld;
    std::unique_ptr<Problem> P;
    for (size_t i = 0; i < child_pool_.size(); i++) {
        Problem &c = *child_pool_[i];
        if (c.cap_at >= need_at && c.cap_c >= need_c && c.cap_b >= need_b && c.cap_k >= K) {
            P = std::move(child_pool_[i]);
            child_pool_.erase(child_pool_.begin() + i);
            break;
        }
    }
    auto release = [&](Problem &q) {   // a child owns dAt and ONE block holding b | c | c1 | sign | var
        if (q.dAt) hipFree(q.dAt);
        if (q.db) hipFree(q.db);
        q.dAt = q.dc = q.dc1 = q.db = q.dsign = nullptr; q.dvar = nullptr;
    };
    if (!P) {
        P.reset(new Problem);
        P->is_child = true;
        // some head-room so that deeper children of the same root reuse the slot
        const int kcap = K + 8, mcap = m0 + kcap, ncap = n0 + kcap, ldcap = (mcap + 1) & ~1;
        P->cap_at = (size_t)(ncap + 1) * ldcap; P->cap_c = (size_t)ncap + 1; P->cap_b = (size_t)ldcap; P->cap_k = kcap;
        // b | c | c1 | sign (doubles) | var (ints) in one device block, mirrored by the pinned staging block: one copy
        const size_t blk = P->cap_b + 2 * P->cap_c + 2 * (size_t)P->cap_k + 8;
        if (dmalloc(&P->dAt, P->cap_at) != hipSuccess || dmalloc(&P->db, blk) != hipSuccess) {
            release(*P);
            return -GOMILP_ERR_DEVICE;
        }
        P->dc = P->db + P->cap_b; P->dc1 = P->dc + P->cap_c; P->dsign = P->dc1 + P->cap_c;
        P->dvar = reinterpret_cast<int32_t *>(P->dsign + P->cap_k);
    }
    P->m = m; P->n = n; P->ld = ld;
    auto fail = [&](int code) -> int64_t { release(*P); return -code; };
#define UP_TRY(expr) do { if ((expr) != hipSuccess) return fail(GOMILP_ERR_DEVICE); } while (0)
    P->hc = R.hc; P->hc.resize(n, 0.0);        // c' = [c, 0]   (subproblem.go:110-114)
    P->hb = R.hb; P->hb.insert(P->hb.end(), rhs, rhs + K);  // b' = [b; h]  (:117-119)
    // one pinned staging block with the layout of the child's device block (capacity offsets)
    Work &w = *w_;
    const size_t blk_bytes = (P->cap_b + 2 * P->cap_c + (size_t)P->cap_k) * sizeof(double) + (size_t)P->cap_k * sizeof(int32_t);
    if (w.child_stage_cap < blk_bytes + 64) {
        if (w.child_stage) hipHostFree(w.child_stage);
        w.child_stage_cap = 2 * (blk_bytes + 64);
        UP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.child_stage), w.child_stage_cap, hipHostMallocDefault));
    }
    double *sb = reinterpret_cast<double *>(w.child_stage);
    double *sc = sb + P->cap_b, *sc1 = sc + P->cap_c, *ss = sc1 + P->cap_c;
    int32_t *sv = reinterpret_cast<int32_t *>(ss + P->cap_k);
    for (int i = 0; i < ld; i++) sb[i] = i < m ? P->hb[i] : 0.0;
    for (int j = 0; j <= n; j++) { sc[j] = j < n ? P->hc[j] : 0.0; sc1[j] = j == n ? 1.0 : 0.0; }
    for (int k = 0; k < K; k++) { ss[k] = sign[k]; sv[k] = var[k]; }
    UP_TRY(hipMemcpyAsync(P->db, sb, blk_bytes, hipMemcpyHostToDevice, stream_));
    launch_child_assemble(R.dAt, R.ld, m0, n0, P->dAt, ld, K, P->dvar, P->dsign, stream_);
    UP_TRY(sync_stream());  // the staging block is reused by the next child
    UP_TRY(hipGetLastError());
#undef UP_TRY
    // column statistics follow from the root's: a branched column gains one entry per constraint on it
    P->nnz = R.nnz; P->lastrow = R.lastrow; P->allone = R.allone;
    P->nnz.resize(n); P->lastrow.resize(n); P->allone.resize(n);
    for (int k = 0; k < K; k++) {
        P->nnz[var[k]] += 1; P->lastrow[var[k]] = m0 + k;
        if (sign[k] != 1.0) P->allone[var[k]] = 0;
        P->nnz[n0 + k] = 1; P->lastrow[n0 + k] = m0 + k; P->allone[n0 + k] = 1;
    }
    // verifyInputs of the child (simplex.go:404-438): branch rows and their slack columns are never empty, so a row verdict of
    // the root stands; a COLUMN verdict is re-derived, because a branch row can fill a column that was empty in the root
    P->verify_status = R.verify_status;
    if (R.verify_status == GOMILP_ERR_ZERO_COLUMN || R.verify_status == GOMILP_ERR_UNBOUNDED) {
        P->verify_status = GOMILP_OK;
        for (int j = 0; j < n && P->verify_status == GOMILP_OK; j++)
            if (P->nnz[j] == 0) P->verify_status = (P->hc[j] < 0) ? GOMILP_ERR_UNBOUNDED : GOMILP_ERR_ZERO_COLUMN;
    }
    // the host copy of [[A0, 0], [G#, I]] is only needed when a solve starts from a non-slack basis: built on demand
    P->hA.clear();
    P->scale_span = R.scale_span;
    P->root = root;
    P->root_ptr = &R;   // (a root resident in another engine of the pool: `root` is -1, the owner keeps it alive)
    P->kvar.assign(var, var + K);
    P->ksign.assign(sign, sign + K);
    P->seconds_upload = now_s() - t0;
    for (size_t i = 0; i < problems_.size(); i++)
        if (!problems_[i]) { problems_[i] = std::move(P); return (int64_t)i; }
    problems_.push_back(std::move(P));
    return (int64_t)problems_.size() - 1;
}

int Engine::free_problem(int64_t id) {
    std::lock_guard<std::mutex> g(mu_);
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id]) return GOMILP_ERR_BAD_SHAPE;
    hipSetDevice(device_);
    if (problems_[id]->is_child) {  // keep the buffers for the next child (no hipFree: it synchronises the device)
        child_pool_.push_back(std::move(problems_[id]));
        problems_[id].reset();
        return GOMILP_OK;
    }
    // a root's buffers are kept for the next upload of a fitting shape (at most 4: two shapes in rotation + slack)
    problems_[id]->hA.clear(); problems_[id]->hA.shrink_to_fit();
    root_pool_.push_back(std::move(problems_[id]));
    problems_[id].reset();
    while (root_pool_.size() > 4) {
        Problem &P = *root_pool_.front();
        hipFree(P.dAt); hipFree(P.dc); hipFree(P.dc1); hipFree(P.db);
        root_pool_.erase(root_pool_.begin());
    }
    return GOMILP_OK;
}

LPArgs Engine::make_args(const Problem &P, int phase, double tol, int nn, const double *cost) {
    Work &w = *w_;
    LPArgs a;
    memset(&a, 0, sizeof(a));
    a.m = P.m; a.ld = P.ld; a.nn = nn; a.phase = phase; a.tol = tol;
    a.At = P.dAt; a.cost = cost; a.b = P.db;
    a.binv_cur = w.binv[cur_]; a.binv_next = w.binv[cur_ ^ 1];
    a.xb = w.xb; a.y = w.yb[ycur_]; a.dvec = w.dvec; a.move = w.move; a.rvec = w.rvec;
    a.basic = w.basic; a.nonbasic = w.nonbasic;
    a.pk_price = w.pk_price; a.pi_price = w.pi_price; a.pk_ratio = w.pk_ratio; a.pi_ratio = w.pi_ratio;
    a.pv_price = w.pv_price; a.pd_ratio = w.pd_ratio; a.pb_ratio = w.pb_ratio;
    a.st = w.st;
    a.trace = (trace_on_ || shadow_trace_) ? w.trace : nullptr;
    a.trace_cap = w.trace_cap;
    return a;
}

void Engine::sync_state_to_device() {
    // a snapshot through the staging ring: the host copy may be rewritten before the stream reaches this upload
    stage_upload(w_->st, w_->st_host, sizeof(DevState));
}

int Engine::refresh_xb_y(const Problem &P, const double *cost) {
    Work &w = *w_;
    launch_matvec_rows(w.binv[cur_], P.ld, P.m, P.db, w.xb, stream_);
    launch_y_from_binv(w.binv[cur_], P.ld, P.m, cost, w.basic, w.yscratch, w.yb[ycur_], stream_);
    launches_ += 3;
    return GOMILP_OK;
}

// host copy of A for the general-basis path; children derive it from their root (subproblem.go:81-139)
bool Engine::ensure_host_A(const Problem &P) {
    if (!P.hA.empty()) return true;
    if (!P.is_child && P.lazy_host) {   // flat call: the copy was skipped at upload, At (column-major, resident) has the same numbers
        const int m = P.m, n = P.n;
        std::vector<double> at((size_t)n * P.ld);
        if (hipMemcpyAsync(at.data(), P.dAt, at.size() * sizeof(double), hipMemcpyDeviceToHost, stream_) != hipSuccess || sync_stream() != hipSuccess) return false;
        P.hA.resize((size_t)m * n);
        for (int j = 0; j < n; j++) for (int i = 0; i < m; i++) P.hA[(size_t)i * n + j] = at[(size_t)j * P.ld + i];
        return true;
    }
    if (!P.is_child || !P.root_ptr) return false;
    const Problem &R = *P.root_ptr;
    // the root of a pool child may live in another worker's engine (gomilp_pool_add_root keeps extra roots in worker 0): its host
    // copy was made at upload (roots are never built lazily), read in place
    if (P.root >= 0 ? !ensure_host_A(R) : R.hA.empty()) return false;
    const int m = P.m, n = P.n, m0 = R.m, n0 = R.n, K = (int)P.kvar.size();
    if ((size_t)m * n > ((size_t)1 << 25)) return false;
    P.hA.assign((size_t)m * n, 0.0);
    for (int i = 0; i < m0; i++) memcpy(&P.hA[(size_t)i * n], &R.hA[(size_t)i * n0], sizeof(double) * (size_t)n0);
    for (int k = 0; k < K; k++) { P.hA[(size_t)(m0 + k) * n + P.kvar[k]] = P.ksign[k]; P.hA[(size_t)(m0 + k) * n + n0 + k] = 1.0; }
    return true;
}

// Small host -> device upload from pageable / short-lived memory without a stream sync of its own: the copy is enqueued
// from a slot of a pinned ring.  A slot may be rewritten only after the stream has been fully synchronised since its
// last use: sync_stream() — the engine's only way to wait for the stream — resets the count, and the ring forces a
// sync itself when it would wrap around without one (no per-upload event: with many worker streams the queue
// packets, not the host, are the bottleneck).
hipError_t Engine::sync_stream() {
    const hipError_t e = hipStreamSynchronize(stream_);
    w_->stage_inflight = 0;
    return e;
}

int Engine::stage_upload(void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return GOMILP_OK;
    Work &w = *w_;
    if (w.stage_inflight >= Work::kStageSlots) HIP_TRY(sync_stream());
    const int slot = w.stage_next;
    if (w.stage_cap[slot] < bytes) {
        // slots are sized once per work-buffer generation (ensure_work); an oversized request takes the plain path
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream_));
        HIP_TRY(sync_stream());
        return GOMILP_OK;
    }
    w.stage_next = (slot + 1) % Work::kStageSlots;
    w.stage_inflight++;
    memcpy(w.stage_buf[slot], src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, w.stage_buf[slot], bytes, hipMemcpyHostToDevice, stream_));
    return GOMILP_OK;
}

int Engine::upload_index_lists(const std::vector<int32_t> &basic, const std::vector<int32_t> &nonbasic) {
    Work &w = *w_;
    if (!basic.empty() && !nonbasic.empty()) {
        // the two lists share one device block (basic at 0, nonbasic at cap_m): one copy, one queue packet
        const size_t gap = (size_t)(w.nonbasic - w.basic);
        std::vector<int32_t> both(gap + nonbasic.size(), 0);
        memcpy(both.data(), basic.data(), basic.size() * sizeof(int32_t));
        memcpy(both.data() + gap, nonbasic.data(), nonbasic.size() * sizeof(int32_t));
        return stage_upload(w.basic, both.data(), both.size() * sizeof(int32_t));
    }
    int rc = stage_upload(w.basic, basic.data(), basic.size() * sizeof(int32_t));
    if (rc != GOMILP_OK) return rc;
    return stage_upload(w.nonbasic, nonbasic.data(), nonbasic.size() * sizeof(int32_t));
}

// ------------------------------------------------------------------------------------------------
// replaceBland (simplex.go:347-383), host-driven: candidates are tried one at a time with the FTRAN
// kernel.  Deviation: the `mat.Cond(abTmp,1) < 1e16` guard of :377 is replaced by the
// pivot-magnitude guard |d_replace| >= dRoundTol that move[replace] < +Inf already implies.
// ------------------------------------------------------------------------------------------------
int Engine::host_bland(const Problem &P, LPArgs &a, gomilp_lp_stats *st) {
    Work &w = *w_;
    const int m = P.m, nn = a.nn;
    std::vector<double> r(nn), move(m);
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.rvec, (size_t)nn * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    for (int j = 0; j < nn; j++) { r[j] = w.h_vec[j]; if (fabs(r[j]) < 1e-13) r[j] = 0; }  // rRoundTol, :252-256
    for (int i = 0; i < nn; i++) {
        if (r[i] > -1e-14) continue;  // blandNegTol, :352
        w.st_host->done = 0; w.st_host->status = ST_RUNNING;
        sync_state_to_device();
        launch_ftran(a, 0, i, -1, stream_);
        launches_++;
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.move, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int k = 0; k < m; k++) move[k] = w.h_vec[k];
        int64_t replace = min_idx(move.data(), m);
        if (move[replace] == std::numeric_limits<double>::infinity()) return GOMILP_ERR_UNBOUNDED;  // computeMove :328
        if (!(fabs(move[replace]) > 1e-12)) {  // blandZeroTol, :362
            replace = -1;
            for (int rp = 0; rp < m; rp++)
                if (!(move[rp] > 1e-12)) { replace = rp; break; }  // :368-379 (cond guard: see header comment)
            if (replace < 0) continue;
        }
        launch_update(a, 0, (int)replace, 0, 1, stream_);
        launches_++;
        cur_ ^= 1;
        if (st) st->bland_steps++;
        // pivots counter is advanced on device by the update kernel
        return GOMILP_OK;
    }
    return GOMILP_ERR_BLAND;
}

// ------------------------------------------------------------------------------------------------
// Pivot loop (simplex.go:233-293): chunks of kernel launches; the kernels stop themselves through
// DevState::done, the host looks at the state after each chunk.
// returns GOMILP_OK when the loop ended at an optimum, else the reference's error class.
// ------------------------------------------------------------------------------------------------
void Engine::account_samples(gomilp_lp_stats *st, const std::vector<int64_t> &sample_t, int64_t executed, int nk) {
    // nk kernels per pivot, each bracketed by its own (start, stop) event pair attached to the dispatch
    if (!st) return;
    Work &w = *w_;
    for (size_t s = 0; s < sample_t.size(); s++) {
        if (sample_t[s] >= executed) break;  // kernels already stopped by DevState::done are not samples
        float ms[3] = {0, 0, 0};
        bool ok = true;
        for (int k = 0; k < nk; k++)
            ok = ok && hipEventElapsedTime(&ms[k], w.sample_ev[(s * 3 + k) * 2], w.sample_ev[(s * 3 + k) * 2 + 1]) == hipSuccess;
        if (!ok) continue;
        if (nk == 3) {
            st->pivot_kernel_seconds[0] += ms[0] * 1e-3; st->pivot_kernel_seconds[1] += ms[1] * 1e-3;
            st->pivot_kernel_seconds[2] += ms[2] * 1e-3;
        } else {
            st->pivot_kernel_seconds[0] += ms[0] * 1e-3; st->pivot_kernel_seconds[2] += ms[1] * 1e-3;
        }
        st->pivot_kernel_seconds[3] += 1.0;  // number of sampled pivots
    }
}

int Engine::run_loop(const Problem &P, int phase, double tol, int nn, const double *cost, gomilp_lp_stats *st) {
    if (fused_ && fused_supported(P.ld)) return run_loop_fused(P, phase, tol, nn, cost, st);
    Work &w = *w_;
    DevState &hs = *w.st_host;
    hs.done = 0; hs.status = ST_RUNNING; hs.pivots = 0; hs.q = hs.p = -1; hs.rq = hs.dp = hs.mv = 0;
    hs.max_pivots = max_pivots_;
    hs.lu_singular = 0;
    sync_state_to_device();  // trace_len is cumulative over the solve
    int64_t since_refresh = 0;
    HIP_TRY(hipEventRecord(w.ev[0], stream_));
    int ret = GOMILP_OK;
    const bool sampling = sample_events_ > 0;
    for (;;) {
        const int64_t before = hs.pivots;
        std::vector<int64_t> sample_t;  // chunk-local pivot index of every sampled pivot
        for (int64_t t = 0; t < chunk_; t++) {
            LPArgs a = make_args(P, phase, tol, nn, cost);
            a.binv_cur = w.binv[(cur_ + t) & 1];
            a.binv_next = w.binv[(cur_ + t + 1) & 1];
            const bool sample = sampling && ((before + t) % sample_events_ == 0);
            hipEvent_t e[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            if (sample) {
                const size_t e0 = sample_t.size() * 6;
                while (w.sample_ev.size() < e0 + 6) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); w.sample_ev.push_back(ev); }
                for (int k = 0; k < 6; k++) e[k] = w.sample_ev[e0 + k];
                sample_t.push_back(t);
            }
            const int gp = launch_price(a, stream_, e[0], e[1]);
            const int gr = launch_ftran(a, gp, -1, -1, stream_, e[2], e[3]);
            launch_update(a, gr, -1, 0, 0, stream_, e[4], e[5]);
            launches_ += 3;
        }
        HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        const int64_t executed = hs.pivots - before;
        cur_ = (int)((cur_ + executed) & 1);
        since_refresh += executed;
        account_samples(st, sample_t, executed, 3);
        if (!hs.done) {
            if (refresh_ > 0 && since_refresh >= refresh_) {
                refresh_xb_y(P, cost);
                since_refresh = 0;
                if (st) st->refreshes++;
            }
            continue;
        }
        if (hs.status == ST_OPTIMAL) break;
        if (hs.status == ST_UNBOUNDED) { ret = GOMILP_ERR_UNBOUNDED; break; }
        if (hs.status == ST_MAX_PIVOTS) { ret = GOMILP_ERR_UNSUPPORTED; break; }
        if (hs.status == ST_NEED_BLAND) {
            LPArgs a = make_args(P, phase, tol, nn, cost);
            int rc = host_bland(P, a, st);
            if (rc != GOMILP_OK) { ret = rc; break; }
            // resume: the update kernel of the Bland step has been enqueued; read the state back after it
            HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(sync_stream());
            hs.done = 0; hs.status = ST_RUNNING;
            sync_state_to_device();
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

// Fused two-kernel pipeline (fused_kernels.hip).  Within a segment (since the last restart) launch index t:
//   K_A(t) reads  y = Y[(y0+t-1)&1], row p from B[(c0+t-1)&1]; writes Y[(y0+t)&1]      (t >= 1; t = 0: reads Y[y0])
//   K_B(t) reads  B[(c0+t-1)&1], writes B[(c0+t)&1]                                     (t >= 1; t = 0: reads B[c0])
// After T committed pivots the current buffers are B[(c0+T)&1], Y[(y0+T)&1] whatever stopped the loop.
int Engine::run_loop_fused(const Problem &P, int phase, double tol, int nn, const double *cost, gomilp_lp_stats *st) {
    Work &w = *w_;
    DevState &hs = *w.st_host;
    hs.done = 0; hs.status = ST_RUNNING; hs.pivots = 0; hs.q = hs.p = -1; hs.rq = hs.dp = hs.mv = 0;
    hs.max_pivots = 0;
    hs.lu_singular = 0;
    hs.theta = 0; hs.ent_cur = hs.ent_prev = hs.lea = -1;
    sync_state_to_device();
    int64_t since_refresh = 0;
    HIP_TRY(hipEventRecord(w.ev[0], stream_));
    int ret = GOMILP_OK;
    const bool sampling = sample_events_ > 0;
    int c0 = cur_, y0 = ycur_;
    int64_t seg_start = 0, tseg = 0;  // pivots committed at segment start, launch index inside the segment
    for (;;) {
        const int64_t before = hs.pivots;
        std::vector<int64_t> sample_t;
        int64_t nlaunch = chunk_;
        if (max_pivots_ > 0) nlaunch = std::min<int64_t>(nlaunch, std::max<int64_t>(1, max_pivots_ - hs.pivots + 1));
        for (int64_t t = 0; t < nlaunch; t++, tseg++) {
            const int pending = tseg > 0;
            LPArgs a = make_args(P, phase, tol, nn, cost);
            const int prev = (int)((tseg + 1) & 1);  // (x + tseg - 1) & 1 == (x + tseg + 1) & 1
            a.binv_cur = w.binv[pending ? ((c0 + prev) & 1) : c0];
            a.binv_next = w.binv[(c0 + tseg) & 1];
            const double *y_in = w.yb[pending ? ((y0 + prev) & 1) : y0];
            double *y_out = w.yb[(y0 + tseg) & 1];
            const bool sample = sampling && ((before + t) % sample_events_ == 0) && pending;
            hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
            if (sample) {
                const size_t e0 = sample_t.size() * 6;
                while (w.sample_ev.size() < e0 + 6) { hipEvent_t ev; HIP_TRY(hipEventCreate(&ev)); w.sample_ev.push_back(ev); }
                for (int k = 0; k < 4; k++) e[k] = w.sample_ev[e0 + k];
                sample_t.push_back(t);
            }
            const int gr = grid_ratio_;
            const int gp = launch_price_fused(a, y_in, y_out, pending, gr, stream_, e[0], e[1]);
            grid_ratio_ = launch_update_ftran_fused(a, pending, gp, stream_, e[2], e[3]);
            launches_ += 2;
        }
        HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        const int64_t executed = hs.pivots - before;
        since_refresh += executed;
        // a sampled launch index t measured K_A(t) (which commits pivot t-1) and K_B(t): valid while t <= executed
        account_samples(st, sample_t, executed + 1, 2);
        const int64_t T = hs.pivots - seg_start;
        if (!hs.done) {
            if (max_pivots_ > 0 && hs.pivots >= max_pivots_) { cur_ = (c0 + T) & 1; ycur_ = (y0 + T) & 1; ret = GOMILP_ERR_UNSUPPORTED; break; }
            if (refresh_ > 0 && since_refresh >= refresh_) {
                cur_ = (int)((c0 + T) & 1); ycur_ = (int)((y0 + T) & 1);
                refresh_xb_y(P, cost);
                since_refresh = 0;
                if (st) st->refreshes++;
            }
            continue;
        }
        cur_ = (int)((c0 + T) & 1);
        ycur_ = (int)((y0 + T) & 1);
        if (hs.status == ST_OPTIMAL) break;
        if (hs.status == ST_UNBOUNDED) { ret = GOMILP_ERR_UNBOUNDED; break; }
        if (hs.status == ST_NEED_BLAND) {
            // the degenerate pivot has not been committed: B^-1, x_B, y and the index lists are those of the
            // current basis; the Bland step runs on the unfused kernels, then a new fused segment starts
            LPArgs a = make_args(P, phase, tol, nn, cost);
            int rc = host_bland(P, a, st);
            if (rc != GOMILP_OK) { ret = rc; break; }
            HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
            HIP_TRY(sync_stream());
            hs.done = 0; hs.status = ST_RUNNING;
            sync_state_to_device();
            c0 = cur_; y0 = ycur_; seg_start = hs.pivots; tseg = 0;
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

// ------------------------------------------------------------------------------------------------
// Final solve x_B = ab^-1 b in gonum's order: LU on the device (k_lu_*), the two triangular solves
// of Dgetrs (lapack/gonum/dgetrs.go:37-45 -> blas/gonum/level3double.go:75-118) on the host, because
// the upper solve is one sequential dependency chain of m^2/2 rounded operations.
// ------------------------------------------------------------------------------------------------
// transpose: the system is ab^T y = rhs (the reference's BTRAN, simplex.go:236: LU of a materialised copy of ab.T()).  The
// column-major image of ab^T is the row-major image of ab, so the two gathers just change places; no column of ab^T is known
// to be a unit vector.  rhs_host (m entries, by basis position; default b): the right-hand side.
int Engine::final_solve(const Problem &P, int, std::vector<double> &x, bool *singular, const int32_t *basic_host, bool transpose,
                        const double *rhs_host) {
    int rc = lu_factor(P, singular, basic_host, transpose);
    if (rc != GOMILP_OK) return rc;
    return lu_solve(P, x, rhs_host);
}

// The factorization half of the final solve: gather, gonum-order LU on the device, the packed factors on the host (and, for the large
// bases, what the row kernel needs on the device).  Everything lu_solve needs stays in lu_cache_ / the work buffers until the next
// factorization: an exact step solves for x_B, for the entering column and for every Bland candidate from ONE factorization of ab
// (the reference factors again each time, simplex.go:289,315,356 — same matrix, same bits).
int Engine::lu_factor(const Problem &P, bool *singular, const int32_t *basic_host, bool transpose) {
    Work &w = *w_;
    const double tf0 = now_s();
    const int m = P.m, ldw = P.ld;
    int nonunit = 0;
    const bool compressed = lu_blocked_ >= 2 && lu_compressed_supported(m);
    lu_cache_.valid = false;
    // the compressed schedule keeps L/U column-major (lu_compressed.hip), the other two row-major
    if (compressed != transpose) launch_luc_gather(P.dAt, P.ld, m, w.basic, w.W, ldw, stream_);
    else {
        if (transpose) HIP_TRY(hipMemsetAsync(w.W, 0, (size_t)m * ldw * sizeof(double), stream_));   // (k_gather_w leaves the padding of a line alone)
        launch_gather_w(P.dAt, P.ld, m, w.basic, w.W, ldw, stream_);
    }
    // unit columns of ab (from the column statistics of the upload): the blocked LU skips their elimination steps
    if (!basic_host) {   // the caller has no host copy of the basis positions yet
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        basic_host = w.h_idx;
    }
    {
        std::vector<int32_t> ur(m);
        for (int pos = 0; pos < m; pos++) {
            const int j = basic_host[pos];
            ur[pos] = (!transpose && j < P.n && P.nnz[j] == 1 && P.allone[j]) ? P.lastrow[j] : -1;
            if (ur[pos] < 0) nonunit++;
        }
        int rcu = stage_upload(w.unitrow, ur.data(), (size_t)m * sizeof(int32_t));
        if (rcu != GOMILP_OK) return rcu;
    }
    LUArgs a;
    a.W = w.W; a.ldw = ldw; a.m = m; a.lpos = w.lpos; a.rowstep = w.rowstep;
    for (int t = 0; t < 2; t++) { a.pk[t] = w.lpk[t]; a.pl[t] = w.lpl[t]; a.pr[t] = w.lpr[t]; }
    a.st = w.st;
    a.unit_row = w.unitrow;
    const bool blocked = compressed || (lu_blocked_ && lu_blocked_supported(m));
    a.dense_flag = blocked ? w.denseflag : nullptr;
    a.ctl = w.luctl; a.Lp = w.luLp; a.Up = w.luUp;
    // look-ahead schedule: where a factorization takes many rounds (measured: 2048 rows 23 rounds 2.20 -> 2.05 ms, 1000 rows 12 rounds
    // 1.10 -> 1.03 ms on the device; 520-row children, 4 rounds: 0.63 -> 0.65 ms, and a wave runs dozens of them side by side)
    // Its launches wait for their own workgroups (bounded), like the loop kernels: two such launches side by side, or one beside a loop
    // kernel, can hold each other's workgroups off the CUs until a wait gives up (measured: four metric LPs finishing together, two of
    // four factorizations fell back after ~50 ms).  So it runs only while this engine holds the device's loop slots, all of them, and
    // only if they are free right now; a pool's workers never ask (knob lu_look).
    a.slots = 1; a.look = (lu_blocked_ >= 3 && lu_look_ && m > 768 && compressed) ? 1 : 0;
    struct LookSlot {
        int dev; bool held;
        LookSlot(int d, bool want) : dev(d), held(want && Engine::loop_try_acquire_all(d)) {}
        ~LookSlot() { drop(); }
        void drop() { if (held) Engine::loop_release(dev, 4, 0); held = false; }
    } look_slot(device_, a.look != 0);
    if (!look_slot.held) a.look = 0;
    // opt-in: the rows of a panel on the workgroups of one XCD (lu_cross.hip) — the plain schedule with that panel
    int cross_G = (lu_cross_ && compressed) ? luc_cross_groups(m, 1) : 0;
    if (cross_G) {
        if (!w.luxrec) {
            HIP_TRY(dmalloc(&w.luxrec, luc_cross_doubles()));
            HIP_TRY(hipMemsetAsync(w.luxrec, 0, luc_cross_doubles() * sizeof(double), stream_));
        }
        a.look = 0; look_slot.drop();
    }
    a.ctl_prev = a.ctl; a.Lp_prev = a.Lp; a.Up_prev = a.Up;
    a.rowsnap = w.rowstep + w.cap_m; a.rowsnap_prev = a.rowsnap;   // (launch_luc_rounds sets the round's parity)
    a.ctl_base = a.ctl; a.round = 0; a.pad3 = bt_fault_ == 2 ? 1 : 0;
    w.st_host->lu_singular = 0;
    if (!(compressed && m <= 128)) sync_state_to_device();   // (small bases: k_luc_init clears the flag on the device, the packed block brings it back)
    lu_rounds_ = 0;
    int32_t *h_dense = w.h_idx + w.cap_m;   // landing place of the dense-step flags (h_idx holds nm + nc entries; lpos lands in front)
    // Small bases: the packed factors are asked for together with the control block, in ONE host round trip and ONE copy — all m
    // columns (the compact list of dense columns would need the flags first; a unit-column step has zero multipliers and zero
    // off-diagonal U entries, which the solves skip like gonum's do: same bits, m*m instead of m*nd doubles over PCIe), with the
    // diagonal, the row positions, the flags and both control blocks behind them (k_luc_pack_small).  An exact step factors twice and
    // small trees are made of round trips and 3 us copies (60 per relaxation before this).  Should the batch of rounds turn out too
    // short, the general path takes over from where the rounds stand.
    bool oneshot = compressed && !cross_G && m <= 128 && luc_pack_small_bytes(m) <= ((size_t)w.cap_m * w.cap_ld + 512) * sizeof(double);   // (the block fits Wd / h_W: ensure_work)
    std::vector<int32_t> dl;
    auto enqueue_pack = [&](int nd2) -> int {
        int rcd = stage_upload(w.dlist, dl.data(), (size_t)nd2 * sizeof(int32_t));
        if (rcd != GOMILP_OK) return rcd;
        const bool split2 = compressed && m >= 1024 && nd2 > 0;
        if (split2) launch_luc_pack_dense(a, w.dlist, nd2, w.rho, w.Wd, w.ludiag, stream_);
        else if (compressed) launch_luc_pack(a, w.dlist, nd2, w.Wd, w.ludiag, stream_);
        else launch_lu_pack(a, w.dlist, nd2, w.Wd, w.ludiag, stream_);
        launches_++;
        if (nd2) HIP_TRY(hipMemcpyAsync(w.h_W, w.Wd, (size_t)(split2 ? nd2 : m) * nd2 * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.ludiag, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.lpos, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        if (compressed && m <= 128)   // (the state block was not uploaded in front of this factorization: only the flag comes back)
            HIP_TRY(hipMemcpyAsync(&w.st_host->lu_singular, &w.st->lu_singular, sizeof(w.st_host->lu_singular), hipMemcpyDeviceToHost, stream_));
        else
            HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        return GOMILP_OK;
    };
    if (compressed) {
        // rounds are data dependent (lu_compressed.hip): enqueue a batch, read the control block, repeat.  The first
        // batch is sized from the number of columns that are dense for sure.
        const int nb = lu_compressed_nb(m, a.slots != 0);
        for (int attempt = 0;; attempt++) {
            launch_luc_init(a, stream_);
            launches_++;
            // measured: steps that do arithmetic ~ 3 x the non-unit columns (each of them usually turns a unit column dense)
            // (the slot panel takes up to nb steps per round whatever the order of the columns; a wrong guess costs one more look at the
            // control block, a generous one a run of empty rounds)
            int batch = a.slots ? std::max(1, (std::min(m, (5 * nonunit) / 2) + nb - 1) / nb) : std::max(1, (3 * nonunit + nb - 1) / nb + 1);
            if (oneshot && GOMILP_DBG_ENV("GOMILP_DEBUG_LU_SHORT")) batch = 1;   // (diagnostic flavour: a first batch that is too short — the small-basis block comes too early and the general path takes over)
            int enq = 0;   // rounds enqueued so far: the look-ahead schedule keeps two control blocks, by round parity
            int k_seen = -1;   // steps done when the control block was last read
            const LUCtl *last = w.luctl_host;
            for (;;) {
                launches_ += cross_G ? launch_luc_rounds_cross(a, w.rho, batch, w.luxrec, cross_G, stream_) : launch_luc_rounds(a, w.rho, batch, enq, stream_);
                enq += batch;
                if (oneshot) {
                    launch_luc_pack_small(a, w.Wd, stream_);
                    launches_++;
                    HIP_TRY(hipMemcpyAsync(w.h_W, w.Wd, luc_pack_small_bytes(m), hipMemcpyDeviceToHost, stream_));
                    HIP_TRY(sync_stream());
                    const double *blk = w.h_W;
                    const int32_t *io = reinterpret_cast<const int32_t *>(blk + (size_t)m * m + m);
                    memcpy(w.h_vec, blk + (size_t)m * m, (size_t)m * sizeof(double));   // diag
                    memcpy(w.h_idx, io, (size_t)m * sizeof(int32_t));                   // lpos
                    memcpy(h_dense, io + m, (size_t)m * sizeof(int32_t));
                    memcpy(w.luctl_host, io + 2 * m, 2 * sizeof(LUCtl));
                    w.st_host->lu_singular = io[2 * m + (int)(2 * sizeof(LUCtl) / sizeof(int32_t))];
                    dl.resize(m);
                    for (int k = 0; k < m; k++) dl[k] = k;
                } else {
                    HIP_TRY(hipMemcpyAsync(w.luctl_host, w.luctl, 2 * sizeof(LUCtl), hipMemcpyDeviceToHost, stream_));
                    // the dense-step flags ride along (final once k_next == m): no separate round trip for them
                    HIP_TRY(hipMemcpyAsync(h_dense, w.denseflag, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
                    HIP_TRY(sync_stream());
                }
                last = w.luctl_host + (a.look ? ((enq - 1) & 1) : 0);
                if (w.luctl_host[0].fault || last->k_next >= m) break;
                // (every round performs at least the step it starts at — a listed column or a bookkeeping step; a batch that moved nothing
                // would repeat for ever: report it instead)
                if (last->k_next <= k_seen) return GOMILP_ERR_DEVICE;
                k_seen = last->k_next;
                oneshot = false;   // (the batch was too short: that pack came too early)
                batch = std::max(4, (int)(((int64_t)(m - last->k_next) * last->rounds) / std::max(1, last->k_next)) + 2);
                if (batch > 64) batch = 64;
            }
            lu_rounds_ = last->rounds;
            if (cross_G && !w.luctl_host[0].fault) { launch_luc_lpos_final(a, stream_); launches_++; }   // (that panel keeps its maps up to the last tied search only)
            look_slot.drop();   // (the rounds are behind the last sync)
            if (!w.luctl_host[0].fault) break;
            // a wait inside a look-ahead launch ran out of patience (its workgroups never became resident together): once more, from the
            // basis, with the whole update behind each panel
            if (attempt > 0 || !(a.look || cross_G)) return GOMILP_ERR_DEVICE;
            a.look = 0; cross_G = 0; oneshot = false;
            lu_look_faults_++; lu_look_fault_ = true;
            if (GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) fprintf(stderr, "final_solve: a look-ahead launch gave up a wait (m %d, rounds enqueued %d, cnt_x %u cnt_u %u cnt_s %u): plain schedule\n", m, enq, w.luctl_host[0].cnt_x, w.luctl_host[0].cnt_u, w.luctl_host[0].cnt_s);
            if (compressed != transpose) launch_luc_gather(P.dAt, P.ld, m, w.basic, w.W, ldw, stream_);
            else {
                if (transpose) HIP_TRY(hipMemsetAsync(w.W, 0, (size_t)m * ldw * sizeof(double), stream_));
                launch_gather_w(P.dAt, P.ld, m, w.basic, w.W, ldw, stream_);
            }
        }
    } else if (blocked) launches_ += launch_lu_blocked(a, w.rho, stream_) + 1;
    else { launch_lu(a, stream_); launches_ += m + 2; }
    // Only the columns whose elimination step did arithmetic carry non-zero L / off-diagonal U entries (a unit-column
    // step has zero multipliers and its column is zero in every earlier pivot row), so the host solves need those
    // columns and the diagonal only: m*(nd+1) doubles cross PCIe instead of m*m.
    int nd;
    bool split;
    int ndense = 0;   // steps that did arithmetic (stats)
    if (oneshot) {    // everything is on the host already
        nd = m; split = false;
        for (int k = 0; k < m; k++) ndense += h_dense[k] != 0;
    } else {
        dl.clear();
        if (blocked) {
            if (!compressed) {
                HIP_TRY(hipMemcpyAsync(h_dense, w.denseflag, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(sync_stream());
            }
            for (int k = 0; k < m; k++) if (h_dense[k]) dl.push_back(k);
        } else {
            for (int k = 0; k < m; k++) dl.push_back(k);
        }
        nd = (int)dl.size();
        ndense = nd;
        // large bases: only the nd x nd part that couples the dense positions goes to the host (lu_compressed.hip,
        // k_luc_pack_dense / k_luc_solve_rows); small ones take one host pass over all rows (one round trip fewer)
        split = compressed && m >= 1024 && nd > 0;
        int rcp = enqueue_pack(nd);
        if (rcp != GOMILP_OK) return rcp;
        HIP_TRY(sync_stream());
    }
    lu_dense_ = ndense;
    HIP_TRY(hipGetLastError());
    const double tf1 = now_s();
    fs_device_ += tf1 - tf0;
    std::vector<int32_t> phys(m);
    for (int R = 0; R < m; R++) phys[w.h_idx[R]] = R;
    const double *diag = w.h_vec;
    // LU.Det() == 0 (mat/lu.go:301, :118-135): exp(sum log|u_ii|) == 0.  The sum of the binary exponents brackets it ( |u| in
    // [2^(e-1), 2^e) ): only a product that could underflow (or a zero / NaN on the diagonal) pays for the m logarithms in gonum's order.
    double logdet = 0;
    {
        long long esum = 0;
        bool plain = true;
        for (int i = 0; i < m; i++) {
            int e = 0;
            const double d = diag[i];
            if (!(fabs(d) > 0) || !std::isfinite(d)) { plain = false; break; }
            (void)frexp(d, &e);
            esum += e;
        }
        // sum log|u_ii| >= (esum - m) ln 2, minus rounding of m additions; exp() is zero below -745.14
        if (plain && (double)(esum - m) * 0.6931471805599453 > -700.0) logdet = 0;   // (exp(logdet) != 0 for sure; the value itself is not used)
        else for (int i = 0; i < m; i++) logdet += log(fabs(diag[phys[i]]));
    }
    *singular = w.st_host->lu_singular != 0 || exp(logdet) == 0;
    if (*singular && GOMILP_DBG_ENV("GOMILP_DEBUG_LOOP")) {
        int nz = 0; double dmin = 1e300;
        for (int i = 0; i < m; i++) { if (diag[phys[i]] == 0) nz++; dmin = std::min(dmin, fabs(diag[phys[i]])); }
        fprintf(stderr, "final_solve: singular (transpose %d, m %d, nd %d, rounds %lld, lu_singular flag %d, logdet %g, zero diagonals %d, min |u_ii| %g, compressed %d)\n",
                (int)transpose, m, nd, (long long)lu_rounds_, (int)w.st_host->lu_singular, logdet, nz, dmin, (int)compressed);
    }
    lu_cache_.m = m; lu_cache_.nd = nd; lu_cache_.split = split; lu_cache_.singular = *singular;
    lu_cache_.phys = phys; lu_cache_.dl = dl;
    lu_cache_.diag.assign(diag, diag + m);   // (h_vec is everybody's landing buffer)
    lu_cache_.args = a;
    lu_cache_.valid = true;
    fs_host_ += now_s() - tf1;
    return GOMILP_OK;
}

// The solve half (Dgetrs, lapack/gonum/dgetrs.go:37-45) from the factors lu_factor left: rhs_host (m entries, by basis position;
// default b).  A singular factorization gives zeros (the caller has the flag).
int Engine::lu_solve(const Problem &P, std::vector<double> &x, const double *rhs_host) {
    Work &w = *w_;
    if (!lu_cache_.valid || lu_cache_.m != P.m) return GOMILP_ERR_DEVICE;
    const double tf1 = now_s();
    const int m = P.m, nd = lu_cache_.nd;
    const bool split = lu_cache_.split;
    const std::vector<int32_t> &phys = lu_cache_.phys, &dl = lu_cache_.dl;
    const double *diag = lu_cache_.diag.data();
    const LUArgs &a = lu_cache_.args;
    const double *rhs = rhs_host ? rhs_host : P.hb.data();
    x.assign(m, 0.0);
    if (lu_cache_.singular) return GOMILP_OK;
    auto term = [](double bi, double va, double xk) { return va != 0 ? (-va) * xk + bi : bi; };
    if (split) {
        const double *rhs_dev = P.db;
        if (rhs_host) {   // the row kernel reads the right-hand side on the device
            int rcr = stage_upload(w.move, rhs_host, (size_t)m * sizeof(double));
            if (rcr != GOMILP_OK) return rcr;
            rhs_dev = w.move;
        }
        // coupled part on the host: row s of h_W is dense position dl[s] restricted to the dense columns
        std::vector<double> xdl(nd), xdu(nd);
        // Dtrsm(Left, Lower, NoTrans, Unit): a row is one chain of dependent rounded additions in ascending t; four rows run side by
        // side over the part of the solution that is known before the first of them, then finish one after the other
        int s2 = 0;
        for (; s2 + 4 <= nd; s2 += 4) {
            const double *r0 = w.h_W + (size_t)s2 * nd, *r1 = r0 + nd, *r2 = r1 + nd, *r3 = r2 + nd;
            double b0 = rhs[phys[dl[s2]]], b1 = rhs[phys[dl[s2 + 1]]], b2 = rhs[phys[dl[s2 + 2]]], b3 = rhs[phys[dl[s2 + 3]]];
            for (int t = 0; t < s2; t++) {
                const double xk = xdl[t];
                b0 = term(b0, r0[t], xk); b1 = term(b1, r1[t], xk); b2 = term(b2, r2[t], xk); b3 = term(b3, r3[t], xk);
            }
            xdl[s2] = b0;
            b1 = term(b1, r1[s2], b0); xdl[s2 + 1] = b1;
            b2 = term(b2, r2[s2], b0); b2 = term(b2, r2[s2 + 1], b1); xdl[s2 + 2] = b2;
            b3 = term(b3, r3[s2], b0); b3 = term(b3, r3[s2 + 1], b1); b3 = term(b3, r3[s2 + 2], b2); xdl[s2 + 3] = b3;
        }
        for (; s2 < nd; s2++) {
            const double *row = w.h_W + (size_t)s2 * nd;
            double bi = rhs[phys[dl[s2]]];
            for (int t = 0; t < s2; t++) bi = term(bi, row[t], xdl[t]);
            xdl[s2] = bi;
        }
        for (int s2 = nd - 1; s2 >= 0; s2--) {   // Dtrsm(Left, Upper, NoTrans, NonUnit)
            const double *row = w.h_W + (size_t)s2 * nd;
            double bi = xdl[s2];
            for (int t = s2 + 1; t < nd; t++) bi = term(bi, row[t], xdu[t]);
            const double tinv = 1 / diag[phys[dl[s2]]];
            xdu[s2] = bi * tinv;
        }
        fs_host_ += now_s() - tf1;
        const double tf2 = now_s();
        double *dxl = w.yscratch, *dxu = w.yscratch + P.ld, *dx = w.yscratch + 2 * (size_t)P.ld;   // 64 * ld doubles
        int rcs = stage_upload(dxl, xdl.data(), (size_t)nd * sizeof(double));
        if (rcs == GOMILP_OK) rcs = stage_upload(dxu, xdu.data(), (size_t)nd * sizeof(double));
        if (rcs != GOMILP_OK) return rcs;
        launch_luc_solve_rows(a, w.dlist, nd, rhs_dev, dxl, dxu, dx, stream_);
        launches_++;
        HIP_TRY(hipMemcpyAsync(w.h_vec, dx, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        for (int i = 0; i < m; i++) x[i] = w.h_vec[i];
        for (int s2 = 0; s2 < nd; s2++) x[dl[s2]] = xdu[s2];
        fs_device_ += now_s() - tf2;
        return GOMILP_OK;
    }
    // Dlaswp: b in logical row order
    for (int i = 0; i < m; i++) x[i] = rhs[phys[i]];
    // The two Dtrsm of Dgetrs, per row in gonum's order: ascending k, zero multipliers skipped, b_i = (-a_ik)*b_k + b_i
    // as a rounded multiply and a rounded add (level3double.go:75-118).  Only the nd columns whose elimination step did
    // arithmetic carry off-diagonal entries, so a row depends on the solution at those "dense" positions only: they are
    // solved first, one after the other; every other row is then independent of the rest and four of them run
    // interleaved (each row is one chain of dependent rounded additions: a single chain leaves the FPU idle).
    std::vector<double> xd(nd);
    std::vector<char> isd(m, 0);
    for (int t = 0; t < nd; t++) isd[dl[t]] = 1;
    std::vector<int> nl;   // logical positions that are not dense, ascending
    nl.reserve(m - nd);
    for (int i = 0; i < m; i++) if (!isd[i]) nl.push_back(i);
    // ---- Dtrsm(Left, Lower, NoTrans, Unit)
    for (int s2 = 0; s2 < nd; s2++) {
        const int i = dl[s2];
        const double *row = w.h_W + (size_t)phys[i] * nd;
        double bi = x[i];
        for (int t = 0; t < s2; t++) bi = term(bi, row[t], xd[t]);
        x[i] = bi; xd[s2] = bi;
    }
    {
        size_t g = 0;
        int cnt = 0;   // dense positions below the current row
        for (; g + 4 <= nl.size(); g += 4) {
            int c[4];
            const double *row[4];
            double acc[4];
            for (int r = 0; r < 4; r++) {
                const int i = nl[g + r];
                while (cnt < nd && dl[cnt] < i) cnt++;
                c[r] = cnt; row[r] = w.h_W + (size_t)phys[i] * nd; acc[r] = x[i];
            }
            const int c0 = c[0];   // c[0] <= c[1] <= c[2] <= c[3]
            for (int t = 0; t < c0; t++) {
                const double xk = xd[t];
                acc[0] = term(acc[0], row[0][t], xk); acc[1] = term(acc[1], row[1][t], xk);
                acc[2] = term(acc[2], row[2][t], xk); acc[3] = term(acc[3], row[3][t], xk);
            }
            for (int r = 1; r < 4; r++)
                for (int t = c0; t < c[r]; t++) acc[r] = term(acc[r], row[r][t], xd[t]);
            for (int r = 0; r < 4; r++) x[nl[g + r]] = acc[r];
        }
        for (; g < nl.size(); g++) {
            const int i = nl[g];
            while (cnt < nd && dl[cnt] < i) cnt++;
            const double *row = w.h_W + (size_t)phys[i] * nd;
            double bi = x[i];
            for (int t = 0; t < cnt; t++) bi = term(bi, row[t], xd[t]);
            x[i] = bi;
        }
    }
    // ---- Dtrsm(Left, Upper, NoTrans, NonUnit): rows from the bottom, ascending k within a row, then * (1/u_ii)
    for (int s2 = nd - 1; s2 >= 0; s2--) {
        const int i = dl[s2];
        const double *row = w.h_W + (size_t)phys[i] * nd;
        double bi = x[i];
        for (int t = s2 + 1; t < nd; t++) bi = term(bi, row[t], xd[t]);
        const double tinv = 1 / diag[phys[i]];
        x[i] = bi * tinv; xd[s2] = x[i];
    }
    {
        size_t g = 0;
        int first = 0;   // first dense position above the current row
        for (; g + 4 <= nl.size(); g += 4) {
            int f[4];
            const double *row[4];
            double acc[4];
            for (int r = 0; r < 4; r++) {
                const int i = nl[g + r];
                while (first < nd && dl[first] < i) first++;
                f[r] = first; row[r] = w.h_W + (size_t)phys[i] * nd; acc[r] = x[i];
            }
            const int f3 = f[3];   // f[0] <= f[1] <= f[2] <= f[3]: the leading terms of rows 0..2 come first, in order
            for (int r = 0; r < 3; r++)
                for (int t = f[r]; t < f3; t++) acc[r] = term(acc[r], row[r][t], xd[t]);
            for (int t = f3; t < nd; t++) {
                const double xk = xd[t];
                acc[0] = term(acc[0], row[0][t], xk); acc[1] = term(acc[1], row[1][t], xk);
                acc[2] = term(acc[2], row[2][t], xk); acc[3] = term(acc[3], row[3][t], xk);
            }
            for (int r = 0; r < 4; r++) {
                const int i = nl[g + r];
                const double tinv = 1 / diag[phys[i]];
                x[i] = acc[r] * tinv;
            }
        }
        for (; g < nl.size(); g++) {
            const int i = nl[g];
            while (first < nd && dl[first] < i) first++;
            const double *row = w.h_W + (size_t)phys[i] * nd;
            double bi = x[i];
            for (int t = first; t < nd; t++) bi = term(bi, row[t], xd[t]);
            const double tinv = 1 / diag[phys[i]];
            x[i] = bi * tinv;
        }
    }
    fs_host_ += now_s() - tf1;
    return GOMILP_OK;
}

// ------------------------------------------------------------------------------------------------
// simplex() — simplex.go:93-302
// ------------------------------------------------------------------------------------------------
int Engine::solve(int64_t id, double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                  int64_t *basis_out, gomilp_lp_stats *stats) {
    std::lock_guard<std::mutex> g(mu_);
    xchg_timeout_ = false;
    int rc = solve_locked(id, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
    if (rc == GOMILP_ERR_DEVICE && xchg_timeout_ && bt_groups_ >= 0) {
        // The workgroups of the multi-workgroup block kernel wait for each other; when one of them was not resident within the
        // bounded wait (a crowded device) the launch gives up.  That says nothing about the problem: once more on the
        // single-workgroup kernels, which depend on nobody.
        const int64_t keep = bt_groups_;
        bt_groups_ = -1;
        rc = solve_locked(id, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
        bt_groups_ = keep;
        if (stats) stats->device_retries = 1;
    }
    return rc;
}

int Engine::solve_locked(int64_t id, double tol, const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x,
                         int64_t *basis_out, gomilp_lp_stats *stats) {
    const double t0 = now_s();
    gomilp_lp_stats local;
    gomilp_lp_stats *st = stats ? stats : &local;
    memset(st, 0, sizeof(*st));
    st->device_id = device_;
    const double nan = std::numeric_limits<double>::quiet_NaN();
    const double inf = std::numeric_limits<double>::infinity();
    if (has_x) *has_x = 0;
    if (opt_f) *opt_f = nan;
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id] || !opt_f || !opt_x || !has_x) return GOMILP_ERR_BAD_SHAPE;
    const Problem &P = *problems_[id];
    st->seconds_upload = P.seconds_upload;
    auto finish = [&](int code) {
        st->seconds_total = now_s() - t0; st->kernel_launches = launches_;
        st->seconds_final_device = fs_device_; st->seconds_final_host = fs_host_;
        st->lu_dense_steps = lu_dense_; st->lu_rounds = lu_rounds_;
        if (lu_look_fault_) st->device_retries = 1;   // (the final solve was repeated with the plain LU schedule)
        return code;
    };
    launches_ = 0;
    fs_device_ = fs_host_ = 0; lu_look_fault_ = false;
    last_trace_.clear();
    last_trace_total_ = 0;
    if (P.verify_status != GOMILP_OK) {  // simplex.go:94-100
        if (P.verify_status == GOMILP_ERR_UNBOUNDED) *opt_f = -inf;
        return finish(P.verify_status);
    }
    const int m = P.m, n = P.n;
    int rc = ensure_work(m, n + 1);
    if (rc != GOMILP_OK) return finish(rc);
    Work &w = *w_;
    w.st_host->trace_len = 0;
    std::vector<double> xb_exact;
    bool singular = false;

    if (m == n) {  // simplex.go:103-119: exactly constrained, one linear solve
        std::vector<int32_t> ident(n);
        for (int j = 0; j < n; j++) ident[j] = j;
        if ((rc = upload_index_lists(ident, {})) != GOMILP_OK) return finish(rc);
        const double t1 = now_s();
        if ((rc = final_solve(P, n, xb_exact, &singular)) != GOMILP_OK) return finish(rc);
        st->seconds_final_solve = now_s() - t1;
        if (singular) return finish(GOMILP_ERR_SINGULAR);
        if (cond_guard_ && !P.hA.empty() && n <= 1024) {   // cond > 1e16 is a Condition too: lp.ErrSingular (simplex.go:109-112)
            st->cond_fallbacks++;
            const double kinf = general_cond_inf(P.hA, n);
            if (kinf > 1e16 || kinf != kinf) return finish(GOMILP_ERR_SINGULAR);
        }
        for (int j = 0; j < n; j++)
            if (xb_exact[j] < 0) return finish(GOMILP_ERR_INFEASIBLE);
        *opt_f = dot_unitary(xb_exact.data(), P.hc.data(), n);
        memcpy(opt_x, xb_exact.data(), sizeof(double) * (size_t)n);
        *has_x = 1;
        return finish(GOMILP_OK);
    }
    if (m > n) return finish(GOMILP_ERR_SINGULAR);  // findLinearlyIndependent cannot reach m columns (:495-497)

    // findLinearlyIndependent (simplex.go:611-637), unit-column fast path: the descending scan meets m distinct
    // unit vectors (always true for GoMILP's [.. | I] standard forms, subproblem.go:81-139); cond == 1 there.
    std::vector<int32_t> basic(m), rho(m);
    bool unit_basis = true;
    if (initial_basic) {
        // supplied basis (simplex.go:147-160; GoMILP itself always passes nil): the caller hands over m entries; an index
        // out of range panics in extractColumns, a singular or infeasible set panics in initializeFromBasic (:447-471)
        unit_basis = false;
        for (int pos = 0; pos < m; pos++) {
            const int64_t j = initial_basic[pos];
            if (j < 0 || j >= n) return finish(GOMILP_ERR_PANIC);
            basic[pos] = (int32_t)j;
        }
    } else {
        std::vector<char> used(m, 0);
        for (int pos = 0; pos < m; pos++) {
            const int j = n - 1 - pos;
            if (!(P.nnz[j] == 1 && P.allone[j]) || used[P.lastrow[j]]) { unit_basis = false; break; }
            rho[pos] = P.lastrow[j]; used[rho[pos]] = 1; basic[pos] = j;
        }
    }
    gen_start_ = !unit_basis;
    gen_binv_dev_ = false;
    badly_scaled_ = P.scale_span > 1e9;
    const int nn_max = n + 1 - m;
    // a non-slack starting basis (equality rows, supplied basis) always takes the tableau pipelines: their set-up accepts
    // any B^-1; the n - m < 2m rule is only the bytes-per-pivot trade-off between the two formulations
    const bool use_tab = tableau_ && ((n - m) < 2 * m || !unit_basis) && (size_t)tab_ld(nn_max) * sizeof(double) <= 64 * 1024;
    std::vector<double> xb(m, 0.0), binv_host;
    bool feasible = true;
    if (unit_basis) {
        // ab = permutation, xb = ab^-1 b exactly (initializeFromBasic, simplex.go:447-471)
        for (int pos = 0; pos < m; pos++) { xb[pos] = P.hb[rho[pos]]; if (xb[pos] < -1e-13) feasible = false; }
    } else {
        // general case (engine_general.cpp): host search over a kept copy of A, small problems on the tableau pipelines
        if (!use_tab || !ensure_host_A(P)) return finish(GOMILP_ERR_UNSUPPORTED);
        const double t_g0 = now_s();
        if (!initial_basic) {
            // 96 rows and more (knob general_min_rows): the scan runs on the device — whole solves with the host / the blocked device search,
            // tools/general_small.py: 64 rows 0.96 / 1.08 ms, 96: 1.54 / 1.51, 128: 2.28 / 1.85, 180: 4.5 / 2.9, 224: 6.8 / 3.4, 300: 11.1 / 4.5
            // (until round 5, five launches per candidate: 224 rows)
            rc = (m >= general_min_rows_ && general_device_) ? find_independent_device(P, basic, &binv_host, &gen_binv_dev_) : general_find_linearly_independent(P.hA, m, n, basic, &binv_host);
            if (rc != GOMILP_OK) return finish(rc);  // ErrSingular, simplex.go:495-497
        }
        const double t_g1 = now_s();
        if (!gen_binv_dev_ && binv_host.size() != (size_t)m * m && !general_basis_inverse(P.hA, m, n, basic, n, std::vector<double>(), binv_host))
            return finish(initial_basic ? GOMILP_ERR_PANIC : GOMILP_ERR_SINGULAR);
        const double t_g2 = now_s();
        // xb = ab^-1 b with the reference's own arithmetic (gonum-order LU on the device): the feasibility test of
        // simplex.go:459-469 then sees the same bits
        if ((rc = upload_index_lists(basic, {})) != GOMILP_OK) return finish(rc);
        bool sing = false;
        if ((rc = final_solve(P, n, xb_exact, &sing)) != GOMILP_OK) return finish(rc);
        if (sing) { feasible = false; }  // "singular" also sends the reference to Phase I (simplex.go:504-507), xb stays zero
        else { xb = xb_exact; for (int pos = 0; pos < m; pos++) if (xb[pos] < -1e-13) feasible = false; }
        if (initial_basic && !feasible) return finish(GOMILP_ERR_PANIC);  // initializeFromBasic errors panic (:156-158)
        if (GOMILP_DBG_ENV("GOMILP_DEBUG_GS")) fprintf(stderr, "general start: search %.2f ms, host inverse %.2f ms, x_B (gonum-order LU) %.2f ms\n", 1e3 * (t_g1 - t_g0), 1e3 * (t_g2 - t_g1), 1e3 * (now_s() - t_g2));
    }
    // small bases starting feasible: record the pivots for the host replay of the reference's condition guards
    // (every pipeline: until round 5 the revised-simplex pipelines — shapes with n - m >= 2m, wide small LPs among them — went without the
    // replay, which is what the one status difference of the badly scaled family, seed 1079, a 2 x 6 LP, came from)
    shadow_trace_ = cond_guard_ && feasible && m <= 64 && !initial_basic && ensure_host_A(P);
    const std::vector<int32_t> basic_start = basic;
    cur_ = 0;
    if (unit_basis && !use_tab) {
        HIP_TRY(hipMemsetAsync(w.binv[0], 0, (size_t)m * P.ld * sizeof(double), stream_));
        if ((rc = stage_upload(w.rho, rho.data(), (size_t)m * sizeof(int32_t))) != GOMILP_OK) return finish(rc);
        launch_set_binv_perm(w.binv[0], P.ld, m, w.rho, stream_);
    }
    ycur_ = 0;
    if (!use_tab) {   // the duals only exist on the revised-simplex pipelines
        HIP_TRY(hipMemsetAsync(w.yb[0], 0, (size_t)P.ld * sizeof(double), stream_));
        HIP_TRY(hipMemsetAsync(w.yb[1], 0, (size_t)P.ld * sizeof(double), stream_));
    }
    {   // x_B with its zero padding in one copy
        std::vector<double> xpad(P.ld, 0.0);
        std::copy(xb.begin(), xb.begin() + m, xpad.begin());
        if ((rc = stage_upload(w.xb, xpad.data(), (size_t)P.ld * sizeof(double))) != GOMILP_OK) return finish(rc);
    }

    // pipeline choice: the explicit tableau moves 16*m*(n-m) bytes per pivot in one launch, the revised form
    // 8*[m(n-m) + 2m^2] in two: the tableau wins while n - m < 2m
    use_bt_ = use_tab && blocked_ && bt_supported(m, nn_max);
    st->pipeline = use_tab ? (use_bt_ ? 3 : 2) : ((fused_ && fused_supported(P.ld)) ? 1 : 0);
    int loop_rc = GOMILP_OK;
    if (use_tab) {
        const int ldt = tab_ld(nn_max);
        if (w.cap_T < (size_t)(m + 3) * ldt || w.cap_ldt < ldt) {   // + 3 rows: the tiled layout pads m to a multiple of 4
            for (double **pp : {&w.T[0], &w.T[1], &w.R[0], &w.R[1], &w.tscratch, &w.btV}) { if (*pp) hipFree(*pp); *pp = nullptr; }
            if (w.srcpos) hipFree(w.srcpos); w.srcpos = nullptr;
            const size_t cap = std::max(w.cap_T, (size_t)(m + 3 + 32 + m / 16) * ldt);   // head-room for deeper children
            const int cl = std::max(w.cap_ldt, ldt);
            HIP_TRY(dmalloc(&w.T[0], cap)); HIP_TRY(dmalloc(&w.T[1], cap));
            HIP_TRY(dmalloc(&w.R[0], (size_t)cl)); HIP_TRY(dmalloc(&w.R[1], (size_t)cl));
            HIP_TRY(dmalloc(&w.tscratch, (size_t)64 * cl)); HIP_TRY(dmalloc(&w.srcpos, (size_t)cl));
            HIP_TRY(dmalloc(&w.btV, (size_t)bt_max_k() * cl));
            w.cap_T = cap; w.cap_ldt = cl; w.cap_btU = 0;
        }
        // the rank-1 u terms are rows of P.ld doubles: sized by the ROW count, which can grow while the T buffers
        // (sized by m * ldt) still fit — a wide problem followed by a taller, narrower one on the same context
        if (w.cap_btU < (size_t)bt_max_k() * (size_t)P.ld) {
            if (w.btU) hipFree(w.btU); w.btU = nullptr;
            const size_t cap = (size_t)bt_max_k() * (size_t)std::max(w.cap_ld, P.ld);
            HIP_TRY(dmalloc(&w.btU, cap));
            w.cap_btU = cap;
        }
        if (use_bt_ && !w.xbuf) {   // exchange records of the multi-workgroup block kernel: zero = no exchange has happened
            HIP_TRY(dmalloc(&w.xbuf, bt_xbuf_doubles()));
            HIP_TRY(hipMemsetAsync(w.xbuf, 0, bt_xbuf_doubles() * sizeof(double), stream_));
        }
        if (!use_bt_) {   // the blocked kernels never read the padding of r
            HIP_TRY(hipMemsetAsync(w.R[0], 0, (size_t)w.cap_ldt * sizeof(double), stream_));
            HIP_TRY(hipMemsetAsync(w.R[1], 0, (size_t)w.cap_ldt * sizeof(double), stream_));
        }
        rc = solve_tableau(P, tol, basic, rho, xb, feasible, st, &loop_rc, unit_basis ? nullptr : &binv_host);
        if (rc != GOMILP_OK) return finish(rc);
    } else {
    std::vector<int32_t> nonbasic;
    auto build_nonbasic = [&](int ncols) {  // simplex.go:174-184: ascending ids not in the basis
        std::vector<char> inb(ncols, 0);
        for (int i = 0; i < m; i++) inb[basic[i]] = 1;
        nonbasic.clear();
        for (int j = 0; j < ncols; j++) if (!inb[j]) nonbasic.push_back(j);
    };

    if (!feasible) {
        // ---- Phase I (simplex.go:529-606) ----
        st->phase1_used = 1;
        const int64_t minidx = min_idx(xb.data(), m);
        // a_{n+1} = b - sum_{i != minidx} a_{basic_i}: for unit columns one exact "- 1" per row (floats.Sub, :536-542)
        std::vector<double> art(P.ld, 0.0);
        for (int k = 0; k < m; k++) art[k] = P.hb[k];
        for (int i = 0; i < m; i++) { if (i == minidx) continue; art[rho[i]] = -1 * 1.0 + art[rho[i]]; }
        bool art_zero = true;
        for (int k = 0; k < m; k++) if (art[k] != 0) { art_zero = false; break; }
        if (art_zero) { st->wrapped_status = GOMILP_ERR_ZERO_COLUMN; return finish(GOMILP_ERR_PHASE1_WRAPPED); }  // verifyInputs of the recursive call
        HIP_TRY(hipMemcpyAsync(P.dAt + (size_t)n * P.ld, art.data(), (size_t)P.ld * sizeof(double), hipMemcpyHostToDevice, stream_));
        HIP_TRY(sync_stream());
        // basis := slack basis with position minidx replaced by the artificial: one forced pivot builds its inverse
        w.st_host->done = 0; w.st_host->status = ST_RUNNING; w.st_host->pivots = 0; w.st_host->max_pivots = 0; w.st_host->rq = 0;
        sync_state_to_device();
        {
            LPArgs a = make_args(P, 1, 1e-10, 0, P.dc1);
            launch_ftran(a, 0, -1, n, stream_);
            launch_update(a, 0, (int)minidx, 1, 0, stream_);
            launches_ += 2;
            cur_ ^= 1;
        }
        basic[minidx] = n;
        build_nonbasic(n + 1);
        if ((rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return finish(rc);
        refresh_xb_y(P, P.dc1);  // xb = ab^-1 b (initializeFromBasic of the recursive call), y = ab^-T cb
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        for (int i = 0; i < m; i++) if (w.h_vec[i] < -1e-13) return finish(GOMILP_ERR_PANIC);  // simplex.go:155-158
        rc = run_loop(P, 1, 1e-10, (int)nonbasic.size(), P.dc1, st);
        if (rc == GOMILP_ERR_DEVICE) return finish(rc);
        if (rc != GOMILP_OK) { st->wrapped_status = rc; return finish(GOMILP_ERR_PHASE1_WRAPPED); }  // :557-559
        HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(w.h_vec, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        int added = -1;
        for (int i = 0; i < m; i++) { basic[i] = w.h_idx[i]; xb[i] = w.h_vec[i]; if (basic[i] == n) added = i; }
        double xart = added >= 0 ? xb[added] : 0.0;
        if (added >= 0 && fabs(xart) > 1e-13 && fabs(xart) < 1e-11) {
            // too close to phaseIZeroTol to trust the updated x_B: take the reference's own value (fresh gonum-order solve)
            if ((rc = final_solve(P, n + 1, xb_exact, &singular)) != GOMILP_OK) return finish(rc);
            if (!singular) xart = xb_exact[added];
        }
        if (fabs(xart) > 1e-12) return finish(GOMILP_ERR_INFEASIBLE);  // phaseIZeroTol, :563-565
        if (added >= 0) {
            // :581-606 the artificial stayed basic at zero: exchange it for the first nonbasic column that keeps
            // the basis nonsingular and feasible.  Guard on the pivot element instead of the LU condition estimate.
            bool exchanged = false;
            std::vector<char> inb(n + 1, 0);
            for (int i = 0; i < m; i++) inb[basic[i]] = 1;
            for (int j = 0; j < n && !exchanged; j++) {
                if (inb[j]) continue;
                w.st_host->done = 0; w.st_host->status = ST_RUNNING; w.st_host->rq = 0;
                sync_state_to_device();
                LPArgs a = make_args(P, 1, 1e-10, 0, P.dc1);
                launch_ftran(a, 0, -1, j, stream_);
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
                launch_update(a, 0, added, 1, 0, stream_);
                launches_++;
                cur_ ^= 1;
                basic[added] = j;
                exchanged = true;
                st->art_exchanges++;
            }
            if (!exchanged) return finish(GOMILP_ERR_INFEASIBLE);  // :606
        }
    }

    // ---- Phase II (simplex.go:169-293) ----
    build_nonbasic(n);
    if ((rc = upload_index_lists(basic, nonbasic)) != GOMILP_OK) return finish(rc);
    if (st->phase1_used) {
        refresh_xb_y(P, P.dc);
    } else {
        launch_y_from_binv(w.binv[cur_], P.ld, m, P.dc, w.basic, w.yscratch, w.yb[ycur_], stream_);
        launches_ += 2;
    }
    loop_rc = run_loop(P, 2, tol, (int)nonbasic.size(), P.dc, st);
    }  // revised-simplex pipelines
    if (loop_rc == GOMILP_ERR_DEVICE) return finish(loop_rc);
    const bool loop_unbounded = loop_rc == GOMILP_ERR_UNBOUNDED;   // :261-263, :272-274 — after the condition guards below
    if (loop_unbounded && !shadow_trace_) { *opt_f = -inf; return finish(loop_rc); }

    // ---- epilogue (simplex.go:296-301): x_B from a fresh gonum-order solve on the final basis ----
    HIP_TRY(hipMemcpyAsync(w.h_idx, w.basic, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(w.h_vec, w.xb, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIP_TRY(sync_stream());
    for (int i = 0; i < m; i++) { basic[i] = w.h_idx[i]; xb[i] = w.h_vec[i]; }
    if (shadow_trace_ && (loop_rc == GOMILP_OK || loop_rc == GOMILP_ERR_BLAND || loop_unbounded)) {
        // ---- the reference's LU.Solve guards (mat/lu.go:301,321), replayed on the host with exact condition numbers
        HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        const int64_t cnt = std::min<int64_t>(w.st_host->trace_len, w.trace_cap);
        std::vector<DevPivot> tr((size_t)cnt);
        if (cnt) HIP_TRY(hipMemcpy(tr.data(), w.trace, (size_t)cnt * sizeof(DevPivot), hipMemcpyDeviceToHost));
        std::vector<std::pair<int, int>> piv;
        for (auto &e : tr) piv.emplace_back((int)e.replace, (int)e.entering);
        std::vector<int32_t> bas = basic_start;
        int cst = GOMILP_OK;
        int64_t evals = 0;
        const int stop = general_condition_replay(P.hA, m, n, bas, piv, loop_unbounded, &cst, &evals);
        st->cond_fallbacks += evals;
        if (cst != GOMILP_OK) {
            // the reference left its loop here with the point of that basis (simplex.go:296-301)
            std::vector<double> xs;
            if (general_solve_basis(P.hA, m, n, bas, P.hb, xs)) {
                std::vector<double> cb(m);
                for (int i = 0; i < m; i++) cb[i] = P.hc[bas[i]];
                *opt_f = dot_unitary(cb.data(), xs.data(), m);
                for (int j = 0; j < n; j++) opt_x[j] = 0;
                for (int i = 0; i < m; i++) opt_x[bas[i]] = xs[i];
                *has_x = 1;
                if (basis_out) for (int i = 0; i < m; i++) basis_out[i] = bas[i];
                st->pivots_phase2 = stop;
                shadow_trace_ = false;
                return finish(cst);
            }
        }
    }
    shadow_trace_ = false;
    if (loop_unbounded) { *opt_f = -inf; return finish(loop_rc); }
    rc = epilogue(P, basic, xb, loop_rc, opt_f, opt_x, has_x, basis_out, st);
    if (rc == GOMILP_ERR_DEVICE) return finish(rc);
    if (trace_on_) {
        HIP_TRY(hipMemcpyAsync(w.st_host, w.st, sizeof(DevState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        last_trace_total_ = w.st_host->trace_len;
        const int64_t cnt = std::min<int64_t>(last_trace_total_, w.trace_cap);
        std::vector<DevPivot> tmp((size_t)cnt);
        if (cnt) HIP_TRY(hipMemcpy(tmp.data(), w.trace, (size_t)cnt * sizeof(DevPivot), hipMemcpyDeviceToHost));
        last_trace_.resize((size_t)cnt);
        for (int64_t i = 0; i < cnt; i++) {
            last_trace_[i].phase = tmp[i].phase; last_trace_[i].bland = tmp[i].bland;
            last_trace_[i].min_idx = tmp[i].min_idx; last_trace_[i].replace = tmp[i].replace;
            last_trace_[i].entering = tmp[i].entering; last_trace_[i].leaving = tmp[i].leaving;
        }
    }
    return finish(rc);
}

// x_B = ab^-1 b from a fresh gonum-order LU of the final basis (the device list w.basic must hold `basic`), z = DotUnitary,
// scatter (simplex.go:296-301).  Returns loop_rc (or mat.Condition when the final basis is exactly singular).
int Engine::epilogue(const Problem &P, std::vector<int32_t> &basic, std::vector<double> &xb, int loop_rc, double *opt_f,
                     double *opt_x, int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *st) {
    const int m = P.m, n = P.n;
    std::vector<double> xb_exact;
    bool singular = false;
    const double t1 = now_s();
    int rc = final_solve(P, n, xb_exact, &singular, basic.data());
    if (rc != GOMILP_OK) return rc;
    st->seconds_final_solve = now_s() - t1;
    if (singular) {
        xb_exact = xb;  // the reference keeps its previous x_B when Det()==0 (mat/lu.go:301); ours is the updated one
        if (loop_rc == GOMILP_OK) loop_rc = GOMILP_ERR_CONDITION;
    }
    double drift = 0;
    for (int i = 0; i < m; i++) drift = std::max(drift, fabs(xb[i] - xb_exact[i]));
    st->drift_xb = drift;
    std::vector<double> cb(m);
    for (int i = 0; i < m; i++) cb[i] = P.hc[basic[i]];
    *opt_f = dot_unitary(cb.data(), xb_exact.data(), m);
    for (int j = 0; j < n; j++) opt_x[j] = 0;
    for (int i = 0; i < m; i++) opt_x[basic[i]] = xb_exact[i];
    *has_x = 1;
    if (basis_out) for (int i = 0; i < m; i++) basis_out[i] = basic[i];
    return loop_rc;
}

bool Engine::root_view(int64_t id, RootView *out) {
    std::lock_guard<std::mutex> g(mu_);
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id]) return false;
    const Problem &P = *problems_[id];
    out->gen.reset();   // a view is reused across roots (gomilp_pool_set_root): never keep the previous root's searched basis
    out->m = P.m; out->n = P.n; out->ld = P.ld; out->dAt = P.dAt; out->dc = P.dc; out->db = P.db;
    out->verify_status = P.verify_status; out->serial = P.serial; out->hb = P.hb; out->hc = P.hc; out->scale_span = P.scale_span;
    out->rho0.assign(P.m, 0);
    out->unit_basis = P.m < P.n;
    std::vector<char> used(P.m, 0);
    for (int pos = 0; pos < P.m && out->unit_basis; pos++) {
        const int j = P.n - 1 - pos;
        if (!(P.nnz[j] == 1 && P.allone[j]) || used[P.lastrow[j]]) { out->unit_basis = false; break; }
        out->rho0[pos] = P.lastrow[j]; used[P.lastrow[j]] = 1;
    }
    out->rm.reset();
    if (out->unit_basis && P.verify_status == GOMILP_OK && (size_t)P.m * (size_t)P.n <= ((size_t)1 << 23) && hipSetDevice(device_) == hipSuccess) {
        if (!stream_ && hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking) != hipSuccess) return true;
        std::shared_ptr<RootView::RowMajor> rm(new RootView::RowMajor);
        rm->lda = (P.n + 1) & ~1;
        if (dmalloc(&rm->dA, (size_t)P.m * rm->lda) == hipSuccess) {
            // "A" of the transposing kernel = At (n rows of ld), its output = m rows of lda: dA[i * lda + j] = At[j * ld + i]
            launch_transpose_in(P.dAt, P.ld, P.n, P.m, rm->dA, rm->lda, stream_);
            if (hipStreamSynchronize(stream_) == hipSuccess) out->rm = rm;
        }
    }
    return true;
}

Engine::RootView::RowMajor::~RowMajor() { if (dA) hipFree(dA); }

Engine::RootView::General::~General() {
    for (void *p : {(void *)dT0, (void *)dxb0, (void *)dbasic0, (void *)dnonbasic0, (void *)dposvar0}) if (p) hipFree(p);
}

bool Engine::root_general(int64_t id, RootView *out) {
    std::lock_guard<std::mutex> g(mu_);
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id]) return false;
    const Problem &P = *problems_[id];
    const int m = P.m, n = P.n;
    if (P.verify_status != GOMILP_OK || m >= n || !((n - m) < 2 * m) || !ensure_host_A(P)) return false;
    if (ensure_work(m, n + 1) != GOMILP_OK) return false;
    Work &w = *w_;
    std::vector<int32_t> basic;
    std::vector<double> binv;
    bool binv_dev = false;
    const int rc = (m >= general_min_rows_ && general_device_) ? find_independent_device(P, basic, &binv, &binv_dev) : general_find_linearly_independent(P.hA, m, n, basic, &binv);
    if (rc != GOMILP_OK || (int)basic.size() != m) return false;
    if (!binv_dev && binv.size() != (size_t)m * m && !general_basis_inverse(P.hA, m, n, basic, n, std::vector<double>(), binv)) return false;
    std::vector<char> inb(n, 0);
    for (int i = 0; i < m; i++) inb[basic[i]] = 1;
    std::vector<int32_t> nonbasic, posvar(n);
    for (int j = 0; j < n; j++) if (!inb[j]) nonbasic.push_back(j);
    const int nn = (int)nonbasic.size();
    for (int i = 0; i < m; i++) posvar[basic[i]] = i;
    for (int jp = 0; jp < nn; jp++) posvar[nonbasic[jp]] = -1 - jp;
    if (upload_index_lists(basic, nonbasic) != GOMILP_OK) return false;
    // x_B of the basis with the reference's own arithmetic (as Engine::solve does for the feasibility test)
    std::vector<double> xb;
    bool sing = false;
    if (final_solve(P, n, xb, &sing, basic.data()) != GOMILP_OK || sing) return false;
    std::shared_ptr<RootView::General> G(new RootView::General);
    G->m = m; G->nn = nn; G->ldt = tab_ld(nn);
    auto ok = [](hipError_t e) { return e == hipSuccess; };
    if (!ok(dmalloc(&G->dT0, (size_t)m * G->ldt)) || !ok(dmalloc(&G->dxb0, (size_t)m)) || !ok(dmalloc(&G->dbasic0, (size_t)m)) ||
        !ok(dmalloc(&G->dnonbasic0, (size_t)std::max(nn, 1))) || !ok(dmalloc(&G->dposvar0, (size_t)n))) return false;
    if (!binv_dev && !ok(hipMemcpy2DAsync(w.binv[0], (size_t)P.ld * sizeof(double), binv.data(), (size_t)m * sizeof(double), (size_t)m * sizeof(double), m, hipMemcpyHostToDevice, stream_))) return false;
    if (!ok(hipMemsetAsync(G->dT0, 0, (size_t)m * G->ldt * sizeof(double), stream_))) return false;
    launch_tab_gemm(w.binv[0], P.ld, P.dAt, P.ld, m, nn, w.nonbasic, G->dT0, G->ldt, false, stream_);
    if (!ok(hipMemcpyAsync(G->dxb0, xb.data(), (size_t)m * sizeof(double), hipMemcpyHostToDevice, stream_)) ||
        !ok(hipMemcpyAsync(G->dbasic0, basic.data(), (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice, stream_)) ||
        !ok(hipMemcpyAsync(G->dnonbasic0, nonbasic.data(), (size_t)nn * sizeof(int32_t), hipMemcpyHostToDevice, stream_)) ||
        !ok(hipMemcpyAsync(G->dposvar0, posvar.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, stream_)) || !ok(sync_stream())) return false;
    out->gen = G;
    return true;
}

int Engine::finish_from_basis(int64_t id, const int32_t *basic_in, const double *xb_updated, int loop_rc, double *opt_f,
                              double *opt_x, int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *stats) {
    std::lock_guard<std::mutex> g(mu_);
    const double t0 = now_s();
    gomilp_lp_stats local;
    gomilp_lp_stats *st = stats ? stats : &local;
    memset(st, 0, sizeof(*st));
    st->device_id = device_;
    if (has_x) *has_x = 0;
    if (opt_f) *opt_f = std::numeric_limits<double>::quiet_NaN();
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id] || !opt_f || !opt_x || !has_x || !basic_in || !xb_updated) return GOMILP_ERR_BAD_SHAPE;
    const Problem &P = *problems_[id];
    launches_ = 0; fs_device_ = fs_host_ = 0; lu_look_fault_ = false;
    int rc = ensure_work(P.m, P.n + 1);
    if (rc != GOMILP_OK) return rc;
    std::vector<int32_t> basic(basic_in, basic_in + P.m);
    std::vector<double> xb(xb_updated, xb_updated + P.m);
    if ((rc = upload_index_lists(basic, {})) != GOMILP_OK) return rc;
    rc = epilogue(P, basic, xb, loop_rc, opt_f, opt_x, has_x, basis_out, st);
    st->seconds_total = now_s() - t0; st->kernel_launches = launches_;
    st->seconds_final_device = fs_device_; st->seconds_final_host = fs_host_;
    st->lu_dense_steps = lu_dense_; st->lu_rounds = lu_rounds_;
    if (lu_look_fault_) st->device_retries = 1;
    return rc;
}

int64_t Engine::last_trace(gomilp_pivot *out, int64_t cap) {
    std::lock_guard<std::mutex> g(mu_);
    const int64_t cnt = std::min<int64_t>((int64_t)last_trace_.size(), cap);
    if (out && cnt > 0) memcpy(out, last_trace_.data(), (size_t)cnt * sizeof(gomilp_pivot));
    return last_trace_total_;
}


// findLinearlyIndependent: the column scan on the device — general_block.hip: 16 / 8 / 4 candidates per four launches (default);
// general_kernels.hip: five launches per candidate (knob general_block = 0, bases beyond 4096 rows, the square step) — one look at the
// state block per chunk of candidates; Q^T and R^-1 live in the two B^-1 buffers of the revised pipelines (free at this point)
int Engine::find_independent_device(const Problem &P, std::vector<int32_t> &basic, std::vector<double> *binv_out, bool *binv_on_device) {
    if (binv_on_device) *binv_on_device = false;
    Work &w = *w_;
    const int m = P.m, n = P.n, ldq = P.ld;
    const double t_gs0 = now_s();
    basic.clear();
    if (!w.gs_state) {
        HIP_TRY(dmalloc(&w.gs_state, 1));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&w.gs_host), sizeof(GsState), hipHostMallocDefault));
    }
    if (w.cap_gs_idx < m) {
        if (w.gs_idx) hipFree(w.gs_idx);
        w.gs_idx = nullptr;
        HIP_TRY(dmalloc(&w.gs_idx, (size_t)m + 64));
        w.cap_gs_idx = m + 64;
    }
    double *QT = w.binv[0], *Rinv = w.binv[1];
    double *wv = w.yscratch, *tv = w.yscratch + P.ld, *yv = w.yscratch + 2 * (size_t)P.ld;   // 64 rows of ld doubles: w, t, 16 partial y
    HIP_TRY(hipMemsetAsync(Rinv, 0, (size_t)m * ldq * sizeof(double), stream_));
    // The scan starts at the slack columns (simplex.go:618).  A unit column e_r after k unit columns with other rows is kept with
    // kappa_1 = 1, one with a row already taken is exactly dependent and dropped; its Householder step only negates a row of Q^T
    // or swaps two (the rank-1 update acts on 0 / +-1 entries: no rounding), so the leading run of unit columns is decided here
    // in O(1) each and the device scan starts behind it with Q^T = that signed permutation, R = R^-1 = diag(beta).
    int col = n - 1, s0 = 0;
    {
        std::vector<int32_t> perm(m), inv(m);
        std::vector<double> sgn(m, 1.0), beta(m, 1.0);
        for (int i = 0; i < m; i++) { perm[i] = i; inv[i] = i; }
        std::vector<int32_t> acc;
        for (; col >= 0 && s0 < m - 1; col--) {
            if (!(P.nnz[col] == 1 && P.allone[col])) break;
            const int r = P.lastrow[col], p = inv[r], k = s0;
            if (p < k) continue;                       // row r already carries an accepted unit column: beta = 0, cond = inf, dropped
            if (p == k) { beta[k] = sgn[k] >= 0 ? -1.0 : 1.0; sgn[k] = -sgn[k]; }
            else {                                     // alpha = 0, beta = -1, v = e_k + sgn[p] e_p: rows k and p trade places
                const int pk = perm[k];
                const double sk = sgn[k], sp = sgn[p];
                beta[k] = -1.0;
                perm[k] = r; sgn[k] = -1.0;
                perm[p] = pk; sgn[p] = -sp * sk;
                inv[r] = k; inv[pk] = p;
            }
            acc.push_back(col);
            s0++;
        }
        if (s0) {
            int rcp = stage_upload(w.gs_idx, acc.data(), (size_t)s0 * sizeof(int32_t));
            if (rcp == GOMILP_OK) rcp = stage_upload(w.lpos, perm.data(), (size_t)m * sizeof(int32_t));
            if (rcp == GOMILP_OK) rcp = stage_upload(yv + 20 * (size_t)P.ld, sgn.data(), (size_t)m * sizeof(double));
            if (rcp == GOMILP_OK) rcp = stage_upload(yv + 21 * (size_t)P.ld, beta.data(), (size_t)m * sizeof(double));
            if (rcp != GOMILP_OK) return rcp;
            launch_gs_init_perm(QT, Rinv, ldq, m, w.lpos, yv + 20 * (size_t)P.ld, yv + 21 * (size_t)P.ld, s0, w.gs_state, stream_);
        } else {
            launch_gs_init(QT, ldq, m, w.gs_state, stream_);
        }
    }
    launches_++;
    bool scan_done = s0 >= m - 1;
    int stop_after_prefix = col;
    // blocked form (general_block.hip): NB candidates per four launches; as many blocks per look at the state as would fill the basis
    // if every candidate were accepted (a rejected candidate costs one more look)
    const int nbw = general_block_ ? gs_block_width(m) : 0;
    int k_known = s0;
    for (; !scan_done;) {
        if (nbw) {
            const int want = std::max(1, std::min(32, (m - 1 - k_known + nbw - 1) / nbw));
            for (int q = 0; q < want && col >= 0; q++) {
                const int nc = std::min(nbw, col + 1);
                launch_gs_block(P.dAt, P.ld, col, nc, QT, Rinv, ldq, m, w.yscratch, w.gs_idx, w.gs_state, stream_);
                col -= nc;
                launches_ += 4;
            }
        } else {
            const int chunk = std::min(col + 1, 128);
            for (int q = 0; q < chunk; q++, col--) launch_gs_candidate(P.dAt + (size_t)col * P.ld, QT, Rinv, ldq, m, wv, tv, yv, col, w.gs_idx, w.gs_state, stream_);
            launches_ += 5 * chunk;
        }
        HIP_TRY(hipMemcpyAsync(w.gs_host, w.gs_state, sizeof(GsState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        if (w.gs_host->done || col < 0) break;
        k_known = w.gs_host->k;
    }
    if (scan_done) {   // the unit columns alone filled m - 1 positions: the device never scanned
        HIP_TRY(hipMemcpyAsync(w.gs_host, w.gs_state, sizeof(GsState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        w.gs_host->stop_col = stop_after_prefix;
    }
    const int k = w.gs_host->k;
    std::vector<int32_t> idx(k);
    if (k) HIP_TRY(hipMemcpy(idx.data(), w.gs_idx, (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost));
    basic = idx;
    if (k < m - 1 || !w.gs_host->done) return GOMILP_ERR_SINGULAR;   // the columns ran out: simplex.go:495-497
    const double t_scan = now_s();
    // the square step: the first remaining candidate is taken tentatively on the device (reflector, new column of R^-1), B^-1 =
    // R^-1 Q^T is formed there, and the host judges kappa_1 of the matrix itself, |C|_1 |C^-1|_1 (mat.Cond of a square matrix
    // goes through its LU in the reference).  A rejected candidate sends the rest of the scan to the host form.
    int cand = w.gs_host->stop_col;
    int rcl = GOMILP_ERR_SINGULAR;
    if (cand >= 0) {
        launch_gs_candidate(P.dAt + (size_t)cand * P.ld, QT, Rinv, ldq, m, wv, tv, yv, cand, w.gs_idx, w.gs_state, stream_, 1);
        launch_gs_binv(Rinv, QT, ldq, m, w.W, stream_);
        // the two 1-norms on the device too (absolute column sums of C^-1 in 8 row slices, of the m basis columns out of At): 72 KB
        // come home instead of the matrix — the 8 MB copy, the host's passes over it and over a strided A, and later the upload of the
        // same B^-1 for the tableau set-up were 1.4 of the 1000-row solve's 17.5 ms
        double *nrm = yv, *csum = yv + 8 * (size_t)P.ld;   // (the block search's W rows: free again)
        launch_gs_norms(w.W, ldq, m, nrm, P.dAt, P.ld, w.gs_idx, cand, csum, stream_);
        launches_ += 7;
        const bool keep_dev = binv_on_device != nullptr;
        if (keep_dev) HIP_TRY(hipMemcpyAsync(QT, w.W, (size_t)m * ldq * sizeof(double), hipMemcpyDeviceToDevice, stream_));   // Q^T has served: B^-1 takes its place (the LU that follows overwrites W)
        double *hn = w.h_W;   // (pinned; 9 rows of ld doubles)
        HIP_TRY(hipMemcpyAsync(hn, nrm, (size_t)9 * P.ld * sizeof(double), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(hipMemcpyAsync(w.gs_host, w.gs_state, sizeof(GsState), hipMemcpyDeviceToHost, stream_));
        HIP_TRY(sync_stream());
        HIP_TRY(hipGetLastError());
        double nC = 0, nI = 0;
        bool finite = w.gs_host->beta_last != 0;
        if (finite) {
            const double *hc = hn + 8 * (size_t)P.ld;
            for (int p = 0; p < m; p++) nC = std::max(nC, hc[p]);   // |C|_1: largest absolute column sum over the m basis columns
            for (int c2 = 0; c2 < m; c2++) {
                double sc = 0;
                for (int sl = 0; sl < 8; sl++) sc += hn[(size_t)sl * P.ld + c2];
                if (!std::isfinite(sc)) finite = false;
                nI = std::max(nI, sc);
            }
        }
        const double cond = finite ? nC * nI : std::numeric_limits<double>::infinity();
        if (!(cond > 1e12)) {   // simplex.go:630
            basic.push_back(cand);
            if (keep_dev) *binv_on_device = true;
            else if (binv_out) {   // (a caller that wants the inverse at home)
                HIP_TRY(hipMemcpyAsync(w.h_W, w.W, (size_t)m * ldq * sizeof(double), hipMemcpyDeviceToHost, stream_));
                HIP_TRY(sync_stream());
                binv_out->resize((size_t)m * m);
                for (int r = 0; r < m; r++) memcpy(binv_out->data() + (size_t)r * m, w.h_W + (size_t)r * ldq, (size_t)m * sizeof(double));
            }
            rcl = GOMILP_OK;
        } else {
            rcl = general_finish_last_column(P.hA, m, n, basic, cand - 1, binv_out);
        }
    }
    if (GOMILP_DBG_ENV("GOMILP_DEBUG_GS")) fprintf(stderr, "gs: m %d scanned %d scan %.2f ms last column %.2f ms\n", m, w.gs_host->scanned, 1e3 * (t_scan - t_gs0), 1e3 * (now_s() - t_scan));
    return rcl;
}

int Engine::debug_find_independent(int64_t id, std::vector<int32_t> &idxs) {
    std::lock_guard<std::mutex> g(mu_);
    if (id < 0 || (size_t)id >= problems_.size() || !problems_[id]) return GOMILP_ERR_BAD_SHAPE;
    const Problem &P = *problems_[id];
    HIP_TRY(hipSetDevice(device_));
    int rc = ensure_work(P.m, P.n + 1);
    if (rc != GOMILP_OK) return rc;
    if (!ensure_host_A(P)) return GOMILP_ERR_UNSUPPORTED;
    return find_independent_device(P, idxs, nullptr);
}

}  // namespace gomilp
