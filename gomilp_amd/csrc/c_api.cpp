// extern "C" surface declared in include/gomilp_lp.h.  Each entry point names the reference
// interface it replaces; see INTEGRATION.md for the cgo binding.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"
#include "engine_batch.hpp"

using gomilp::BatchEngine;
using gomilp::Engine;

struct gomilp_ctx {
    std::unique_ptr<Engine> eng;
};

// One worker = one Engine (stream + work buffers) with its own resident copy of the root and a persistent host thread;
// the pivot loops of a wave run device-batched on the pool's BatchEngine (engine_batch.hpp), the workers take what is
// left per relaxation: the final gonum-order solve of a finished basis, or a whole solve where the batched schedule
// hands a relaxation back.
struct gomilp_pool {
    int device = 0;
    std::vector<std::unique_ptr<Engine>> eng;
    std::vector<int64_t> root;  // root problem id inside each engine
    int64_t m0 = 0, n0 = 0;
    int batched = 1;            // knob: 0 = every relaxation through a worker's single-relaxation engine (round-1 path)
    std::unique_ptr<BatchEngine> batch;
    std::unique_ptr<BatchEngine> batch2;   // second schedule (split waves, waves of large relaxations; created on first use)
    std::unique_ptr<BatchEngine> batchx[2];   // third and fourth schedule for waves of large relaxations (split_large)
    int cond_guard = 1, exact_degenerate = 1, sample_batch = 0, batch_loop = 1, batch_res = 0, batch_virt = 1;   // knob values kept for batch2: both halves of a split wave decide alike
    int split_large = 1;        // knob: waves of >= 4 large relaxations run as two interleaved schedules
    int split_phase = 1;        // knob: relaxations that start feasible (Phase II from the slack basis: the long pivot chains of a wave) and
                                // relaxations that need Phase I (on a B&B frontier mostly proved infeasible within a few pivots) run as two
                                // schedules side by side: a block step of the wide group (hundreds of tableaus gathered / updated) no longer
                                // sits between two blocks of a long chain
    int large_loop = 0;         // knob: relaxations of 1025..2048 rows / columns run on the workers' persistent loop kernels (four at a
                                // time, each with its pivot workgroups on an XCD of its own) instead of the batched launch pairs.  Off:
                                // measured 342 k pivots/s for 4 metric LPs against 391 k batched — four updates streaming at once
                                // raise the latency of every chain's agent-scope reads and polls
    gomilp::WarmStore warm;     // final states kept for warm starts (gomilp_frontier_solve_warm), shared by both schedules
    Engine::RootView view;      // of eng[0]'s root (all workers hold the same data)
    // further roots (gomilp_pool_add_root): resident in worker 0's engine only, read in place by the others
    std::vector<int64_t> extra_root;
    std::vector<std::unique_ptr<Engine::RootView>> extra_view;
    std::mutex call_mu;         // one wave at a time per pool
    // A schedule that runs beside the calling thread's (the wide half of a split wave, the second half of a wave of large LPs) gets a
    // thread that LIVES with the pool: a std::thread per wave cost, once in ~50 waves, milliseconds (stack mapped and unmapped in a
    // process full of pinned and device-mapped memory: 6-10 ms waves among 4.0 ms ones in tools/wave_outliers.py).
    struct Aux {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> task;
        bool has = false, done = true, stop = false;
        void loop() {
            for (;;) {
                std::function<void()> f;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || has; });
                    if (stop && !has) return;
                    f = std::move(task); has = false;
                }
                f();
                { std::lock_guard<std::mutex> lk(mu); done = true; }
                cv.notify_all();
            }
        }
        void run(std::function<void()> f) {
            { std::lock_guard<std::mutex> lk(mu); task = std::move(f); has = true; done = false; }
            if (!th.joinable()) th = std::thread([this] { loop(); });
            cv.notify_all();
        }
        void wait() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return done; }); }
        ~Aux() {
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            if (th.joinable()) th.join();
        }
    };
    Aux aux[3];
    // persistent workers
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv, cv_idle;
    std::deque<std::function<void(int)>> queue;
    int busy = 0;
    bool stop = false;

    void worker(int w) {
        for (;;) {
            std::function<void(int)> task;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                task = std::move(queue.front());
                queue.pop_front();
                busy++;
            }
            task(w);
            {
                std::lock_guard<std::mutex> lk(mu);
                busy--;
                if (queue.empty() && busy == 0) cv_idle.notify_all();
            }
        }
    }
    void submit(std::function<void(int)> f) {
        { std::lock_guard<std::mutex> lk(mu); queue.push_back(std::move(f)); }
        cv.notify_one();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&] { return queue.empty() && busy == 0; });
    }
    ~gomilp_pool() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : threads) t.join();
    }
};

#ifdef GOMILP_DEBUG
namespace gomilp { void luc_stamps_read(unsigned long long *out); void res_stamps_read(unsigned long long *out); void lux_stamps_read(unsigned long long *out); }
#endif

extern "C" {

const char *gomilp_version(void) { return "gomilp_amd 0.1 (gfx950)"; }
int gomilp_device_count(void) { return gomilp::device_count(); }
const char *gomilp_compiled_arch(void) { return gomilp::compiled_arch(); }

gomilp_ctx *gomilp_ctx_create(int device, int *status) {
    int n = gomilp::device_count();
    if (n <= 0) { if (status) *status = GOMILP_ERR_DEVICE; return nullptr; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n) { if (status) *status = GOMILP_ERR_DEVICE; return nullptr; }
    gomilp_ctx *c = new gomilp_ctx;
    c->eng.reset(new Engine(device));
    if (status) *status = GOMILP_OK;
    return c;
}
void gomilp_ctx_destroy(gomilp_ctx *ctx) { delete ctx; }
int gomilp_ctx_device(const gomilp_ctx *ctx) { return ctx ? ctx->eng->device() : -1; }
int gomilp_ctx_set(gomilp_ctx *ctx, const char *key, int64_t value) {
    if (!ctx || !key) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->set(key, value);
}
int64_t gomilp_lp_upload(gomilp_ctx *ctx, const double *c, const double *A, int64_t lda, const double *b, int64_t m,
                         int64_t n) {
    if (!ctx) return -GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->upload(c, A, lda, b, m, n);
}
int gomilp_lp_free(gomilp_ctx *ctx, int64_t problem) {
    if (!ctx) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->free_problem(problem);
}
int gomilp_lp_solve_resident(gomilp_ctx *ctx, int64_t problem, double tol, const int64_t *initial_basic, double *opt_f,
                             double *opt_x, int32_t *has_x, int64_t *basis_out, gomilp_lp_stats *stats) {
    if (!ctx) return GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->solve(problem, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
}
int64_t gomilp_lp_upload_child(gomilp_ctx *ctx, int64_t root_problem, int32_t K, const int32_t *var, const double *sign,
                               const double *rhs) {
    if (!ctx) return -GOMILP_ERR_BAD_SHAPE;
    return ctx->eng->upload_child(root_problem, K, var, sign, rhs);
}

gomilp_pool *gomilp_pool_create(int device, int workers, int *status) {
    int n = gomilp::device_count();
    if (n <= 0) { if (status) *status = GOMILP_ERR_DEVICE; return nullptr; }
    if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
    if (device >= n || workers < 1 || workers > 64) { if (status) *status = device >= n ? GOMILP_ERR_DEVICE : GOMILP_ERR_BAD_SHAPE; return nullptr; }
    gomilp_pool *p = new gomilp_pool;
    p->device = device;
    for (int w = 0; w < workers; w++) {
        p->eng.emplace_back(new Engine(device));
        p->root.push_back(-1);
        // the workers finish the relaxations of a wave side by side, next to the batched schedule's own persistent launches: their final
        // solves keep the LU schedule that waits for nobody (lu_compressed.hip: look-ahead launches wait for their own workgroups)
        p->eng.back()->set("lu_look", 0);
    }
    p->batch.reset(new BatchEngine(device));
    for (int w = 0; w < workers; w++) p->threads.emplace_back([p, w] { p->worker(w); });
    if (status) *status = GOMILP_OK;
    return p;
}
void gomilp_pool_destroy(gomilp_pool *pool) { delete pool; }

int gomilp_pool_set(gomilp_pool *pool, const char *key, int64_t value) {
    if (!pool || !key) return GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(pool->call_mu);
    if (std::string(key) == "batched") { pool->batched = value ? 1 : 0; return GOMILP_OK; }
    if (std::string(key) == "split_large") { pool->split_large = value ? 1 : 0; return GOMILP_OK; }
    if (std::string(key) == "split_phase") { pool->split_phase = value ? 1 : 0; return GOMILP_OK; }
    if (std::string(key) == "batch_virt") {   // wide waves on virtual tableaus for their first block (default 1)
        pool->batch_virt = value ? 1 : 0;
        pool->batch->set_virt(value != 0);
        if (pool->batch2) pool->batch2->set_virt(value != 0);
        for (auto &bx : pool->batchx) if (bx) bx->set_virt(value != 0);
        return GOMILP_OK;
    }
    if (std::string(key) == "batch_res") {   // block steps of narrow waves in the register-resident kernel (opt-in, default 0: k_b_loop / launch pairs)
        pool->batch_res = value ? 1 : 0;
        pool->batch->set_res(value != 0);
        if (pool->batch2) pool->batch2->set_res(value != 0);
        for (auto &bx : pool->batchx) if (bx) bx->set_res(value != 0);
        return GOMILP_OK;
    }
    if (std::string(key) == "batch_loop") {   // block steps of narrow waves in the persistent loop kernel (default 1)
        pool->batch_loop = value ? 1 : 0;
        pool->batch->set_loop(value != 0);
        if (pool->batch2) pool->batch2->set_loop(value != 0);
        for (auto &bx : pool->batchx) if (bx) bx->set_loop(value != 0);
        return GOMILP_OK;
    }
    if (std::string(key) == "large_loop") { pool->large_loop = value ? 1 : 0; return GOMILP_OK; }
    if (std::string(key) == "sample_batch") {
        pool->sample_batch = value != 0;
        pool->batch->set_sampling(value != 0);
        if (pool->batch2) pool->batch2->set_sampling(value != 0);
        for (auto &bx : pool->batchx) if (bx) bx->set_sampling(value != 0);
        return GOMILP_OK;
    }
#ifdef GOMILP_DEBUG
    if (std::string(key) == "bt_fault") { pool->batch->set_fault((int)value); if (pool->batch2) pool->batch2->set_fault((int)value); }   // (and the workers' engines below)
#endif
    if (std::string(key) == "cond_guard") {
        pool->cond_guard = (int)value;
        pool->batch->set_cond_guard((int)value);
        if (pool->batch2) pool->batch2->set_cond_guard((int)value);
        for (auto &bx : pool->batchx) if (bx) bx->set_cond_guard((int)value);
    }
    if (std::string(key) == "exact_degenerate") {   // (and the workers' engines below)
        pool->exact_degenerate = (int)value;
        pool->batch->set_exact_degenerate((int)value);
        if (pool->batch2) pool->batch2->set_exact_degenerate((int)value);
        for (auto &bx : pool->batchx) if (bx) bx->set_exact_degenerate((int)value);
    }
    int rc = GOMILP_OK;
    for (auto &e : pool->eng) { const int r = e->set(key, value); if (r != GOMILP_OK) rc = r; }
    return rc;
}

int gomilp_pool_set_root(gomilp_pool *pool, const double *c0, const double *A0, int64_t lda, const double *b0, int64_t m0,
                         int64_t n0) {
    if (!pool) return GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(pool->call_mu);
    for (auto id : pool->extra_root) pool->eng[0]->free_problem(id);
    pool->extra_root.clear(); pool->extra_view.clear();
    pool->warm.clear();
    for (size_t w = 0; w < pool->eng.size(); w++) {
        if (pool->root[w] >= 0) pool->eng[w]->free_problem(pool->root[w]);
        int64_t id = pool->eng[w]->upload(c0, A0, lda, b0, m0, n0);
        if (id < 0) { pool->root[w] = -1; return (int)-id; }
        pool->root[w] = id;
    }
    pool->m0 = m0; pool->n0 = n0;
    if (!pool->eng[0]->root_view(pool->root[0], &pool->view)) return GOMILP_ERR_DEVICE;
    if (!pool->view.unit_basis && pool->view.verify_status == GOMILP_OK) pool->eng[0]->root_general(pool->root[0], &pool->view);   // equality rows: one column search per root
    // first-touch allocations of every worker (work buffers, child slot, final-solve workspace) happen here, not inside the
    // first waves: each worker finishes one dummy child (8 slack rows x_0 <= 1e30) from its slack basis
    if (pool->view.unit_basis && pool->view.verify_status == GOMILP_OK) {
        const int W = (int)pool->eng.size(), K = 8;
        std::atomic<int> started(0);
        for (int t = 0; t < W; t++)
            pool->submit([pool, W, K, &started](int w) {
                started.fetch_add(1);
                while (started.load() < W) std::this_thread::yield();   // every worker takes exactly one of the W tasks
                Engine &E = *pool->eng[w];
                std::vector<int32_t> var(K, 0);
                std::vector<double> sign(K, 1.0), rhs(K, 1e30);
                const int64_t id = E.upload_child(pool->root[w], K, var.data(), sign.data(), rhs.data());
                if (id < 0) return;
                const int m = pool->view.m + K, n = pool->view.n + K;
                std::vector<int32_t> basic(m);
                for (int pos = 0; pos < m; pos++) basic[pos] = n - 1 - pos;
                std::vector<double> xb(m, 0.0), x(n, 0.0);
                double z = 0; int32_t hx = 0;
                E.finish_from_basis(id, basic.data(), xb.data(), GOMILP_OK, &z, x.data(), &hx, nullptr, nullptr);
                E.free_problem(id);
            });
        pool->drain();
    }
    return GOMILP_OK;
}

}  // extern "C"

// warm: parent / tag / keep per relaxation (each nullable), budget; null: a cold wave
struct WarmArgs { const int64_t *parent, *tag; const int32_t *keep; int32_t budget; };
static int frontier_solve_impl(gomilp_pool *pool, int64_t count, const int32_t *root_of, const int64_t *koff, const int32_t *var,
                               const double *sign, const double *rhs, double tol, double *z_out, double *x_out, int64_t ldx,
                               int32_t *status_out, int32_t *has_x_out, gomilp_frontier_stats *stats, const WarmArgs *wa) {
    if (!pool || count < 0 || !koff || !z_out || !x_out || !status_out || !has_x_out) return GOMILP_ERR_BAD_SHAPE;
    for (auto r : pool->root) if (r < 0) return GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> call_guard(pool->call_mu);
    const auto t0 = std::chrono::steady_clock::now();
    const int nroots = 1 + (int)pool->extra_root.size();
    std::vector<const Engine::RootView *> views(nroots);
    views[0] = &pool->view;
    for (int r = 1; r < nroots; r++) views[r] = pool->extra_view[r - 1].get();
    for (int64_t i = 0; i < count; i++) {
        const int ri = root_of ? root_of[i] : 0;
        if (ri < 0 || ri >= nroots || views[ri]->n > ldx) return GOMILP_ERR_BAD_SHAPE;
    }
    auto rootn = [&](int64_t i) -> int64_t { return views[root_of ? root_of[i] : 0]->n; };
    // child i of root r on worker w: root 0 lives in every worker's engine, the others in worker 0's
    auto upload_child = [&](int w, int64_t i) -> int64_t {
        const int ri = root_of ? root_of[i] : 0;
        const int64_t k0 = koff[i], K = koff[i + 1] - koff[i];
        Engine &E = *pool->eng[w];
        if (ri == 0) return E.upload_child(pool->root[w], (int)K, var + k0, sign + k0, rhs + k0);
        return E.upload_child_of(*pool->eng[0], pool->extra_root[ri - 1], (int)K, var + k0, sign + k0, rhs + k0);
    };
    const int W = (int)pool->eng.size();
    std::vector<gomilp_frontier_stats> ws(W);
    for (auto &S : ws) S = gomilp_frontier_stats();
    for (int64_t i = 0; i < count; i++) { z_out[i] = NAN; has_x_out[i] = 0; status_out[i] = GOMILP_ERR_DEVICE; }
    auto busy_since = [](std::chrono::steady_clock::time_point s0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - s0).count(); };
    // a whole relaxation on worker w's engine (the round-1 path; also the fall-back of the batched schedule)
    auto full_solve = [&](int w, int64_t i) {
        const auto s0 = std::chrono::steady_clock::now();
        Engine &E = *pool->eng[w];
        gomilp_frontier_stats &S = ws[w];
        const int64_t K = koff[i + 1] - koff[i], n0 = rootn(i);
        int64_t id = upload_child(w, i);
        if (id < 0) { status_out[i] = (int32_t)-id; return; }
        std::vector<double> x((size_t)(n0 + K), 0.0);
        gomilp_lp_stats st;
        int32_t hx = 0;
        double z = NAN;
        const int rc = E.solve(id, tol, nullptr, &z, x.data(), &hx, nullptr, &st);
        E.free_problem(id);
        status_out[i] = rc; z_out[i] = z; has_x_out[i] = hx;
        if (hx) for (int64_t j = 0; j < n0; j++) x_out[i * ldx + j] = x[j];  // subproblem.go:157-159
        S.relaxations++; S.pivots_phase1 += st.pivots_phase1; S.pivots_phase2 += st.pivots_phase2;
        S.bland_steps += st.bland_steps; S.phase1_runs += st.phase1_used; S.kernel_launches += st.kernel_launches;
        S.seconds_busy_sum += busy_since(s0);
    };
    // the epilogue of a relaxation whose pivot loops ran device-batched: final gonum-order solve of its basis
    auto finish = [&](int w, int64_t i, const int32_t *basic, const double *xb, int loop_rc) {
        const auto s0 = std::chrono::steady_clock::now();
        Engine &E = *pool->eng[w];
        gomilp_frontier_stats &S = ws[w];
        const int64_t K = koff[i + 1] - koff[i], n0 = rootn(i);
        int64_t id = upload_child(w, i);
        if (id < 0) { status_out[i] = (int32_t)-id; return; }
        std::vector<double> x((size_t)(n0 + K), 0.0);
        gomilp_lp_stats st;
        int32_t hx = 0;
        double z = NAN;
        const double t_up = busy_since(s0);
        const int rc = E.finish_from_basis(id, basic, xb, loop_rc, &z, x.data(), &hx, nullptr, &st);
        E.free_problem(id);
        if (GOMILP_DBG_ENV("GOMILP_DEBUG_TASKS") && busy_since(s0) > 5e-3)
            fprintf(stderr, "slow finish: worker %d child %lld upload %.2f ms finish %.2f ms (device %.2f host %.2f) rounds %lld dense %lld\n", w, (long long)i, 1e3 * t_up,
                    1e3 * st.seconds_total, 1e3 * st.seconds_final_device, 1e3 * st.seconds_final_host, (long long)st.lu_rounds, (long long)st.lu_dense_steps);
        status_out[i] = rc; z_out[i] = z; has_x_out[i] = hx;
        if (hx) for (int64_t j = 0; j < n0; j++) x_out[i * ldx + j] = x[j];
        S.kernel_launches += st.kernel_launches;
        S.seconds_busy_sum += busy_since(s0);
    };
    int K_max = 0;
    for (int64_t i = 0; i < count; i++) K_max = std::max<int>(K_max, (int)(koff[i + 1] - koff[i]));
    gomilp_frontier_stats agg = gomilp_frontier_stats();
    BatchEngine::Stats bs;
    bool use_batch = pool->batched && count > 0;
    bool any_p1 = false;
    for (int r = 0; r < nroots && !any_p1; r++) {
        if (views[r]->gen && !views[r]->unit_basis) any_p1 = true;
        for (double v : views[r]->hb) if (v < -1e-13) { any_p1 = true; break; }
    }
    for (int64_t k = koff[0]; k < koff[count] && !any_p1; k++) if (rhs[k] < -1e-13) any_p1 = true;
    for (int r = 0; r < nroots && use_batch; r++) use_batch = pool->batch->eligible(*views[r], K_max, any_p1);
    {
        // 1025..2048 rows / columns, knob "large_loop": the workers' persistent loop kernels instead (four share the device,
        // Engine::loop_acquire)
        int m_big = 0, nn_big = 0;
        for (int r = 0; r < nroots; r++) { m_big = std::max(m_big, views[r]->m + K_max); nn_big = std::max(nn_big, views[r]->n - views[r]->m + (any_p1 ? 1 : 0)); }
        const int need = std::max(m_big, gomilp::batch_ldt(nn_big));
        if (use_batch && pool->large_loop && need > 1024 && need <= 2048) use_batch = false;
    }
    if (use_batch) {
        std::mutex agg_mu;
        std::vector<int64_t> cold_again;   // warm starts that spent their dual-pivot budget: solved cold by this same call
        auto on_done_at = [&](int64_t i, const BatchEngine::Outcome &o, const int32_t *basic, const double *xb) {
            std::lock_guard<std::mutex> lk(agg_mu);
            if (o.warm) { agg.warm_started++; agg.pivots_dual += o.pivd; }
            if (o.stage == gomilp::BS_COLD) { agg.warm_fallbacks++; cold_again.push_back(i); return; }
            if (o.stage != gomilp::BS_DONE) {   // a path the device schedule does not cover
                agg.host_fallbacks++;
                pool->submit([&full_solve, i](int w) { full_solve(w, i); });
                return;
            }
            agg.relaxations++; agg.batched_relaxations++;
            agg.pivots_phase1 += o.piv1; agg.pivots_phase2 += o.piv2; agg.bland_steps += o.bland; agg.phase1_runs += o.phase1_used;
            if (o.status == GOMILP_OK || o.status == GOMILP_ERR_BLAND) {
                const int rc0 = o.status;
                pool->submit([&finish, i, basic, xb, rc0](int w) { finish(w, i, basic, xb, rc0); });
            } else {
                status_out[i] = o.status;
                if (o.status == GOMILP_ERR_UNBOUNDED) z_out[i] = -INFINITY;   // simplex.go:261-263
            }
        };
        auto on_done = [&](int64_t i, const BatchEngine::Outcome &o, const int32_t *basic, const double *xb) { on_done_at(i, o, basic, xb); };
        // Large relaxations (8-workgroup block kernel, rank-16 update): a block step of one schedule is [block kernels of all its
        // relaxations side by side] then [their updates one after the other]; two schedules of half the wave each, on two
        // streams, put one half's updates under the other half's block kernels.
        int m_big = 0, nn_big = 0;
        for (int r = 0; r < nroots; r++) { m_big = std::max(m_big, views[r]->m + K_max); nn_big = std::max(nn_big, views[r]->n - views[r]->m + (any_p1 ? 1 : 0)); }
        const bool two = pool->split_large && count >= 4 && gomilp::bt_batch_k(m_big, gomilp::batch_ldt(nn_big)) == 16 && !wa;
        int rc = GOMILP_OK;
        auto make_batch2 = [&] {
            if (!pool->batch2) {
                pool->batch2.reset(new BatchEngine(pool->device));
                pool->batch2->set_cond_guard(pool->cond_guard);
                pool->batch2->set_exact_degenerate(pool->exact_degenerate);
                pool->batch2->set_sampling(pool->sample_batch != 0);
                pool->batch2->set_loop(pool->batch_loop != 0);
                pool->batch2->set_res(pool->batch_res != 0);
                pool->batch2->set_virt(pool->batch_virt != 0);
            }
        };
        auto merge_stats = [&](const BatchEngine::Stats &bs2, bool serial) {
            bs.launches += bs2.launches; bs.blocks += bs2.blocks; bs.blocks_sampled += bs2.blocks_sampled;
            bs.supersteps = serial ? bs.supersteps + bs2.supersteps : std::max(bs.supersteps, bs2.supersteps);
            bs.seconds_inner += bs2.seconds_inner; bs.seconds_update += bs2.seconds_update;
            bs.seconds_total = serial ? bs.seconds_total + bs2.seconds_total : std::max(bs.seconds_total, bs2.seconds_total);
            bs.warm_kept += bs2.warm_kept;
        };
        // a subset of the wave as a schedule of its own (the schedules take contiguous lists)
        struct Group { std::vector<int32_t> root_of, var, keep; std::vector<int64_t> koff, parent, tag; std::vector<double> sign, rhs; };
        // (sized once, rows copied in runs: an 8192-wide wave has 100 k branch rows — element-wise push_backs were 1.5 ms of it, in front of
        // the first kernel)
        auto gather = [&](const std::vector<int64_t> &idx, Group &g) {
            const size_t ng = idx.size();
            size_t tot = 0;
            for (int64_t i : idx) tot += (size_t)(koff[i + 1] - koff[i]);
            g.koff.resize(ng + 1); g.root_of.resize(ng); g.parent.resize(ng); g.tag.resize(ng); g.keep.resize(ng);
            g.var.resize(std::max<size_t>(tot, 1)); g.sign.resize(std::max<size_t>(tot, 1)); g.rhs.resize(std::max<size_t>(tot, 1));   // (valid pointers for K = 0 everywhere)
            g.var[0] = 0; g.sign[0] = 0; g.rhs[0] = 0;
            size_t at = 0;
            g.koff[0] = 0;
            for (size_t t = 0; t < ng; t++) {
                const int64_t i = idx[t];
                const size_t K = (size_t)(koff[i + 1] - koff[i]);
                g.root_of[t] = root_of ? root_of[i] : 0;
                if (K) {
                    memcpy(&g.var[at], var + koff[i], K * sizeof(int32_t));
                    memcpy(&g.sign[at], sign + koff[i], K * sizeof(double));
                    memcpy(&g.rhs[at], rhs + koff[i], K * sizeof(double));
                }
                at += K;
                g.koff[t + 1] = (int64_t)at;
                g.parent[t] = wa && wa->parent ? wa->parent[i] : -1;
                g.tag[t] = wa && wa->tag ? wa->tag[i] : -1;
                g.keep[t] = wa && wa->keep && wa->tag && wa->tag[i] >= 0 ? wa->keep[i] : 0;
            }
        };
        auto run_group = [&](BatchEngine &be, const std::vector<int64_t> &idx, const Group &g, bool start_warm, BatchEngine::Stats *bsx) -> int {
            gomilp::WarmSpec ws;
            ws.store = &pool->warm; ws.parent = g.parent.data(); ws.tag = g.tag.data(); ws.keep = g.keep.data();
            ws.dual_budget = wa ? wa->budget : 0; ws.start_warm = start_warm;
            auto od = [&, idxp = &idx](int64_t i, const BatchEngine::Outcome &o, const int32_t *basic, const double *xb) { on_done_at((*idxp)[(size_t)i], o, basic, xb); };
            return be.run_roots(views.data(), nroots, g.root_of.data(), (int64_t)idx.size(), g.koff.data(), g.var.data(), g.sign.data(), g.rhs.data(), tol, od, bsx,
                                wa ? &ws : nullptr);
        };
        // ---- warm starts first: the relaxations whose parent's final state is resident
        std::vector<int64_t> cold_idx;
        bool subset = false;   // the cold part is a proper subset of the call (or carries keep flags): gathered lists
        if (wa) {
            std::vector<int64_t> warm_idx;
            for (int64_t i = 0; i < count; i++) {
                const bool w = wa->parent && wa->parent[i] >= 0 && koff[i + 1] - koff[i] >= 1 && pool->warm.find(wa->parent[i]) != nullptr;
                (w ? warm_idx : cold_idx).push_back(i);
            }
            subset = true;
            if (!warm_idx.empty()) {
                Group gw;
                gather(warm_idx, gw);
                rc = run_group(*pool->batch, warm_idx, gw, true, &bs);
                if (rc != GOMILP_OK) { pool->drain(); return rc; }
                std::lock_guard<std::mutex> lk(agg_mu);
                for (int64_t i : cold_again) cold_idx.push_back(i);
                cold_again.clear();
                std::sort(cold_idx.begin(), cold_idx.end());
            }
        }
        // relaxations that start feasible (slack basis, b >= 0, every branch right-hand side >= 0: Phase II at once) against those that
        // need Phase I
        std::vector<int64_t> grp_f, grp_p;
        const int64_t ncold = subset ? (int64_t)cold_idx.size() : count;
        if (pool->split_phase && !two && ncold >= 16) {
            std::vector<char> root_ok(nroots, 0);
            for (int r = 0; r < nroots; r++) {
                bool ok = views[r]->unit_basis;
                for (double v : views[r]->hb) if (v < -1e-13) { ok = false; break; }
                root_ok[r] = ok ? 1 : 0;
            }
            for (int64_t t = 0; t < ncold; t++) {
                const int64_t i = subset ? cold_idx[(size_t)t] : t;
                bool f = root_ok[root_of ? root_of[i] : 0] != 0;
                for (int64_t k = koff[i]; k < koff[i + 1] && f; k++) if (rhs[k] < -1e-13) f = false;
                (f ? grp_f : grp_p).push_back(i);
            }
        }
        BatchEngine::Stats bsc;   // the cold part
        if (ncold == 0) {
            // (every relaxation of the call was a warm start that stayed warm)
        } else if (!grp_f.empty() && !grp_p.empty()) {
            Group gf, gp;
            gather(grp_f, gf);   // (the wide group's lists are put together on its own thread, below: the long chains start at once)
            make_batch2();
            BatchEngine::Stats bs2;
            int rc2 = GOMILP_OK;
            pool->batch->set_low_priority(true);   // the wide group yields to the long chains
            pool->batch->set_loop_share(2); pool->batch2->set_loop_share(2);   // two schedules with persistent launches: half the loop slots each
            // (the long chains on the calling thread: they are the critical path of the wave and start without waiting for a thread to come up)
            pool->aux[0].run([&] {
                hipSetDevice(pool->device);
                gather(grp_p, gp);
                rc = run_group(*pool->batch, grp_p, gp, false, &bsc);
            });
            rc2 = run_group(*pool->batch2, grp_f, gf, false, &bs2);
            pool->aux[0].wait();
            pool->batch->set_low_priority(false);
            pool->batch->set_loop_share(1); pool->batch2->set_loop_share(1);
            if (rc == GOMILP_OK) rc = rc2;
            merge_stats(bsc, true); merge_stats(bs2, false);
        } else if (subset) {
            Group gc;
            gather(cold_idx, gc);
            rc = run_group(*pool->batch, cold_idx, gc, false, &bsc);
            merge_stats(bsc, true);
        } else if (two) {
            // up to four schedules, each with a contiguous quarter of the wave, each on its own stream and host thread, their block kernels on
            // XCDs of their own (BatchEngine::set_xcd_offset): one schedule's updates and control steps run under the other schedules' blocks
            make_batch2();
            const int nsched = (int)std::min<int64_t>(2, count);   // (four schedules — one per LP of a wave of four — measured 363 k pivots / s against 408-440 k for one or two: the chains slow each other through the memory system)
            BatchEngine *be[4] = {pool->batch.get(), pool->batch2.get(), nullptr, nullptr};
            for (int t = 2; t < nsched; t++) {
                if (!pool->batchx[t - 2]) {
                    pool->batchx[t - 2].reset(new BatchEngine(pool->device));
                    pool->batchx[t - 2]->set_cond_guard(pool->cond_guard);
                    pool->batchx[t - 2]->set_exact_degenerate(pool->exact_degenerate);
                    pool->batchx[t - 2]->set_sampling(pool->sample_batch != 0);
                    pool->batchx[t - 2]->set_loop(pool->batch_loop != 0);
                    pool->batchx[t - 2]->set_res(pool->batch_res != 0);
                    pool->batchx[t - 2]->set_virt(pool->batch_virt != 0);
                }
                be[t] = pool->batchx[t - 2].get();
            }
            int64_t lo[5];
            for (int t = 0; t <= nsched; t++) lo[t] = count * t / nsched;
            BatchEngine::Stats bsx[4];
            int rcx[4] = {GOMILP_OK, GOMILP_OK, GOMILP_OK, GOMILP_OK};
            auto run_part = [&](int t) {
                hipSetDevice(pool->device);
                const int64_t off = lo[t];
                auto od = [&, off](int64_t i, const BatchEngine::Outcome &o, const int32_t *basic, const double *xb) { on_done_at(i + off, o, basic, xb); };
                be[t]->set_xcd_offset((int)((2 * t) & 7));
                rcx[t] = be[t]->run_roots(views.data(), nroots, root_of ? root_of + off : nullptr, lo[t + 1] - off, koff + off, var, sign, rhs, tol, od, &bsx[t]);
                be[t]->set_xcd_offset(0);
            };
            for (int t = 1; t < nsched; t++) pool->aux[t - 1].run([&run_part, t] { run_part(t); });
            run_part(0);
            for (int t = 1; t < nsched; t++) pool->aux[t - 1].wait();
            bs = bsx[0];
            for (int t = 0; t < nsched; t++) { if (rc == GOMILP_OK) rc = rcx[t]; if (t) merge_stats(bsx[t], false); }
        } else {
            rc = pool->batch->run_roots(views.data(), nroots, root_of, count, koff, var, sign, rhs, tol, on_done, &bs);
        }
        agg.warm_kept = bs.warm_kept;
        pool->drain();
        if (rc != GOMILP_OK) return rc;
    } else {
        for (int64_t i = 0; i < count; i++) pool->submit([&full_solve, i](int w) { full_solve(w, i); });
        pool->drain();
    }
    if (stats) {
        *stats = agg;
        for (auto &S : ws) {
            stats->relaxations += S.relaxations; stats->pivots_phase1 += S.pivots_phase1; stats->pivots_phase2 += S.pivots_phase2;
            stats->bland_steps += S.bland_steps; stats->phase1_runs += S.phase1_runs; stats->kernel_launches += S.kernel_launches;
            stats->seconds_busy_sum += S.seconds_busy_sum;
        }
        stats->kernel_launches += bs.launches;
        stats->supersteps = bs.supersteps; stats->seconds_batch = bs.seconds_total;
        stats->blocks = bs.blocks; stats->blocks_sampled = bs.blocks_sampled;
        stats->seconds_inner_kernels = bs.seconds_inner; stats->seconds_update_kernels = bs.seconds_update;
        stats->warm_kept = bs.warm_kept;
        stats->workers = W; stats->device_id = pool->device;
        stats->seconds_total = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    return GOMILP_OK;
}

extern "C" {

int gomilp_frontier_solve_roots(gomilp_pool *pool, int64_t count, const int32_t *root_of, const int64_t *koff, const int32_t *var,
                                const double *sign, const double *rhs, double tol, double *z_out, double *x_out, int64_t ldx,
                                int32_t *status_out, int32_t *has_x_out, gomilp_frontier_stats *stats) {
    return frontier_solve_impl(pool, count, root_of, koff, var, sign, rhs, tol, z_out, x_out, ldx, status_out, has_x_out, stats, nullptr);
}

int gomilp_frontier_solve_warm(gomilp_pool *pool, int64_t count, const int64_t *koff, const int32_t *var, const double *sign,
                               const double *rhs, const int64_t *parent, const int64_t *tag, const int32_t *keep, int32_t dual_budget,
                               double tol, double *z_out, double *x_out, int32_t *status_out, int32_t *has_x_out,
                               gomilp_frontier_stats *stats) {
    if (!pool) return GOMILP_ERR_BAD_SHAPE;
    WarmArgs wa{parent, tag, keep, dual_budget};
    return frontier_solve_impl(pool, count, nullptr, koff, var, sign, rhs, tol, z_out, x_out, pool->n0, status_out, has_x_out, stats, &wa);
}

int gomilp_pool_release_warm(gomilp_pool *pool, int64_t tag) {
    if (!pool) return GOMILP_ERR_BAD_SHAPE;
    // (behind the pool's call lock like every other entry point: an entry released while a warm wave runs would go back to the free list and
    // could be recycled by that wave's own harvest — the copy of a kept child into it would overwrite a parent tableau the wave still reads)
    std::lock_guard<std::mutex> g(pool->call_mu);
    if (tag < 0) pool->warm.clear(); else pool->warm.release(tag);
    return GOMILP_OK;
}

int gomilp_frontier_solve(gomilp_pool *pool, int64_t count, const int64_t *koff, const int32_t *var, const double *sign,
                          const double *rhs, double tol, double *z_out, double *x_out, int32_t *status_out,
                          int32_t *has_x_out, gomilp_frontier_stats *stats) {
    if (!pool) return GOMILP_ERR_BAD_SHAPE;
    return gomilp_frontier_solve_roots(pool, count, nullptr, koff, var, sign, rhs, tol, z_out, x_out, pool->n0, status_out, has_x_out, stats);
}

// The root relaxation (subproblem.go:172) on the pool's first worker.
int gomilp_pool_solve_root(gomilp_pool *pool, double tol, double *opt_f, double *opt_x, int32_t *has_x, gomilp_lp_stats *stats) {
    if (!pool || pool->root[0] < 0) return GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(pool->call_mu);
    return pool->eng[0]->solve(pool->root[0], tol, nullptr, opt_f, opt_x, has_x, nullptr, stats);
}

int gomilp_pool_add_root(gomilp_pool *pool, const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n) {
    if (!pool) return -GOMILP_ERR_BAD_SHAPE;
    std::lock_guard<std::mutex> g(pool->call_mu);
    if (pool->root[0] < 0) return -GOMILP_ERR_BAD_SHAPE;   // gomilp_pool_set_root first
    const int64_t id = pool->eng[0]->upload(c, A, lda, b, m, n);
    if (id < 0) return (int)id;
    std::unique_ptr<Engine::RootView> v(new Engine::RootView);
    if (!pool->eng[0]->root_view(id, v.get())) return -GOMILP_ERR_DEVICE;
    if (!v->unit_basis && v->verify_status == GOMILP_OK) pool->eng[0]->root_general(id, v.get());
    pool->extra_root.push_back(id);
    pool->extra_view.push_back(std::move(v));
    return (int)pool->extra_root.size();
}

// diagnostic (host only, no device): the column search of findLinearlyIndependent as the engine performs it for non-slack
// starting bases; fast = 1: incremental QR (O(m^2 n)), 0: a fresh exact condition number per candidate (O(m^4))
int64_t gomilp_debug_find_independent(const double *A, int64_t lda, int64_t m, int64_t n, int64_t *idx_out, int fast) {
    if (!A || !idx_out || m <= 0 || n <= 0 || lda < n || m > 4096 || n > (1 << 20)) return -GOMILP_ERR_BAD_SHAPE;
    std::vector<double> a((size_t)m * n);
    for (int64_t i = 0; i < m; i++) for (int64_t j = 0; j < n; j++) a[(size_t)i * n + j] = A[i * lda + j];
    std::vector<int32_t> idx;
    if (fast) gomilp::general_find_linearly_independent(a, (int)m, (int)n, idx);
    else gomilp::general_find_linearly_independent_slow(a, (int)m, (int)n, idx);
    for (size_t k = 0; k < idx.size(); k++) idx_out[k] = idx[k];
    return (int64_t)idx.size();
}

// diagnostic (host only): the condition-number estimate the engine takes its mat.Condition verdict on beyond the exact screen
// (Engine::cond_check): |B|_1 times the Hager / Higham estimate of |B^-1|_1 (inf = 0; gonum: the duals' solve with ab^T), or
// |B|_inf times the estimate of |B^-1|_inf (inf = 1: the solves with ab).  -1: singular to working precision.
double gomilp_debug_cond_estimate(const double *B, int64_t n, int inf) {
    if (!B || n <= 0 || n > 4096) return -1.0;
    std::vector<double> b((size_t)n * n), inv;
    for (int64_t i = 0; i < n * n; i++) b[(size_t)i] = B[i];
    if (!gomilp::general_invert(b, (int)n, inv)) return -1.0;
    double norm = 0;
    if (!inf) { for (int64_t j = 0; j < n; j++) { double s2 = 0; for (int64_t i = 0; i < n; i++) s2 += fabs(b[(size_t)(i * n + j)]); norm = std::max(norm, s2); } }
    else { for (int64_t i = 0; i < n; i++) { double s2 = 0; for (int64_t j = 0; j < n; j++) s2 += fabs(b[(size_t)(i * n + j)]); norm = std::max(norm, s2); } }
    return norm * gomilp::inverse_norm1_estimate(inv, (int)n, inf != 0);
}

// diagnostic: the same search with the scan on the device (general_kernels.hip), on a problem resident in `ctx`
int gomilp_debug_gonum_lu_cond(const double *M, int64_t n, int transposed, double *cond) {
    if (!M || !cond || n < 1 || n > gomilp::kGonumCondMax) return -1;
    bool dz = false;
    if (!gomilp::gonum_lu_cond(M, (int)n, (int)n, transposed != 0, cond, &dz)) return -1;
    return dz ? 1 : 0;
}

int64_t gomilp_debug_find_independent_device(gomilp_ctx *ctx, int64_t problem, int64_t *idx_out, int64_t cap) {
    if (!ctx || !idx_out) return -GOMILP_ERR_BAD_SHAPE;
    std::vector<int32_t> idx;
    const int rc = ctx->eng->debug_find_independent(problem, idx);
    if (rc != GOMILP_OK && rc != GOMILP_ERR_SINGULAR) return -rc;
    for (size_t k = 0; k < idx.size() && (int64_t)k < cap; k++) idx_out[k] = idx[k];
    return (int64_t)idx.size();
}

// launches of the loop kernel's pivot role with replicated reduced costs so far (tests: the kernel under test really ran)
long long gomilp_debug_loop_rep_launches(void) { return gomilp::bt_loop_rep_launches(); }
#ifdef GOMILP_DEBUG
// diagnostic flavour only: cycle sums of the final-solve panel kernel (lu_compressed.hip), 4 waves x 16 segments
void gomilp_debug_luc_stamps(unsigned long long *out) { gomilp::luc_stamps_read(out); }
void gomilp_debug_lux_stamps(unsigned long long *out) { gomilp::lux_stamps_read(out); }
void gomilp_debug_res_stamps(unsigned long long *out) { gomilp::res_stamps_read(out); }
#endif

int64_t gomilp_lp_last_trace(gomilp_ctx *ctx, gomilp_pivot *out, int64_t cap) {
    if (!ctx) return -1;
    return ctx->eng->last_trace(out, cap);
}

// lp.Simplex drop-in (simplex.go:88).  The reference's solveWorker goroutines (tree.go:98-100,196-205) call it concurrently:
// every call checks a context (engine + stream + recycled buffers) out of a per-device free list and puts it back, so N
// callers run on N streams; a context is created when the list is empty (at most kFlatMaxCtx per device, then callers
// wait).  The call uploads, solves and releases: nothing of the caller's memory is retained (cgo rules).
namespace {
constexpr int kFlatMaxCtx = 32;
struct FlatPool {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<gomilp_ctx *> free_list;
    int created = 0;
};
FlatPool &flat_pool(int dev) {
    static std::mutex mu;
    static std::map<int, FlatPool *> pools;   // never destroyed: contexts live as long as the process (HIP tears down at exit)
    std::lock_guard<std::mutex> g(mu);
    auto it = pools.find(dev);
    if (it == pools.end()) it = pools.emplace(dev, new FlatPool).first;
    return *it->second;
}
}  // namespace

int gomilp_lp_simplex(const double *c, const double *A, int64_t lda, const double *b, int64_t m, int64_t n, double tol,
                      const int64_t *initial_basic, double *opt_f, double *opt_x, int32_t *has_x, int64_t *basis_out,
                      gomilp_lp_stats *stats) {
    if (has_x) *has_x = 0;
    if (opt_f) *opt_f = NAN;
    if (!c || !A || !b || !opt_f || !opt_x || !has_x || m <= 0 || n <= 0 || lda < n) return GOMILP_ERR_BAD_SHAPE;
    int dev = 0;
    if (gomilp::device_count() <= 0 || hipGetDevice(&dev) != hipSuccess) return GOMILP_ERR_DEVICE;
    FlatPool &fp = flat_pool(dev);
    gomilp_ctx *ctx = nullptr;
    {
        std::unique_lock<std::mutex> lk(fp.mu);
        for (;;) {
            if (!fp.free_list.empty()) { ctx = fp.free_list.back(); fp.free_list.pop_back(); break; }
            if (fp.created < kFlatMaxCtx) { fp.created++; break; }   // create outside the lock
            fp.cv.wait(lk);
        }
    }
    if (!ctx) {
        int st = 0;
        ctx = gomilp_ctx_create(dev, &st);
        if (!ctx) {
            { std::lock_guard<std::mutex> lk(fp.mu); fp.created--; }
            fp.cv.notify_one();
            return st;
        }
    }
    int rc;
    int64_t id = ctx->eng->upload(c, A, lda, b, m, n, true);   // (lazy host copy: Engine::upload)
    if (id < 0) rc = (int)-id;
    else {
        rc = gomilp_lp_solve_resident(ctx, id, tol, initial_basic, opt_f, opt_x, has_x, basis_out, stats);
        gomilp_lp_free(ctx, id);
    }
    { std::lock_guard<std::mutex> lk(fp.mu); fp.free_list.push_back(ctx); }
    fp.cv.notify_one();
    return rc;
}

}  // extern "C"
