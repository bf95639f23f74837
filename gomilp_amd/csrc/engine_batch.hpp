// Device-batched wave of LP relaxations: ONE stream, a fixed schedule of launches with grid.x / grid.z = relaxation
// (batch_kernels.hip, bt_kernels.hip), O(1) host round trips per superstep for the WHOLE wave instead of a host thread +
// stream per relaxation.  Replaces, for one FIFO level of the tree, the solveWorker goroutine pool of
// /root/reference/tree.go:98-100,196-205 (each worker: subProblem.solve -> lp.Simplex, subproblem.go:141-159).
#pragma once
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "engine.hpp"

namespace gomilp {

// Warm start (opt-in; SURVEY.md §8f-1, /root/reference/README.md TODO "initiate the simplex at solution of parent", the reference's hook:
// initialBasic, simplex.go:147-161): final states of relaxations the caller asked to keep, by caller-chosen tag (the B&B node id).  A
// later relaxation that names such a tag as its parent and carries ONE more branch row starts from that basis (batch_kernels.hip
// k_b_setup_warm).  Shared by the schedules of a pool; entries are recycled through a free list.
struct WarmEntry {
    int m = 0, n = 0, nn = 0, ldt = 0, K = 0;
    uint64_t root_serial = 0;
    double *T = nullptr, *xb = nullptr;                              // tableau (4x4 tiles, m4 x ldt), updated x_B
    int32_t *basic = nullptr, *nonbasic = nullptr, *posvar = nullptr;
    size_t cap_t = 0, cap_m = 0, cap_nn = 0, cap_n = 0;
    ~WarmEntry();
};
class WarmStore {
   public:
    std::shared_ptr<WarmEntry> find(int64_t tag);
    // an entry with room for the given shape (recycled or new; nullptr: out of device memory)
    std::shared_ptr<WarmEntry> acquire(int m4, int ldt, int m, int nn, int n);
    void put(int64_t tag, std::shared_ptr<WarmEntry> e);
    void release(int64_t tag);
    void clear();
    size_t size();

   private:
    std::mutex mu_;
    std::map<int64_t, std::shared_ptr<WarmEntry>> by_tag_;
    std::vector<std::shared_ptr<WarmEntry>> free_;
};
// per-wave description of the warm start: parent[i] (tag, or < 0: none) / keep[i] / tag[i] per relaxation
struct WarmSpec {
    WarmStore *store = nullptr;
    const int64_t *parent = nullptr, *tag = nullptr;
    const int32_t *keep = nullptr;
    int dual_budget = 0;    // dual pivots before a warm relaxation is handed back (BS_COLD)
    bool start_warm = true; // false: keep only (every relaxation starts cold)
};

class BatchEngine {
   public:
    struct Outcome {
        int stage = BS_HOST;       // BS_DONE: `status` is final (GOMILP_OK / ERR_BLAND: basis + x_B ready for the final solve);
                                   // BS_HOST: the single-relaxation engine must solve this child
        int status = 0, wrapped = 0, phase1_used = 0;
        int64_t piv1 = 0, piv2 = 0, bland = 0, pivd = 0;   // (pivd: dual pivots of a warm start)
        int warm = 0;                                      // started from its parent's basis
    };
    struct Stats {
        int64_t launches = 0, supersteps = 0, blocks = 0, loop_launches = 0, res_launches = 0, virt_blocks = 0, warm_started = 0, warm_kept = 0;
        double seconds_setup = 0, seconds_total = 0;
        double seconds_inner = 0, seconds_update = 0;   // HIP-event time of the sampled block launches (set_sampling)
        int64_t blocks_sampled = 0;
        const char *inner_kernel = "";
    };
    void set_sampling(bool on) { sampling_ = on; }
    // stream priority of the schedule (before its first run): a wave split into a long-chain group and a wide group gives the long
    // chains the higher priority, so their one-workgroup launches are dispatched ahead of the wide group's hundreds of workgroups
    void set_low_priority(bool low) { low_priority_ = low; }
    void set_xcd_offset(int x) { xcd_off_ = x & 7; }   // XCD of list position 0 of the 8-workgroup block kernel (several schedules side by side)
    void set_fault(int v) { fault_ = v; }   // diagnostic flavour: the first update workgroup of every relaxation of a loop launch leaves at once
    void set_res(bool on) { res_ = on; }     // block steps of narrow waves in the register-resident kernel (res_kernels.hip k_b_res; opt-in: correct, but 4.4 us per exchange round against the loop kernel's 3.6 us per pivot — DESIGN.md section 2.5d)
    // several schedules of one pool side by side (a split wave): each sizes its persistent launches for 1 / share of the device's loop slots —
    // every workgroup of such a launch must be resident, and two schedules that each plan for the whole device wait for each other's CUs
    void set_loop_share(int share) { loop_share_ = share < 1 ? 1 : share; }
    void set_virt(bool on) { virt_ = on; }   // wide waves: set-up pivot and first block on computed tableau entries, only the survivors' tableaus written (default on)
    void set_loop(bool on) { loop_ = on; }   // block steps in the persistent loop kernel where the active relaxations fit one launch (default on)
    void set_cond_guard(int v) { cond_guard_ = v; }   // as the engine knob of the same name
    void set_exact_degenerate(int v) { exact_degenerate_ = v; }   // as the engine knob of the same name
    // called on the thread that runs the wave as soon as child i is terminal; basic / xb (m_i entries, host memory, valid
    // until the next run) are non-null for BS_DONE children whose status needs the final solve
    using DoneFn = std::function<void(int64_t i, const Outcome &, const int32_t *basic, const double *xb)>;

    explicit BatchEngine(int device);
    ~BatchEngine();
    // can children of this root with up to K_max branch rows take the batched path?
    // (phase1: some relaxation of the wave starts infeasible — the Phase-I tableau is one column wider)
    bool eligible(const Engine::RootView &R, int K_max, bool phase1) const;
    int run(const Engine::RootView &R, int64_t count, const int64_t *koff, const int32_t *var, const double *sign,
            const double *rhs, double tol, const DoneFn &on_done, Stats *stats) {
        const Engine::RootView *one = &R;
        return run_roots(&one, 1, nullptr, count, koff, var, sign, rhs, tol, on_done, stats);
    }
    // relaxation i is a child (K_i >= 0 rows) of roots[root_of[i]] (root_of == nullptr: all of roots[0]); the wave is
    // ONE batch: independent LPs of similar shape are children with K = 0 of different roots
    int run_roots(const Engine::RootView *const *roots, int nroots, const int32_t *root_of, int64_t count, const int64_t *koff,
                  const int32_t *var, const double *sign, const double *rhs, double tol, const DoneFn &on_done, Stats *stats,
                  const WarmSpec *warm = nullptr);

   private:
    struct Buf;
    int ensure(int nlp, int m_max, int n_max, int ldt1, int64_t ktot);
    int device_;
    bool sampling_ = false, low_priority_ = false, loop_ = true, res_ = false, virt_ = true;
    int exact_degenerate_ = 1, xcd_off_ = 0, fault_ = 0, loop_share_ = 1;
    int cond_guard_ = 1;
    hipStream_t stream_ = nullptr, stream_hi_ = nullptr, stream_lo_ = nullptr, copy_stream_ = nullptr;   // stream_: the one this run uses
    Buf *b_;
};

}  // namespace gomilp
