// Private to the engine translation units: small host helpers and the device work-buffer struct.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <limits>

#include "engine.hpp"

namespace gomilp {

namespace {

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

#ifdef GOMILP_DEBUG
// diagnostic flavour: say which call failed
#define HIP_TRY(expr)                                                                                                   \
    do {                                                                                                                \
        hipError_t _e = (expr);                                                                                         \
        if (_e != hipSuccess) {                                                                                         \
            fprintf(stderr, "gomilp: %s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e));              \
            return GOMILP_ERR_DEVICE;                                                                                   \
        }                                                                                                               \
    } while (0)
#else
#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) return GOMILP_ERR_DEVICE; \
    } while (0)
#endif

// floats.MinIdx (floats/floats.go:458-474)
int64_t min_idx(const double *s, int64_t n) {
    double mn = std::numeric_limits<double>::quiet_NaN();
    int64_t ind = 0;
    for (int64_t i = 0; i < n; i++) {
        const double v = s[i];
        if (v != v) continue;
        if (v < mn || mn != mn) { mn = v; ind = i; }
    }
    return ind;
}

// f64.DotUnitary (internal/asm/f64/dot_amd64.s:43-92): 4 interleaved partial sums, tail into lane 0.
// Used for z = cb.xb (simplex.go:296) so the objective is bit-identical given identical xb.
double dot_unitary(const double *x, const double *y, int64_t n) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int64_t i = 0;
    for (; i + 4 <= n; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    for (; i < n; i++) s0 += x[i] * y[i];
    return (s0 + s2) + (s1 + s3);
}

template <typename T>
hipError_t dmalloc(T **p, size_t count) {
    return hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(T));
}

}  // namespace

struct Engine::Work {
    int cap_m = 0, cap_ld = 0, cap_cols = 0;
    double *binv[2] = {nullptr, nullptr};
    double *yb[2] = {nullptr, nullptr};
    double *xb = nullptr, *dvec = nullptr, *move = nullptr, *rvec = nullptr, *yscratch = nullptr, *W = nullptr;
    int32_t *basic = nullptr, *nonbasic = nullptr, *lpos = nullptr, *rowstep = nullptr, *rho = nullptr;
    unsigned long long *pk_price = nullptr, *pk_ratio = nullptr, *lpk[2] = {nullptr, nullptr};
    unsigned int *pi_price = nullptr, *pi_ratio = nullptr, *lpl[2] = {nullptr, nullptr}, *lpr[2] = {nullptr, nullptr};
    unsigned int *pv_price = nullptr, *pb_ratio = nullptr;
    double *pd_ratio = nullptr, *px_ratio = nullptr;
    double *T[2] = {nullptr, nullptr}, *R[2] = {nullptr, nullptr}, *tscratch = nullptr;  // tableau pipelines
    double *btU = nullptr, *btV = nullptr;  // blocked tableau: rank-1 terms of the running block
    double *xbuf = nullptr;                 // multi-workgroup block kernel: exchange records (btg_kernels.hip)
    int64_t loop_launches = 0;              // launches of the persistent loop kernel so far (launch parity)
    double *luxrec = nullptr;   // exchange records of the cross-workgroup LU panel (lu_cross.hip), zeroed once
    GsState *gs_state = nullptr, *gs_host = nullptr;   // device column search (general_kernels.hip): state block + pinned mirror
    int32_t *gs_idx = nullptr; int cap_gs_idx = 0;
    int32_t *srcpos = nullptr;
    int32_t *unitrow = nullptr;  // final solve: unit-column rows per basis position
    int32_t *denseflag = nullptr, *dlist = nullptr;  // final solve: steps that did arithmetic / their compact list
    double *ludiag = nullptr;   // diagonal of U by physical row
    double *luLp = nullptr, *luUp = nullptr;  // compressed LU: compact multiplier / U-row panels of the running round
    double *Wd = nullptr;       // packed dense columns of L\\U for the host solves (m x nd)
    size_t cap_T = 0;  // doubles per T buffer
    size_t cap_btU = 0;  // doubles in btU (kBtMaxK rows of the padded row count)
    unsigned long long *stamps = nullptr, *stamps_host = nullptr;   // diagnostic build of the block kernel (knob "bt_stamps")
    int cap_ldt = 0;
    DevState *st = nullptr;
    DevState *st_host = nullptr;  // pinned
    LUCtl *luctl = nullptr, *luctl_host = nullptr;  // compressed LU schedule: control block (device / pinned)
    DevPivot *trace = nullptr;
    int64_t trace_cap = 0;
    double *h_W = nullptr;  // pinned, cap_m * cap_ld
    double *h_vec = nullptr;  // pinned, max(cap_ld, cap_cols)
    double *h_chk = nullptr;  // pinned, cap_ld: x_B of the Phase-I starting vertex (checked after the loop)
    int32_t *h_idx = nullptr; // pinned, max(cap_m, cap_cols)
    // pinned staging ring for small host -> device uploads whose source is pageable / short-lived: the copy is enqueued
    // from a ring slot; a slot is reused only after a full stream sync has happened since (Engine::sync_stream counts)
    static constexpr int kStageSlots = 32;
    char *stage_buf[kStageSlots] = {};
    size_t stage_cap[kStageSlots] = {};
    int stage_next = 0;
    int stage_inflight = 0;   // uploads enqueued since the last full stream sync (Engine::sync_stream)
    char *child_stage = nullptr;  // pinned staging block of upload_child
    size_t child_stage_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    DevState *pipe_state[2] = {nullptr, nullptr};   // pinned: state after each of the two chunks in flight (blocked pipeline)
    hipEvent_t pipe_ev[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> sample_ev;  // pairs around sampled kernels

    void release() {
        for (auto &p : binv) { if (p) hipFree(p); p = nullptr; }
        for (double **p : {&xb, &yb[0], &yb[1], &dvec, &move, &rvec, &yscratch, &W, &ludiag, &Wd, &luLp, &luUp}) { if (*p) hipFree(*p); *p = nullptr; }
        nonbasic = nullptr;   // lives in the same block as `basic`
        for (int32_t **p : {&basic, &lpos, &rowstep, &rho, &unitrow, &denseflag, &dlist}) { if (*p) hipFree(*p); *p = nullptr; }
        if (h_W) hipHostFree(h_W); h_W = nullptr;
        if (h_vec) hipHostFree(h_vec); h_vec = nullptr;
        if (h_chk) hipHostFree(h_chk); h_chk = nullptr;
        if (h_idx) hipHostFree(h_idx); h_idx = nullptr;
        cap_m = cap_ld = cap_cols = 0;
    }
    void release_all() {
        release();
        for (auto **p : {&pk_price, &pk_ratio, &lpk[0], &lpk[1]}) { if (*p) hipFree(*p); *p = nullptr; }
        for (auto **p : {&pi_price, &pi_ratio, &lpl[0], &lpl[1], &lpr[0], &lpr[1], &pv_price, &pb_ratio}) { if (*p) hipFree(*p); *p = nullptr; }
        if (pd_ratio) hipFree(pd_ratio); pd_ratio = nullptr;
        if (px_ratio) hipFree(px_ratio); px_ratio = nullptr;
        for (double **p : {&T[0], &T[1], &R[0], &R[1], &tscratch, &btU, &btV, &xbuf}) { if (*p) hipFree(*p); *p = nullptr; }
        if (srcpos) hipFree(srcpos); srcpos = nullptr;
        cap_T = 0; cap_ldt = 0; cap_btU = 0;
        if (luxrec) hipFree(luxrec); luxrec = nullptr;
        if (gs_state) hipFree(gs_state); gs_state = nullptr;
        if (gs_host) hipHostFree(gs_host); gs_host = nullptr;
        if (gs_idx) hipFree(gs_idx); gs_idx = nullptr; cap_gs_idx = 0;
        if (stamps) hipFree(stamps); stamps = nullptr;
        if (stamps_host) hipHostFree(stamps_host); stamps_host = nullptr;
        if (st) hipFree(st); st = nullptr;
        if (st_host) hipHostFree(st_host); st_host = nullptr;
        if (child_stage) hipHostFree(child_stage); child_stage = nullptr; child_stage_cap = 0;
        if (trace) hipFree(trace); trace = nullptr;
        for (auto &e : ev) { if (e) hipEventDestroy(e); e = nullptr; }
        for (auto &e : sample_ev) hipEventDestroy(e);
        sample_ev.clear();
        for (int t = 0; t < 2; t++) {
            if (pipe_state[t]) hipHostFree(pipe_state[t]); pipe_state[t] = nullptr;
            if (pipe_ev[t]) hipEventDestroy(pipe_ev[t]); pipe_ev[t] = nullptr;
        }
        for (int t = 0; t < kStageSlots; t++) {
            if (stage_buf[t]) hipHostFree(stage_buf[t]); stage_buf[t] = nullptr; stage_cap[t] = 0;
        }
        if (luctl) hipFree(luctl); luctl = nullptr;
        if (luctl_host) hipHostFree(luctl_host); luctl_host = nullptr;
    }
};


}  // namespace gomilp
