// Fused two-kernel pivot pipeline for gfx950 (the fast path when the padded row length is 64*NV double2).
//
// Per pivot the reference does pricing + 3 LU solves (simplex.go:233-293).  The explicit-inverse form needs
// pricing (read A_N), FTRAN (read B^-1) and the rank-1 update (read+write B^-1).  Here the update of pivot
// t-1 and the FTRAN of pivot t share ONE pass over B^-1:
//
//   K_A(t)  k_price_fused : reduce the ratio-test partials of pivot t-1 -> leaving row p; commit pivot t-1
//                           (index swap, trace, counters); y_t = y_{t-1} + (r_q/d_p) * row_p(B^-1_{t-1}) built
//                           directly in LDS; price all nonbasic columns with y_t; partial argmin.
//   K_B(t)  k_update_ftran_fused : reduce the pricing partials -> entering q (or OPTIMAL); one streaming pass
//                           over B^-1: row_i <- row_i - (d_i/d_p) row_p (update t-1, written to the other
//                           buffer), d'_i = row_i_new . a_q (FTRAN t), x_B update, ratio test, partial argmin.
//
// HBM traffic per pivot: 8*[m(n-m) + 2 m^2] bytes instead of 8*[m(n-m) + 3 m^2].
// Every wave owns whole rows; all NV 16-byte loads of a row are issued before the first use (NV KiB in flight
// per wave).  Cross-workgroup data only crosses kernel boundaries; buffers that one kernel both reads globally
// and rewrites (B^-1, y) are ping-ponged by the host.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

// Both kernels keep their dependent global-load chain to three round trips (at m = 2048 the fixed part of
// a pivot kernel, not the streaming, is what limits the rate):
//   round 1  state block + this thread's partials + (K_A) nonbasic[pos] / (K_B) first B^-1 row and old row p
//   round 2  reduction in registers/LDS; the winner carries its payload (d_p, basic[p] / variable id, and the
//            value itself is decoded from the sort key), so no load depends on the winning index
//   round 3  (K_A) row p of B^-1 and y  /  (K_B) entering column a_q

template <int NV>
__global__ __launch_bounds__(kBlock) void k_price_fused(LPArgs a, const double *__restrict__ y_in,
                                                        double *__restrict__ y_out, int pending, int nparts_ratio) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    __shared__ ArgMinP sm[kWavesPerBlock];
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    DevState *st = a.st;
    constexpr int ld2 = NV * 64;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    // ---- round 1: everything that depends on nothing
    const int done = st->done;
    const int q_prev = pending ? st->q : -1;
    const double rq_prev = st->rq;
    ArgMinP best;
    best.k = ~0ull; best.i = 0xFFFFFFFFu; best.u = 0; best.d = 0;
    if (pending) {
        for (int t = threadIdx.x; t < nparts_ratio; t += kBlock) {
            ArgMinP c;
            c.k = a.pk_ratio[t]; c.i = a.pi_ratio[t]; c.u = a.pb_ratio[t]; c.d = a.pd_ratio[t];
            aminp_take(best, c);
        }
    }
    int j0 = -1;
    const bool pre = (wave < a.nn) && (wave != q_prev);
    if (pre) j0 = a.nonbasic[wave];
    if (done) return;
    // this wave's first column goes in flight now; the one wave that owns position q_prev must wait for `lea`
    double2 v[NV];
    if (pre) {
        const double2 *r2 = reinterpret_cast<const double2 *>(a.At + (size_t)j0 * a.ld);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = r2[lane + 64 * k];
    }
    int lea = -1;
    if (pending) {
        // ---- round 2: leaving row of pivot t-1
        block_argminp(best, sm);
        const int p = (int)best.i;
        const double mv = orddecode(best.k);
        if (mv == __builtin_inf()) {  // simplex.go:328-330
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_UNBOUNDED; st->p = p; st->mv = mv; }
            return;
        }
        if (mv <= 0) {  // simplex.go:269 -> Bland
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_NEED_BLAND; st->p = p; st->mv = mv; }
            return;
        }
        const double dpv = best.d;
        const double alpha = rq_prev / dpv;
        lea = (int)best.u;
        // ---- round 3: y_t = y_{t-1} + (r_q/d_p) * row_p(B^-1_{t-1}), built directly in LDS
        const double2 *rp = reinterpret_cast<const double2 *>(a.binv_cur + (size_t)p * a.ld);
        const double2 *yi = reinterpret_cast<const double2 *>(y_in);
        double2 *yo = reinterpret_cast<double2 *>(y_out);
        for (int c = threadIdx.x; c < ld2; c += kBlock) {
            double2 w = yi[c];
            const double2 r = rp[c];
            w.x = w.x + alpha * r.x;
            w.y = w.y + alpha * r.y;
            svec[c] = w;
            if (blockIdx.x == 0) yo[c] = w;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            // commit pivot t-1 (simplex.go:280); basic[p] itself is rewritten by K_B(t)
            const int ent = st->ent_cur;
            st->p = p; st->dp = dpv; st->mv = mv; st->theta = a.xb[p] / dpv;
            st->ent_prev = ent; st->lea = lea;
            a.nonbasic[q_prev] = lea;
            if (a.trace && st->trace_len < a.trace_cap) {
                DevPivot &t = a.trace[st->trace_len];
                t.phase = a.phase; t.bland = 0; t.min_idx = q_prev; t.replace = p; t.entering = ent; t.leaving = lea;
            }
            st->trace_len += 1;
            st->pivots += 1;
        }
        __syncthreads();
    } else {
        stage_vec(svec, y_in, ld2);
    }
    unsigned long long bk = ~0ull;
    unsigned int bi = 0xFFFFFFFFu, bv = 0;
    for (int pos = wave; pos < a.nn; pos += nwaves) {
        int j = j0;
        if (!(pre && pos == wave)) {
            j = (pos == q_prev) ? lea : a.nonbasic[pos];
            const double2 *r2 = reinterpret_cast<const double2 *>(a.At + (size_t)j * a.ld);
#pragma unroll
            for (int k = 0; k < NV; k++) v[k] = r2[lane + 64 * k];
        }
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const double2 s0 = svec[lane + 64 * k];
            acc[k & 3] += v[k].x * s0.x + v[k].y * s0.y;
        }
        const double dot = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
        const double r = a.cost[j] - dot;
        if (lane == 0) a.rvec[pos] = r;
        const unsigned long long key = ordkey(r);
        if (key < bk || (key == bk && (unsigned int)pos < bi)) { bk = key; bi = (unsigned int)pos; bv = (unsigned int)j; }
    }
    // workgroup winner with its variable id
    ArgMinP mine;
    mine.k = bk; mine.i = bi; mine.u = bv; mine.d = 0;
    block_argminp(mine, sm);
    if (threadIdx.x == 0) { a.pk_price[blockIdx.x] = mine.k; a.pi_price[blockIdx.x] = mine.i; a.pv_price[blockIdx.x] = mine.u; }
    (void)sk; (void)si;
}

template <int NV>
__global__ __launch_bounds__(kBlock) void k_update_ftran_fused(LPArgs a, int pending, int nparts_price) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    __shared__ ArgMinP sm[kWavesPerBlock];
    DevState *st = a.st;
    constexpr int ld2 = NV * 64;
    double2 *svecA = svec;        // entering column a_q
    double2 *svecP = svec + ld2;  // old row p of B^-1
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    // ---- round 1: state, partials, first row, old row p
    const int done = st->done;
    const int p = pending ? st->p : -1;
    const double dpv = pending ? st->dp : 1.0;
    const double theta = pending ? st->theta : 0.0;
    ArgMinP best;
    best.k = ~0ull; best.i = 0xFFFFFFFFu; best.u = 0; best.d = 0;
    for (int t = threadIdx.x; t < nparts_price; t += kBlock) {
        ArgMinP c;
        c.k = a.pk_price[t]; c.i = a.pi_price[t]; c.u = a.pv_price[t]; c.d = 0;
        aminp_take(best, c);
    }
    if (done) return;
    double2 v[NV];
    if (wave < a.m) {
        const double2 *src = reinterpret_cast<const double2 *>(a.binv_cur + (size_t)wave * a.ld);
#pragma unroll
        for (int k = 0; k < NV; k++) v[k] = src[lane + 64 * k];
    }
    if (pending) {
        const double2 *rp = reinterpret_cast<const double2 *>(a.binv_cur + (size_t)p * a.ld);
        for (int c = threadIdx.x; c < ld2; c += kBlock) svecP[c] = rp[c];
    }
    // ---- round 2: entering column of pivot t
    block_argminp(best, sm);
    const int q = (int)best.i;
    const int var = (int)best.u;
    const double rq = orddecode(best.k);
    const bool optimal = (rq >= -a.tol);  // simplex.go:248
    // ---- round 3: a_q
    if (!optimal) {
        const double2 *aq = reinterpret_cast<const double2 *>(a.At + (size_t)var * a.ld);
        for (int c = threadIdx.x; c < ld2; c += kBlock) svecA[c] = aq[c];
    }
    __syncthreads();
    if (!pending && optimal) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_OPTIMAL; st->q = q; st->rq = rq; }
        return;
    }
    unsigned long long bk = ~0ull;
    unsigned int bi = 0xFFFFFFFFu;
    double bd = 0;
    for (int i = wave; i < a.m; i += nwaves) {
        if (i != wave) {
            const double2 *src = reinterpret_cast<const double2 *>(a.binv_cur + (size_t)i * a.ld);
#pragma unroll
            for (int k = 0; k < NV; k++) v[k] = src[lane + 64 * k];
        }
        const double di_old = pending ? a.dvec[i] : 0.0;
        double xbi = a.xb[i];
        if (pending) {
            double2 *dst = reinterpret_cast<double2 *>(a.binv_next + (size_t)i * a.ld);
            if (i == p) {
#pragma unroll
                for (int k = 0; k < NV; k++) { v[k].x = v[k].x / dpv; v[k].y = v[k].y / dpv; dst[lane + 64 * k] = v[k]; }
                xbi = theta;
            } else {
                const double f = di_old / dpv;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const double2 rp = svecP[lane + 64 * k];
                    v[k].x = v[k].x - f * rp.x;
                    v[k].y = v[k].y - f * rp.y;
                    dst[lane + 64 * k] = v[k];
                }
                xbi = xbi - theta * di_old;
            }
            if (lane == 0) a.xb[i] = xbi;
        }
        if (!optimal) {
            double acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NV; k++) {
                const double2 s0 = svecA[lane + 64 * k];
                acc[k & 3] += v[k].x * s0.x + v[k].y * s0.y;
            }
            const double dnew = wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
            double d = -dnew;                  // simplex.go:319
            if (fabs(d) < 1e-13) d = 0;        // :321-325
            const double mv = (d >= 0) ? __builtin_inf() : xbi / fabs(d);  // :334-340
            if (lane == 0) { a.dvec[i] = dnew; a.move[i] = mv; }
            const unsigned long long key = ordkey(mv);
            if (key < bk || (key == bk && (unsigned int)i < bi)) { bk = key; bi = (unsigned int)i; bd = dnew; }
        }
    }
    if (!optimal) {
        ArgMinP mine;
        mine.k = bk; mine.i = bi; mine.d = bd;
        // basic[] is stable in this kernel except position p, rewritten below by (0,0) to ent_prev: read it through
        // the same substitution so that the payload is the post-commit value
        mine.u = (bi == 0xFFFFFFFFu) ? 0u : (unsigned int)(((int)bi == p) ? st->ent_prev : a.basic[bi]);
        block_argminp(mine, sm);
        if (threadIdx.x == 0) {
            a.pk_ratio[blockIdx.x] = mine.k; a.pi_ratio[blockIdx.x] = mine.i; a.pd_ratio[blockIdx.x] = mine.d;
            a.pb_ratio[blockIdx.x] = mine.u;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (pending) a.basic[p] = st->ent_prev;
        st->q = q; st->rq = rq;
        if (optimal) { st->done = 1; st->status = ST_OPTIMAL; }
        else st->ent_cur = var;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers.  ev_start/ev_stop (nullable) are attached to the dispatch itself
// (hipExtLaunchKernelGGL), so their elapsed time is the kernel's own duration, not launch gaps.
// ------------------------------------------------------------------------------------------------

bool fused_supported(int ld) {
    if (ld % 128 != 0) return false;
    const int nv = ld / 128;
    return nv == 1 || nv == 2 || nv == 4 || nv == 8 || nv == 16 || nv == 32;
}

static inline int grid_rows(int rows) {
    int g = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > kMaxPartials) g = kMaxPartials;
    return g < 1 ? 1 : g;
}

template <int NV>
static int launch_price_fused_t(const LPArgs &a, const double *y_in, double *y_out, int pending, int nparts_ratio,
                                hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int g = grid_rows(a.nn);
    hipExtLaunchKernelGGL((k_price_fused<NV>), dim3(g), dim3(kBlock), (size_t)a.ld * sizeof(double), s, e0, e1, 0, a, y_in,
                          y_out, pending, nparts_ratio);
    return g;
}
template <int NV>
static int launch_uf_fused_t(const LPArgs &a, int pending, int nparts_price, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int g = grid_rows(a.m);
    if (NV == 32) {  // 2 x 32 KiB of staged vectors + the reduction scratch exceed the default 64 KiB dynamic-LDS cap
        lds_attr_once(reinterpret_cast<const void *>(&k_update_ftran_fused<NV>), 2 * 4096 * (int)sizeof(double));
    }
    hipExtLaunchKernelGGL((k_update_ftran_fused<NV>), dim3(g), dim3(kBlock), (size_t)a.ld * 2 * sizeof(double), s, e0, e1, 0,
                          a, pending, nparts_price);
    return g;
}

int launch_price_fused(const LPArgs &a, const double *y_in, double *y_out, int pending, int nparts_ratio, hipStream_t s,
                       hipEvent_t e0, hipEvent_t e1) {
    switch (a.ld / 128) {
        case 1: return launch_price_fused_t<1>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
        case 2: return launch_price_fused_t<2>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
        case 4: return launch_price_fused_t<4>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
        case 8: return launch_price_fused_t<8>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
        case 16: return launch_price_fused_t<16>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
        case 32: return launch_price_fused_t<32>(a, y_in, y_out, pending, nparts_ratio, s, e0, e1);
    }
    return -1;
}
int launch_update_ftran_fused(const LPArgs &a, int pending, int nparts_price, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    switch (a.ld / 128) {
        case 1: return launch_uf_fused_t<1>(a, pending, nparts_price, s, e0, e1);
        case 2: return launch_uf_fused_t<2>(a, pending, nparts_price, s, e0, e1);
        case 4: return launch_uf_fused_t<4>(a, pending, nparts_price, s, e0, e1);
        case 8: return launch_uf_fused_t<8>(a, pending, nparts_price, s, e0, e1);
        case 16: return launch_uf_fused_t<16>(a, pending, nparts_price, s, e0, e1);
        case 32: return launch_uf_fused_t<32>(a, pending, nparts_price, s, e0, e1);
    }
    return -1;
}

}  // namespace gomilp
