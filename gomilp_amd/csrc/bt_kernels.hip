// Blocked tableau pivoting with deferred rank-K updates (gfx950) — the default pipeline when n - m < 2m.
//
// A simplex pivot only LOOKS at one column (the entering one: ratio test) and one row (the leaving one:
// reduced-cost update) of the tableau T = B^-1 A_N; the rank-1 update of all m*(n-m) entries is bookkeeping that
// can be deferred.  So pivots run in blocks of K:
//
//   k_bt_inner2  ONE workgroup performs up to K = 8 complete pivots (shapes up to 2048 rows / nonbasic columns; larger
//   k_bt_inner   ones use the older k_bt_inner with K = 16 and the block terms re-read through L2).  It reads the
//                needed column and row of the stale T and corrects them with the block's earlier rank-1 terms,
//                   T_cur = T_stale + sum_j u_j v_j'^T,
//                takes every decision exactly like the other pipelines (first-index argmin, 1e-13 rounding, unbounded /
//                degenerate tests, Bland rule of simplex.go:347-383 in-kernel), and emits u_k (m) and v_k' (n-m) per
//                pivot.  No grid-wide step, no launch per pivot.  k_bt_inner2: own terms in registers, two barriers
//                per pivot, T in 4x4 tiles (see the comments at the kernel).
//   k_bt_update_tiled / k_bt_update
//                all CUs: T += sum_k u_k v_k'^T in one streaming read+write pass (K multiply-adds per element, in
//                place — an element depends only on its own old value), 8 rows of loads in flight per lane.
//
// HBM traffic per pivot: 16*m*(n-m)/K + O(K*(m + n-m)) bytes instead of 16*m*(n-m).
// One pivot as a pure rank-1 term:  with d = column q, v = row p, d_p = pivot element:
//   u_i = -d_i/d_p (i != p), u_p = 1/d_p - 1;  v'_j = v_j (j != q), v'_q = d_p + 1
//   => T + u v'^T has row p = v/d_p, rows i = T_i - (d_i/d_p) v, and column q = the leaving variable's column.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "device_types.h"
#include "kernels_common.h"
#include "bt_loop.h"
#include "batch_dev.h"

namespace gomilp {

constexpr int kBtThreads = 1024;
constexpr int kBtWaves = kBtThreads / 64;
constexpr int kBtMaxK = 32;
constexpr int kStampSegs = 16;   // == kBtStampSegs (engine.hpp)

// RI / CJ: rows / columns per thread (m <= RI*1024, n-m <= CJ*1024).
// KREG > 0: the block's rank-1 terms of a thread's OWN rows / columns live in registers (newest first, shifted
// every pivot), foreign scalars travel through LDS — the single CU that runs this kernel then touches global memory
// only for the stale column, the stale row and the u_k / v_k stores.  KREG = 0: terms are re-read from global (L2).
template <int NT, int RI, int CJ, int KREG>
__global__ __launch_bounds__(NT) void k_bt_inner(BTArgs a) {
    constexpr int kBtThreads = NT;   // shadows the namespace-level default inside this kernel
    constexpr int kBtWaves = NT / 64;
    constexpr int KR = KREG > 0 ? KREG : 1;
    double ureg[RI][KR], vreg[CJ][KR];
#pragma unroll
    for (int s = 0; s < RI; s++)
#pragma unroll
        for (int j = 0; j < KR; j++) ureg[s][j] = 0;
#pragma unroll
    for (int s = 0; s < CJ; s++)
#pragma unroll
        for (int j = 0; j < KR; j++) vreg[s][j] = 0;
    extern __shared__ __attribute__((aligned(16))) double sh[];
    double *r_s = sh;                 // ldt
    double *xb_s = sh + a.ldt;        // ldu
    int *basic_s = reinterpret_cast<int *>(sh + a.ldt + a.ldu);  // m   (the commit is LDS-only: no dependent global loads
    int *nonbasic_s = basic_s + a.ldu;                           // nn   on the critical path; written back at the end)
    __shared__ BtCand sm2[2 * 16];
    int sm_par = 0;
    __shared__ double vq[kBtMaxK], up[kBtMaxK];
    __shared__ double s_bcast[2];
    DevState *st = a.st;
    const int tid = threadIdx.x;
    if (st->done) {
        if (tid == 0) st->kdone = 0;
        return;
    }
    for (int j = tid; j < a.ldt; j += kBtThreads) r_s[j] = a.r[j];
    for (int i = tid; i < a.ldu; i += kBtThreads) xb_s[i] = a.xb[i];
    for (int i = tid; i < a.m; i += kBtThreads) basic_s[i] = a.basic[i];
    for (int j = tid; j < a.nn; j += kBtThreads) nonbasic_s[j] = a.nonbasic[j];
    __syncthreads();
    const double inf = __builtin_inf();
    int kd = 0, status = ST_RUNNING, blands = 0;

    // column q of the current tableau for this thread's rows
    auto column = [&](int q, int k, double (&dcol)[RI]) {
        if (KREG > 0) {
            // the owner of column q publishes its K newest-first v entries (static register indices)
            if (tid == (q & (kBtThreads - 1))) {
                const int sq = q / kBtThreads;
#pragma unroll
                for (int s = 0; s < CJ; s++)
                    if (s == sq) {
#pragma unroll
                        for (int j = 0; j < KR; j++) vq[j] = vreg[s][j];
                    }
            }
        } else if (tid < k) {
            vq[tid] = __hip_atomic_load(a.V + (size_t)tid * a.ldt + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * kBtThreads;
            double d = 0;
            if (i < a.m) {
                d = a.T[(size_t)i * a.ldt + q];
                if (KREG > 0) {
#pragma unroll
                    for (int j = 0; j < KR; j++) d += ureg[s][j] * vq[j];
                } else {
                    // batches of 8 independent loads: a plain `for (j < k)` serialises one L2 round trip per term
                    for (int j0 = 0; j0 < k; j0 += 8) {
                        double uu[8];
#pragma unroll
                        for (int t = 0; t < 8; t++) uu[t] = (j0 + t < k) ? a.U[(size_t)(j0 + t) * a.ldu + i] : 0.0;
#pragma unroll
                        for (int t = 0; t < 8; t++) d += uu[t] * ((j0 + t < k) ? vq[j0 + t] : 0.0);
                    }
                }
            }
            dcol[s] = d;
        }
    };
    // ratio vector (simplex.go:321-340) and its first-index argmin, winner carries d_i
    auto ratio = [&](const double (&dcol)[RI], double (&mvv)[RI]) -> BtCand {
        BtCand c;
        c.k = ~0ull; c.i = 0xFFFFFFFFu;
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * kBtThreads;
            mvv[s] = inf;
            if (i < a.m) {
                double d = -dcol[s];
                if (fabs(d) < 1e-13) d = 0;
                mvv[s] = (d >= 0) ? inf : xb_s[i] / fabs(d);
                BtCand b;
                b.k = ordkey(mvv[s]); b.i = (unsigned int)i;
                bt_take(c, b);
            }
        }
        bt_block_argmin<kBtWaves>(c, sm2 + kBtWaves * (sm_par ^= 1));
        return c;
    };

    for (int k = 0; k < a.kmax; k++) {
        const bool forced = (k == 0 && a.forced_q >= 0);
        int q, p;
        double rq, dpv = 1.0;
        bool bland = false;
        double dcol[RI], mvv[RI];
        if (!forced) {
            // ---- entering position: first index of min r (simplex.go:247)
            BtCand c;
            c.k = ~0ull; c.i = 0xFFFFFFFFu;
            for (int j = tid; j < a.nn; j += kBtThreads) {
                BtCand b;
                b.k = ordkey(r_s[j]); b.i = (unsigned int)j;
                bt_take(c, b);
            }
            bt_block_argmin<kBtWaves>(c, sm2 + kBtWaves * (sm_par ^= 1));
            q = (int)c.i;
            rq = r_s[q];
            if (a.guard == inf && !(k == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // strict mode (knob exact_degenerate = 3): the stop test too is the host's, on fresh reduced costs
            if (rq >= -a.tol) { status = ST_OPTIMAL; break; }  // simplex.go:248
            column(q, k, dcol);
            BtCand w = ratio(dcol, mvv);
            p = (int)w.i;
            const double mv = orddecode(w.k);
            if (mv == inf) { status = ST_UNBOUNDED; break; }  // simplex.go:328-330
            if (a.guard > 0 && mv <= a.guard && !(k == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // degenerate (or nearly): decided on a fresh x_B
            if (mv <= 0) {
                // ---- replaceBland (simplex.go:347-383): candidates in position order with r_i <= -1e-14 after the
                // 1e-13 rounding of :252-256; the mat.Cond guard of :377 is replaced by |d| >= 1e-13
                bland = true;
                blands++;
                int cand = -1;
                bool found = false;
                for (;;) {
                    BtCand f;
                    f.k = ~0ull; f.i = 0xFFFFFFFFu;
                    for (int j = tid; j < a.nn; j += kBtThreads) {
                        if (j <= cand) continue;
                        double rv = r_s[j];
                        if (fabs(rv) < 1e-13) rv = 0;
                        if (!(rv > -1e-14)) { f.k = 0; f.i = (unsigned int)j; break; }  // first such j of this thread
                    }
                    bt_block_argmin<kBtWaves>(f, sm2 + kBtWaves * (sm_par ^= 1));
                    if (f.i == 0xFFFFFFFFu) break;  // candidates exhausted -> ErrBland
                    cand = (int)f.i;
                    column(cand, k, dcol);
                    BtCand w2 = ratio(dcol, mvv);
                    const double mv2 = orddecode(w2.k);
                    if (mv2 == inf) { status = ST_UNBOUNDED; break; }  // computeMove inside Bland, :356-360
                    if (fabs(mv2) > 1e-12) { q = cand; p = (int)w2.i; found = true; break; }  // :362
                    BtCand g;
                    g.k = ~0ull; g.i = 0xFFFFFFFFu;
#pragma unroll
                    for (int s = 0; s < RI; s++) {
                        const int i = tid + s * kBtThreads;
                        if (i < a.m && !(mvv[s] > 1e-12)) { BtCand b; b.k = 0; b.i = (unsigned int)i; bt_take(g, b); }
                    }
                    bt_block_argmin<kBtWaves>(g, sm2 + kBtWaves * (sm_par ^= 1));
                    if (g.i != 0xFFFFFFFFu) { q = cand; p = (int)g.i; found = true; break; }  // :368-379
                }
                if (status == ST_UNBOUNDED) break;
                if (!found) { status = ST_BLAND_FAILED; break; }
                rq = r_s[q];
            }
        } else {
            q = a.forced_q; p = a.forced_p;
            rq = a.forced_nocommit ? 0.0 : r_s[q];   // a set-up pivot leaves the reduced costs alone (they are rebuilt); a pivot the host decided on fresh solves (exact_step) is a pivot like any other
            column(q, k, dcol);
        }
        // ---- row p of the current tableau for this thread's columns; its owner publishes the pivot element d_p and
        // x_B[p] (so the reductions carry no payload) and, when the block terms live in registers, its u entries
        if (tid == (p & (kBtThreads - 1))) {
            const int sp = p / kBtThreads;
#pragma unroll
            for (int s = 0; s < RI; s++)
                if (s == sp) {
                    s_bcast[0] = dcol[s];
                    s_bcast[1] = xb_s[p];
                    if (KREG > 0) {
#pragma unroll
                        for (int j = 0; j < KR; j++) up[j] = ureg[s][j];
                    }
                }
        }
        if (KREG == 0 && tid < k) up[tid] = __hip_atomic_load(a.U + (size_t)tid * a.ldu + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        dpv = s_bcast[0];
        const double mult = rq / dpv;
        const double theta = s_bcast[1] / dpv;
        double *Vk = a.V + (size_t)k * a.ldt;
        double *Uk = a.U + (size_t)k * a.ldu;
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * kBtThreads;
            if (j < a.ldt) {
                double v = 0;
                if (j < a.nn) {
                    v = a.T[(size_t)p * a.ldt + j];
                    if (KREG > 0) {
#pragma unroll
                        for (int jj = 0; jj < KR; jj++) v += up[jj] * vreg[s][jj];
                    } else {
                        for (int j0 = 0; j0 < k; j0 += 8) {
                            double vv8[8];
#pragma unroll
                            for (int t = 0; t < 8; t++) vv8[t] = (j0 + t < k) ? a.V[(size_t)(j0 + t) * a.ldt + j] : 0.0;
#pragma unroll
                            for (int t = 0; t < 8; t++) v += ((j0 + t < k) ? up[j0 + t] : 0.0) * vv8[t];
                        }
                    }
                    // reduced costs (positional): r_j - (r_q/d_p) v_j ; the leaving variable takes slot q
                    r_s[j] = (j == q) ? -mult : r_s[j] - mult * v;
                }
                const double vprime = (j == q) ? dpv + 1.0 : v;
                Vk[j] = vprime;
                if (KREG > 0) {
#pragma unroll
                    for (int jj = KR - 1; jj > 0; jj--) vreg[s][jj] = vreg[s][jj - 1];
                    vreg[s][0] = vprime;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * kBtThreads;
            if (i < a.ldu) {
                double u = 0;
                if (i < a.m) {
                    u = (i == p) ? 1.0 / dpv - 1.0 : -dcol[s] / dpv;
                    xb_s[i] = (i == p) ? theta : xb_s[i] - theta * dcol[s];
                }
                Uk[i] = u;
                if (KREG > 0) {
#pragma unroll
                    for (int jj = KR - 1; jj > 0; jj--) ureg[s][jj] = ureg[s][jj - 1];
                    ureg[s][0] = u;
                }
            }
        }
        if (tid == 0 && !(forced && a.forced_nocommit)) {  // simplex.go:280
            const int ent = nonbasic_s[q], lea = basic_s[p];
            basic_s[p] = ent; nonbasic_s[q] = lea;
            if (a.trace && st->trace_len < a.trace_cap) {
                DevPivot &tr = a.trace[st->trace_len];
                tr.phase = a.phase; tr.bland = bland ? 1 : 0; tr.min_idx = q; tr.replace = p; tr.entering = ent; tr.leaving = lea;
            }
            st->trace_len += 1;
            st->pivots += 1;
        }
        kd = k + 1;
        if (KREG == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // u_k / v_k reach L2 before the sc1 loads of other waves
        __syncthreads();
    }
    for (int j = tid; j < a.ldt; j += kBtThreads) a.r[j] = r_s[j];
    for (int i = tid; i < a.ldu; i += kBtThreads) a.xb[i] = xb_s[i];
    for (int i = tid; i < a.m; i += kBtThreads) a.basic[i] = basic_s[i];
    for (int j = tid; j < a.nn; j += kBtThreads) a.nonbasic[j] = nonbasic_s[j];
    if (tid == 0) {
        st->kdone = kd;
        st->bland_steps += blands;
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

// ---- register-resident variant (m <= RI*NT, ldt <= CJ*NT, KR block terms in registers) ----------------------------
// k_bt_inner above is VALU-issue bound (about 1600 instructions per wave and pivot, 4 waves per SIMD): (key, index)
// candidates dragged through 64-bit integer compares, five barriers, r / x_B traffic through LDS.  This variant:
//  * a thread's OWN block terms u_k[i], v_k'[j] live in registers (newest first, shifted each pivot) — except the v'
//    terms of the last VL column slots, which sit in an LDS ring over k where 128 VGPRs per thread (NT = 1024) are not
//    enough; its r_j and x_B[i] in LDS slots only the owner touches: no barrier guards any of it;
//  * first-index argmin without index payloads: v_min_f64 over DPP row rotations gives the minimum M, then
//    ballot(value == M) + ff1 gives the first lane, slot by slot — exactly floats.MinIdx (NaN never wins because
//    v_min_f64 returns the other operand, -0 == +0, first index among equals);
//  * the lane that owns a wave's winner publishes, next to the wave's (M, index), the scalars everybody needs about
//    it (r_q and v'_k[q]; d_p, x_B[p] and u_k[p]), so the two reductions are the only two barriers of a pivot;
//  * one division per pivot for the rank-1 term (u_i = d_i * (-1/d_p)) and fused multiply-adds for the block
//    corrections: these are the engine's own running quantities, not values the reference defines bit by bit
//    (the returned x comes from the gonum-order solve of the final basis).
__device__ __forceinline__ unsigned int tile_off(unsigned int i, unsigned int j, unsigned int ldt);
struct BtWin { double m; unsigned int i; };   // minimum and the first index that attains it (0xFFFFFFFF: none, all NaN)

// LOOP: the pivot role of the batched persistent loop kernel (k_b_loop below): up to a.nblocks blocks of KR / 2 pivots in one launch; the
// registers hold the running block's terms AND the previous block's (not yet in the tableau buffer this block reads: the update
// workgroups of the same launch apply them beside this block) — the protocol of k_bt_loop (bt_loop.h) with ONE pivot workgroup.
// VIRT: the tableau does not exist in HBM yet (BatchLP::virt > 0: the set-up pivot and the first block of a wide wave): the column and the row
// a pivot looks at are computed — T0 from the root's resident A and the child's branch rows, plus the set-up pivot's rank-1 term (batch_dev.h
// b_virt_entry: the bits a read of the materialised tableau would return); the block's terms go on top as always
template <int NT, int RI, int CJ, int KR, int VL, bool STAMP = false, bool LOOP = false, bool VIRT = false>
__device__ __forceinline__ void bt_inner2_body(const BTArgs &a, const int nupd = 0, const BatchLP *vl = nullptr) {
    constexpr int NW = NT / 64;
    static_assert(!LOOP || (VL == 0 && !STAMP && (KR & 1) == 0), "loop mode: all terms in registers, KR / 2 lagging + KR / 2 current");
    static_assert(!VIRT || (!LOOP && !STAMP), "virtual tableau: the plain block kernel only");
    constexpr int KB = LOOP ? KR / 2 : KR;   // pivots per block
    // STAMP: diagnostic build (context knob "bt_stamps"): every wave sums the shader cycles it spends in each segment of
    // a pivot (s_memtime, cdna_hip_programming.md §7 "In-kernel stamps") and adds them to a.stamps[wave][segment]; the
    // loads are waited for where a segment ends, so this build's run time is not the product kernel's
    unsigned long long tacc[kStampSegs] = {};
    unsigned long long tprev = 0;
    auto stamp = [&](auto seg) {
        if constexpr (STAMP) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            constexpr int S = decltype(seg)::value;
            if (S >= 0) tacc[S >= 0 ? S : 0] += t - tprev;
            tprev = t;
        }
    };
    auto drain = [&]() { if constexpr (STAMP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
#define BT_STAMP(S) stamp(std::integral_constant<int, S>())
    extern __shared__ __attribute__((aligned(16))) double sh2[];
    // fixed offsets (not a.ldu / a.ldt): every slot address is 8*tid + a compile-time constant, one VGPR for all of them
    double *xb_s = sh2;              // RI*NT  x_B and r: every thread touches only its own rows / columns (no barrier
    double *r_s = sh2 + RI * NT;     // CJ*NT  needed); kept out of the register file, which the block terms fill
    int *basic_s = reinterpret_cast<int *>(sh2 + RI * NT + CJ * NT);  // RI*NT  only thread 0 touches the lists in the loop
    int *nonbasic_s = basic_s + RI * NT;                              // CJ*NT
    // the block terms v'_k of the LAST VL column slots live in LDS (a ring over k: no shifting), the others in registers:
    // at NT = 1024 the register file holds 128 VGPRs per thread, 16 short of what all terms in registers need
    double *vl_s = reinterpret_cast<double *>(nonbasic_s + CJ * NT);   // VL*KR*NT, element (slot, ring position, thread)
    constexpr int CR = CJ - VL;   // column slots with register-resident terms
    int vhead = 0;                // ring position of the newest term
    // virtual tableau: the set-up pivot's term u0 (by row), v0' (by column) for the scalar look-ups (own rows / columns: registers)
    double *u0_s = vl_s + VL * KR * NT;   // RI*NT
    double *v0_s = u0_s + RI * NT;        // CJ*NT
    double u0r[RI], v0r[CJ];
    __shared__ double redMA[16], redMB[16];
    __shared__ unsigned int redIA[16], redIB[16];
    __shared__ double payA[16][KR + 1];  // per wave: r_q, v'_k[q]
    __shared__ double payB[16][KR + 2];  // per wave: d_p, x_B[p], u_k[p]
    __shared__ double red2[16];          // guard mode: per-wave runner-up of the ratio test
    DevState *st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    // prologue: every load is issued before the first one is waited for (the `done` test used to add a full memory
    // round trip in front of the state loads of every launch)
    const int done = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double inf = __builtin_inf();
    const unsigned int ldt = (unsigned int)a.ldt;
    const char *Tb = reinterpret_cast<const char *>(a.T);   // byte offsets in 32 bits: saddr + voffset addressing
    // loop mode: the buffer the update workgroups finished two blocks ago — they write it from other CUs while this launch runs: agent scope
    auto ldT = [&](unsigned int elem) -> double {
        if constexpr (LOOP) return ld_agent(reinterpret_cast<const double *>(Tb + (elem << 3)));
        else return *reinterpret_cast<const double *>(Tb + (elem << 3));
    };
    // loop mode: counters and buffer choice handed from launch to launch (bt_loop.h; G = 1 pivot workgroup)
    const int sel0 = LOOP ? (a.par ? st->tsel2[1] : st->tsel2[0]) : 0;
    const double *hdr_in = a.xbuf + 1 + 5 * (LOOP ? a.par : 0);
    double *hdr_out = a.xbuf + 1 + 5 * (LOOP ? (a.par ^ 1) : 0);
    unsigned int blk_base = 0, upd_base[4] = {0, 0, 0, 0};
    if constexpr (LOOP) {
        blk_base = (unsigned int)(unsigned long long)hdr_in[0];
#pragma unroll
        for (int j = 0; j < 4; j++) upd_base[j] = (unsigned int)(unsigned long long)hdr_in[1 + j];
    }
    unsigned int *blk_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync), *upd_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync + 16);
    auto hand_on = [&](int nb) {   // counters in step for the next launch after nb blocks
        hdr_out[0] = (double)(unsigned int)(blk_base + (unsigned int)nb);
#pragma unroll
        for (int j = 0; j < 4; j++) hdr_out[1 + j] = (double)(unsigned int)(upd_base[j] + (unsigned int)nupd * (unsigned int)(nb > j ? (nb - 1 - j) / 4 + 1 : 0));
    };
    __shared__ int s_ok;
    double x0[RI], r0[CJ];
    int b0[RI], n0[CJ];
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = tid + s * NT;
        x0[s] = i < a.m ? a.xb[i] : 0.0;
        b0[s] = i < a.m ? a.basic[i] : 0;
    }
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        r0[s] = j < a.nn ? a.r[j] : inf;   // padding never wins an argmin
        n0[s] = j < a.nn ? a.nonbasic[j] : 0;
    }
    if (done) {
        if constexpr (LOOP) {
            // a launch behind the end of the loop: release the update workgroups (they wait for block 0; nothing to apply) and keep
            // counters and buffer choice in step
            if (tid == 0) {
                __hip_atomic_store(&st->kdone2[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_fetch_add(blk_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                st->tsel2[a.par ^ 1] = sel0;
                st->loop_blocks = 0;
                hand_on(1);
            }
        } else if (tid == 0) st->kdone = 0;
        return;
    }
    double ureg[RI][KR], vreg[CR > 0 ? CR : 1][KR];
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = tid + s * NT;
        xb_s[i] = x0[s];
        basic_s[i] = b0[s];
#pragma unroll
        for (int j = 0; j < KR; j++) ureg[s][j] = 0;
    }
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        r_s[j] = r0[s];
        nonbasic_s[j] = n0[s];
#pragma unroll
        for (int j2 = 0; j2 < KR; j2++) {
            if (s < CR) vreg[s < CR ? s : 0][j2] = 0;
            else vl_s[((s - CR) * KR + j2) * NT + tid] = 0;
        }
    }
    if (tid < KR + 1) payA[0][tid] = 0;   // a host-chosen first pivot reads v'_k[q] = 0 from here
    if constexpr (VIRT) {
        const bool t0 = vl->virt_t0 != 0;
        const double *U8 = a.U + (size_t)8 * a.ldu, *V8 = a.V + (size_t)8 * a.ldt;   // the set-up pivot's term: row 8 (the block's own terms take rows 0 .. 7)
#pragma unroll
        for (int s = 0; s < RI; s++) { const int i = tid + s * NT; u0r[s] = (t0 && i < a.m) ? U8[i] : 0.0; u0_s[i] = u0r[s]; }
#pragma unroll
        for (int s = 0; s < CJ; s++) { const int j = tid + s * NT; v0r[s] = (t0 && j < a.ldt) ? V8[j] : 0.0; v0_s[j] = v0r[s]; }
    }
    __syncthreads();
    int kd = 0, status = ST_RUNNING, blands = 0;
    // thread 0 keeps the pivot / trace counters in registers: a global read-modify-write per pivot would stall its wave
    // (and, at the next barrier, everybody) for a memory round trip
    long long trace_len = 0, npiv = 0;
    if (tid == 0) { trace_len = st->trace_len; npiv = st->pivots; }

    // ---- floats.MinIdx over N slots per thread (slot s of thread t is index t + s*NT)
    auto wave_first_min = [&](auto &val, auto nslots) -> BtWin {
        constexpr int N = decltype(nslots)::value;
        double x = val[0];
#pragma unroll
        for (int s = 1; s < N; s++) x = vmin_f64(x, val[s]);
        BtWin w;
        w.m = wave_min_f64(x);
        w.i = 0xFFFFFFFFu;
#pragma unroll
        for (int s = N - 1; s >= 0; s--) {
            const unsigned long long mask = __ballot(val[s] == w.m);
            if (mask) w.i = (unsigned int)(s * NT + wbase + __builtin_ctzll(mask));
        }
        return w;
    };
    auto block_first_min = [&](const double *redM, const unsigned int *redI) -> BtWin {
        const double x = lane < NW ? redM[lane] : inf;
        const unsigned int ii = lane < NW ? redI[lane] : 0xFFFFFFFFu;
        BtWin f;
        f.m = readlane_f64(row_min_f64(x), 15);
        const unsigned int key = (x == f.m) ? ii : 0xFFFFFFFFu;
        f.i = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(key), 15);
        return f;
    };
    // entering column: returns (min, q); r_q and v'_k[q] through payA
    auto reduce_cols = [&](double (&val)[CJ], double &rq, const double *&vq) -> BtWin {
        const BtWin w = wave_first_min(val, std::integral_constant<int, CJ>());
#pragma unroll
        for (int s = 0; s < CJ; s++)
            if ((unsigned int)(tid + s * NT) == w.i) {
                payA[wv][0] = r_s[tid + s * NT];
#pragma unroll
                for (int j = 0; j < KR; j++)   // newest first
                    payA[wv][1 + j] = s < CR ? vreg[s < CR ? s : 0][j] : vl_s[((s - CR) * KR + ((vhead - j) & (KR - 1))) * NT + tid];
            }
        if (lane == 0) { redMA[wv] = w.m; redIA[wv] = w.i; }
        BT_STAMP(0);    // r from LDS, wave-level first-min, payload
        __syncthreads();
        BT_STAMP(1);    // barrier A
        const BtWin f = block_first_min(redMA, redIA);
        const int ww = (f.i & (NT - 1)) >> 6;
        rq = payA[ww][0];
        vq = &payA[ww][1];
        BT_STAMP(2);    // block-level first-min A
        return f;
    };
    // leaving row: returns (min, p); d_p, x_B[p] and u_k[p] through payB
    auto reduce_rows = [&](double (&val)[RI], const double (&dcol)[RI], double &dp, double &xp, const double *&up) -> BtWin {
        const BtWin w = wave_first_min(val, std::integral_constant<int, RI>());
#pragma unroll
        for (int s = 0; s < RI; s++)
            if ((unsigned int)(tid + s * NT) == w.i) {
                payB[wv][0] = dcol[s];
                payB[wv][1] = xb_s[tid + s * NT];
#pragma unroll
                for (int j = 0; j < KR; j++) payB[wv][2 + j] = ureg[s][j];
            }
        if (lane == 0) { redMB[wv] = w.m; redIB[wv] = w.i; }
        BT_STAMP(5);    // ratios, wave-level first-min, payload
        __syncthreads();
        BT_STAMP(6);    // barrier B
        const BtWin f = block_first_min(redMB, redIB);
        const int ww = (f.i & (NT - 1)) >> 6;
        dp = payB[ww][0];
        xp = payB[ww][1];
        up = &payB[ww][2];
        BT_STAMP(7);    // block-level first-min B
        return f;
    };
    // column q of the current tableau for this thread's rows
    auto column = [&](int q, const double *vq, double (&dcol)[RI]) {
        if constexpr (STAMP) {
#pragma unroll
            for (int s = 0; s < RI; s++) {
                const int i = tid + s * NT;
                const unsigned int ic = (unsigned int)(i < a.m ? i : a.m - 1);
                dcol[s] = ldT(tile_off(ic, (unsigned int)q, ldt));
            }
            drain();
            BT_STAMP(3);   // column q of the stale tableau: issue -> data back
#pragma unroll
            for (int s = 0; s < RI; s++) {
                const int i = tid + s * NT;
                double d = dcol[s];
#pragma unroll
                for (int j = 0; j < KR; j++) d = __builtin_fma(ureg[s][j], vq[j], d);
                dcol[s] = i < a.m ? d : 0.0;
            }
            BT_STAMP(4);   // block corrections of the column
            return;
        }
        double v0q = 0;
        if constexpr (VIRT) v0q = v0_s[q];
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            const unsigned int ic = (unsigned int)(i < a.m ? i : a.m - 1);
            double d;
            if constexpr (VIRT) d = b_virt_entry<false>(*vl, (int)ic, q, a.nn, i < a.m ? u0r[s] : u0_s[ic], v0q);
            else d = ldT(tile_off(ic, (unsigned int)q, ldt));   // T in 4x4 tiles (k_bt_tile)
#pragma unroll
            for (int j = 0; j < KR; j++) d = __builtin_fma(ureg[s][j], vq[j], d);
            dcol[s] = i < a.m ? d : 0.0;
        }
    };
    // ratio vector (simplex.go:321-340); rows beyond m carry +Inf.  Branch-free: the quotient is formed for every row
    // and discarded where the reference does not divide (x/0 is harmless here), so the division chains of a thread's
    // rows interleave instead of running one after the other inside exec-masked branches
    auto ratios = [&](const double (&dcol)[RI], double (&mvv)[RI]) {
        double quot[RI], dn[RI];
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            double d = -dcol[s];
            if (fabs(d) < 1e-13) d = 0;
            dn[s] = d;
            quot[s] = div_pos(xb_s[i], fabs(d));   // == xb / |d| bit for bit (kernels_common.h); discarded where d == 0
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            mvv[s] = (dn[s] >= 0 || i >= a.m) ? inf : quot[s];
        }
    };

    int nbe = 0, cur0 = 0;   // loop mode: blocks run by this launch; first U / V row of the running block
    if constexpr (VIRT) { if (vl->virt == 2) cur0 = 8; }   // the set-up pivot's term is kept beside the first block's (rows 0 .. 7): both are applied when the tableau is written out
    bool dead = false;       // loop mode: an update counter made no progress within the limit
    for (int blk = 0; blk < (LOOP ? a.nblocks : 1); blk++) {
    if constexpr (LOOP) {
        // block blk reads the tableau after blk - 1 blocks: the update of block blk - 2 must be through (nupd arrivals per block)
        if (blk >= 2) {
            if (wv == 0) {
                const int cj = (blk - 2) & 3;
                const unsigned int ub = cj == 0 ? upd_base[0] : cj == 1 ? upd_base[1] : cj == 2 ? upd_base[2] : upd_base[3];
                const bool ok = spin_counter(upd_cnt + 16 * cj * 2, ub + (unsigned int)nupd * (unsigned int)((blk - 2) / 4 + 1), 0, 4 * kBLoopSpinLimit);   // (no sleep between these polls: ~0.4 us each)
                if (lane == 0) s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ok) { dead = true; status = ST_XCHG_TIMEOUT; }
            __syncthreads();   // (s_ok is rewritten by the next block's wait)
            if (dead) break;
        }
        Tb = reinterpret_cast<const char *>(((sel0 ^ (blk > 0 ? blk - 1 : 0)) & 1) ? a.Tbuf[1] : a.Tbuf[0]);
        cur0 = (blk & 1) * KB;
        if (blk > 0) {   // entries 0..KB-1 = the previous block (now lagging), KB.. = the block before it: in the tableau by now
#pragma unroll
            for (int s = 0; s < RI; s++) {
#pragma unroll
                for (int j = KB; j < KR; j++) ureg[s][j] = 0;
            }
#pragma unroll
            for (int s = 0; s < CJ; s++) {
#pragma unroll
                for (int j = KB; j < KR; j++) vreg[s < CR ? s : 0][j] = 0;
            }
        }
        kd = 0;
    }
    for (int k = 0; k < a.kmax; k++) {
        BT_STAMP(-1);
        const bool forced = (k == 0 && blk == 0 && a.forced_q >= 0);
        int q, p;
        double rq = 0, dpv = 1.0, xbp = 0;
        const double *vq = &payA[0][1], *up = &payB[0][2];
        bool bland = false;
        double dcol[RI];
        if (!forced) {
            // ---- entering position: first index of min r (simplex.go:247)
            BtWin fq;
            {
                double rv[CJ];
#pragma unroll
                for (int s = 0; s < CJ; s++) rv[s] = r_s[tid + s * NT];
                fq = reduce_cols(rv, rq, vq);
                if (a.guard > 0 && !(k == 0 && blk == 0 && a.exact_once)) {
                    // Guard mode (BTArgs::guard).  The reference takes this decision on reduced costs recomputed from a fresh LU
                    // (simplex.go:236-248): a minimum within the guard of the stop threshold, or two columns within the guard of each
                    // other (integer data: exact ties), is decided by that solve's rounding noise — the host repeats it (ST_NEED_EXACT)
                    double x2 = inf;
#pragma unroll
                    for (int s = 0; s < CJ; s++) x2 = vmin_f64(x2, (unsigned int)(tid + s * NT) == fq.i ? inf : rv[s]);
                    x2 = wave_min_f64(x2);
                    __syncthreads();   // (everyone has read payA / redMA of the reduction above before red2 reuses the barrier slot)
                    if (lane == 0) red2[wv] = x2;
                    __syncthreads();
                    const double y2 = lane < NW ? red2[lane] : inf;
                    const double r2 = readlane_f64(row_min_f64(y2), 15);
                    const double rq0 = rq;   // (uniform)
                    // at the stop threshold itself only the drift of the updated reduced costs matters (1e-12; a Phase-I optimum has
                    // many reduced costs at zero, 1e-10 above its threshold: nothing to re-decide there); a tie matters only when
                    // the loop goes on
                    // (an infinite guard — knob exact_degenerate = 3, strict — stops in front of EVERY decision, the stop test included)
                    if (a.guard == inf || fabs(rq0 + a.tol) <= 1e-12 || (!(rq0 >= -a.tol) && r2 - rq0 <= a.guard * fmax(1.0, fabs(rq0)))) { status = ST_NEED_EXACT; break; }
                }
            }
            q = (int)fq.i;
            if (fq.i >= (unsigned int)a.nn) { q = 0; rq = __builtin_nan(""); }  // every r_j is NaN: MinIdx returns 0
            if (rq >= -a.tol) { status = ST_OPTIMAL; break; }  // simplex.go:248
            column(q, vq, dcol);
            BtWin w;
            double mv2 = inf;   // guard mode: the runner-up ratio
            {
                double mvv[RI];
                ratios(dcol, mvv);
                w = reduce_rows(mvv, dcol, dpv, xbp, up);
                if (a.guard > 0) {   // (uniform) second-smallest ratio: a tie, exact or nearly, is broken by the fresh x_B's rounding noise too
                    double x2 = inf;
#pragma unroll
                    for (int s = 0; s < RI; s++) x2 = vmin_f64(x2, (unsigned int)(tid + s * NT) == w.i ? inf : mvv[s]);
                    x2 = wave_min_f64(x2);
                    if (lane == 0) red2[wv] = x2;
                    __syncthreads();
                    const double y2 = lane < NW ? red2[lane] : inf;
                    mv2 = readlane_f64(row_min_f64(y2), 15);
                }
            }
            p = (int)w.i;
            const double mv = w.m;
            if (mv == inf || w.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }  // simplex.go:328-330
            // degenerate (or nearly), or two rows within 1e-9 of each other: decided on a fresh gonum-order x_B (DevTypes: BTArgs::guard)
            // (a winning pivot element below the guard too: the updated tableau drifts by ~1e-12 on nearly dependent rows, where the fresh
            // column has an exact zero that the reference's 1e-13 rounding removes from the test — pivoting there gave singular bases)
            if (a.guard > 0 && !(k == 0 && blk == 0 && a.exact_once) && (mv <= a.guard || mv2 - mv <= a.guard * fmax(1.0, fabs(mv)) || fabs(dpv) <= a.guard)) { status = ST_NEED_EXACT; break; }
            if (a.cguard > 0 && fabs(dpv) <= a.cguard && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // (BTArgs::cguard)
            if (mv <= 0) {
                // ---- replaceBland (simplex.go:347-383): candidates in position order with r_i <= -1e-14 after the
                // 1e-13 rounding of :252-256; the mat.Cond guard of :377 is replaced by |d| >= 1e-13
                bland = true;
                blands++;
                int cand = -1;
                bool found = false;
                for (;;) {
                    double fl[CJ];
#pragma unroll
                    for (int s = 0; s < CJ; s++) {
                        const int j = tid + s * NT;
                        double rv = r_s[j];
                        if (fabs(rv) < 1e-13) rv = 0;
                        fl[s] = (j < a.nn && j > cand && !(rv > -1e-14)) ? 0.0 : inf;
                    }
                    double rqc;
                    const BtWin fc = reduce_cols(fl, rqc, vq);
                    if (fc.m != 0.0) break;  // candidates exhausted -> ErrBland
                    cand = (int)fc.i;
                    column(cand, vq, dcol);
                    BtWin w2;
                    {
                        double mvv[RI];
                        ratios(dcol, mvv);
                        w2 = reduce_rows(mvv, dcol, dpv, xbp, up);
                    }
                    const double mv2 = w2.m;
                    if (mv2 == inf || w2.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }  // computeMove inside Bland, :356-360
                    if (fabs(mv2) > 1e-12) { q = cand; p = (int)w2.i; rq = rqc; found = true; break; }  // :362
                    double gl[RI];
                    ratios(dcol, gl);
#pragma unroll
                    for (int s = 0; s < RI; s++) {
                        const int i = tid + s * NT;
                        gl[s] = (i < a.m && !(gl[s] > 1e-12)) ? 0.0 : inf;
                    }
                    __syncthreads();   // two row reductions in a row: everyone is done with the first one's LDS slots
                    const BtWin gw = reduce_rows(gl, dcol, dpv, xbp, up);
                    if (gw.m == 0.0) { q = cand; p = (int)gw.i; rq = rqc; found = true; break; }  // :368-379
                }
                if (status == ST_UNBOUNDED) break;
                if (!found) { status = ST_BLAND_FAILED; break; }
            }
        } else {
            // set-up pivot chosen by the host: first pivot of a block, so the block terms are all zero
            q = a.forced_q; p = a.forced_p;
            rq = a.forced_nocommit ? 0.0 : r_s[q];   // a set-up pivot leaves the reduced costs alone (they are rebuilt); a pivot the host decided on fresh solves (exact_step) is a pivot like any other
            column(q, vq, dcol);   // vq -> the zeros written before the loop
            double gl[RI];
#pragma unroll
            for (int s = 0; s < RI; s++) gl[s] = (tid + s * NT == p) ? 0.0 : inf;
            reduce_rows(gl, dcol, dpv, xbp, up);
        }
        // ---- row p of the current tableau for this thread's columns, reduced costs, block terms
        // one exact reciprocal per pivot; r_q / d_p and x_B[p] / d_p as products with it (engine-own running quantities, like
        // u_i = d_i * (-1/d_p): two IEEE divisions = 26 instructions less per wave and pivot)
        const double rinv = 1.0 / dpv, nrinv = -rinv;
        const double mult = rq * rinv;
        const double theta = xbp * rinv;
        char *Vk = reinterpret_cast<char *>(a.V + (size_t)(cur0 + k) * a.ldt);
        char *Uk = reinterpret_cast<char *>(a.U + (size_t)(cur0 + k) * a.ldu);
        auto st_term = [&](char *row, unsigned int idx, double val) {   // loop mode: read by the update workgroups of the same launch
            if constexpr (LOOP) st_agent(reinterpret_cast<double *>(row + (idx << 3)), val);
            else *reinterpret_cast<double *>(row + (idx << 3)) = val;
        };
        // the row loads go out first: the u terms / x_B update below need nothing from them and run under their latency
        double vrow[CJ];
        double u0p = 0;
        if constexpr (VIRT) u0p = u0_s[p];
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            if constexpr (VIRT) vrow[s] = j < a.ldt ? b_virt_entry<true>(*vl, p, j, a.nn, u0p, v0r[s]) : 0.0;
            else vrow[s] = j < a.ldt ? ldT(tile_off((unsigned int)p, (unsigned int)j, ldt)) : 0.0;   // columns nn..ldt of T are zero
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            if (i < a.ldu) {
                const double u = (i == p) ? rinv - 1.0 : dcol[s] * nrinv;   // rows >= m: dcol = 0
                if (i < a.m) xb_s[i] = (i == p) ? theta : __builtin_fma(-theta, dcol[s], xb_s[i]);
                st_term(Uk, (unsigned int)i, u);
#pragma unroll
                for (int jj = KR - 1; jj > 0; jj--) ureg[s][jj] = ureg[s][jj - 1];
                ureg[s][0] = u;
            }
        }
        BT_STAMP(10);  // row loads issued; u terms, x_B, u store, term shift
        if constexpr (STAMP) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RI) : "memory");   // the RI u stores behind the loads may stay in flight
            BT_STAMP(8);   // what is left of the row-load latency
        }
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            if (j < a.ldt) {
                double v = vrow[s];
                if (s < CR) {
#pragma unroll
                    for (int jj = 0; jj < KR; jj++) v = __builtin_fma(up[jj], vreg[s < CR ? s : 0][jj], v);
                } else {
#pragma unroll
                    for (int jj = 0; jj < KR; jj++) v = __builtin_fma(up[jj], vl_s[((s - CR) * KR + ((vhead - jj) & (KR - 1))) * NT + tid], v);
                }
                // reduced costs (positional): r_j - (r_q/d_p) v_j ; the leaving variable takes slot q
                r_s[j] = (j == q) ? -mult : __builtin_fma(-mult, v, r_s[j]);
                const double vprime = (j == q) ? dpv + 1.0 : v;
                st_term(Vk, (unsigned int)j, vprime);
                if (s < CR) {
#pragma unroll
                    for (int jj = KR - 1; jj > 0; jj--) vreg[s < CR ? s : 0][jj] = vreg[s < CR ? s : 0][jj - 1];
                    vreg[s < CR ? s : 0][0] = vprime;
                } else {
                    vl_s[((s - CR) * KR + ((vhead + 1) & (KR - 1))) * NT + tid] = vprime;   // becomes the newest once vhead advances
                }
            }
        }
        vhead = (vhead + 1) & (KR - 1);
        BT_STAMP(9);   // row corrections, reduced costs, v' store, term shift
        // a host-chosen set-up pivot may leave the lists alone (forced_nocommit 1: the host uploads new ones) or exchange
        // them without being counted or traced as a pivot of the loop (2)
        if (tid == 0 && forced && a.forced_nocommit >= 2) {
            const int ent = nonbasic_s[q], lea = basic_s[p];
            basic_s[p] = ent; nonbasic_s[q] = lea;
        }
        if (forced && a.forced_nocommit == 3) status = ST_FORCED_DONE;   // batched schedule: this order runs once (kmax = 1)
        if (tid == 0 && !(forced && a.forced_nocommit)) {  // simplex.go:280
            const int ent = nonbasic_s[q], lea = basic_s[p];
            basic_s[p] = ent; nonbasic_s[q] = lea;
            if (a.trace && trace_len < a.trace_cap) {
                DevPivot &tr = a.trace[trace_len];
                tr.phase = a.phase; tr.bland = bland ? 1 : 0; tr.min_idx = q; tr.replace = p; tr.entering = ent; tr.leaving = lea;
            }
            trace_len += 1;
            npiv += 1;
        }
        kd = k + 1;
    }
    nbe = blk + 1;
    if constexpr (LOOP) {
        // hand the block to the update workgroups: every term store of this workgroup has landed (agent scope), then the pivot count
        // (and the end of the loop), then the arrival that releases them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(&st->kdone2[blk & 1], kd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (status != ST_RUNNING) {
                __hip_atomic_store(&st->status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&st->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(blk_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (status != ST_RUNNING) break;
    }
    }
    if constexpr (STAMP) {
        if (a.stamps && lane == 0) {
#pragma unroll
            for (int sg = 0; sg < kStampSegs; sg++) a.stamps[wv * kStampSegs + sg] += tacc[sg];
            if (wv == 0) a.stamps[16 * kStampSegs] += (unsigned long long)kd;   // pivots behind the sums
        }
    }
#undef BT_STAMP
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        if (j < a.ldt) a.r[j] = j < a.nn ? r_s[j] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = tid + s * NT;
        if (i < a.ldu) a.xb[i] = xb_s[i];
    }
    __syncthreads();
    for (int i = tid; i < a.m; i += NT) a.basic[i] = basic_s[i];
    for (int j = tid; j < a.nn; j += NT) a.nonbasic[j] = nonbasic_s[j];
    if constexpr (VIRT) {
        // A block that spent all its pivots looks at the next stop test too: a relaxation whose Phase I is over after exactly kmax pivots would
        // otherwise stay "running" until the next launch finds min r >= -tol first thing — and have its tableau written out for that one look
        // (29 % of an 8192-wide wave with 13 branch rows against the 13 % that really go on).  Only the plain stop test: anything a guard might
        // want to re-decide is left to the next launch, which starts with the same selection.
        if (status == ST_RUNNING && kd == a.kmax && kd > 0 && a.guard == 0) {
            double rv[CJ];
#pragma unroll
            for (int s = 0; s < CJ; s++) rv[s] = r_s[tid + s * NT];
            double rq1;
            const double *vq1;
            const BtWin f1 = reduce_cols(rv, rq1, vq1);
            if (f1.i < (unsigned int)a.nn && rq1 >= -a.tol) status = ST_OPTIMAL;
            __syncthreads();
        }
        // will anybody read this relaxation's tableau?  Not when its Phase I ended in this block with the artificial above the zero tolerance (infeasible,
        // or the host path) or with a wrapped error — the tests k_b_ctrl takes next: then it is never written (k_b_gather mode 3)
        int deadflag = 0;
        if (a.phase == 1 && status != ST_RUNNING) {
            if (status == ST_UNBOUNDED || status == ST_BLAND_FAILED) deadflag = 1;
            else if (status == ST_OPTIMAL) {
#pragma unroll
                for (int s = 0; s < RI; s++) {
                    const int i = tid + s * NT;
                    if (i < a.m && basic_s[i] == vl->n && fabs(xb_s[i]) > 1e-13) deadflag = 1;
                }
            }
        }
        deadflag = __syncthreads_or(deadflag);
        if (tid == 0) st->dead1 = deadflag;
    }
    if (tid == 0) {
        st->trace_len = trace_len;
        st->pivots = npiv;
        if constexpr (LOOP) {
            if (dead) {
                // the update workgroups stopped arriving: nothing after this point can be trusted to be applied — the status says so
                // (the batched schedule hands the relaxation to a worker, which starts from the root data)
                st->tsel2[a.par ^ 1] = sel0;
                st->loop_blocks = nbe;
                hand_on(nbe);
            } else {
                // the update workgroups apply every block with pivots before the launch ends: the tableau after them
                const int napplied = kd > 0 ? nbe : nbe - 1;
                st->tsel2[a.par ^ 1] = sel0 ^ (napplied & 1);
                st->loop_blocks = nbe;
                hand_on(nbe);
            }
        } else st->kdone = kd;
        st->bland_steps += blands;
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

// ---- dual-simplex block kernel (warm start, opt-in: gomilp_frontier_solve_warm; first built in round 2 against the ROOT's tableau, withdrawn in
// round 3, back in round 4 against the PARENT's) -------------------------------------------------------------------------------------------
// A child starts from its parent's OPTIMAL basis + the slack of its one new branch row (/root/reference/README.md TODO
// "initiate the simplex at solution of parent?"; simplex.go:147-161 is the hook the reference has for it): that basis is
// dual feasible (reduced costs >= 0) and primal infeasible only in the violated branch rows.  Each pivot here is the mirror
// image of the primal one — leaving row first (first index of min x_B, stop when >= -tol), then the dual ratio test over
// row p (first index of min r_j / -T[p][j] over T[p][j] < -1e-13; none: the child is infeasible), then column q — with the
// same block terms, reductions and update formulas, so the rank-K update kernel and the primal kernel continue from its
// state unchanged.  This mode does NOT follow the reference's pivot path: results agree in z and in every branching
// decision, not bit by bit.
template <int NT, int RI, int CJ, int KR, int VL>
__device__ __forceinline__ void bt_inner2_dual_body(const BTArgs &a) {
    constexpr int NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) double sh2[];
    double *xb_s = sh2;
    double *r_s = sh2 + RI * NT;
    int *basic_s = reinterpret_cast<int *>(sh2 + RI * NT + CJ * NT);
    int *nonbasic_s = basic_s + RI * NT;
    double *vl_s = reinterpret_cast<double *>(nonbasic_s + CJ * NT);
    constexpr int CR = CJ - VL;
    int vhead = 0;
    __shared__ double redMA[16], redMB[16];
    __shared__ unsigned int redIA[16], redIB[16];
    __shared__ double payA[16][KR + 2];  // per wave: r_q, T[p][q], v'_k[q]
    __shared__ double payB[16][KR + 1];  // per wave: x_B[p], u_k[p]
    DevState *st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int done = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double inf = __builtin_inf();
    const unsigned int ldt = (unsigned int)a.ldt;
    const char *Tb = reinterpret_cast<const char *>(a.T);
    auto ldT = [&](unsigned int elem) -> double { return *reinterpret_cast<const double *>(Tb + (elem << 3)); };
    if (done) {
        if (tid == 0) st->kdone = 0;
        return;
    }
    double ureg[RI][KR], vreg[CR > 0 ? CR : 1][KR];
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = tid + s * NT;
        xb_s[i] = i < a.m ? a.xb[i] : inf;   // padding never wins the leaving-row argmin
        basic_s[i] = i < a.m ? a.basic[i] : 0;
#pragma unroll
        for (int j = 0; j < KR; j++) ureg[s][j] = 0;
    }
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        r_s[j] = j < a.nn ? a.r[j] : inf;
        nonbasic_s[j] = j < a.nn ? a.nonbasic[j] : 0;
#pragma unroll
        for (int j2 = 0; j2 < KR; j2++) {
            if (s < CR) vreg[s < CR ? s : 0][j2] = 0;
            else vl_s[((s - CR) * KR + j2) * NT + tid] = 0;
        }
    }
    __syncthreads();
    int kd = 0, status = ST_RUNNING;
    long long npiv = 0;
    if (tid == 0) npiv = st->pivots;

    auto wave_first_min = [&](auto &val, auto nslots) -> BtWin {
        constexpr int N = decltype(nslots)::value;
        double x = val[0];
#pragma unroll
        for (int s = 1; s < N; s++) x = vmin_f64(x, val[s]);
        BtWin w;
        w.m = wave_min_f64(x);
        w.i = 0xFFFFFFFFu;
#pragma unroll
        for (int s = N - 1; s >= 0; s--) {
            const unsigned long long mask = __ballot(val[s] == w.m);
            if (mask) w.i = (unsigned int)(s * NT + wbase + __builtin_ctzll(mask));
        }
        return w;
    };
    auto block_first_min = [&](const double *redM, const unsigned int *redI) -> BtWin {
        const double x = lane < NW ? redM[lane] : inf;
        const unsigned int ii = lane < NW ? redI[lane] : 0xFFFFFFFFu;
        BtWin f;
        f.m = readlane_f64(row_min_f64(x), 15);
        const unsigned int key = (x == f.m) ? ii : 0xFFFFFFFFu;
        f.i = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(key), 15);
        return f;
    };

    for (int k = 0; k < a.kmax; k++) {
        // ---- leaving row: first index of min x_B; stop when the basis is primal feasible
        BtWin fp;
        double xbp = 0;
        const double *up = &payB[0][1];
        {
            double xv[RI];
#pragma unroll
            for (int s = 0; s < RI; s++) xv[s] = xb_s[tid + s * NT];
            const BtWin w = wave_first_min(xv, std::integral_constant<int, RI>());
#pragma unroll
            for (int s = 0; s < RI; s++)
                if ((unsigned int)(tid + s * NT) == w.i) {
                    payB[wv][0] = xv[s];
#pragma unroll
                    for (int j = 0; j < KR; j++) payB[wv][1 + j] = ureg[s][j];
                }
            if (lane == 0) { redMB[wv] = w.m; redIB[wv] = w.i; }
            __syncthreads();
            fp = block_first_min(redMB, redIB);
            const int ww = (fp.i & (NT - 1)) >> 6;
            xbp = payB[ww][0];
            up = &payB[ww][1];
        }
        if (!(fp.m < -a.tol) || fp.i >= (unsigned int)a.m) { status = ST_OPTIMAL; break; }
        const int p = (int)fp.i;
        // ---- row p of the current tableau, dual ratio test
        double vrow[CJ];
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            double v = j < a.ldt ? ldT(tile_off((unsigned int)p, (unsigned int)j, ldt)) : 0.0;
            if (s < CR) {
#pragma unroll
                for (int jj = 0; jj < KR; jj++) v = __builtin_fma(up[jj], vreg[s < CR ? s : 0][jj], v);
            } else {
#pragma unroll
                for (int jj = 0; jj < KR; jj++) v = __builtin_fma(up[jj], vl_s[((s - CR) * KR + ((vhead - jj) & (KR - 1))) * NT + tid], v);
            }
            vrow[s] = v;
        }
        BtWin fq;
        double rq = 0, dpv = 1.0;
        const double *vq = &payA[0][2];
        {
            double ratio[CJ];
#pragma unroll
            for (int s = 0; s < CJ; s++) {
                const int j = tid + s * NT;
                const double v = vrow[s];
                ratio[s] = (j < a.nn && v < -1e-13) ? r_s[j] / (-v) : inf;
            }
            const BtWin w = wave_first_min(ratio, std::integral_constant<int, CJ>());
#pragma unroll
            for (int s = 0; s < CJ; s++)
                if ((unsigned int)(tid + s * NT) == w.i) {
                    payA[wv][0] = r_s[tid + s * NT];
                    payA[wv][1] = vrow[s];
#pragma unroll
                    for (int j = 0; j < KR; j++)
                        payA[wv][2 + j] = s < CR ? vreg[s < CR ? s : 0][j] : vl_s[((s - CR) * KR + ((vhead - j) & (KR - 1))) * NT + tid];
                }
            if (lane == 0) { redMA[wv] = w.m; redIA[wv] = w.i; }
            __syncthreads();
            fq = block_first_min(redMA, redIA);
            const int ww = (fq.i & (NT - 1)) >> 6;
            rq = payA[ww][0];
            dpv = payA[ww][1];
            vq = &payA[ww][2];
        }
        if (fq.m == inf || fq.i >= (unsigned int)a.nn) { status = ST_DUAL_INFEASIBLE; break; }   // no entry of row p can restore x_B[p] >= 0
        const int q = (int)fq.i;
        // ---- column q: loads first, the column-side updates run under their latency
        double dcol[RI];
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            const unsigned int ic = (unsigned int)(i < a.m ? i : a.m - 1);
            dcol[s] = ldT(tile_off(ic, (unsigned int)q, ldt));
        }
        const double mult = rq / dpv;
        const double theta = xbp / dpv;
        const double rinv = 1.0 / dpv, nrinv = -rinv;
        char *Vk = reinterpret_cast<char *>(a.V + (size_t)k * a.ldt);
        char *Uk = reinterpret_cast<char *>(a.U + (size_t)k * a.ldu);
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            if (j < a.ldt) {
                const double v = vrow[s];
                if (j < a.nn) r_s[j] = (j == q) ? -mult : __builtin_fma(-mult, v, r_s[j]);
                const double vprime = (j == q) ? dpv + 1.0 : v;
                *reinterpret_cast<double *>(Vk + ((unsigned int)j << 3)) = vprime;
                if (s < CR) {
#pragma unroll
                    for (int jj = KR - 1; jj > 0; jj--) vreg[s < CR ? s : 0][jj] = vreg[s < CR ? s : 0][jj - 1];
                    vreg[s < CR ? s : 0][0] = vprime;
                } else {
                    vl_s[((s - CR) * KR + ((vhead + 1) & (KR - 1))) * NT + tid] = vprime;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = tid + s * NT;
            double d = dcol[s];
#pragma unroll
            for (int j = 0; j < KR; j++) d = __builtin_fma(ureg[s][j], vq[j], d);   // vq: the v' terms of column q BEFORE this pivot
            d = i < a.m ? d : 0.0;
            if (i < a.ldu) {
                const double u = (i == p) ? rinv - 1.0 : d * nrinv;
                if (i < a.m) xb_s[i] = (i == p) ? theta : __builtin_fma(-theta, d, xb_s[i]);
                *reinterpret_cast<double *>(Uk + ((unsigned int)i << 3)) = u;
#pragma unroll
                for (int jj = KR - 1; jj > 0; jj--) ureg[s][jj] = ureg[s][jj - 1];
                ureg[s][0] = u;
            }
        }
        vhead = (vhead + 1) & (KR - 1);
        if (tid == 0) {
            const int ent = nonbasic_s[q], lea = basic_s[p];
            basic_s[p] = ent; nonbasic_s[q] = lea;
            npiv += 1;
        }
        kd = k + 1;   // (two barriers per pivot are enough: every LDS slot written before barrier X of pivot k+1 was last read before barrier X' of pivot k that all waves passed)
    }
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        if (j < a.ldt) a.r[j] = j < a.nn ? r_s[j] : 0.0;
    }
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = tid + s * NT;
        if (i < a.ldu) a.xb[i] = i < a.m ? xb_s[i] : 0.0;
    }
    __syncthreads();
    for (int i = tid; i < a.m; i += NT) a.basic[i] = basic_s[i];
    for (int j = tid; j < a.nn; j += NT) a.nonbasic[j] = nonbasic_s[j];
    if (tid == 0) {
        st->pivots = npiv;
        st->kdone = kd;
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

template <int NT, int RI, int CJ, int KR, int VL>
__global__ __launch_bounds__(NT) void k_bt_inner2_dual_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.x >= *count) return;
    const BatchLP &lp = lps[ids[blockIdx.x]];
    if (lp.stage != BS_DUAL) return;
    const BTArgs a = lp.bt;
    bt_inner2_dual_body<NT, RI, CJ, KR, VL>(a);
}

template <int NT, int RI, int CJ, int KR, int VL, bool STAMP = false>
__global__ __launch_bounds__(NT) void k_bt_inner2(BTArgs a) {
    bt_inner2_body<NT, RI, CJ, KR, VL, STAMP>(a);
}
// one workgroup per relaxation of a wave (device-batched frontier): the argument block comes from HBM, where the
// control kernel rewrites it at phase changes; relaxations in a terminal stage leave at once
template <int NT, int RI, int CJ, int KR, int VL>
__global__ __launch_bounds__(NT) void k_bt_inner2_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.x >= *count) return;   // the grid is sized from an older (larger) count of active relaxations
    const BatchLP &lp = lps[ids[blockIdx.x]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_DUAL || stage == BS_COLD) return;   // (BS_DUAL: the dual kernel's)
    const BTArgs a = lp.bt;
    bt_inner2_body<NT, RI, CJ, KR, VL, false>(a);
}

// the same while the wave's tableaus are virtual (BatchLP::virt > 0: engine_batch.cpp)
template <int NT, int RI, int CJ, int KR, int VL>
__global__ __launch_bounds__(NT) void k_bt_inner2_virt_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.x >= *count) return;
    const BatchLP &lp = lps[ids[blockIdx.x]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_DUAL || stage == BS_COLD || lp.virt <= 0) return;
    const BTArgs a = lp.bt;
    bt_inner2_body<NT, RI, CJ, KR, VL, false, false, true>(a, 0, &lp);
}

// ---- batched persistent loop kernel (round 4) -----------------------------------------------------------------------------------------
// The launch pairs above put the rank-8 update of EVERY block on the chain of a relaxation (26 us of pivots, then 7 us of update, then a
// launch boundary), and a wave's long chains wait at every block step for whatever else the step carries.  Here the relaxation at
// position s of the active list gets ONE pivot workgroup (bt_inner2_body in loop mode: 8 current + 8 lagging terms in registers, the
// tableau read from the buffer the update finished two blocks ago) and NU update workgroups (bt_loop.h: matrix cores, ping-pong between
// the relaxation's two tableau buffers) for up to `nblocks` blocks of one launch — the protocol of k_bt_loop with one pivot workgroup,
// counters and buffer choice per relaxation (its own exchange buffer and DevState).  Placement: block b = x + 8 j runs on XCD x (blocks
// are dealt round-robin over the XCDs); position s = (j / (1 + NU)) * 8 + x, role j % (1 + NU): all workgroups of a relaxation share one
// L2.  Every workgroup of the launch must be resident (they wait for each other): the host launches at most one workgroup per CU.
// Stages with a host-chosen or no pivot (BS_FORCED, BS_P2_START, BS_EXCH: kmax < 8) run one block.
template <int NT, int RI, int KB, int NU>
__global__ __launch_bounds__(NT) void k_b_loop(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, int nblocks, int par) {
    const unsigned int x8 = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const int slot = (int)((j / (1u + NU)) * 8u + x8), role = (int)(j % (1u + NU));
    if (slot >= *count) return;
    const BatchLP &lp = lps[ids[slot]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_COLD) return;
    BTArgs a = lp.bt;
    a.Tbuf[0] = lp.T[0]; a.Tbuf[1] = lp.T[1];
    a.loop = 1; a.par = par;
    // (the host counts blocks of 8 pivots: a launch with shorter blocks runs more of them)
    if (a.kmax >= KB) { a.kmax = KB; a.nblocks = nblocks * (8 / KB); }
    else a.nblocks = 1;
#ifdef GOMILP_DEBUG
    if (a.fault && role == 1) return;   // test hook (diagnostic flavour only): an update workgroup that never takes part -> the pivot workgroup gives up (ST_XCHG_TIMEOUT)
#endif
    if (role == 0) bt_inner2_body<NT, RI, RI, 2 * KB, 0, false, true>(a, NU);
    else bt_loop_update_role<NT, KB, true>(a, role - 1, NU, 1);
}

// T[i, j] += sum_{k < kdone} U[k][i] * V[k][j]  — in place, one streaming pass.
// Workgroup = 4 waves x 128 columns (one double2 per lane) over `rows_per_wg` rows; each lane keeps its V column
// pair for all k in registers, the u scalars of the row block sit in LDS.
template <int KMAX>
__global__ __launch_bounds__(kBlock) void k_bt_update(BTArgs a, int rows_per_wg) {
    __shared__ double us[KMAX][64];
    const int kd = a.st->kdone;
    if (kd <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c2 = (blockIdx.x * kWavesPerBlock + wv) * 64 + lane;  // double2 column index
    const int ld2 = a.ldt >> 1;
    const int i0 = blockIdx.y * rows_per_wg;
    const int nrows = min(rows_per_wg, a.m - i0);
    for (int idx = threadIdx.x; idx < KMAX * 64; idx += kBlock) {
        const int k = idx / 64, rr = idx % 64;
        us[k][rr] = (k < kd && rr < nrows) ? a.U[(size_t)k * a.ldu + i0 + rr] : 0.0;
    }
    double2 vv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        vv[k].x = 0; vv[k].y = 0;
        if (k < kd && c2 < ld2) vv[k] = reinterpret_cast<const double2 *>(a.V + (size_t)k * a.ldt)[c2];
    }
    __syncthreads();
    if (c2 >= ld2) return;
    // UNR rows per trip: independent 16-byte loads in flight per lane before the first use (bytes in flight per CU set
    // the speed of this kernel, see k_bt_update_tiled)
    constexpr int UNR = KMAX <= 8 ? 8 : 4;
    int rr = 0;
    for (; rr + UNR <= nrows; rr += UNR) {
        double2 *cell = reinterpret_cast<double2 *>(a.T + (size_t)(i0 + rr) * a.ldt) + c2;
        double2 t[UNR];
#pragma unroll
        for (int x = 0; x < UNR; x++) t[x] = cell[(size_t)x * ld2];
#pragma unroll
        for (int x = 0; x < UNR; x++) {
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const double u = us[k][rr + x];
                t[x].x += u * vv[k].x;
                t[x].y += u * vv[k].y;
            }
        }
#pragma unroll
        for (int x = 0; x < UNR; x++) cell[(size_t)x * ld2] = t[x];
    }
    for (; rr < nrows; rr++) {
        double2 *cell = reinterpret_cast<double2 *>(a.T + (size_t)(i0 + rr) * a.ldt) + c2;
        double2 t = *cell;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const double u = us[k][rr];
            t.x += u * vv[k].x;
            t.y += u * vv[k].y;
        }
        *cell = t;
    }
}


// ---- 4x4-tile layout of T for the register-resident inner kernel ---------------------------------------------------
// One workgroup reads a whole column (ratio test) and a whole row (reduced costs) of T per pivot.  Row-major T makes
// the column a gather of m separate 128-byte lines (8 useful bytes each) through ONE CU's L1 — 256 KB of fill traffic
// per pivot at m = 2048, the largest item of the pivot's critical path.  In 4x4 tiles (one tile = one 128-byte line,
// tile (I, J) at ((I * ldt/4) + J) * 16, element (i&3)*4 + (j&3)) a column and a row both touch m/4 resp. (n-m)/4
// lines: 4x less fill for the column, 4x more for the (cheap) row, half in total.  The tableau set-up kernels
// (tableau_kernels.hip: tab_idx) read and write either layout, so the blocked pipeline is tiled from the start;
// k_bt_tile converts only when a caller mixes pipelines.
__device__ __forceinline__ unsigned int tile_off(unsigned int i, unsigned int j, unsigned int ldt) {
    return ((i >> 2) * (ldt >> 2) + (j >> 2)) * 16u + ((i & 3u) << 2) + (j & 3u);
}

// one thread = one 32-byte piece (4 consecutive columns of one row); threads follow the tiled side's memory order
__global__ __launch_bounds__(256) void k_bt_tile(const double *__restrict__ src, double *__restrict__ dst, int m, int ldt, int to_tiles) {
    const unsigned int c = blockIdx.x * 256u + threadIdx.x;        // piece index on the tiled side
    const unsigned int per_tilerow = (unsigned int)ldt;             // (ldt/4 tiles) * 4 pieces
    const unsigned int I = c / per_tilerow, rem = c % per_tilerow;
    const unsigned int J = rem >> 2, r = rem & 3u;
    const unsigned int i = I * 4u + r, j = J * 4u;
    const unsigned int m4 = ((unsigned int)m + 3u) & ~3u;
    if (i >= m4) return;
    const size_t toff = (size_t)c * 4u, roff = (size_t)i * (unsigned int)ldt + j;
    if (to_tiles) {
        double2 lo = make_double2(0, 0), hi = make_double2(0, 0);
        if (i < (unsigned int)m) { lo = *reinterpret_cast<const double2 *>(src + roff); hi = *reinterpret_cast<const double2 *>(src + roff + 2); }
        *reinterpret_cast<double2 *>(dst + toff) = lo;
        *reinterpret_cast<double2 *>(dst + toff + 2) = hi;
    } else if (i < (unsigned int)m) {
        *reinterpret_cast<double2 *>(dst + roff) = *reinterpret_cast<const double2 *>(src + toff);
        *reinterpret_cast<double2 *>(dst + roff + 2) = *reinterpret_cast<const double2 *>(src + toff + 2);
    }
}

// T += sum_k u_k v_k'^T on the tiled layout.  Thread = one 16-byte half piece (tile column J, row-in-tile r, half h),
// looping over `tilerows_per_wg` tile rows: consecutive lanes touch consecutive 16 bytes (the access shape of
// k_bt_update); v_k'[4J+2h], v_k'[4J+2h+1] in registers, u in LDS.
template <int KMAX>
__device__ __forceinline__ void bt_update_tiled_body(const BTArgs &a, int tilerows_per_wg, unsigned int bx, unsigned int by) {
    __shared__ double us[KMAX][64];
    const int kd = a.st->kdone;   // tested below, after the loads that do not depend on it are in flight
    const unsigned int cx = bx * kBlock + threadIdx.x;   // half piece within a tile row: 0 .. 2*ldt-1
    const unsigned int J = cx >> 3, r = (cx >> 1) & 3u, h = cx & 1u;
    const int I0 = by * tilerows_per_wg;
    const int m4 = (a.m + 3) & ~3;
    const int nI = min(tilerows_per_wg, m4 / 4 - I0);
    for (int idx = threadIdx.x; idx < KMAX * 64; idx += kBlock) {
        const int k = idx / 64, rr = idx % 64;
        const int row = I0 * 4 + rr;
        const double uval = (k < a.kmax && rr < nI * 4 && row < a.m) ? a.U[(size_t)k * a.ldu + row] : 0.0;
        us[k][rr] = k < kd ? uval : 0.0;   // rows of U beyond the pivots of this block are stale
    }
    const bool inb = cx < 2u * (unsigned int)a.ldt;
    double2 vv[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        vv[k] = make_double2(0, 0);
        if (k < a.kmax && inb) vv[k] = *reinterpret_cast<const double2 *>(a.V + (size_t)k * a.ldt + J * 4u + h * 2u);
    }
    if (kd <= 0) return;
#pragma unroll
    for (int k = 0; k < KMAX; k++)
        if (k >= kd) vv[k] = make_double2(0, 0);   // rows of V beyond the pivots of this block are stale
    __syncthreads();
    if (!inb) return;
    double2 *cell = reinterpret_cast<double2 *>(a.T) + (size_t)I0 * (unsigned int)a.ldt * 2u + cx;
    const size_t step = (size_t)a.ldt * 2u;   // double2 per tile row
    // UNR tile rows per trip: UNR independent 16-byte loads in flight per lane before the first use (the kernel runs out
    // of the Infinity Cache at m = 2048: bytes in flight per CU, not issue rate, set its speed)
    constexpr int UNR = 8;
    int ii = 0;
    for (; ii + UNR <= nI; ii += UNR, cell += UNR * step) {
        double2 t[UNR];
#pragma unroll
        for (int x = 0; x < UNR; x++) t[x] = cell[x * step];
#pragma unroll
        for (int x = 0; x < UNR; x++) {
#pragma unroll
            for (int k = 0; k < KMAX; k++) {
                const double u = us[k][(ii + x) * 4 + r];
                t[x].x += u * vv[k].x;
                t[x].y += u * vv[k].y;
            }
        }
#pragma unroll
        for (int x = 0; x < UNR; x++) cell[x * step] = t[x];
    }
    for (; ii < nI; ii++, cell += step) {
        double2 t = *cell;
#pragma unroll
        for (int k = 0; k < KMAX; k++) {
            const double u = us[k][ii * 4 + r];
            t.x += u * vv[k].x;
            t.y += u * vv[k].y;
        }
        *cell = t;
    }
}

template <int KMAX>
__global__ __launch_bounds__(kBlock) void k_bt_update_tiled(BTArgs a, int tilerows_per_wg) {
    bt_update_tiled_body<KMAX>(a, tilerows_per_wg, blockIdx.x, blockIdx.y);
}
// Batched form: 1-D grid, block L works on the relaxation at position L % nlp_pad of the active list, tile L / nlp_pad.
// nlp_pad is a multiple of 8 and blocks are dealt round-robin over the 8 XCDs, so every block of the relaxation at list
// position b — and block b of the batched inner kernel — runs on the XCD b % 8: the relaxation's tableau (2 MB at
// 520 x 512) stays in ONE XCD's L2 between the update and the column / row reads of the next block (speed only: nothing
// depends on the placement; the list is rebuilt once per superstep).
template <int KMAX>
__global__ __launch_bounds__(kBlock) void k_bt_update_tiled_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count,
                                                                  int nlp_pad, int gx, int ntiles, int tilerows_per_wg, int xcd_local) {
    const unsigned int L = blockIdx.x;
    // xcd_local (small tableaus, many relaxations): relaxation-minor order keeps one relaxation on one XCD (see above);
    // otherwise tile-minor: the tiles of a large tableau spread over all 8 XCDs (one XCD's 32 CUs would stream it 8x slower)
    const unsigned int li = xcd_local ? L % (unsigned int)nlp_pad : L / (unsigned int)ntiles;
    const unsigned int tile = xcd_local ? L / (unsigned int)nlp_pad : L % (unsigned int)ntiles;
    if ((int)li >= *count) return;
    const BatchLP &lp = lps[ids[li]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_COLD) return;
    if (stage == BS_P1 && lp.st->done) {
        // Phase I ended inside this block step.  With the artificial still basic above the zero tolerance the relaxation is infeasible (or
        // goes to the host path, which starts from the root data), and a wrapped error ends it too: k_b_ctrl, next in the stream, decides
        // exactly so, and nobody reads this tableau again — on a B&B frontier that is most of a wide wave, each a 2.4 MB tableau streamed
        // for nothing (2048 children: 2.4 of 11.5 ms).  The tests of k_b_ctrl, on the same lists.
        const int status = lp.st->status;
        if (status == ST_UNBOUNDED || status == ST_BLAND_FAILED) return;
        if (status == ST_OPTIMAL) {
            __shared__ int s_skip;
            if (threadIdx.x == 0) s_skip = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < lp.m; i += kBlock)
                if (lp.basic[i] == lp.n && fabs(lp.xb[i]) > 1e-13) s_skip = 1;
            __syncthreads();
            if (s_skip) return;
        }
    }
    const BTArgs a = lp.bt;
    const unsigned int bx = tile % (unsigned int)gx, by = tile / (unsigned int)gx;
    if (bx * kBlock >= 2u * (unsigned int)a.ldt || (int)by * tilerows_per_wg * 4 >= ((a.m + 3) & ~3)) return;
    bt_update_tiled_body<KMAX>(a, tilerows_per_wg, bx, by);
}

// Rank-16 update on the matrix cores.  With K = 16 the update does 2 flop per byte moved and the VALU form above runs at
// 4.8 TB/s (13.9 us for the 67 MB of a 2048 x 2048 tableau; the rank-8 form streams at 6.8 TB/s): the 16 LDS reads and 32
// multiply-adds per 16 bytes, not the memory system, set its pace.  A 16 x 16 block of T in the 4x4-tile layout is exactly
// the C/D operand of v_mfma_f64_16x16x4_f64 (lane l, register r <-> row (l >> 4) + 4 r, column l & 15: tile row r, row
// l >> 4 of the tile, tile column (l & 15) >> 2, column l & 3 of the tile), so for each r the 64 lanes read the 512
// contiguous bytes of four neighbouring tiles; A = u_k[row] (lane: row l & 15, k = 4 s + (l >> 4)), B = v'_k[column], four
// MFMAs (k = 0..15) per block.  One wave = a strip of 16 rows x `cw` column blocks, the u operands loaded once.
typedef double bt_d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bt_update_mfma16_body(const BTArgs &a, int cw, int bx, int strip) {
    const int kd = a.st->kdone;
    if (kd <= 0) return;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ncb = a.ldt >> 4;
    const int cb0 = (bx * 4 + wv) * cw;
    if (cb0 >= ncb) return;
    const int cb1 = min(cb0 + cw, ncb);
    const int row = strip * 16 + l15;
    double av[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const int k = 4 * s + l4;
        av[s] = (k < kd && row < a.m) ? a.U[(size_t)k * a.ldu + row] : 0.0;   // rows of U beyond the pivots of this block are stale
    }
    const int ntr = (a.m + 3) >> 2;   // tile rows that exist
    const size_t trow = (size_t)(a.ldt >> 2) * 16;   // doubles per tile row
    double *base = a.T + (size_t)(strip * 4) * trow + (size_t)(l15 >> 2) * 16 + l4 * 4 + (l15 & 3);
    bool valid[4];
#pragma unroll
    for (int r = 0; r < 4; r++) valid[r] = strip * 4 + r < ntr;
    constexpr int UN = 4;   // column blocks in flight per wave: 16 loads of 8 bytes per lane before the first MFMA
    for (int cb = cb0; cb < cb1; cb += UN) {
        bt_d4 c[UN];
        double bv[UN][4];
#pragma unroll
        for (int x = 0; x < UN; x++) {
            const bool in = cb + x < cb1;
#pragma unroll
            for (int r = 0; r < 4; r++) c[x][r] = (in && valid[r]) ? base[(size_t)r * trow + (size_t)(cb + x) * 64] : 0.0;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const int k = 4 * s + l4;
                bv[x][s] = (in && k < kd) ? a.V[(size_t)k * a.ldt + (cb + x) * 16 + l15] : 0.0;
            }
        }
#pragma unroll
        for (int x = 0; x < UN; x++) {
#pragma unroll
            for (int s = 0; s < 4; s++) c[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[x][s], c[x], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < UN; x++) {
            if (cb + x < cb1) {
#pragma unroll
                for (int r = 0; r < 4; r++)
                    if (valid[r]) base[(size_t)r * trow + (size_t)(cb + x) * 64] = c[x][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_bt_update_mfma16(BTArgs a, int cw) { bt_update_mfma16_body(a, cw, (int)blockIdx.x, (int)blockIdx.y); }
// batched form: block L works on the relaxation at list position L / ntiles, tile L % ntiles (tile-minor: the tiles of a large
// tableau spread over all XCDs)
__global__ __launch_bounds__(256) void k_bt_update_mfma16_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, int gx, int ntiles, int cw) {
    const unsigned int li = blockIdx.x / (unsigned int)ntiles, tile = blockIdx.x % (unsigned int)ntiles;
    if ((int)li >= *count) return;
    const BatchLP &lp = lps[ids[li]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST) return;
    const BTArgs a = lp.bt;
    const int bx = (int)(tile % (unsigned int)gx), strip = (int)(tile / (unsigned int)gx);
    if (strip * 16 >= ((a.m + 3) & ~3)) return;
    bt_update_mfma16_body(a, cw, bx, strip);
}

// ---- launch wrappers ---------------------------------------------------------------------------

int bt_max_k() { return kBtMaxK; }

// Thread count: the per-pivot reductions are instruction-issue bound and every wave repeats them, so the kernel runs
// with ONE wave per SIMD (256 threads) whenever the rows/columns per thread still fit in registers.
// Register-resident block terms cost (RI + CJ) * KREG doubles per thread.
struct BtCfg { int nt, ri, cj, kreg; };
static BtCfg bt_cfg(int m, int ldt, int force) {   // force: context knob "bt_nt" (0 = by shape)
    auto per = [](int x, int nt) { return (x + nt - 1) / nt; };
    for (int nt : {256, 512, 1024}) {
        if (force && nt != force) continue;
        const int r = std::max(per(m, nt), per(ldt, nt));
        if (!force) {
            // measured on gfx950 (round 2, k_bt_inner2): 512 threads with up to 4 rows + 4 columns per thread hold every block term
            // in registers without spilling (244 VGPRs of the 256 a thread of a 512-thread workgroup may use) — 37.7 us per 8
            // pivots at m = 2048; 1024 threads x 2 need an LDS ring for one column slot and still spill 10 VGPRs: 40.7 us
            if (nt == 256) continue;
            if (nt == 512 && r > 4) continue;
        }
        if (r <= 2) return {nt, 2, 2, 8};
        if (r <= 4) return {nt, 4, 4, nt <= 512 ? 8 : 0};
        if (r <= 8) return {nt, 8, 8, nt <= 256 ? 8 : 0};
    }
    return {1024, 8, 8, 0};
}
bool bt_supported(int m, int nn) {  // r, x_B (doubles) and the two index lists (ints) live in LDS: 12 bytes per row + column
    const long ldt = ((nn + 511) / 512) * 512, ldu = (m + 1) & ~1;
    return m <= 8 * 1024 && ldt <= 8 * 1024 && (ldt + ldu) * 12 <= 140 * 1024;
}
int bt_reg_k(int m, int ldt, int nt_force) { return bt_cfg(m, ldt, nt_force).kreg; }
template <int NT>
static void bt_launch_nt(const BTArgs &a, const BtCfg &c, bool reg, size_t lds, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
#define GOMILP_BT_LAUNCH(RI, CJ, KR) hipExtLaunchKernelGGL((k_bt_inner<NT, RI, CJ, KR>), dim3(1), dim3(NT), lds, s, e0, e1, 0, a)
    if (a.tiled) {   // register-resident kernel on the tiled layout (the engine converted T: bt_tiled())
        constexpr int VL2 = NT >= 1024 ? 1 : 0;   // terms of the last column slot in LDS where the register budget is 128
        const size_t lds2 = (size_t)(c.ri + c.cj) * NT * (sizeof(double) + sizeof(int));
        if (c.ri == 2) {
            const size_t lds = lds2 + (size_t)VL2 * 8 * NT * sizeof(double);
            if (lds > 64 * 1024) {
                lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner2<NT, 2, 2, 8, VL2>), 140 * 1024);
            }
            if (a.stamps) {
                if constexpr (NT >= 512) {   // diagnostic build: the two instances the headline sizes run
                    lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner2<NT, 2, 2, 8, VL2, true>), 140 * 1024);
                    hipExtLaunchKernelGGL((k_bt_inner2<NT, 2, 2, 8, VL2, true>), dim3(1), dim3(NT), lds, s, e0, e1, 0, a);
                    return;
                }
            }
            hipExtLaunchKernelGGL((k_bt_inner2<NT, 2, 2, 8, VL2>), dim3(1), dim3(NT), lds, s, e0, e1, 0, a);
            return;
        }
        if constexpr (NT <= 512) {
            if (c.ri == 4) { hipExtLaunchKernelGGL((k_bt_inner2<NT, 4, 4, 8, 0>), dim3(1), dim3(NT), lds2, s, e0, e1, 0, a); return; }
        }
        return;   // unreachable: bt_tiled() admits exactly the two shapes above
    }
    if (c.ri == 2) { if (reg) GOMILP_BT_LAUNCH(2, 2, 8); else GOMILP_BT_LAUNCH(2, 2, 0); }
    else if (c.ri == 4) { if (reg && NT <= 512) GOMILP_BT_LAUNCH(4, 4, (NT <= 512 ? 8 : 0)); else GOMILP_BT_LAUNCH(4, 4, 0); }
    else { if (reg && NT <= 256) GOMILP_BT_LAUNCH(8, 8, (NT <= 256 ? 8 : 0)); else GOMILP_BT_LAUNCH(8, 8, 0); }
#undef GOMILP_BT_LAUNCH
}
void launch_bt_inner_groups(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1);   // btg_kernels.hip
void launch_bt_inner(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (a.groups > 0 && a.tiled) { launch_bt_inner_groups(a, s, e0, e1); return; }
    const size_t lds = (size_t)(a.ldt + a.ldu) * sizeof(double) + (size_t)(a.ldu + a.ldt) * sizeof(int);
    if (lds > 64 * 1024) {
        lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner<1024, 4, 4, 0>), 140 * 1024);
        lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner<1024, 8, 8, 0>), 140 * 1024);
    }
    const BtCfg c = bt_cfg(a.m, a.ldt, a.nt_force);
    const bool reg = c.kreg > 0 && a.kmax <= c.kreg;
    if (c.nt == 256) bt_launch_nt<256>(a, c, reg, lds, s, e0, e1);
    else if (c.nt == 512) bt_launch_nt<512>(a, c, reg, lds, s, e0, e1);
    else bt_launch_nt<1024>(a, c, reg, lds, s, e0, e1);
}
// true when launch_bt_inner picks the register-resident kernel, which works on the tiled layout of T
bool bt_tiled(int m, int ldt, int kmax, int nt_force, bool old_only) {
    const BtCfg c = bt_cfg(m, ldt, nt_force);
    if (old_only || !(c.kreg > 0 && kmax <= c.kreg)) return false;
    return c.ri == 2 || (c.ri == 4 && c.nt <= 512);
}
// ---- batched launches (device-batched frontier, engine_batch.cpp): one configuration for the whole wave, chosen from
// the largest relaxation; the register-resident kernel on the tiled layout only
BtGroupCfg bt_group_cfg(int m, int ldt, int knob);   // btg_kernels.hip
void launch_bt_inner_groups_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, const BtGroupCfg &c, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int xcd_off);
// pivots per block of the batched schedule for a wave of this shape class: 16 where the 8-workgroup kernel runs, else 8
int bt_batch_k(int m_max, int ldt_max) {
    const BtGroupCfg g = bt_group_cfg(m_max, ldt_max, 0);
    return (g.groups == 8 && g.ri == 1) ? 16 : 8;
}
bool bt_batch_supported(int m_max, int ldt_max) {
    if (bt_batch_k(m_max, ldt_max) == 16) return true;
    if (!bt_tiled(m_max, ldt_max, 8, 0, false)) return false;
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    return c.ri == 2 || (c.ri == 4 && c.nt == 512);
}
template <int NT, int RI, int VL>
static void bt_inner_batch_nt(const BatchLP *lps, const int *ids, const int *count, int nlp, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = (size_t)(RI + RI) * NT * (sizeof(double) + sizeof(int)) + (size_t)VL * 8 * NT * sizeof(double);
    if (lds > 64 * 1024) lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner2_batch<NT, RI, RI, 8, VL>), 140 * 1024);
    hipExtLaunchKernelGGL((k_bt_inner2_batch<NT, RI, RI, 8, VL>), dim3(nlp), dim3(NT), lds, s, e0, e1, 0, lps, ids, count);
}
// ids / count: the active list of the previous control step (device); nlp: an upper bound of *count the host knows
void launch_bt_inner_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int xcd_off) {
    if (bt_batch_k(m_max, ldt_max) == 16) { launch_bt_inner_groups_batch(lps, ids, count, nlp, bt_group_cfg(m_max, ldt_max, 0), s, e0, e1, xcd_off); return; }
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    if (c.ri == 2) { if (c.nt == 512) bt_inner_batch_nt<512, 2, 0>(lps, ids, count, nlp, s, e0, e1); else bt_inner_batch_nt<1024, 2, 1>(lps, ids, count, nlp, s, e0, e1); }
    else bt_inner_batch_nt<512, 4, 0>(lps, ids, count, nlp, s, e0, e1);
}
// virtual-tableau form (the 512-thread, two rows + two columns per thread instance: relaxations of up to 1024 rows / columns)
bool bt_virt_batch_supported(int m_max, int ldt_max) {
    if (bt_batch_k(m_max, ldt_max) != 8) return false;
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    return c.ri == 2 && c.nt == 512;
}
void launch_bt_inner_virt_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t lds = (size_t)(2 + 2) * 512 * (sizeof(double) + sizeof(int)) + (size_t)(2 + 2) * 512 * sizeof(double);
    hipExtLaunchKernelGGL((k_bt_inner2_virt_batch<512, 2, 2, 8, 0>), dim3(nlp), dim3(512), lds, s, e0, e1, 0, lps, ids, count);
}
// batched persistent loop kernel: relaxations of up to 1024 rows / columns (two rows + two columns per thread)
constexpr int kBLoopNU = 7;   // update workgroups per relaxation: 1 + 7 = 8 workgroups, four relaxations per XCD, 32 per launch
// relaxations per launch: one workgroup per CU at most, and one relaxation's worth of CUs per XCD left to the other streams (the second
// schedule of a split wave, the workers' final solves): a loop launch holds its CUs for a whole superstep
int b_loop_slots(int ncu) { return 8 * std::max(1, ncu / (8 * (1 + kBLoopNU)) - 1); }
bool b_loop_supported(int m_max, int ldt_max) {
    if (bt_batch_k(m_max, ldt_max) != 8) return false;
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    return c.ri == 2 && c.nt == 512 && (ldt_max & 63) == 0;
}
// Blocks of 4 pivots inside a launch: 4 current + 4 lagging terms per row / column (184 VGPRs).  Blocks of 8 — 16 terms — spill ~50
// dwords per lane into the pivot loop of a 512-thread workgroup: measured 4.9 instead of 3.4 ms for the heaviest child of the C5 wave.
void launch_b_loop(const BatchLP *lps, const int *ids, const int *count, int nlp, int nblocks, int par, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const unsigned int grid = 8u * (unsigned int)((nlp + 7) / 8) * (1u + kBLoopNU);
    const size_t lds = (size_t)(2 + 2) * 512 * (sizeof(double) + sizeof(int));
    hipExtLaunchKernelGGL((k_b_loop<512, 2, 4, kBLoopNU>), dim3(grid), dim3(512), lds, s, e0, e1, 0, lps, ids, count, nblocks, par);
}
// dual pivots of the relaxations in stage BS_DUAL (warm start); the others leave at once
template <int NT, int RI, int VL>
static void bt_inner_dual_batch_nt(const BatchLP *lps, const int *ids, const int *count, int nlp, hipStream_t s) {
    const size_t lds = (size_t)(RI + RI) * NT * (sizeof(double) + sizeof(int)) + (size_t)VL * 8 * NT * sizeof(double);
    if (lds > 64 * 1024) lds_attr_once(reinterpret_cast<const void *>(&k_bt_inner2_dual_batch<NT, RI, RI, 8, VL>), 140 * 1024);
    hipLaunchKernelGGL((k_bt_inner2_dual_batch<NT, RI, RI, 8, VL>), dim3(nlp), dim3(NT), lds, s, lps, ids, count);
}
bool bt_dual_batch_supported(int m_max, int ldt_max) { return bt_batch_k(m_max, ldt_max) == 8 && bt_batch_supported(m_max, ldt_max); }
void launch_bt_inner_dual_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s) {
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    if (c.ri == 2) { if (c.nt == 512) bt_inner_dual_batch_nt<512, 2, 0>(lps, ids, count, nlp, s); else bt_inner_dual_batch_nt<1024, 2, 1>(lps, ids, count, nlp, s); }
    else bt_inner_dual_batch_nt<512, 4, 0>(lps, ids, count, nlp, s);
}
const char *bt_batch_kernel_name(int m_max, int ldt_max) {
    if (bt_batch_k(m_max, ldt_max) == 16) return bt_group_cfg(m_max, ldt_max, 0).nt == 256 ? "k_bt_innerG_batch<8,256,1,16>" : "k_bt_innerG_batch<8,512,1,16>";
    const BtCfg c = bt_cfg(m_max, ldt_max, 0);
    return c.ri == 2 ? (c.nt == 512 ? "k_bt_inner2_batch<512,2,2,8,0>" : "k_bt_inner2_batch<1024,2,2,8,1>") : "k_bt_inner2_batch<512,4,4,8,0>";
}
void launch_bt_update_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, int m_max, int ldt_max, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (bt_batch_k(m_max, ldt_max) == 16) {   // rank-16 update on the matrix cores, one relaxation after the other inside one launch
        const int ncb = ldt_max >> 4, cw = 4;
        const int gxm = (ncb + 4 * cw - 1) / (4 * cw), gym = (m_max + 15) / 16;
        hipExtLaunchKernelGGL(k_bt_update_mfma16_batch, dim3((unsigned int)(gxm * gym * nlp)), dim3(256), 0, s, e0, e1, 0, lps, ids, count, gxm, gxm * gym, cw);
        return;
    }
    const int gx = (2 * ldt_max + kBlock - 1) / kBlock;
    const int ntr = (m_max + 3) / 4;
    const bool xcd_local = (size_t)m_max * (size_t)ldt_max * sizeof(double) <= ((size_t)3 << 20);   // fits one XCD's 4 MB L2
    int tr = 16;   // tile rows per workgroup; at least 32 workgroups per relaxation (the CUs of one XCD), 512 for a large one
    while (tr > 4 && gx * ((ntr + tr - 1) / tr) < (xcd_local ? 32 : 512)) tr >>= 1;
    const int gy = (ntr + tr - 1) / tr;
    const int nlp_pad = (nlp + 7) & ~7;
    hipExtLaunchKernelGGL((k_bt_update_tiled_batch<8>), dim3((unsigned int)(gx * gy * nlp_pad)), dim3(kBlock), 0, s, e0, e1, 0, lps, ids, count, nlp_pad, gx,
                          gx * gy, tr, xcd_local ? 1 : 0);
}

void launch_bt_tile(const double *src, double *dst, int m, int ldt, bool to_tiles, hipStream_t s) {
    const size_t pieces = (size_t)((m + 3) / 4) * 4 * (size_t)(ldt / 4);
    hipLaunchKernelGGL(k_bt_tile, dim3((unsigned int)((pieces + 255) / 256)), dim3(256), 0, s, src, dst, m, ldt, to_tiles ? 1 : 0);
}
void launch_bt_update(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (a.tiled) {
        const int gx = (2 * a.ldt + kBlock - 1) / kBlock;
        const int ntr = (a.m + 3) / 4;
        int tr = 16;   // tile rows per workgroup (64 rows); aim for >= 512 workgroups
        while (tr > 4 && gx * ((ntr + tr - 1) / tr) < 512) tr >>= 1;
        dim3 grid(gx, (ntr + tr - 1) / tr);
        if (a.kmax <= 8) hipExtLaunchKernelGGL((k_bt_update_tiled<8>), grid, dim3(kBlock), 0, s, e0, e1, 0, a, tr);
        else if (a.kmax == 16 && !a.old_only && a.upd_valu == 0) {   // matrix cores (knob "bt_upd_valu" = 1: the VALU form)
            // column blocks per wave: 4 = one trip of 16 loads per lane; measured at 2048 x 2048: 1 -> 19.7 us, 2 -> 14.2, 4 -> 12.5,
            // 8 -> 14.5, 16 -> 16.6 (the VALU form: 13.9); at 4096 x 4096: 40.9 us against 46
            const int ncb = a.ldt >> 4, cw = 4;
            dim3 gm((unsigned int)((ncb + 4 * cw - 1) / (4 * cw)), (unsigned int)((a.m + 15) / 16));
            hipExtLaunchKernelGGL(k_bt_update_mfma16, gm, dim3(256), 0, s, e0, e1, 0, a, cw);
        } else hipExtLaunchKernelGGL((k_bt_update_tiled<16>), grid, dim3(kBlock), 0, s, e0, e1, 0, a, tr);
        return;
    }
    const int ld2 = a.ldt / 2;
    const int gx = (ld2 + kWavesPerBlock * 64 - 1) / (kWavesPerBlock * 64);
    int rows = 64;
    // aim for >= 512 workgroups
    while (rows > 8 && gx * ((a.m + rows - 1) / rows) < 512) rows >>= 1;
    dim3 grid(gx, (a.m + rows - 1) / rows);
    if (a.kmax <= 8) hipExtLaunchKernelGGL((k_bt_update<8>), grid, dim3(kBlock), 0, s, e0, e1, 0, a, rows);
    else if (a.kmax <= 16) hipExtLaunchKernelGGL((k_bt_update<16>), grid, dim3(kBlock), 0, s, e0, e1, 0, a, rows);
    else hipExtLaunchKernelGGL((k_bt_update<32>), grid, dim3(kBlock), 0, s, e0, e1, 0, a, rows);
}

}  // namespace gomilp
