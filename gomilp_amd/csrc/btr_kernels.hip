// Pivot role of the persistent loop kernel with REPLICATED reduced costs (round 5; gfx950).
//
// k_bt_loop (btg_kernels.hip) spreads rows AND columns of a relaxation over its G pivot workgroups, so both argmins of a pivot
// (entering column, simplex.go:247; leaving row, :262-266) are exchanges through the XCD's L2 — two of them, ~2400 cycles each
// (workgroup stage, post, poll, pick), in a chain of ~9500 cycles per pivot.  Here only the ROWS are spread: every workgroup keeps
// the whole reduced-cost vector, the nonbasic list and the block's v' terms of ALL columns (NT threads x CPT columns), updates
// them from the pivot row it reads in full — the same arithmetic on the same values in every workgroup, so the copies never
// differ — and finds the entering column with one workgroup-local reduction: no post, no poll.  What is left of the exchange
// protocol is the leaving row (rows, x_B, the u terms stay with their owners: workgroup g holds rows [g RPG, (g + 1) RPG), one
// per thread of its first RPG / 64 waves).  The update workgroups of the launch are those of k_bt_loop (bt_loop.h): the u term of
// a row is stored by its owner, the v' term of a column by the workgroup whose slice it lies in.
//
// Where the block terms live (512 threads x 4 columns x 16 terms do not fit 256 registers): the v' terms of the RUNNING block in LDS
// (Vc: [term][column], 128 KB at 2048 columns — every thread reads and writes its own columns, and the row waves find the terms of
// the entering column there, so no workgroup ever loads a current v' term of a foreign column from memory), the LAGGING v' terms in
// registers under static indices (copied out of Vc when a block ends), the u terms of a workgroup's rows in LDS (Uc: two halves,
// lagging / current by block parity).  No register shifts (k_bt_loop moves 2 x 15 registers per row and column and pivot).
// Term traffic between workgroups: the u terms of row p and the lagging v' terms of column q come from the U / V rows in global
// memory.  Every wave awaits its own term stores before the barrier in front of a post, and a post is the only thing that lets
// another workgroup move on, so whatever a reader loads has landed (the newest lagging term in the first pivot of a block is the one
// exception: it is taken from Vc, which still holds the previous block there).  The wait is free in practice — the stores were
// issued a column phase earlier.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>

#include "device_types.h"
#include "kernels_common.h"
#include "bt_loop.h"

namespace gomilp {

namespace {
struct RWin { double m; unsigned int i; double p0, p1, p2; };   // winner of a reduction: value, first index, its scalars
}  // namespace

// G pivot workgroups x NT threads; CPT columns per thread (ldt <= NT * CPT), one row per thread of the first RPG = NT * CPT / G
// threads (m <= NT * CPT); blocks of KB pivots
template <int G, int NT, int CPT, int KB, bool STAMP>
__device__ __forceinline__ void bt_loopR_body(const BTArgs &a, const int g, const int nupd) {
    constexpr int NW = NT / 64;
    constexpr int NC = NT * CPT;
    constexpr int RPG = NC / G;
    constexpr int RW = RPG / 64;   // waves that hold rows
    static_assert(G == 16, "16 records per exchange");
    static_assert((NW == 4 || NW == 8) && RPG % 64 == 0 && RW >= 1 && RW <= NW && (CPT == 2 || CPT == 4) && KB == 8, "shape");
#ifdef GOMILP_DEBUG
    if (a.fault && g == 1) return;   // test hook (diagnostic flavour only): a workgroup that never takes part -> the others must give up (ST_XCHG_TIMEOUT)
#endif
    __shared__ double redMc[NW];
    __shared__ unsigned int redIc[NW];
    __shared__ double payc[NW][2];
    __shared__ double redMr[NW];
    __shared__ unsigned int redIr[NW];
    __shared__ double payr[NW][4];
    __shared__ int s_ok;
    extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
    double2 *Vc = reinterpret_cast<double2 *>(dyn_lds);   // [KB][CPT / 2][NT] pairs: columns CPT tid + 2 h, + 1 of term t at (t * CPT / 2 + h) * NT + tid
    double *Uc = dyn_lds + (size_t)KB * NC;               // [2][KB][RPG]: half (blk & 1) holds the running block's u terms of this workgroup's rows
    auto vc_at = [&](int t, int q) -> double { return reinterpret_cast<const double *>(Vc + ((t * (CPT / 2) + ((q % CPT) >> 1)) * NT + q / CPT))[q & 1]; };
    __shared__ unsigned long long s_acc[STAMP ? NW : 1][16];
    unsigned long long tprev = 0;
    auto stamp = [&](int seg) {
        if constexpr (STAMP) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (seg >= 0 && (threadIdx.x & 63) == 0) s_acc[threadIdx.x >> 6][seg] += t - tprev;
            tprev = t;
        }
    };
    DevState *st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const int done = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long xs = (long long)a.xbuf[0];   // written by workgroup 0 at the end of the previous launch
    xpair *recs = reinterpret_cast<xpair *>(a.xbuf + kXHeader);   // [parity][G][kXSlots]
    const double inf = __builtin_inf();
    const unsigned int ldt = (unsigned int)a.ldt;
    const char *Tb = reinterpret_cast<const char *>(a.T);
    auto ldT = [&](unsigned int elem) -> double { return ld_agent(reinterpret_cast<const double *>(Tb + (elem << 3))); };
    unsigned int myxcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
    myxcc &= 0xFu;
    bool fast = false, first = true;
    // rows: thread tid < RPG of workgroup g holds row g RPG + tid
    const bool roww = wv < RW;   // (wave-uniform)
    const int irow = g * RPG + tid;
    double xbv = 0.0;
    int basv = 0;
    if (roww) {
        xbv = irow < a.m ? a.xb[irow] : 0.0;
        basv = irow < a.m ? a.basic[irow] : 0;
    }
    // columns: every workgroup holds all of them, thread tid columns CPT tid ..
    const int j0 = CPT * tid;
    double rv[CPT];
    int nbasv[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) {
        const int j = j0 + c;
        rv[c] = j < a.nn ? a.r[j] : inf;   // padding never wins an argmin
        nbasv[c] = j < a.nn ? a.nonbasic[j] : 0;
    }
    const bool vowner = (tid / (NT / G)) == g;   // this workgroup's slice of the v' rows the update workgroups read
    int cur0 = 0, lag0 = KB, nl = 0, hl = 1;
    const int sel0 = a.par ? st->tsel2[1] : st->tsel2[0];
    const double *hdr_in = a.xbuf + 1 + 5 * a.par;
    double *hdr_out = a.xbuf + 1 + 5 * (a.par ^ 1);
    const unsigned int blk_base = (unsigned int)(unsigned long long)hdr_in[0];
    unsigned int upd_base[4];
#pragma unroll
    for (int j = 0; j < 4; j++) upd_base[j] = (unsigned int)(unsigned long long)hdr_in[1 + j];
    unsigned int *blk_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync), *upd_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync + 16);
    auto hand_on = [&](int nb) {   // counters in step for the next launch after nb blocks (bt_innerG_body)
        hdr_out[0] = (double)(unsigned int)(blk_base + (unsigned int)G * (unsigned int)nb);
#pragma unroll
        for (int j = 0; j < 4; j++) hdr_out[1 + j] = (double)(unsigned int)(upd_base[j] + (unsigned int)nupd * (unsigned int)(nb > j ? (nb - 1 - j) / 4 + 1 : 0));
    };
    if (done) {
        // a launch enqueued behind the end of the loop: release the update workgroups, keep counters and buffer choice in step
        if (tid == 0) {
            if (g == 0) {
                __hip_atomic_store(&st->kdone2[0], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(blk_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g == 0) {
                st->tsel2[a.par ^ 1] = sel0;
                st->loop_blocks = 0;
                hand_on(1);
            }
        }
        return;
    }
    double vlag[CPT][KB];
#pragma unroll
    for (int t = 0; t < KB; t++) {
#pragma unroll
        for (int c = 0; c < CPT; c++) vlag[c][t] = 0;
    }
    for (int e = tid; e < 2 * KB * RPG; e += NT) Uc[e] = 0.0;
    if constexpr (STAMP) { if (tid < NW * 16) s_acc[tid >> 4][tid & 15] = 0; }
    __syncthreads();
    int kd = 0, status = ST_RUNNING, blands = 0;
    bool dead = false;
    long long trace_len = 0, npiv = 0;
    if (g == 0 && tid == 0) { trace_len = st->trace_len; npiv = st->pivots; }

    // entering column (floats.MinIdx over all columns): local.  p0 = r_q, p1 = the entering variable
    auto local_cols = [&](const double (&val)[CPT]) -> RWin {
        stamp(-1);
        double x = val[0];
#pragma unroll
        for (int c = 1; c < CPT; c++) x = vmin_f64(x, val[c]);
        const double wm = wave_min_f64(x);
        const unsigned long long mask = __ballot(x == wm);
        int cf = CPT - 1;   // first of this thread's columns that attains the minimum (if any does)
#pragma unroll
        for (int c = CPT - 2; c >= 0; c--) cf = (val[c] == wm) ? c : cf;
        unsigned int wi = 0xFFFFFFFFu;
        if (mask) {   // (uniform)
            const int L = __builtin_ctzll(mask);
            wi = (unsigned int)(CPT * (wbase + L) + __builtin_amdgcn_readlane(cf, L));
            if (lane == L) {
                double rsel = rv[0];
                int nsel = nbasv[0];
#pragma unroll
                for (int c = 1; c < CPT; c++) { rsel = cf == c ? rv[c] : rsel; nsel = cf == c ? nbasv[c] : nsel; }
                payc[wv][0] = rsel;
                payc[wv][1] = (double)nsel;
            }
        }
        if (lane == 0) { redMc[wv] = wm; redIc[wv] = wi; }
        stamp(0);
        __syncthreads();
        stamp(1);
        // every wave: lane w holds wave w's (minimum, first index); two / three DPP minima inside the first quad / half row
        const double mw = lane < NW ? redMc[lane & (NW - 1)] : inf;
        const double iw = lane < NW ? (double)redIc[lane & (NW - 1)] : 4294967295.0;
        double xm = mw;
        xm = vmin_f64(xm, dpp_f64<0xB1>(xm));
        xm = vmin_f64(xm, dpp_f64<0x4E>(xm));
        if constexpr (NW == 8) xm = vmin_f64(xm, dpp_f64<0x141>(xm));
        const bool mine = lane < NW && mw == xm;
        double km = mine ? iw : 4294967295.0;
        km = vmin_f64(km, dpp_f64<0xB1>(km));
        km = vmin_f64(km, dpp_f64<0x4E>(km));
        if constexpr (NW == 8) km = vmin_f64(km, dpp_f64<0x141>(km));
        const unsigned int mk = (unsigned int)(__ballot(mine && iw == km) & ((1ull << NW) - 1ull));
        const int ww = mk ? __builtin_ctz(mk) : 0;   // (every minimum NaN: wave 0's entry)
        RWin r;
        r.m = readlane_f64(mw, ww);
        r.i = (unsigned int)readlane_f64(iw, ww);
        r.p0 = payc[ww][0];
        r.p1 = payc[ww][1];
        r.p2 = 0;
        stamp(2);
        return r;
    };
    // leaving row (first index of the minimum over all rows): the exchange of bt_innerG_body, G = 16.  p0 = d_p, p1 = x_B[p], p2 = the
    // leaving variable.  Every wave awaits its own term stores before the barrier in front of the post (see the head of this file)
    auto xchg_rows = [&](const double val, const double dcolv) -> RWin {
        stamp(4);
        if (roww) {
            const double wm = wave_min_f64(val);
            const unsigned long long mask = __ballot(val == wm);
            unsigned int wi = 0xFFFFFFFFu;
            if (mask) {
                const int L = __builtin_ctzll(mask);
                wi = (unsigned int)(g * RPG + wbase + L);
                if (lane == L) { payr[wv][0] = dcolv; payr[wv][1] = xbv; payr[wv][2] = (double)basv; payr[wv][3] = 0.0; }
            }
            if (lane == 0) { redMr[wv] = wm; redIr[wv] = wi; }
        }
        stamp(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(6);
        xs += 1;
        const double seqd = (double)xs;
        const int par = (int)(xs & 1);
        if (wv == 0) {
            double fm = redMr[0], v0 = payr[0][lane < 5 && lane >= 2 ? lane - 2 : 3];
            unsigned int fi = redIr[0];
#pragma unroll
            for (int w2 = 1; w2 < RW; w2++) {
                const double mw = redMr[w2];
                const unsigned int iw = redIr[w2];
                const double pw = payr[w2][lane < 5 && lane >= 2 ? lane - 2 : 3];
                // (value, index) lexicographic; a NaN minimum (index 0xFFFFFFFF) never replaces anything
                const bool take = (mw < fm) || (mw == fm && iw < fi) || (fm != fm && mw == mw);
                fm = take ? mw : fm;
                fi = take ? iw : fi;
                v0 = take ? pw : v0;
            }
            if (lane < kXSlots) {
                xpair v;
                v.x = seqd;
                v.y = lane == 0 ? fm : lane == 1 ? (double)fi : lane == 5 ? (double)myxcc : v0;
                xstore(recs + ((size_t)(par * G + g) * kXSlots + lane), v, fast);
            }
        }
        stamp(7);
        // 16 records: two 16-byte loads per lane, both slots of record l & 15 — lanes 0..15 get (minimum, first index) of record l, the
        // other rows the winner's scalars (slots 2, 3, 4) and the XCC ids (slot 5)
        const int rec = lane & 15, grp = lane >> 4;
        const bool actB = grp < 3;
        const xpair *base = recs + (size_t)par * G * kXSlots + rec * kXSlots;
        const xpair *srcA = base + (grp == 0 ? 0 : grp + 1), *srcB = base + (grp == 0 ? 1 : grp == 1 ? 5 : grp == 2 ? 6 : 0);
        xpair got[2];
        int spins = 0;
        for (;;) {
            XLoad<2>::run(srcA, srcB, got, fast);
            if (__all(got[0].x == seqd && (!actB || got[1].x == seqd))) break;
            if (++spins > kXSpinLimit) { dead = true; break; }
        }
        stamp(8);
        const double v1 = got[0].y, v2 = got[1].y;
        double xm = (lane < 16) ? v1 : inf;
        xm = vmin_f64(xm, dpp_f64<0xB1>(xm));
        xm = vmin_f64(xm, dpp_f64<0x4E>(xm));
        xm = vmin_f64(xm, dpp_f64<0x141>(xm));   // row_half_mirror
        xm = vmin_f64(xm, dpp_f64<0x140>(xm));   // row_mirror: lanes 0..15 all hold the minimum
        const bool mine = lane < 16 && v1 == xm;
        const unsigned int mk0 = (unsigned int)(__ballot(mine) & 0xFFFFull);
        int gw;
        double bi;
        if (__builtin_popcount(mk0) == 1) {   // (uniform) one record attains the minimum — the usual case
            gw = __builtin_ctz(mk0);
            bi = readlane_f64(v2, gw);
        } else {
            double km = mine ? v2 : 4294967295.0;
            km = vmin_f64(km, dpp_f64<0xB1>(km));
            km = vmin_f64(km, dpp_f64<0x4E>(km));
            km = vmin_f64(km, dpp_f64<0x141>(km));
            km = vmin_f64(km, dpp_f64<0x140>(km));
            const unsigned int mk = (unsigned int)(__ballot(mine && v2 == km) & 0xFFFFull);
            gw = mk ? __builtin_ctz(mk) : 0;
            bi = readlane_f64(km, 0);
        }
        RWin r;
        r.m = readlane_f64(xm, 0);
        r.i = (unsigned int)bi;
        r.p0 = readlane_f64(v1, 16 + gw);
        r.p1 = readlane_f64(v1, 32 + gw);
        r.p2 = readlane_f64(v1, 48 + gw);
        if (first) {
            fast = !dead && __all(grp != 1 || v2 == (double)myxcc);
            first = false;
        }
        stamp(9);
        return r;
    };
    // entry (irow, q) of the current tableau (row waves): the buffer + 8 lagging + the block's k current terms; the v' terms of column
    // q: lane l < KB fetches lagging term l, lane KB + t current term t; the newest one comes from this workgroup's own copy
    auto column = [&](int q, int k) -> double {
        const unsigned int ic = (unsigned int)(irow < a.m ? irow : a.m - 1);
        const double d0 = ldT(tile_off_g(ic, (unsigned int)q, ldt));
        double tv = 0.0;
        if (lane < KB) {
            if (nl > 0) {
                if (k == 0 && lane == KB - 1) tv = vc_at(KB - 1, q);   // (its store by the slice's owner may still be in flight)
                else tv = ld_agent(a.V + (size_t)(lag0 + lane) * a.ldt + q);
            }
        } else if (lane - KB < k) tv = vc_at(lane - KB, q);
        // this row's u terms: lagging half, current half (terms beyond k: not of this block)
        const double *ul = Uc + (size_t)(hl * KB) * RPG + tid, *uc = Uc + (size_t)((hl ^ 1) * KB) * RPG + tid;
        double ulv[KB], ucv[KB];
#pragma unroll
        for (int t = 0; t < KB; t++) { ulv[t] = ul[t * RPG]; ucv[t] = uc[t * RPG]; }
        if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp(11); }
        double acc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < KB; t++) acc[t & 3] = __builtin_fma(ulv[t], readlane_f64(tv, t), acc[t & 3]);
#pragma unroll
        for (int t = 0; t < KB; t++) acc[t & 3] = __builtin_fma(t < k ? ucv[t] : 0.0, readlane_f64(tv, KB + t), acc[t & 3]);
        const double d = d0 + ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        return irow < a.m ? d : 0.0;
    };
    // ratio (simplex.go:321-340), branch-free
    auto ratio = [&](double dcolv) -> double {
        double d = -dcolv;
        if (fabs(d) < 1e-13) d = 0;
        const double quot = div_pos(xbv, fabs(d));
        return (d >= 0 || irow >= a.m) ? inf : quot;
    };

    int nbe = 0;   // blocks run by this launch
    for (int blk = 0; blk < a.nblocks; blk++) {
        // block blk reads the tableau after blk - 1 blocks: the update of block blk - 2 must be through (nupd arrivals per block)
        if (blk >= 2) {
            if (wv == 0) {
                const int cj = (blk - 2) & 3;
                const unsigned int ub = cj == 0 ? upd_base[0] : cj == 1 ? upd_base[1] : cj == 2 ? upd_base[2] : upd_base[3];
                const bool ok = spin_counter(upd_cnt + 16 * cj * 2, ub + (unsigned int)nupd * (unsigned int)((blk - 2) / 4 + 1), 0);
                if (lane == 0) s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ok) { dead = true; break; }
        }
        Tb = reinterpret_cast<const char *>(((sel0 ^ (blk > 0 ? blk - 1 : 0)) & 1) ? a.Tbuf[1] : a.Tbuf[0]);
        cur0 = (blk & 1) * KB; lag0 = cur0 ^ KB; nl = blk > 0 ? KB : 0;
        hl = (blk & 1) ^ 1;   // the half of Uc with the lagging u terms (the other one takes this block's)
        if (blk > 0) {   // the previous block's terms are the lagging ones now; the block before it is in the tableau
#pragma unroll
            for (int t = 0; t < KB; t++) {
#pragma unroll
                for (int h = 0; h < CPT / 2; h++) {
                    const double2 w2 = Vc[(t * (CPT / 2) + h) * NT + tid];
                    vlag[2 * h][t] = w2.x; vlag[2 * h + 1][t] = w2.y;
                }
            }
        }
        kd = 0;
        for (int k = 0; k < a.kmax; k++) {
            const bool forced = (k == 0 && blk == 0 && a.forced_q >= 0);
            int q, p, ent = 0, lea = 0;
            double rq = 0, dpv = 1.0, xbp = 0;
            bool bland = false;
            double dcolv = 0.0;
            if (!forced) {
                const RWin fq = local_cols(rv);
                q = (int)fq.i; rq = fq.p0; ent = (int)fq.p1;
                if (fq.i >= (unsigned int)a.nn) { q = 0; rq = __builtin_nan(""); }   // every r_j is NaN: MinIdx returns 0
                if (a.guard == inf && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // strict mode: every decision is the host's
                if (rq >= -a.tol) { status = ST_OPTIMAL; break; }                    // simplex.go:248
                if (roww) dcolv = column(q, k);
                const RWin w = xchg_rows(roww ? ratio(dcolv) : inf, dcolv);
                if (dead) break;
                p = (int)w.i; dpv = w.p0; xbp = w.p1; lea = (int)w.p2;
                const double mv = w.m;
                if (mv == inf || w.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // simplex.go:328-330
                if (a.guard > 0 && (mv <= a.guard || fabs(dpv) <= a.guard) && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }
                if (a.cguard > 0 && fabs(dpv) <= a.cguard && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }
                if (mv <= 0) {
                    // replaceBland (simplex.go:347-383): candidates in position order
                    bland = true;
                    blands++;
                    int cand = -1;
                    bool found = false;
                    for (;;) {
                        double fl[CPT];
#pragma unroll
                        for (int c = 0; c < CPT; c++) {
                            const int j = j0 + c;
                            double rr = rv[c];
                            if (fabs(rr) < 1e-13) rr = 0;
                            fl[c] = (j < a.nn && j > cand && !(rr > -1e-14)) ? 0.0 : inf;
                        }
                        const RWin fc = local_cols(fl);
                        if (fc.m != 0.0) break;   // candidates exhausted -> ErrBland
                        cand = (int)fc.i;
                        const double rqc = fc.p0;
                        const int entc = (int)fc.p1;
                        if (roww) dcolv = column(cand, k);
                        const RWin w2 = xchg_rows(roww ? ratio(dcolv) : inf, dcolv);
                        if (dead) break;
                        if (w2.m == inf || w2.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // :356-360
                        if (fabs(w2.m) > 1e-12) {   // :362
                            q = cand; p = (int)w2.i; rq = rqc; ent = entc; dpv = w2.p0; xbp = w2.p1; lea = (int)w2.p2;
                            found = true;
                            break;
                        }
                        double gl2 = roww ? ratio(dcolv) : inf;
                        gl2 = (roww && irow < a.m && !(gl2 > 1e-12)) ? 0.0 : inf;
                        const RWin gw = xchg_rows(gl2, dcolv);
                        if (dead) break;
                        if (gw.m == 0.0) {   // :368-379
                            q = cand; p = (int)gw.i; rq = rqc; ent = entc; dpv = gw.p0; xbp = gw.p1; lea = (int)gw.p2;
                            found = true;
                            break;
                        }
                    }
                    if (dead || status == ST_UNBOUNDED) break;
                    if (!found) { status = ST_BLAND_FAILED; break; }
                }
            } else {
                // a pivot chosen by the host (first pivot of a launch: no current terms yet)
                q = a.forced_q; p = a.forced_p; rq = 0;
                double fl[CPT];
#pragma unroll
                for (int c = 0; c < CPT; c++) fl[c] = (j0 + c == q) ? 0.0 : inf;
                const RWin fc = local_cols(fl);
                ent = (int)fc.p1;
                if (!a.forced_nocommit) rq = fc.p0;   // a pivot the host decided on fresh solves (exact_step): a pivot like any other
                if (roww) dcolv = column(q, k);
                const RWin gw = xchg_rows((roww && irow == p) ? 0.0 : inf, dcolv);
                if (dead) break;
                dpv = gw.p0; xbp = gw.p1; lea = (int)gw.p2;
            }
            // ---- row p for ALL columns (every workgroup the same), reduced costs, block terms
            stamp(-1);
            const double rinv = 1.0 / dpv, nrinv = -rinv;
            const double mult = rq * rinv;
            const double theta = xbp * rinv;
            double *Vk = a.V + (size_t)(cur0 + k) * a.ldt;
            double *Uk = a.U + (size_t)(cur0 + k) * a.ldu;
            const bool commit_lists = !(forced && a.forced_nocommit) || (forced && a.forced_nocommit >= 2);
            double vrow[CPT];
            if (j0 < a.ldt) {
                const double *src = reinterpret_cast<const double *>(Tb) + tile_off_g((unsigned int)p, (unsigned int)j0, ldt);
#pragma unroll
                for (int c = 0; c < CPT; c++) vrow[c] = ld_agent(src + c);
            } else {
#pragma unroll
                for (int c = 0; c < CPT; c++) vrow[c] = 0.0;
            }
            // u_t[p] of the lagging and the block's k earlier pivots (all landed: awaited by their owner before its post of this pivot)
            double tu = 0.0;
            if (lane < KB) { if (nl > 0) tu = ld_agent(a.U + (size_t)(lag0 + lane) * a.ldu + p); }
            else if (lane - KB < k) tu = ld_agent(a.U + (size_t)(cur0 + lane - KB) * a.ldu + p);
            if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(12); }
            if (roww && irow < a.ldu) {
                const double unew = (irow == p) ? rinv - 1.0 : dcolv * nrinv;
                if (irow < a.m) xbv = (irow == p) ? theta : __builtin_fma(-theta, dcolv, xbv);
                st_agent(Uk + irow, unew);
                Uc[(size_t)((hl ^ 1) * KB + k) * RPG + tid] = unew;
                if (irow == p && commit_lists) basv = ent;
            }
            double cs[CPT];   // the running block's part of the correction: terms out of Vc
#pragma unroll
            for (int c = 0; c < CPT; c++) cs[c] = 0.0;
            for (int t = 0; t < k; t++) {
                const double ut = readlane_f64(tu, KB + t);
#pragma unroll
                for (int h = 0; h < CPT / 2; h++) {
                    const double2 w2 = Vc[(t * (CPT / 2) + h) * NT + tid];
                    cs[2 * h] = __builtin_fma(ut, w2.x, cs[2 * h]);
                    cs[2 * h + 1] = __builtin_fma(ut, w2.y, cs[2 * h + 1]);
                }
            }
            double vp[CPT];
#pragma unroll
            for (int c = 0; c < CPT; c++) {
                const int j = j0 + c;
                double acc[4] = {0, 0, 0, 0};
#pragma unroll
                for (int t = 0; t < KB; t++) acc[t & 3] = __builtin_fma(readlane_f64(tu, t), vlag[c][t], acc[t & 3]);
                const double v = vrow[c] + (((acc[0] + acc[1]) + (acc[2] + acc[3])) + cs[c]);
                rv[c] = (j == q) ? -mult : __builtin_fma(-mult, v, rv[c]);
                vp[c] = (j == q) ? dpv + 1.0 : v;
                if (j == q && commit_lists) nbasv[c] = lea;
            }
            // this pivot's v' terms: the workgroup's copy, and the row the update workgroups read
#pragma unroll
            for (int h = 0; h < CPT / 2; h++) {
                double2 w2; w2.x = vp[2 * h]; w2.y = vp[2 * h + 1];
                Vc[(k * (CPT / 2) + h) * NT + tid] = w2;
            }
            if (vowner && j0 < a.ldt) {
#pragma unroll
                for (int c = 0; c < CPT; c++) st_agent(Vk + j0 + c, vp[c]);
            }
            stamp(10);
            if (forced && a.forced_nocommit == 3) status = ST_FORCED_DONE;
            if (g == 0 && tid == 0 && !(forced && a.forced_nocommit)) {   // simplex.go:280
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
        // hand the block to the update workgroups: every term store of this workgroup has landed (agent scope), workgroup 0 publishes
        // the pivot count (and the end of the loop) BEFORE its arrival, the arrivals of all G workgroups release them
        if (dead) status = ST_XCHG_TIMEOUT;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (g == 0) {
                __hip_atomic_store(&st->kdone2[blk & 1], kd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (status != ST_RUNNING) {
                    __hip_atomic_store(&st->status, status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&st->done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(blk_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (status != ST_RUNNING) break;
    }
    if (dead) status = ST_XCHG_TIMEOUT;
    if constexpr (STAMP) {
        if (a.stamps && g == 0 && lane == 0) {
            for (int sg = 0; sg < 16; sg++) a.stamps[wv * 16 + sg] += s_acc[wv][sg];
            if (wv == 0) a.stamps[16 * 16] += (unsigned long long)kd;
        }
    }
    if (g == 0) {   // (every workgroup holds the same reduced costs and nonbasic list)
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            const int j = j0 + c;
            if (j < a.ldt) a.r[j] = j < a.nn ? rv[c] : 0.0;
            if (j < a.nn) a.nonbasic[j] = nbasv[c];
        }
    }
    if (roww) {
        if (irow < a.ldu) a.xb[irow] = xbv;
        if (irow < a.m) a.basic[irow] = basv;
    }
    if (tid == 0 && (g == 0 || dead)) {
        if (g == 0) {
            st->trace_len = trace_len;
            st->pivots = npiv;
            // the update workgroups apply every block with pivots before the launch ends: the tableau after them
            const int napplied = kd > 0 ? nbe : nbe - 1;
            st->tsel2[a.par ^ 1] = sel0 ^ (napplied & 1);
            st->loop_blocks = nbe;
            hand_on(nbe);
            st->bland_steps += blands;
            a.xbuf[0] = (double)xs;
        }
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

// Blocks x, x + 8, ..., x + 8 (G - 1) — one XCD under the round-robin placement of blocks — are the pivot workgroups, every other block
// of the grid an update workgroup (bt_loop.h).  All workgroups of the launch must be resident: one per CU, and a 512-thread workgroup with
// this kernel's registers fills its CU — such a launch owns the device (Engine::loop_try_acquire_all).
template <int G, int NT, int CPT, int KB, bool STAMP = false>
__global__ __launch_bounds__(NT) void k_bt_loopR(BTArgs a) {
    const int b = (int)blockIdx.x, x = a.xcd & 7;
    const int nupd = (a.upd_cap > 0 && a.upd_cap < (int)gridDim.x - G) ? a.upd_cap : (int)gridDim.x - G;
    if ((b & 7) == x && (b >> 3) < G) { bt_loopR_body<G, NT, CPT, KB, STAMP>(a, b >> 3, nupd); return; }
    const int before = b <= x ? 0 : min(G, ((b - x - 1) >> 3) + 1);   // pivot blocks in front of block b
    if (b - before >= nupd) return;
    bt_loop_update_role<NT, KB>(a, b - before, nupd, G);
}

// ---- host side ---------------------------------------------------------------------------------
// shapes: up to 2048 rows and 2048 tableau columns — 16 x 512 threads x 4 columns; up to 1024 x 1024 — 16 x 256 threads x 4 columns
bool bt_loop_rep_supported(int m, int ldt) { return (m <= 2048 && ldt <= 2048 && m > 0); }
int bt_loop_rep_threads(int m, int ldt) { return (m <= 1024 && ldt <= 1024) ? 256 : 512; }
template <int NT, int CPT, int KB> static constexpr int rep_lds_bytes() { return (KB * NT * CPT + 2 * KB * (NT * CPT / 16)) * 8; }
static std::atomic<long long> g_rep_launches{0};
long long bt_loop_rep_launches() { return g_rep_launches.load(); }   // (tests: which pivot role ran)
void launch_bt_loop_rep(const BTArgs &a, int ncu, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    g_rep_launches++;
    const unsigned int grid = (unsigned int)std::max(std::max(ncu, 8 * 16), 136);
    if (a.group_nt == 256) {
        constexpr int lds = rep_lds_bytes<256, 4, 8>();
        lds_attr_once(reinterpret_cast<const void *>(k_bt_loopR<16, 256, 4, 8>), lds);
        hipExtLaunchKernelGGL((k_bt_loopR<16, 256, 4, 8>), dim3(grid), dim3(256), lds, s, e0, e1, 0, a);
        return;
    }
    constexpr int lds = rep_lds_bytes<512, 4, 8>();
    if (a.stamps) {   // diagnostic build
        lds_attr_once(reinterpret_cast<const void *>(k_bt_loopR<16, 512, 4, 8, true>), lds);
        hipExtLaunchKernelGGL((k_bt_loopR<16, 512, 4, 8, true>), dim3(grid), dim3(512), lds, s, e0, e1, 0, a);
    } else {
        lds_attr_once(reinterpret_cast<const void *>(k_bt_loopR<16, 512, 4, 8>), lds);
        hipExtLaunchKernelGGL((k_bt_loopR<16, 512, 4, 8>), dim3(grid), dim3(512), lds, s, e0, e1, 0, a);
    }
}

}  // namespace gomilp
