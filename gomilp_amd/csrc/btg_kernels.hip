// Block kernel of the blocked tableau pipeline spread over G workgroups of ONE XCD (gfx950).
//
// k_bt_inner2 (bt_kernels.hip) runs the K pivots of a block on a single CU: at 2048 rows every pivot drags 128 KB through
// one L1 (a column and a row of T in 4x4 tiles) and about 600 VALU instructions per wave through 4 SIMDs, 4.7 us per pivot;
// 4096-row shapes do not fit its registers at all.  Here G = 2 / 4 / 8 workgroups share a block: workgroup g owns the rows
// and columns (s*G + g)*NT + t — x_B, r, the index lists and the block's rank-1 terms u_k[i], v'_k[j] of exactly those, in
// LDS / registers as before — so loads, corrections and term shifts shrink by G and K = 16 terms fit in registers.
// What stays global are the two argmins of a pivot (entering column: simplex.go:247, leaving row: :262-266).  Each becomes
// an EXCHANGE through the XCD's L2:
//   * every workgroup reduces its own candidates (v_min_f64 over DPP + ballot, as in k_bt_inner2) and wave 0 posts a
//     record {min, first index, three scalars about the winner (r_q, the entering variable; or d_p, x_B[p], the leaving
//     variable), the workgroup's XCC id} — one 128-byte line of 8 slots {sequence number, value}, one 16-byte store per
//     lane, so a slot can never be seen half-written;
//   * every wave polls the G records (loads that L1 never serves) until every slot carries this exchange's sequence
//     number and takes the lexicographic minimum (value, index) — floats.MinIdx over the whole vector — itself: one
//     barrier per exchange, no LDS hop;
//   * the block terms of foreign rows / columns (v'_j[q], u_j[p]) are read from the U / V rows of the running block, which
//     the owners write anyway for the update kernel (agent-scope stores, awaited before the owner's next post).
// Records are double buffered by sequence parity: a workgroup can be at most one exchange ahead of the slowest one.
// Measured (tools/xsync_bench.hip, MI355X): 0.64 us per bare exchange at G = 4 on one XCD, 0.95 us across XCDs; the launch
// therefore uses blocks 0, 8, 16, ... of a grid of 8*G (blocks are dealt round-robin over the 8 XCDs; the others leave at
// once).  Placement is a speed matter only: the first exchange of a launch uses agent-scope accesses and carries the XCC
// ids; only if all are equal do the record accesses drop to the L2-only forms (see xstore).
// Every workgroup takes the same decisions from the same exchanged values, so control flow never diverges between them; a
// poll that sees no progress for ~1 s (a workgroup never got a CU) ends the launch with ST_XCHG_TIMEOUT in every workgroup.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "device_types.h"
#include "kernels_common.h"
#include "bt_loop.h"

namespace gomilp {

namespace {

struct XWin { double m; unsigned int i; double p0, p1, p2, p3; };   // winner of an exchange: value, first index, its scalars (p3: its newest block term)
struct BtWinG { double m; unsigned int i; };                   // a wave's own winner

}  // namespace

// G workgroups x NT threads, RI rows and RI columns per thread (m <= G*NT*RI, ldt <= G*NT*RI), KR block terms in registers
// LOOP: the pivot role of the persistent loop kernel (k_bt_loop): up to a.nblocks blocks of 8 pivots in one launch, the terms
// of the previous block carried along as lagging terms while the update workgroups of the same launch apply them
template <int G, int NT, int RI, int KR, bool STAMP, bool LOOP = false>
__device__ __forceinline__ void bt_innerG_body(const BTArgs &a, const int g, const int nupd = 0) {
    constexpr int NW = NT / 64;
    static_assert(!LOOP || KR == 16 || KR == 24 || KR == 32, "loop mode: KR / 2 lagging + KR / 2 current terms");
    constexpr int KB = LOOP ? KR / 2 : KR;   // pivots per block
    static_assert(G == 2 || G == 4 || G == 8 || G == 16, "G");
#ifdef GOMILP_DEBUG
    if (a.fault && g == 1) return;   // test hook (diagnostic flavour only): a workgroup that never takes part -> the others must give up (ST_XCHG_TIMEOUT)
#endif
    __shared__ double redM[16];
    __shared__ unsigned int redI[16];
    __shared__ double pay[16][4];
    __shared__ int s_ok;
    // STAMP: diagnostic build (knob "bt_stamps"): cycles per pivot segment and wave of workgroup 0 (s_memtime), summed in LDS
    __shared__ unsigned long long s_acc[STAMP ? 16 : 1][16];
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
    // the tableau this block reads: a.T, or in loop mode the buffer the update workgroups finished two blocks ago (they write
    // it with agent-scope stores from other XCDs while this launch runs: agent-scope loads)
    const char *Tb = reinterpret_cast<const char *>(a.T);
    auto ldT = [&](unsigned int elem) -> double {
        if constexpr (LOOP) return ld_agent(reinterpret_cast<const double *>(Tb + (elem << 3)));
        else return *reinterpret_cast<const double *>(Tb + (elem << 3));
    };
    // block terms of OTHER workgroups' rows / columns come from the U / V rows of the running block in global memory:
    // agent-scope (sc1) stores by the owner, agent-scope loads here — never a stale L1 line, correct on any XCD
    auto ld_term = [&](const double *p) -> double { return ld_agent(p); };
    auto st_term = [&](double *p, double v) { st_agent(p, v); };
    auto gidx = [&](int s) -> int { return (s * G + g) * NT + tid; };   // row / column index of this thread's slot s
    // `fast` is decided by the first exchange of the launch, which carries every workgroup's XCC id: all equal -> the
    // same-XCD record accesses from then on (every workgroup sees the same ids, so all switch together)
    unsigned int myxcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
    myxcc &= 0xFu;
    bool fast = false, first = true;
    double xbv[RI], rv[RI];   // x_B, r and the index lists of this thread's rows / columns: registers
    int basv[RI], nbasv[RI];
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = gidx(s);
        xbv[s] = i < a.m ? a.xb[i] : 0.0;
        basv[s] = i < a.m ? a.basic[i] : 0;
        rv[s] = i < a.nn ? a.r[i] : inf;   // padding never wins an argmin
        nbasv[s] = i < a.nn ? a.nonbasic[i] : 0;
    }
    // loop mode: rows [cur0, cur0 + KB) of U / V take the terms of the running block, rows [lag0, lag0 + nl) hold those of the
    // previous one (not yet in the tableau this block reads)
    int cur0 = 0, lag0 = KB, nl = 0;
    const int sel0 = LOOP ? (a.par ? st->tsel2[1] : st->tsel2[0]) : 0;
    const double *hdr_in = a.xbuf + 1 + 5 * (LOOP ? a.par : 0);
    double *hdr_out = a.xbuf + 1 + 5 * (LOOP ? (a.par ^ 1) : 0);
    const unsigned int blk_base = LOOP ? (unsigned int)(unsigned long long)hdr_in[0] : 0u;
    unsigned int upd_base[4];
#pragma unroll
    for (int j = 0; j < 4; j++) upd_base[j] = LOOP ? (unsigned int)(unsigned long long)hdr_in[1 + j] : 0u;
    unsigned int *blk_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync), *upd_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync + 16);
    // counters in step for the next launch after nb blocks: G arrivals per block, nupd per block on the counter of its index mod 4
    auto hand_on = [&](int nb) {
        hdr_out[0] = (double)(unsigned int)(blk_base + (unsigned int)G * (unsigned int)nb);
#pragma unroll
        for (int j = 0; j < 4; j++) hdr_out[1 + j] = (double)(unsigned int)(upd_base[j] + (unsigned int)nupd * (unsigned int)(nb > j ? (nb - 1 - j) / 4 + 1 : 0));
    };
    if (done) {
        if constexpr (LOOP) {
            // a launch enqueued behind the end of the loop: the update workgroups wait for block 0 — release them (no pivots to
            // apply) and keep counters and buffer choice in step for the next launch
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
        } else if (g == 0 && tid == 0) st->kdone = 0;
        return;
    }
    double ureg[RI][KR], vreg[RI][KR];
#pragma unroll
    for (int s = 0; s < RI; s++) {
#pragma unroll
        for (int j = 0; j < KR; j++) { ureg[s][j] = 0; vreg[s][j] = 0; }
    }
    if constexpr (STAMP) { if (tid < 256) s_acc[tid >> 4][tid & 15] = 0; }
    __syncthreads();
    int kd = 0, status = ST_RUNNING, blands = 0;
    bool dead = false;
    long long trace_len = 0, npiv = 0;
    if (g == 0 && tid == 0) { trace_len = st->trace_len; npiv = st->pivots; }

    auto wave_first_min = [&](const double (&val)[RI]) -> BtWinG {
        double x = val[0];
#pragma unroll
        for (int s = 1; s < RI; s++) x = vmin_f64(x, val[s]);
        BtWinG w;
        w.m = wave_min_f64(x);
        w.i = 0xFFFFFFFFu;
#pragma unroll
        for (int s = RI - 1; s >= 0; s--) {
            const unsigned long long mask = __ballot(val[s] == w.m);
            if (mask) w.i = (unsigned int)((s * G + g) * NT + wbase + __builtin_ctzll(mask));
        }
        return w;
    };
    // The exchange.  In: this workgroup's per-wave winners in redM / redI / pay.  Wave 0 reduces them and posts the
    // workgroup's record; EVERY wave then polls the G records itself (one 16-byte load per lane: lane l reads slot l & 7 of
    // record l >> 3), picks the winner with one wave-wide min and keeps its scalars as uniform values — no second barrier,
    // no LDS hop.  A wave can pass the poll only after wave 0 has posted, i.e. after it has read redM / redI / pay, so the
    // next reduction may overwrite them without another barrier.
    auto xchg = [&](int which) -> XWin {
        stamp(which * 5 + 0);
        // this wave's term stores have reached L2 / memory before the post.  Loop mode posts without that wait: the NEWEST term of the
        // candidate travels in the record (slot 6), readers take the older ones from U / V, and those were awaited before the row phase
        // that stored the newest (see there) — the stores of a pivot drain under the next pivot's exchanges instead of in front of them
        if constexpr (!LOOP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(which * 5 + 1);
        xs += 1;
        const double seqd = (double)xs;
        const int par = (int)(xs & 1);
        if (wv == 0) {
            // the workgroup's winner over its NW waves (uniform LDS reads, every lane computes the same): first index among the
            // waves that attain the minimum; lanes 0..7 post {sequence number, value}
            double fm, v0;
            unsigned int fi;
            if constexpr (LOOP && (NW == 4 || NW == 8)) {
                // lane w holds wave w's (minimum, first index): two DPP minima inside the first quad / half row instead of a chain
                // of dependent LDS reads and selects (the serial form cost wave 0 ~1000 cycles per exchange, the others ~240)
                const double mw = lane < NW ? redM[lane & (NW - 1)] : inf;
                const double iw = lane < NW ? (double)redI[lane & (NW - 1)] : 4294967295.0;
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
                const int ww = mk ? __builtin_ctz(mk) : 0;   // (every minimum NaN: wave 0's entry, as the serial form)
                fm = readlane_f64(mw, ww);
                fi = (unsigned int)readlane_f64(iw, ww);
                v0 = pay[ww][lane < 5 && lane >= 2 ? lane - 2 : 3];
            } else {
                fm = redM[0];
                fi = redI[0];
                v0 = pay[0][lane < 5 && lane >= 2 ? lane - 2 : 3];
#pragma unroll
                for (int w2 = 1; w2 < NW; w2++) {
                    const double mw = redM[w2];
                    const unsigned int iw = redI[w2];
                    const double pw = pay[w2][lane < 5 && lane >= 2 ? lane - 2 : 3];
                    // (value, index) lexicographic; a NaN minimum (index 0xFFFFFFFF) never replaces anything
                    const bool take = (mw < fm) || (mw == fm && iw < fi) || (fm != fm && mw == mw);
                    fm = take ? mw : fm;
                    fi = take ? iw : fi;
                    v0 = take ? pw : v0;
                }
            }
            if (lane < kXSlots) {
                xpair v;
                v.x = seqd;
                v.y = lane == 0 ? fm : lane == 1 ? (double)fi : lane == 5 ? (double)myxcc : v0;
                xstore(recs + ((size_t)(par * G + g) * kXSlots + lane), v, fast);
            }
            if constexpr (STAMP && !LOOP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        stamp(which * 5 + 2);
        for (int dl = 0; dl < a.poll_delay; dl++) __builtin_amdgcn_s_sleep(1);
        if constexpr (G == 16) {
            // 16 records: two 16-byte loads per lane, both slots of the same record l & 15 — lanes 0..15 get (minimum, first index)
            // of record l directly, the other rows the winner's scalars (slots 2, 3, 4) and the XCC ids (slot 5)
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
            stamp(which * 5 + 3);
            const double val = got[0].y, val2 = got[1].y;
            double xm = (lane < 16) ? val : inf;
            xm = vmin_f64(xm, dpp_f64<0xB1>(xm));
            xm = vmin_f64(xm, dpp_f64<0x4E>(xm));
            xm = vmin_f64(xm, dpp_f64<0x141>(xm));   // row_half_mirror
            xm = vmin_f64(xm, dpp_f64<0x140>(xm));   // row_mirror: lanes 0..15 all hold the minimum
            const bool mine = lane < 16 && val == xm;
            const unsigned int mk0 = (unsigned int)(__ballot(mine) & 0xFFFFull);
            int gw;
            double bi;
            if (__builtin_popcount(mk0) == 1) {   // (uniform) one record attains the minimum — the usual case: its index, no second reduction
                gw = __builtin_ctz(mk0);
                bi = readlane_f64(val2, gw);
            } else {
                double km = mine ? val2 : 4294967295.0;
                km = vmin_f64(km, dpp_f64<0xB1>(km));
                km = vmin_f64(km, dpp_f64<0x4E>(km));
                km = vmin_f64(km, dpp_f64<0x141>(km));
                km = vmin_f64(km, dpp_f64<0x140>(km));
                const unsigned int mk = (unsigned int)(__ballot(mine && val2 == km) & 0xFFFFull);
                gw = mk ? __builtin_ctz(mk) : 0;
                bi = readlane_f64(km, 0);
            }
            XWin r;
            r.m = readlane_f64(xm, 0);
            r.i = (unsigned int)bi;
            r.p0 = readlane_f64(val, 16 + gw);
            r.p1 = readlane_f64(val, 32 + gw);
            r.p2 = readlane_f64(val, 48 + gw);
            r.p3 = readlane_f64(val2, 32 + gw);
            if (first) {
                fast = !dead && __all(grp != 1 || val2 == (double)myxcc);
                first = false;
            }
            stamp(which * 5 + 4);
            return r;
        }
        // lane l reads slot l >> 3 of record l & 7: the G minima sit in lanes 0..G-1, their indices in lanes 8.., the scalars behind
        const bool act = (lane & 7) < G && (lane >> 3) < 7;
        const xpair *src = recs + (size_t)par * G * kXSlots + (act ? (lane & 7) * kXSlots + (lane >> 3) : 0);
        xpair got[1];
        int spins = 0;
        for (;;) {
            XLoad<1>::run(src, got, fast);
            if (__all(!act || got[0].x == seqd)) break;
            if (++spins > kXSpinLimit) { dead = true; break; }
        }
        stamp(which * 5 + 3);
        // lexicographic minimum of (value, first index) over the G records; a NaN value never wins (v_min_f64): min over
        // lanes 0..7 by three DPP steps inside the half row
        const double val = got[0].y;
        double xm = (lane < G) ? val : inf;
        xm = vmin_f64(xm, dpp_f64<0xB1>(xm));    // quad_perm [1,0,3,2]
        xm = vmin_f64(xm, dpp_f64<0x4E>(xm));    // quad_perm [2,3,0,1]
        xm = vmin_f64(xm, dpp_f64<0x141>(xm));   // row_half_mirror
        // (lanes 0..7 all hold the minimum now.)  Smallest index among the records that attain it, without a trip through
        // scalar registers per candidate: the index of record l (lane 8 + l) is shifted into lane l, lanes that do not attain
        // the minimum offer +big, three more DPP steps, one ballot finds the owner
        const double idxl = dpp_f64<0x108>(val);   // row_shl:8
        const bool mine = lane < G && val == xm;
        double km = mine ? idxl : 4294967295.0;
        km = vmin_f64(km, dpp_f64<0xB1>(km));
        km = vmin_f64(km, dpp_f64<0x4E>(km));
        km = vmin_f64(km, dpp_f64<0x141>(km));
        const unsigned int mk = (unsigned int)(__ballot(mine && idxl == km) & 0xFFull);
        const int gw = mk ? __builtin_ctz(mk) : 0;
        const double bm = readlane_f64(xm, 0), bi = readlane_f64(km, 0);
        XWin r;
        r.m = bm;
        r.i = (unsigned int)bi;
        r.p0 = readlane_f64(val, 16 + gw);
        r.p1 = readlane_f64(val, 24 + gw);
        r.p2 = readlane_f64(val, 32 + gw);
        r.p3 = readlane_f64(val, 48 + gw);
        if (first) {   // slot 5 of every record: the XCC the workgroup runs on
            fast = !dead && __all(!(act && (lane >> 3) == 5) || val == (double)myxcc);
            first = false;
        }
        stamp(which * 5 + 4);
        return r;
    };
    // entering column: (min, q); payload r_q, the entering variable
    auto reduce_cols = [&](const double (&val)[RI]) -> XWin {
        const BtWinG w = wave_first_min(val);
#pragma unroll
        for (int s = 0; s < RI; s++)
            if ((unsigned int)gidx(s) == w.i) {
                pay[wv][0] = rv[s];
                pay[wv][1] = (double)nbasv[s];
                pay[wv][3] = vreg[s][0];   // the newest v' of this column (slots 6 / 7 of the record)
            }
        if (lane == 0) { redM[wv] = w.m; redI[wv] = w.i; }
        return xchg(0);
    };
    // leaving row: (min, p); payload d_p, x_B[p], the leaving variable
    auto reduce_rows = [&](const double (&val)[RI], const double (&dcol)[RI]) -> XWin {
        const BtWinG w = wave_first_min(val);
#pragma unroll
        for (int s = 0; s < RI; s++)
            if ((unsigned int)gidx(s) == w.i) {
                pay[wv][0] = dcol[s];
                pay[wv][1] = xbv[s];
                pay[wv][2] = (double)basv[s];
                pay[wv][3] = ureg[s][0];   // the newest u of this row
            }
        if (lane == 0) { redM[wv] = w.m; redI[wv] = w.i; }
        return xchg(1);
    };
    // column q of the current tableau for this thread's rows; v'_j[q] of the block's k earlier pivots (newest first) from V
    auto column = [&](int q, int k, double (&dcol)[RI], double newest) {
        double d0[RI];
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = gidx(s);
            const unsigned int ic = (unsigned int)(i < a.m ? i : a.m - 1);
            d0[s] = ldT(tile_off_g(ic, (unsigned int)q, ldt));
        }
        double vq[KR];
        if constexpr (LOOP) {
            // lane l fetches term l (newest first: this block's k terms, then the nl lagging ones): ONE load instruction
            // instead of 16 with scalar addresses each; the terms reach the multiply-adds as scalar operands
            const int l = lane & 31;   // (KR <= 32; lanes beyond the terms in use fetch nothing)
            const int trow = l < k ? cur0 + k - 1 - l : lag0 + nl - 1 - (l - k);
            double tv = l < k + nl ? ld_term(a.V + (size_t)trow * a.ldt + q) : 0.0;
            if (l == 0 && k + nl > 0) tv = newest;   // (its store may still be in flight: the owner sent it with the record)
            if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(11); }   // column + term loads: issue -> data
#pragma unroll
            for (int jj = 0; jj < KR; jj++) vq[jj] = readlane_f64(tv, jj);
        } else {
#pragma unroll
            for (int jj = 0; jj < KR; jj++) vq[jj] = jj < k ? ld_term(a.V + (size_t)(k - 1 - jj) * a.ldt + q) : 0.0;
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            double d;
            if constexpr (LOOP) {   // four partial sums, the tableau entry added last: the multiply-adds run under its load latency
                double acc[4] = {0, 0, 0, 0};
#pragma unroll
                for (int j = 0; j < KR; j++) acc[j & 3] = __builtin_fma(ureg[s][j], vq[j], acc[j & 3]);
                d = d0[s] + ((acc[0] + acc[1]) + (acc[2] + acc[3]));
            } else {
                d = d0[s];
#pragma unroll
                for (int j = 0; j < KR; j++) d = __builtin_fma(ureg[s][j], vq[j], d);
            }
            dcol[s] = gidx(s) < a.m ? d : 0.0;
        }
    };
    // ratio vector (simplex.go:321-340), branch-free as in k_bt_inner2
    auto ratios = [&](const double (&dcol)[RI], double (&mvv)[RI]) {
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = gidx(s);
            double d = -dcol[s];
            if (fabs(d) < 1e-13) d = 0;
            const double quot = div_pos(xbv[s], fabs(d));
            mvv[s] = (d >= 0 || i >= a.m) ? inf : quot;
        }
    };

    int nbe = 0;   // loop mode: blocks run by this launch
    for (int blk = 0; blk < (LOOP ? a.nblocks : 1); blk++) {
    if constexpr (LOOP) {
        // block blk reads the tableau after blk - 1 blocks: the update of block blk - 2 must be through (nupd arrivals per block)
        if (blk >= 2) {   // (one verdict per workgroup: a wave that gave up alone would leave the others at a barrier)
            if (wv == 0) {
                const int cj = (blk - 2) & 3;
                const unsigned int ub = cj == 0 ? upd_base[0] : cj == 1 ? upd_base[1] : cj == 2 ? upd_base[2] : upd_base[3];
                const bool ok = spin_counter(upd_cnt + 16 * cj * 2, ub + (unsigned int)nupd * (unsigned int)((blk - 2) / 4 + 1), 0);
                if (lane == 0) s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (!s_ok) { dead = true; break; }
        }
        Tb = reinterpret_cast<const char *>(((sel0 ^ (blk > 0 ? blk - 1 : 0)) & 1) ? a.Tbuf[1] : a.Tbuf[0]);   // (no dynamic index into the argument block: scratch)
        cur0 = (blk & 1) * KB; lag0 = cur0 ^ KB; nl = blk > 0 ? KB : 0;
        if (blk > 0) {   // entries 0..KB-1 = the previous block (now lagging), KB.. = the block before it: in the tableau by now
#pragma unroll
            for (int s = 0; s < RI; s++) {
#pragma unroll
                for (int j = KB; j < KR; j++) { ureg[s][j] = 0; vreg[s][j] = 0; }
            }
        }
        kd = 0;
    }
    for (int k = 0; k < a.kmax; k++) {
        stamp(-1);
        const bool forced = (k == 0 && blk == 0 && a.forced_q >= 0);
        int q, p, ent = 0, lea = 0;
        double rq = 0, dpv = 1.0, xbp = 0, unew = 0;   // (unew: the newest u term of row p, out of the winner's record)
        bool bland = false;
        double dcol[RI];
        if (!forced) {
            const XWin fq = reduce_cols(rv);
            if (dead) break;
            q = (int)fq.i; rq = fq.p0; ent = (int)fq.p1;
            if (fq.i >= (unsigned int)a.nn) { q = 0; rq = __builtin_nan(""); }   // every r_j is NaN: MinIdx returns 0
            if (a.guard == inf && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // strict mode (knob exact_degenerate = 3): every decision is the host's, on fresh solves
            if (rq >= -a.tol) { status = ST_OPTIMAL; break; }                    // simplex.go:248
            column(q, k, dcol, fq.p3);
            XWin w;
            {
                double mvv[RI];
                ratios(dcol, mvv);
                w = reduce_rows(mvv, dcol);
            }
            if (dead) break;
            p = (int)w.i; dpv = w.p0; xbp = w.p1; lea = (int)w.p2; unew = w.p3;
            const double mv = w.m;
            if (mv == inf || w.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // simplex.go:328-330
            if (a.guard > 0 && (mv <= a.guard || fabs(dpv) <= a.guard) && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // degenerate (or nearly): decided on a fresh x_B
            if (a.cguard > 0 && fabs(dpv) <= a.cguard && !(k == 0 && blk == 0 && a.exact_once)) { status = ST_NEED_EXACT; break; }   // (BTArgs::cguard)
            if (mv <= 0) {
                // replaceBland (simplex.go:347-383), as in k_bt_inner2: candidates in position order
                bland = true;
                blands++;
                int cand = -1;
                bool found = false;
                for (;;) {
                    double fl[RI];
#pragma unroll
                    for (int s = 0; s < RI; s++) {
                        const int j = gidx(s);
                        double rr = rv[s];
                        if (fabs(rr) < 1e-13) rr = 0;
                        fl[s] = (j < a.nn && j > cand && !(rr > -1e-14)) ? 0.0 : inf;
                    }
                    const XWin fc = reduce_cols(fl);
                    if (dead) break;
                    if (fc.m != 0.0) break;   // candidates exhausted -> ErrBland
                    cand = (int)fc.i;
                    const double rqc = fc.p0;
                    const int entc = (int)fc.p1;
                    column(cand, k, dcol, fc.p3);
                    XWin w2;
                    {
                        double mvv[RI];
                        ratios(dcol, mvv);
                        w2 = reduce_rows(mvv, dcol);
                    }
                    if (dead) break;
                    if (w2.m == inf || w2.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // :356-360
                    if (fabs(w2.m) > 1e-12) {   // :362
                        q = cand; p = (int)w2.i; rq = rqc; ent = entc; dpv = w2.p0; xbp = w2.p1; lea = (int)w2.p2; unew = w2.p3;
                        found = true;
                        break;
                    }
                    double gl2[RI];
                    ratios(dcol, gl2);
#pragma unroll
                    for (int s = 0; s < RI; s++) gl2[s] = (gidx(s) < a.m && !(gl2[s] > 1e-12)) ? 0.0 : inf;
                    const XWin gw = reduce_rows(gl2, dcol);
                    if (dead) break;
                    if (gw.m == 0.0) {   // :368-379
                        q = cand; p = (int)gw.i; rq = rqc; ent = entc; dpv = gw.p0; xbp = gw.p1; lea = (int)gw.p2; unew = gw.p3;
                        found = true;
                        break;
                    }
                }
                if (dead || status == ST_UNBOUNDED) break;
                if (!found) { status = ST_BLAND_FAILED; break; }
            }
        } else {
            // set-up pivot chosen by the host (first pivot of a block: all block terms are zero)
            q = a.forced_q; p = a.forced_p; rq = 0;   // (a set-up pivot leaves the reduced costs alone: they are rebuilt)
            double fl[RI];
#pragma unroll
            for (int s = 0; s < RI; s++) fl[s] = (gidx(s) == q) ? 0.0 : inf;
            const XWin fc = reduce_cols(fl);
            if (dead) break;
            ent = (int)fc.p1;
            if (!a.forced_nocommit) rq = fc.p0;   // a pivot the host decided on fresh solves (exact_step): a pivot like any other
            column(q, k, dcol, fc.p3);
            double gl2[RI];
#pragma unroll
            for (int s = 0; s < RI; s++) gl2[s] = (gidx(s) == p) ? 0.0 : inf;
            const XWin gw = reduce_rows(gl2, dcol);
            if (dead) break;
            dpv = gw.p0; xbp = gw.p1; lea = (int)gw.p2; unew = gw.p3;
        }
        // ---- row p for this thread's columns, reduced costs, block terms (formulas of k_bt_inner2)
        const double rinv = 1.0 / dpv, nrinv = -rinv;
        const double mult = rq * rinv;
        const double theta = xbp * rinv;
        double *Vk = a.V + (size_t)(cur0 + k) * a.ldt;
        double *Uk = a.U + (size_t)(cur0 + k) * a.ldu;
        const bool commit_lists = !(forced && a.forced_nocommit) || (forced && a.forced_nocommit >= 2);
        double vrow[RI];
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int j = gidx(s);
            vrow[s] = j < a.ldt ? ldT(tile_off_g((unsigned int)p, (unsigned int)j, ldt)) : 0.0;
        }
        double up[KR];   // u_j[p] of the block's k earlier pivots, newest first (loop mode: then the lagging ones)
        if constexpr (LOOP) {
            const int l = lane & 31;   // (KR <= 32; lanes beyond the terms in use fetch nothing)
            const int trow = l < k ? cur0 + k - 1 - l : lag0 + nl - 1 - (l - k);
            double tv = l < k + nl ? ld_term(a.U + (size_t)trow * a.ldu + p) : 0.0;
            if (l == 0 && k + nl > 0) tv = unew;
            if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(12); }   // row + term loads: issue -> data
#pragma unroll
            for (int jj = 0; jj < KR; jj++) up[jj] = readlane_f64(tv, jj);
        } else {
#pragma unroll
            for (int jj = 0; jj < KR; jj++) up[jj] = jj < k ? ld_term(a.U + (size_t)(k - 1 - jj) * a.ldu + p) : 0.0;
        }
        // loop mode: the previous pivot's term stores have landed before this pivot's are issued — by this pivot's posts (which carry
        // only the newest term) every older term is in U / V for the readers; in practice they landed during the exchanges
        if constexpr (LOOP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int i = gidx(s);
            if (i < a.ldu) {
                const double u = (i == p) ? rinv - 1.0 : dcol[s] * nrinv;
                if (i < a.m) xbv[s] = (i == p) ? theta : __builtin_fma(-theta, dcol[s], xbv[s]);
                st_term(Uk + i, u);
#pragma unroll
                for (int jj = KR - 1; jj > 0; jj--) ureg[s][jj] = ureg[s][jj - 1];
                ureg[s][0] = u;
                if (i == p && commit_lists) basv[s] = ent;
            }
        }
#pragma unroll
        for (int s = 0; s < RI; s++) {
            const int j = gidx(s);
            if (j < a.ldt) {
                double v;
                if constexpr (LOOP) {
                    double acc[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int jj = 0; jj < KR; jj++) acc[jj & 3] = __builtin_fma(up[jj], vreg[s][jj], acc[jj & 3]);
                    v = vrow[s] + ((acc[0] + acc[1]) + (acc[2] + acc[3]));
                } else {
                    v = vrow[s];
#pragma unroll
                    for (int jj = 0; jj < KR; jj++) v = __builtin_fma(up[jj], vreg[s][jj], v);
                }
                rv[s] = (j == q) ? -mult : __builtin_fma(-mult, v, rv[s]);
                const double vprime = (j == q) ? dpv + 1.0 : v;
                st_term(Vk + j, vprime);
#pragma unroll
                for (int jj = KR - 1; jj > 0; jj--) vreg[s][jj] = vreg[s][jj - 1];
                vreg[s][0] = vprime;
                if (j == q && commit_lists) nbasv[s] = lea;
            }
        }
        if constexpr (STAMP && !LOOP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    if constexpr (LOOP) {
        // hand the block to the update workgroups: every term store of this workgroup has landed (agent scope), workgroup 0
        // publishes the pivot count (and the end of the loop) BEFORE its arrival, the arrivals of all G workgroups release them
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
    }
    if (dead) status = ST_XCHG_TIMEOUT;
    if constexpr (STAMP) {
        if (a.stamps && g == 0 && lane == 0) {
            for (int sg = 0; sg < 16; sg++) a.stamps[wv * 16 + sg] += s_acc[wv][sg];
            if (wv == 0) a.stamps[16 * 16] += (unsigned long long)kd;
        }
    }
#pragma unroll
    for (int s = 0; s < RI; s++) {
        const int i = gidx(s);
        if (i < a.ldt) a.r[i] = i < a.nn ? rv[s] : 0.0;
        if (i < a.ldu) a.xb[i] = xbv[s];
        if (i < a.m) a.basic[i] = basv[s];
        if (i < a.nn) a.nonbasic[i] = nbasv[s];
    }
    if (tid == 0 && (g == 0 || dead)) {
        if (g == 0) {
            st->trace_len = trace_len;
            st->pivots = npiv;
            if constexpr (LOOP) {
                // the update workgroups apply every block with pivots before the launch ends: the tableau after them
                const int napplied = kd > 0 ? nbe : nbe - 1;
                st->tsel2[a.par ^ 1] = sel0 ^ (napplied & 1);
                st->loop_blocks = nbe;
                hand_on(nbe);
            } else {
                st->kdone = kd;
            }
            st->bland_steps += blands;
            a.xbuf[0] = (double)xs;
        }
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

template <int G, int NT, int RI, int KR, bool STAMP = false>
__global__ __launch_bounds__(NT) void k_bt_innerG(BTArgs a) {
    if (blockIdx.x & 7u) return;   // blocks 0, 8, 16, ...: all on XCD 0
    bt_innerG_body<G, NT, RI, KR, STAMP>(a, (int)(blockIdx.x >> 3));
}
// ---- persistent loop kernel (update role: bt_loop.h) ---------------------------------------------------------------------------------
// Blocks 0, 8, ..., 8 (G - 1) — one XCD under the round-robin placement of blocks — are the pivot workgroups, every other block
// of the grid an update workgroup.  All workgroups of the launch must be resident (they wait for each other): the grid is one
// workgroup per CU (launch_bt_loop), and every wait is bounded.
template <int G, int NT, int RI, bool STAMP = false, int KB = 8>
__global__ __launch_bounds__(NT) void k_bt_loop(BTArgs a) {
    const int b = (int)blockIdx.x, x = a.xcd & 7;
    // update workgroups that take part (knob "loop_upd": fewer of them spread a block's traffic over more of the block time)
    const int nupd = (a.upd_cap > 0 && a.upd_cap < (int)gridDim.x - G) ? a.upd_cap : (int)gridDim.x - G;
    if ((b & 7) == x && (b >> 3) < G) { bt_innerG_body<G, NT, RI, 2 * KB, STAMP, true>(a, b >> 3, nupd); return; }
    const int before = b <= x ? 0 : min(G, ((b - x - 1) >> 3) + 1);   // pivot blocks in front of block b
    if (b - before >= nupd) return;
    bt_loop_update_role<NT, KB>(a, b - before, nupd, G);
}

// Batched form (device-batched waves of large relaxations, engine_batch.cpp): the relaxation at position p of the active list
// runs on XCD p % 8 — block b = x + 8 j serves position (j / G) * 8 + x as its workgroup j % G — so up to 8 relaxations
// advance at once, each inside one L2.  Blocks are dispatched in index order and all G workgroups of a relaxation lie in one
// run of 8 G blocks, so a relaxation whose first workgroup is resident gets the others as soon as slots free up.
template <int G, int NT, int RI, int KR>
__global__ __launch_bounds__(NT) void k_bt_innerG_batch(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, int xcd_off) {
    // (xcd_off: list position 0 runs on that XCD — several schedules side by side keep their block kernels apart)
    const unsigned int x8 = ((blockIdx.x & 7u) + 8u - (unsigned int)xcd_off) & 7u, j = blockIdx.x >> 3;
    const int lpos = (int)((j / G) * 8u + x8);
    if (lpos >= *count) return;
    const BatchLP &lp = lps[ids[lpos]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST) return;
    const BTArgs a = lp.bt;
    bt_innerG_body<G, NT, RI, KR, false>(a, (int)(j % G));
}

// Tried and dropped (round 3, loop mode): one record per WAVE (no LDS stage, no barrier: 32 records at 2048 rows, 64 at 4096, three
// resp. six 16-byte loads per lane and poll).  Same pivots, but 35.1 instead of 32.3 us per 8 pivots at 2048 rows and 114 instead
// of 52.6 at 4096: what the workgroup stage costs (stamps: ~1000 cycles in wave 0) the wider poll costs again (1600 instead of
// 760 cycles from post to "all seen").
// Tried and dropped (round 5, loop mode): 16 pivot workgroups of ONE wave, two rows / columns per thread (k_bt_loop<16,64,2,8>: no workgroup
// stage and no real barrier in front of a post).  Same pivots, 4.45-4.68 instead of 3.87 us per pivot at 2048 rows (grid 256 / 136): the
// wave's doubled row / column work costs more than the stage saved.
// Tried and dropped (round 2): every participant ONE wave on its own CU (no barrier, no LDS hop; terms in LDS by position,
// prefetched under the load latency).  Bit-identical results, but slower at 2048 rows: 8 waves x 4 rows 107.7 us per 16
// pivots, 16 waves x 2 rows 97.0 us, against 80.2 us for 4 workgroups of 8 waves: one wave issues its ~1500 instructions
// per pivot alone on a SIMD, and an exchange among 16 participants waits for the slowest of 16.

// ---- host side ---------------------------------------------------------------------------------
size_t bt_xbuf_doubles() { return (size_t)kXSync + kXSyncDoubles; }

// workgroups for a shape (knob "bt_groups": -1 = never, 0 = by shape, 2 / 4 / 8 = forced where the shape fits)
BtGroupCfg bt_group_cfg(int m, int ldt, int knob) {
    BtGroupCfg c = {0, 0, 0};
    if (knob < 0) return c;
    const int need = m > ldt ? m : ldt;
    auto fits = [&](int G, int nt, int ri) { return (long)G * nt * ri >= need; };
    if (knob == 16) {   // the loop-kernel shapes also below their default range (1024 rows and less: measurement knob)
        if (need <= 2048) c = {8, 256, 1};
        else if (need <= 4096) c = {8, 512, 1};
        return c;
    }
    if (knob == 2 || knob == 4 || knob == 8) {
        const int G = knob;
        if (fits(G, 512, 1)) c = {G, 512, 1};
        else if (fits(G, 512, 2)) c = {G, 512, 2};
        return c;
    }
    if (knob != 0) return c;
    // measured per 16 pivots at 2048 x 2048 (MI355X): 8 workgroups x 256 threads 70.9 us, 4 x 512 80.2, 4 x 256 x 2 rows 75.9,
    // 8 x 128 x 2 rows 87.1, 2 x 512 x 2 rows 91.2 — against 2 x 37.2 us of k_bt_inner2<512,4,4,8> and a rank-8 update more
    // per 16 pivots; at 4096 x 4096: 8 x 512 83.9 us, 8 x 256 x 2 rows 95.1.  One row + one column per thread, 8 workgroups.
    if (need <= 1024) return c;   // one workgroup holds these (k_bt_inner2<512,2,2,8>: 3.1 us per pivot; 8 workgroups x 128 threads: 4.1)
    if (need <= 2048) return {8, 256, 1};
    if (need <= 4096) return {8, 512, 1};
    if (need <= 8192) return {8, 512, 2};
    return c;
}

template <int G, int NT, int RI>
static void btg_launch(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if constexpr (RI == 1) {
        if (a.stamps) { hipExtLaunchKernelGGL((k_bt_innerG<G, NT, RI, 16, true>), dim3(8 * G), dim3(NT), 0, s, e0, e1, 0, a); return; }
    }
    hipExtLaunchKernelGGL((k_bt_innerG<G, NT, RI, 16>), dim3(8 * G), dim3(NT), 0, s, e0, e1, 0, a);
}
void launch_bt_inner_groups(const BTArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int G = a.groups, ri = a.group_ri;
    if (a.group_nt == 256) { btg_launch<8, 256, 1>(a, s, e0, e1); return; }
    if (G == 2) { if (ri == 1) btg_launch<2, 512, 1>(a, s, e0, e1); else btg_launch<2, 512, 2>(a, s, e0, e1); }
    else if (G == 4) { if (ri == 1) btg_launch<4, 512, 1>(a, s, e0, e1); else btg_launch<4, 512, 2>(a, s, e0, e1); }
    else { if (ri == 1) btg_launch<8, 512, 1>(a, s, e0, e1); else btg_launch<8, 512, 2>(a, s, e0, e1); }
}
// persistent loop kernel: one workgroup per CU (the pivot workgroups among them); only the one-row-per-thread instances
bool bt_loop_supported(const BtGroupCfg &c) { return c.groups >= 2 && c.ri == 1; }
void launch_bt_loop(const BTArgs &a, int ncu, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int G = a.groups;
    const unsigned int grid = (unsigned int)std::max(ncu, 8 * G);
    if (a.group_nt == 256 && G == 8) {
        if (a.stamps) hipExtLaunchKernelGGL((k_bt_loop<8, 256, 1, true>), dim3(grid), dim3(256), 0, s, e0, e1, 0, a);   // diagnostic build
        else hipExtLaunchKernelGGL((k_bt_loop<8, 256, 1>), dim3(grid), dim3(256), 0, s, e0, e1, 0, a);
        return;
    }
    if (G == 16 && a.group_nt == 128) { hipExtLaunchKernelGGL((k_bt_loop<16, 128, 1, false, 8>), dim3(std::max(grid, 136u)), dim3(128), 0, s, e0, e1, 0, a); return; }   // up to 2048 rows: 16 x 128 threads, blocks of 8
    if (G == 16 && a.kmax == 12) { hipExtLaunchKernelGGL((k_bt_loop<16, 256, 1, false, 12>), dim3(std::max(grid, 136u)), dim3(256), 0, s, e0, e1, 0, a); return; }   // knob loop_k = 12
    if (G == 16) { hipExtLaunchKernelGGL((k_bt_loop<16, 256, 1, false, 16>), dim3(std::max(grid, 128u)), dim3(256), 0, s, e0, e1, 0, a); return; }   // 4096 rows: 16 x 256 threads, blocks of 16 (pivot blocks: x + 8 j, j < 16, x <= 6)
    if (G == 2) hipExtLaunchKernelGGL((k_bt_loop<2, 512, 1>), dim3(grid), dim3(512), 0, s, e0, e1, 0, a);
    else if (G == 4) hipExtLaunchKernelGGL((k_bt_loop<4, 512, 1>), dim3(grid), dim3(512), 0, s, e0, e1, 0, a);
    else if (a.kmax == 16) hipExtLaunchKernelGGL((k_bt_loop<8, 512, 1, false, 16>), dim3(grid), dim3(512), 0, s, e0, e1, 0, a);   // blocks of 16 pivots
    else hipExtLaunchKernelGGL((k_bt_loop<8, 512, 1>), dim3(grid), dim3(512), 0, s, e0, e1, 0, a);
}
// batched launch: the whole wave has the shape class of its largest relaxation (8 workgroups; 256 or 512 threads)
void launch_bt_inner_groups_batch(const BatchLP *lps, const int *ids, const int *count, int nlp, const BtGroupCfg &c, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int xcd_off) {
    const unsigned int grid = 64u * (unsigned int)((nlp + 7) / 8);
    if (c.nt == 256) hipExtLaunchKernelGGL((k_bt_innerG_batch<8, 256, 1, 16>), dim3(grid), dim3(256), 0, s, e0, e1, 0, lps, ids, count, xcd_off);
    else hipExtLaunchKernelGGL((k_bt_innerG_batch<8, 512, 1, 16>), dim3(grid), dim3(512), 0, s, e0, e1, 0, lps, ids, count, xcd_off);
}
const char *bt_group_kernel_name(int G, int ri) {   // (512-thread instances)
    static const char *names[3][2] = {{"k_bt_innerG<2,512,1,16>", "k_bt_innerG<2,512,2,16>"}, {"k_bt_innerG<4,512,1,16>", "k_bt_innerG<4,512,2,16>"},
                                      {"k_bt_innerG<8,512,1,16>", "k_bt_innerG<8,512,2,16>"}};
    return names[G == 2 ? 0 : G == 4 ? 1 : 2][ri == 1 ? 0 : 1];
}

}  // namespace gomilp
