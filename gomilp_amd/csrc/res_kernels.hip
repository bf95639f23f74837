// Register-resident tableau pivoting for relaxations of up to 640 rows (gfx950) — the long chains of a device-batched wave.
//
// Every block kernel of bt_kernels.hip / btg_kernels.hip reads one column and one row of a tableau that lives in HBM / L2, corrects
// them with the rank-1 terms that are not applied yet, and leaves the streaming update to other workgroups: two dependent memory
// round trips, 4 + 4 lagging terms and (over several workgroups) two exchanges per pivot — 4.5 us per pivot on a 520 x 512 tableau
// of 2.1 MB.  Such a tableau fits the REGISTER FILES of a few CUs (512 KB of VGPRs each).  Here a relaxation is G workgroups of
// 512 threads on one XCD:
//   * workgroup g owns the rows [g R, (g + 1) R) (R = rows per workgroup, a multiple of 4, <= 40); thread t holds column t (and
//     t + 512) of those rows in registers: 40 doubles per column, read with a uniform dynamic index (v_movrels) when a row leaves;
//   * the reduced costs r and the nonbasic list are REPLICATED in every workgroup (thread t: its columns), updated with the same
//     instructions from the same operands — the entering column (floats.MinIdx over r, simplex.go:247) needs NO exchange;
//   * x_B and the basic list of the workgroup's rows sit in lanes 0 .. R-1 of EVERY wave (each wave repeats the ratio test of
//     simplex.go:321-340 for the workgroup's rows: 40 quotients — cheaper than a barrier);
//   * one barrier per column selection: every wave dumps the column of ITS OWN best candidate to LDS next to (value, index), then
//     everybody reads the winner's;
//   * ONE exchange through the XCD's L2 per ratio test: every workgroup posts {min ratio, row, d_p, x_B[p], leaving variable,
//     runner-up, and the Bland rule's first zero-level row} in one 128-byte record — and, speculatively, the tableau ROW of its own
//     candidate (one 16-byte granule {sequence number, value} per column); everybody polls the G records, takes the
//     lexicographic minimum (value, index) = floats.MinIdx over the whole ratio vector, and reads the winner's row: the rank-1 update
//     T += u v'^T is applied in registers at once.  No ping-pong pair, no update role, no lagging terms.
// Decisions are those of bt_inner2_body (bt_kernels.hip) line by line: Dantzig rule, stop test, 1e-13 roundings, unbounded test,
// the guards that hand a pivot to the host (ST_NEED_EXACT), replaceBland (simplex.go:347-383) with its 1e-14 / 1e-12 tolerances,
// host-chosen first pivots.  The running quantities (T, r, x_B) are the engine's own (fused multiply-adds, one reciprocal per
// pivot); the returned x comes from the gonum-order solve of the final basis as everywhere.
// Everything that crosses workgroups carries the sequence number of its exchange in the SAME 16-byte store as the value (a reader
// can only consume what it has seen arrive), is double buffered by exchange parity (a workgroup can be at most one exchange ahead of
// the slowest), and every wait is bounded (ST_XCHG_TIMEOUT -> the schedule hands the relaxation to a worker).  The first exchange of
// a launch uses agent-scope accesses and carries the XCC ids; if they agree the rest uses the same-XCD forms (bt_loop.h xstore).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "device_types.h"
#include "kernels_common.h"
#include "bt_loop.h"

namespace gomilp {

constexpr int kResNT = 512;          // threads per workgroup
constexpr int kResRMax = 40;         // rows per workgroup
constexpr int kResGMax = 16;         // workgroups per relaxation (records per exchange)
constexpr int kResCols = 1024;       // columns a slot's row buffers are laid out for
constexpr int kResSpin = 400000;     // polls (~0.3 us each) before a workgroup gives up
constexpr unsigned int kResNone = 0xFFFFFFFFu;
// slot buffer (xpairs): records [parity][kResGMax][kXSlots], then candidate rows [parity][kResGMax][kResCols]
constexpr size_t kResRecPairs = (size_t)2 * kResGMax * kXSlots;
constexpr size_t kResSlotPairs = kResRecPairs + (size_t)2 * kResGMax * kResCols;

typedef double rvec8 __attribute__((ext_vector_type(8)));
typedef double rvec2 __attribute__((ext_vector_type(2)));

namespace {

struct RWin { double m; unsigned int i; };
__device__ __forceinline__ double rpack(unsigned int lo, unsigned int hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ unsigned int rlo(double v) { return (unsigned int)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned int rhi(double v) { return (unsigned int)((unsigned long long)__double_as_longlong(v) >> 32); }

// element l (uniform) of a thread's register column (kResRMax / 8 = 5 tuples of 8 doubles), the tuple by a uniform branch, the element by v_movrels
// (the tuples come BY VALUE: through a reference the element access is canonicalised into a scalar load at a dynamic address before the
// function is inlined, and the whole array stays in scratch; the asm statements keep the branches from being merged into one access)
__device__ __forceinline__ double res_rowval(const rvec8 t0, const rvec8 t1, const rvec8 t2, const rvec8 t3, const rvec8 t4, int l) {
    const int t = __builtin_amdgcn_readfirstlane(l >> 3);
    int e = __builtin_amdgcn_readfirstlane(l & 7);
    asm volatile("" : "+s"(e));   // (an index the compiler cannot bound: otherwise a tuple that is loaded for this one use is turned into a scalar load at a dynamic address)
    double v;
    if (t == 0) { v = t0[e]; asm volatile("" : "+v"(v)); }
    else if (t == 1) { v = t1[e]; asm volatile("" : "+v"(v)); }
    else if (t == 2) { v = t2[e]; asm volatile("" : "+v"(v)); }
    else if (t == 3) { v = t3[e]; asm volatile("" : "+v"(v)); }
    else { v = t4[e]; asm volatile("" : "+v"(v)); }
    return v;
}

// what a workgroup knows about its own rows after a ratio test
struct ResLocal {
    double m;          // min ratio (+Inf: no candidate)
    int l;             // its local row (-1: none)
    double dp, xp;     // pivot element, x_B of that row
    int lea;           // its basic variable
    double ru;         // runner-up ratio (guard mode)
    int lb;            // Bland rule: first local row with ratio <= 1e-12 (-1: none)
    double dpB, xpB;
    int leaB;
};
// outcome of an exchange (uniform in every wave of every workgroup)
struct ResWin {
    double m; unsigned int i; int gw; double dp, xp; int lea; double mv2;
    unsigned int posted;      // row the winner workgroup posted speculatively
    unsigned int bi; int gb; double dpB, xpB; int leaB; unsigned int postedB;   // Bland rule: first zero-level row over all workgroups
};

template <int CJ>
__device__ __forceinline__ void res_body(const BTArgs &a, const int g, const int G, const int npiv, const double seq0, xpair *__restrict__ xslot) {
    constexpr int NT = kResNT, NW = NT / 64, RM = kResRMax, NV = RM / 8;
    __shared__ __attribute__((aligned(16))) double colbuf[2][NW][RM];
    __shared__ __attribute__((aligned(16))) double ubuf[NW][RM];
    __shared__ double redM[2][NW], payR[2][NW], red2[NW];
    __shared__ unsigned int redI[2][NW];
    __shared__ int payN[2][NW];
    DevState *st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const double inf = __builtin_inf();
    const unsigned int ldt = (unsigned int)a.ldt;
    const int done = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int R = 4 * ((a.m + 4 * G - 1) / (4 * G)), row0 = g * R;   // (the host launches this kernel only where R <= kResRMax)
    xpair *recs = xslot, *rows = xslot + kResRecPairs;
    unsigned int myxcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
    myxcc &= 0xFu;
    if (done) {
        if (g == 0 && tid == 0) st->kdone = 0;
        return;
    }
    // ---- state: the slab of the tableau, r and the nonbasic list by column, x_B and the basic list by row (lane l of every wave)
    rvec8 T[CJ][NV];
    double r[CJ];
    int nbv[CJ];
    const double *Tg = a.T;
#pragma unroll
    for (int s = 0; s < CJ; s++) {
        const int j = tid + s * NT;
        r[s] = j < a.nn ? a.r[j] : inf;   // padding never wins an argmin
        nbv[s] = j < a.nn ? a.nonbasic[j] : 0;
#pragma unroll
        for (int t = 0; t < NV; t++) {
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const int l = t * 8 + e, i = row0 + l;
                T[s][t][e] = (l < R && i < a.m && (unsigned int)j < ldt) ? Tg[tile_off_g((unsigned int)i, (unsigned int)j, ldt)] : 0.0;
            }
        }
    }
    const int irow = row0 + lane;
    const bool rvalid = lane < R && irow < a.m;
    double xbl = rvalid ? a.xb[irow] : 0.0;
    int basl = rvalid ? a.basic[irow] : 0;
    long long trace_len = 0, npv = 0;
    if (g == 0 && tid == 0) { trace_len = st->trace_len; npv = st->pivots; }
    int kd = 0, status = ST_RUNNING, blands = 0;
    bool dead = false, fast = false, first = true;
    int selpar = 0;
    long long ex = 0;   // exchanges of this launch

    auto wave_first_min = [&](const double (&val)[CJ]) -> RWin {
        double x = val[0];
#pragma unroll
        for (int s = 1; s < CJ; s++) x = vmin_f64(x, val[s]);
        RWin w;
        w.m = wave_min_f64(x);
        w.i = kResNone;
#pragma unroll
        for (int s = CJ - 1; s >= 0; s--) {
            const unsigned long long mask = __ballot(val[s] == w.m);
            if (mask) w.i = (unsigned int)(s * NT + wbase + __builtin_ctzll(mask));
        }
        return w;
    };
    // Column selection: first index of the minimum of val over all columns (every workgroup computes the same).  Each wave dumps the
    // slab of ITS candidate column to LDS beside (value, index, r_j, variable); one barrier; everybody reads the winner's.  The
    // buffers alternate: a wave can reach its next selection while a slower one still reads this one's.
    auto select = [&](const double (&val)[CJ], double &rq, int &ent, const double *&dcolp) -> RWin {
        selpar ^= 1;
        const RWin w = wave_first_min(val);
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            if ((unsigned int)(tid + s * NT) == w.i) {
                payR[selpar][wv] = r[s];
                payN[selpar][wv] = nbv[s];
                rvec2 *dst = reinterpret_cast<rvec2 *>(&colbuf[selpar][wv][0]);
#pragma unroll
                for (int t = 0; t < NV; t++) {
                    if (t * 8 < R) {
#pragma unroll
                        for (int h = 0; h < 4; h++) dst[t * 4 + h] = rvec2{T[s][t][2 * h], T[s][t][2 * h + 1]};
                    }
                }
            }
        }
        if (lane == 0) { redM[selpar][wv] = w.m; redI[selpar][wv] = w.i; }
        __syncthreads();
        const double x = lane < NW ? redM[selpar][lane] : inf;
        const unsigned int ii = lane < NW ? redI[selpar][lane] : kResNone;
        RWin f;
        f.m = readlane_f64(row_min_f64(x), 15);
        const unsigned int key = (x == f.m) ? ii : kResNone;
        f.i = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(key), 15);
        const int ww = (int)((f.i & (unsigned int)(NT - 1)) >> 6);
        rq = payR[selpar][ww];
        ent = payN[selpar][ww];
        dcolp = &colbuf[selpar][ww][0];
        return f;
    };
    // Ratio test of simplex.go:321-340 over this workgroup's rows, in every wave alike (lane l = local row l): d' = -d rounded at
    // 1e-13, move = x_B / |d'| where d' < 0, +Inf elsewhere.  forced_p >= 0: a host-chosen leaving row — only its owner bids
    auto ratio_stage = [&](const double *dcolp, const int forced_p, const bool want_ru, const bool want_b, double &dl) -> ResLocal {
        dl = (lane < RM) ? dcolp[lane < RM ? lane : 0] : 0.0;
        if (!rvalid) dl = 0.0;
        double dn = -dl;
        if (fabs(dn) < 1e-13) dn = 0;
        const double quot = div_pos(xbl, fabs(dn));   // == x_B / |d'| bit for bit; discarded where the reference does not divide
        double mv = (dn >= 0 || !rvalid) ? inf : quot;
        if (forced_p >= 0) mv = (rvalid && irow == forced_p) ? 0.0 : inf;
        ResLocal L;
        L.m = wave_min_f64(mv);
        const unsigned long long mask = __ballot(mv == L.m);
        const bool has = L.m < inf && mask != 0ull;   // (a NaN minimum: no candidate, like +Inf)
        L.l = has ? (int)__builtin_ctzll(mask) : -1;
        if (!has) L.m = inf;
        const int lr = L.l < 0 ? 0 : L.l;
        L.dp = readlane_f64(dl, lr);
        L.xp = readlane_f64(xbl, lr);
        L.lea = __builtin_amdgcn_readlane(basl, lr);
        L.ru = inf;
        if (want_ru) L.ru = wave_min_f64(lane == L.l ? inf : mv);
        L.lb = -1; L.dpB = 0; L.xpB = 0; L.leaB = 0;
        if (want_b) {
            const unsigned long long mb = __ballot(rvalid && !(mv > 1e-12));   // simplex.go:368-379: rows with move <= blandZeroTol, in order
            if (mb) {
                L.lb = (int)__builtin_ctzll(mb);
                L.dpB = readlane_f64(dl, L.lb);
                L.xpB = readlane_f64(xbl, L.lb);
                L.leaB = __builtin_amdgcn_readlane(basl, L.lb);
            }
        }
        return L;
    };
    // this workgroup's tableau row l into its row buffer of the given parity: one {sequence number, value} granule per column
    auto post_row = [&](const int l, const int par, const double seqd) {
        xpair *dst = rows + ((size_t)(par * kResGMax + g) * kResCols);
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            if ((unsigned int)j < ldt) {
                xpair v;
                v.x = seqd;
                v.y = res_rowval(T[s][0], T[s][1], T[s][2], T[s][3], T[s][4], l);
                xstore(dst + j, v, fast);
            }
        }
    };
    // The exchange: post this workgroup's record (wave 0) and its candidate row (every wave its columns), poll the G records.
    auto exchange = [&](const ResLocal &L, const bool blandx) -> ResWin {
        ex += 1;
        const double seqd = seq0 + (double)ex;
        const int par = (int)(ex & 1);
        const int prow = (blandx && L.lb >= 0) ? L.lb : L.l;
        if (prow >= 0) post_row(prow, par, seqd);
        if (wv == 0 && lane < kXSlots) {
            const unsigned int gi = L.l >= 0 ? (unsigned int)(row0 + L.l) : kResNone;
            const unsigned int gb = L.lb >= 0 ? (unsigned int)(row0 + L.lb) : kResNone;
            xpair v;
            v.x = seqd;
            v.y = lane == 0 ? L.m : lane == 1 ? rpack(gi, ((unsigned int)L.lea & 0xFFFFFFu) | (myxcc << 24)) : lane == 2 ? L.dp : lane == 3 ? L.xp : lane == 4 ? L.ru
                  : lane == 5 ? rpack(gb, (unsigned int)L.leaB & 0xFFFFFFu) : lane == 6 ? L.dpB : L.xpB;
            xstore(recs + ((size_t)(par * kResGMax + g) * kXSlots + lane), v, fast);
        }
        // lane l reads slots 2 (l >> 4) and 2 (l >> 4) + 1 of record l & 15: lanes 0..15 hold (min, index | variable) of record l,
        // lanes 16..31 (d_p, x_B[p]), lanes 32..47 (runner-up, Bland row | variable), lanes 48..63 (Bland d_p, x_B[p])
        const int rec = lane & 15, grp = lane >> 4;
        const bool act = rec < G;
        const xpair *base = recs + (size_t)(par * kResGMax + (act ? rec : 0)) * kXSlots + 2 * grp;
        xpair got[2];
        int spins = 0;
        for (;;) {
            XLoad<2>::run(base, base + 1, got, fast);
            if (__all(!act || (got[0].x == seqd && got[1].x == seqd))) break;
            if (++spins > kResSpin) { dead = true; break; }
        }
        ResWin W;
        const double v0 = got[0].y, v1 = got[1].y;
        const bool row0l = lane < 16 && act;
        const double xm = readlane_f64(row_min_f64(row0l ? v0 : inf), 0);
        const bool mine = row0l && v0 == xm;
        const unsigned int idxl = rlo(v1);
        const unsigned int kmin = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(mine ? idxl : kResNone), 0);
        const unsigned int mk = (unsigned int)(__ballot(mine && idxl == kmin) & 0xFFFFull);
        const int gw = mk ? __builtin_ctz(mk) : 0;
        W.m = mk ? xm : inf;
        W.i = mk ? kmin : kResNone;
        W.gw = gw;
        W.lea = (int)(__builtin_amdgcn_readlane((int)rhi(v1), gw) & 0xFFFFFF);
        W.dp = readlane_f64(v0, 16 + gw);
        W.xp = readlane_f64(v1, 16 + gw);
        const double rug = readlane_f64(v0, 32 + gw);
        W.mv2 = readlane_f64(row_min_f64(lane < 16 ? (lane == gw ? rug : (act ? v0 : inf)) : inf), 0);
        const unsigned int bgw = (unsigned int)__builtin_amdgcn_readlane((int)rlo(v1), 32 + gw);
        W.posted = (blandx && bgw != kResNone) ? bgw : W.i;
        // Bland rule: smallest zero-level row over the workgroups (lanes 32..47 hold each record's)
        W.bi = kResNone; W.gb = 0; W.dpB = 0; W.xpB = 0; W.leaB = 0; W.postedB = kResNone;
        if (blandx) {
            const bool row2l = grp == 2 && act;
            const unsigned int bl = rlo(v1);
            const unsigned int bmin = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(row2l ? bl : kResNone), 32);
            const unsigned int mb = (unsigned int)((__ballot(row2l && bl == bmin) >> 32) & 0xFFFFull);
            if (bmin != kResNone && mb) {
                W.gb = __builtin_ctz(mb);
                W.bi = bmin;
                W.leaB = (int)(__builtin_amdgcn_readlane((int)rhi(v1), 32 + W.gb) & 0xFFFFFF);
                W.dpB = readlane_f64(v0, 48 + W.gb);
                W.xpB = readlane_f64(v1, 48 + W.gb);
                W.postedB = bmin;   // (a workgroup with a zero-level row posts that row)
            }
        }
        if (first) {   // every workgroup on the same XCC: the same-XCD record accesses from here on (all see the same ids: all switch together)
            const unsigned int xl = (rhi(v1) >> 24) & 0xFu;
            fast = !dead && __all(!row0l || xl == myxcc);
            first = false;
        }
        return W;
    };
    // row `need` of workgroup gwin for this thread's columns: the row it posted with the exchange, or (the Bland rule took another of
    // its rows than it had guessed: a negative ratio beside zero-level rows) a second post — as a FULL exchange with empty bids, so that
    // no workgroup runs two exchanges ahead of one that still polls the records of this parity
    auto fetch_row = [&](const int gwin, const unsigned int need, const unsigned int posted, double (&v)[CJ]) {
        if (posted != need) {   // (uniform over the relaxation)
            ResLocal L;
            L.m = inf; L.l = g == gwin ? (int)need - row0 : -1; L.dp = 0; L.xp = 0; L.lea = 0; L.ru = inf; L.lb = -1; L.dpB = 0; L.xpB = 0; L.leaB = 0;
            (void)exchange(L, false);
            if (dead) return;
        }
        const double seqd = seq0 + (double)ex;
        const xpair *src = rows + ((size_t)((int)(ex & 1) * kResGMax + gwin) * kResCols);
        const int j0 = tid, j1 = tid + NT;
        const bool in0 = (unsigned int)j0 < ldt, in1 = CJ > 1 && (unsigned int)j1 < ldt;
        xpair got[CJ];
        int spins = 0;
        for (;;) {
            if constexpr (CJ == 1) {
                XLoad<1>::run(src + (in0 ? j0 : 0), got, fast);
                if (__all(!in0 || got[0].x == seqd)) break;
            } else {
                XLoad<2>::run(src + (in0 ? j0 : 0), src + (in1 ? j1 : 0), got, fast);
                if (__all((!in0 || got[0].x == seqd) && (!in1 || got[1].x == seqd))) break;
            }
            if (++spins > kResSpin) { dead = true; break; }
        }
#pragma unroll
        for (int s = 0; s < CJ; s++) v[s] = ((unsigned int)(tid + s * NT) < ldt) ? got[s].y : 0.0;
    };

    for (int k = 0; k < npiv; k++) {
        const bool forced = (k == 0 && a.forced_q >= 0);
        const bool free1 = (k == 0 && a.exact_once);   // the first pivot decides on the r / x_B the host just refreshed
        int q = 0, p = 0, ent = 0, lea = 0, gwin = 0;
        double rq = 0, dpv = 1.0, xbp = 0, dl = 0;
        unsigned int posted = kResNone;
        const double *dcolp = &colbuf[0][0][0];
        bool bland = false;
        if (!forced) {
            // ---- entering position: first index of min r (simplex.go:247)
            RWin fq = select(r, rq, ent, dcolp);
            if (a.guard > 0 && !free1) {
                // guard mode (BTArgs::guard; bt_kernels.hip bt_inner2_body): a minimum within the guard of the stop threshold, or two
                // columns within the guard of each other, is decided by the rounding noise of the reference's fresh solve
                double x2 = inf;
#pragma unroll
                for (int s = 0; s < CJ; s++) x2 = vmin_f64(x2, (unsigned int)(tid + s * NT) == fq.i ? inf : r[s]);
                x2 = wave_min_f64(x2);
                __syncthreads();
                if (lane == 0) red2[wv] = x2;
                __syncthreads();
                const double y2 = lane < NW ? red2[lane] : inf;
                const double r2 = readlane_f64(row_min_f64(y2), 15);
                if (a.guard == inf || fabs(rq + a.tol) <= 1e-12 || (!(rq >= -a.tol) && r2 - rq <= a.guard * fmax(1.0, fabs(rq)))) { status = ST_NEED_EXACT; break; }
            }
            q = (int)fq.i;
            if (fq.i >= (unsigned int)a.nn) {   // every r_j is NaN: MinIdx returns 0
                double z0[CJ];
#pragma unroll
                for (int s = 0; s < CJ; s++) z0[s] = (tid + s * NT == 0) ? 0.0 : inf;
                double dummy;
                select(z0, dummy, ent, dcolp);
                q = 0; rq = __builtin_nan("");
            }
            if (rq >= -a.tol) { status = ST_OPTIMAL; break; }   // simplex.go:248
            const ResLocal L = ratio_stage(dcolp, -1, a.guard > 0, false, dl);
            const ResWin W = exchange(L, false);
            if (dead) break;
            p = (int)W.i; dpv = W.dp; xbp = W.xp; lea = W.lea; gwin = W.gw; posted = W.posted;
            const double mv = W.m;
            if (mv == inf || W.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // simplex.go:328-330
            if (a.guard > 0 && !free1 && (mv <= a.guard || W.mv2 - mv <= a.guard * fmax(1.0, fabs(mv)) || fabs(dpv) <= a.guard)) { status = ST_NEED_EXACT; break; }
            if (a.cguard > 0 && fabs(dpv) <= a.cguard && !free1) { status = ST_NEED_EXACT; break; }   // (BTArgs::cguard)
            if (mv <= 0) {
                // ---- replaceBland (simplex.go:347-383): candidates in position order with r_i <= -1e-14 after the 1e-13 rounding of
                // :252-256; per candidate ONE exchange that carries both the minimum ratio (:362) and the first zero-level row (:368-379)
                bland = true;
                blands++;
                int cand = -1;
                bool found = false;
                for (;;) {
                    double fl[CJ];
#pragma unroll
                    for (int s = 0; s < CJ; s++) {
                        const int j = tid + s * NT;
                        double rv = r[s];
                        if (fabs(rv) < 1e-13) rv = 0;
                        fl[s] = (j < a.nn && j > cand && !(rv > -1e-14)) ? 0.0 : inf;
                    }
                    double rqc;
                    int entc;
                    const RWin fc = select(fl, rqc, entc, dcolp);
                    if (fc.m != 0.0) break;   // candidates exhausted -> ErrBland
                    cand = (int)fc.i;
                    const ResLocal L2 = ratio_stage(dcolp, -1, false, true, dl);
                    const ResWin W2 = exchange(L2, true);
                    if (dead) break;
                    if (W2.m == inf || W2.i >= (unsigned int)a.m) { status = ST_UNBOUNDED; break; }   // computeMove inside Bland, :356-360
                    if (fabs(W2.m) > 1e-12) {   // :362
                        q = cand; rq = rqc; ent = entc; p = (int)W2.i; dpv = W2.dp; xbp = W2.xp; lea = W2.lea; gwin = W2.gw; posted = W2.posted; found = true;
                        break;
                    }
                    if (W2.bi != kResNone) {     // :368-379
                        q = cand; rq = rqc; ent = entc; p = (int)W2.bi; dpv = W2.dpB; xbp = W2.xpB; lea = W2.leaB; gwin = W2.gb; posted = W2.postedB; found = true;
                        break;
                    }
                }
                if (dead || status == ST_UNBOUNDED) break;
                if (!found) { status = ST_BLAND_FAILED; break; }
            }
        } else {
            // set-up pivot chosen by the host
            q = a.forced_q; p = a.forced_p;
            double z0[CJ];
#pragma unroll
            for (int s = 0; s < CJ; s++) z0[s] = (tid + s * NT == q) ? 0.0 : inf;
            double rq0;
            select(z0, rq0, ent, dcolp);
            rq = a.forced_nocommit ? 0.0 : rq0;   // a set-up pivot leaves the reduced costs alone (they are rebuilt); a pivot the host decided on fresh solves is a pivot like any other
            const ResLocal L = ratio_stage(dcolp, p, false, false, dl);
            const ResWin W = exchange(L, false);
            if (dead) break;
            dpv = W.dp; xbp = W.xp; lea = W.lea; gwin = W.gw; posted = W.posted;
            if (W.i != (unsigned int)p) { dead = true; break; }   // (a row outside the tableau: never ordered)
        }
        // ---- row p for this thread's columns, then the rank-1 update in registers
        double vrow[CJ];
        fetch_row(gwin, (unsigned int)p, posted, vrow);
        if (dead) break;
        const double rinv = 1.0 / dpv, nrinv = -rinv;
        const double mult = rq * rinv;
        const double theta = xbp * rinv;
        {
            const double u = (rvalid && irow == p) ? rinv - 1.0 : dl * nrinv;   // rows beyond m: d = 0
            if (rvalid) xbl = (irow == p) ? theta : __builtin_fma(-theta, dl, xbl);
            if (lane < RM) ubuf[wv][lane < RM ? lane : 0] = u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double vp[CJ];
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            const double v = vrow[s];
            // reduced costs (positional): r_j - (r_q / d_p) v_j; the leaving variable takes slot q
            if (j < a.nn) r[s] = (j == q) ? -mult : __builtin_fma(-mult, v, r[s]);
            vp[s] = (j == q) ? dpv + 1.0 : v;
        }
        {
            const rvec2 *us = reinterpret_cast<const rvec2 *>(&ubuf[wv][0]);
#pragma unroll
            for (int t = 0; t < NV; t++) {
                if (t * 8 < R) {
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        const rvec2 uu = us[t * 4 + h];
#pragma unroll
                        for (int s = 0; s < CJ; s++) {
                            T[s][t][2 * h] = __builtin_fma(uu[0], vp[s], T[s][t][2 * h]);
                            T[s][t][2 * h + 1] = __builtin_fma(uu[1], vp[s], T[s][t][2 * h + 1]);
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // a host-chosen set-up pivot may leave the lists alone (forced_nocommit 1: the host uploads new ones) or exchange them without
        // being counted or traced as a pivot of the loop (2, 3)
        const bool commit = !(forced && a.forced_nocommit);
        if (commit || (forced && a.forced_nocommit >= 2)) {
#pragma unroll
            for (int s = 0; s < CJ; s++) if (tid + s * NT == q) nbv[s] = lea;
            if (rvalid && irow == p) basl = ent;
        }
        if (forced && a.forced_nocommit == 3) status = ST_FORCED_DONE;   // batched schedule: this order runs once
        if (commit && g == 0 && tid == 0) {   // simplex.go:280
            if (a.trace && trace_len < a.trace_cap) {
                DevPivot &tr = a.trace[trace_len];
                tr.phase = a.phase; tr.bland = bland ? 1 : 0; tr.min_idx = q; tr.replace = p; tr.entering = ent; tr.leaving = lea;
            }
            trace_len += 1;
            npv += 1;
        }
        kd = k + 1;
        if (status != ST_RUNNING) break;
    }
    if (dead) status = ST_XCHG_TIMEOUT;
    // ---- write back: the slab, x_B and the basic list of this workgroup's rows; workgroup 0: r, the nonbasic list, the state
    if (kd > 0 && !dead) {
        double *Tw = a.T;
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
#pragma unroll
            for (int t = 0; t < NV; t++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int l = t * 8 + e, i = row0 + l;
                    if (l < R && i < a.m && (unsigned int)j < ldt) Tw[tile_off_g((unsigned int)i, (unsigned int)j, ldt)] = T[s][t][e];
                }
            }
            if (g == 0 && (unsigned int)j < ldt) {
                a.r[j] = j < a.nn ? r[s] : 0.0;
                if (j < a.nn) a.nonbasic[j] = nbv[s];
            }
        }
        if (wv == 0 && rvalid) { a.xb[irow] = xbl; a.basic[irow] = basl; }
    }
    if (g == 0 && tid == 0) {
        st->trace_len = trace_len;
        st->pivots = npv;
        st->kdone = 0;   // nothing is pending: the tableau in a.T is current
        st->bland_steps += blands;
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

}  // namespace

// Relaxation at position `slot` of the active list = G workgroups on one XCD (blocks are dealt round-robin over the 8 XCDs: block
// b = x + 8 j runs on XCD x; slot = (j / G) * 8 + x, workgroup j % G).  Every workgroup of a relaxation must be resident (they wait for
// each other): the host launches at most 8 slots x 16 workgroups per schedule — half the CUs of every XCD (engine_batch.cpp).
// nb: blocks of 8 pivots the host budgets for this launch; stages with a host-chosen or no pivot (kmax < 8) run kmax pivots.
__global__ __launch_bounds__(kResNT) void k_b_res(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, const int G, const int nb,
                                                  const double seq0, xpair *__restrict__ xbase) {
    const unsigned int x8 = blockIdx.x & 7u, jb = blockIdx.x >> 3;
    const int slot = (int)((jb / (unsigned int)G) * 8u + x8), g = (int)(jb % (unsigned int)G);
    if (slot >= *count) return;
    const BatchLP &lp = lps[ids[slot]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_COLD || stage == BS_DUAL) return;
    const BTArgs a = lp.bt;
    const int npiv = a.kmax >= 8 ? nb * 8 : a.kmax;
#ifdef GOMILP_DEBUG
    if (a.fault && g == 1) return;   // test hook (diagnostic flavour only): a workgroup that never takes part -> the others give up (ST_XCHG_TIMEOUT)
#endif
    xpair *xslot = xbase + (size_t)slot * kResSlotPairs;
    if (a.ldt <= kResNT) res_body<1>(a, g, G, npiv, seq0, xslot);
    else res_body<2>(a, g, G, npiv, seq0, xslot);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
size_t b_res_slot_bytes() { return kResSlotPairs * sizeof(xpair); }
int b_res_max_slots() { return 8; }
// workgroups per relaxation for a wave whose largest relaxation has m_max rows (0: the kernel does not take this shape)
int b_res_groups(int m_max, int ldt_max) {
    if (ldt_max > kResCols || ldt_max > 2 * kResNT || (ldt_max & 63) != 0) return 0;
    for (int G : {8, 16}) {
        if (G == 8 && m_max > 256) continue;   // (8 workgroups: the cheaper exchange, where 32 rows per workgroup are enough)
        if (4 * ((m_max + 4 * G - 1) / (4 * G)) <= kResRMax) return G;
    }
    return 0;
}
void launch_b_res(const BatchLP *lps, const int *ids, const int *count, int nlp, int G, int nb, double seq0, void *xbase, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const unsigned int grid = 8u * (unsigned int)((nlp + 7) / 8) * (unsigned int)G;
    hipExtLaunchKernelGGL(k_b_res, dim3(grid), dim3(kResNT), 0, s, e0, e1, 0, lps, ids, count, G, nb, seq0, reinterpret_cast<xpair *>(xbase));
}

}  // namespace gomilp
