// Register-resident tableau pivoting for relaxations of up to 640 rows and 512 (+ 1) nonbasic columns (gfx950) — the long chains of a
// device-batched wave.
//
// Every block kernel of bt_kernels.hip / btg_kernels.hip reads one column and one row of a tableau that lives in HBM / L2, corrects
// them with the rank-1 terms that are not applied yet, and leaves the streaming update to other workgroups: two dependent memory
// round trips, 4 + 4 lagging terms and (over several workgroups) two exchanges per pivot — 4.5 us per pivot on a 520 x 512 tableau
// of 2.1 MB.  Such a tableau fits the REGISTER FILES of a few CUs (512 KB of VGPRs each).  Here a relaxation is G workgroups of
// 256 threads (one wave per SIMD) on one XCD:
//   * workgroup g owns the rows [g R, (g + 1) R) (R = rows per workgroup, a multiple of 4, <= 40); thread t holds the columns t and
//     t + 256 of those rows in registers: 2 x 40 doubles, read with a uniform dynamic index (v_movrels) when a row leaves; a 513th
//     column — the artificial of Phase I (simplex.go:532-551) — lives in LDS, one lane per row;
//   * the reduced costs r and the nonbasic list are REPLICATED in every workgroup (thread t: its columns), updated with the same
//     instructions from the same operands — the entering column (floats.MinIdx over r, simplex.go:247) needs NO exchange;
//   * x_B and the basic list of the workgroup's rows sit in lanes 0 .. R-1 of EVERY wave (each wave repeats the ratio test of
//     simplex.go:321-340 for the workgroup's rows on a SIMD of its own: nobody waits for a broadcast);
//   * two barriers per column selection: the waves' candidates (value, index, r_j, variable) through LDS, then the slab of the winning
//     column, written by the one lane that holds it;
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
// The pivot loop is ONE round — select a column, ratio test, exchange — behind a small mode switch (Dantzig / Bland candidate / host-
// chosen / second post), every stage with a single call site: the first version inlined each stage four times, 50 KB of code per
// instance and 1200 spilled scalar registers, and ran at 7 us per pivot.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "device_types.h"
#include "kernels_common.h"
#include "bt_loop.h"

namespace gomilp {

constexpr int kResNT = 256;          // threads per workgroup: ONE wave per SIMD
constexpr int kResCJ = 2;            // register columns per thread: 512 columns
constexpr int kResRMax = 40;         // rows per workgroup
constexpr int kResGMax = 16;         // workgroups per relaxation (records per exchange)
constexpr int kResCols = 1024;       // granules of a slot's row buffer (512 + the extra column, rounded up)
constexpr int kResSpin = 400000;     // polls (~0.3 us each) before a workgroup gives up
constexpr unsigned int kResNone = 0xFFFFFFFFu;
// slot buffer (xpairs): records [parity][kResGMax][kXSlots], then candidate rows [parity][kResGMax][kResCols]
constexpr size_t kResRecPairs = (size_t)2 * kResGMax * kXSlots;
constexpr size_t kResSlotPairs = kResRecPairs + (size_t)2 * kResGMax * kResCols;

typedef double rvec8 __attribute__((ext_vector_type(8)));
typedef double rvec2 __attribute__((ext_vector_type(2)));

#ifdef GOMILP_DEBUG
// diagnostic flavour: cycles (s_memtime) per segment of a pivot, summed over the launches of a process by the waves of workgroup 0 of
// every relaxation: [wave][segment] (32 per wave); [0][31] = exchanges, [0][30] = pivots, [0][29] = polls that found a record missing,
// [0][28] = exchanges in the same-XCD access mode; gomilp_debug_res_stamps() hands them out
__device__ unsigned long long g_res_stamps[4 * 32];
#define RES_STAMP(S)                                                                      \
    do {                                                                                  \
        unsigned long long t_;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if ((S) >= 0) tacc[(S) >= 0 ? (S) : 0] += t_ - tprev;                             \
        tprev = t_;                                                                       \
    } while (0)
#define RES_COUNT(S, V) do { tacc[S] += (unsigned long long)(V); } while (0)
#else
#define RES_STAMP(S) do { } while (0)
#define RES_COUNT(S, V) do { } while (0)
#endif

namespace {

__device__ __forceinline__ double rpack(unsigned int lo, unsigned int hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
__device__ __forceinline__ unsigned int rlo(double v) { return (unsigned int)(unsigned long long)__double_as_longlong(v); }
__device__ __forceinline__ unsigned int rhi(double v) { return (unsigned int)((unsigned long long)__double_as_longlong(v) >> 32); }

// element l (uniform) of a thread's register column (kResRMax / 8 = 5 tuples of 8 doubles): the tuple by a uniform branch, the element by
// v_movrels.  (The tuples come BY VALUE: through a reference the element access is canonicalised into a scalar load at a dynamic address
// before the function is inlined, and the whole array stays in scratch; the index passes through an asm statement so that the compiler
// cannot bound it — a tuple that is loaded for this one use would otherwise be turned into such a scalar load too; the asm statements
// behind the accesses keep the branches from being merged into one access through a selected pointer.)
__device__ __forceinline__ double res_rowval(const rvec8 t0, const rvec8 t1, const rvec8 t2, const rvec8 t3, const rvec8 t4, int l) {
    const int t = __builtin_amdgcn_readfirstlane(l >> 3);
    int e = __builtin_amdgcn_readfirstlane(l & 7);
    asm volatile("" : "+s"(e));
    double v;
    if (t == 0) { v = t0[e]; asm volatile("" : "+v"(v)); }
    else if (t == 1) { v = t1[e]; asm volatile("" : "+v"(v)); }
    else if (t == 2) { v = t2[e]; asm volatile("" : "+v"(v)); }
    else if (t == 3) { v = t3[e]; asm volatile("" : "+v"(v)); }
    else { v = t4[e]; asm volatile("" : "+v"(v)); }
    return v;
}

enum : int { RM_DANTZIG = 0, RM_BLAND = 1, RM_FORCED = 2, RM_REPOST = 3 };

}  // namespace

// Relaxation at position `slot` of the active list = G workgroups on one XCD (blocks are dealt round-robin over the 8 XCDs: block
// b = x + 8 j runs on XCD x; slot = (j / G) * 8 + x, workgroup j % G).  Every workgroup of a relaxation must be resident (they wait for
// each other): the host launches at most 8 slots x 16 workgroups per schedule — half the CUs of every XCD (engine_batch.cpp).
// nb: blocks of 8 pivots the host budgets for this launch; stages with a host-chosen or no pivot (kmax < 8) run kmax pivots.
__global__ __launch_bounds__(kResNT) void k_b_res(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, const int G, const int nb,
                                                  const double seq0, xpair *__restrict__ xbase) {
    constexpr int NT = kResNT, NW = NT / 64, RM = kResRMax, NV = RM / 8, CJ = kResCJ, XC = CJ * NT;   // XC: index of the column that lives in LDS
    __shared__ __attribute__((aligned(16))) double colbuf[2][NW][RM];
    __shared__ __attribute__((aligned(16))) double ubuf[NW][RM];
    __shared__ __attribute__((aligned(16))) double xcol[RM];
    __shared__ double redM[2][NW], payR[2][NW], red2[NW];
    __shared__ unsigned int redI[2][NW];
    __shared__ int payN[2][NW];
    const unsigned int x8 = blockIdx.x & 7u, jb = blockIdx.x >> 3;
    const int slot = (int)((jb / (unsigned int)G) * 8u + x8), g = (int)(jb % (unsigned int)G);
    if (slot >= *count) return;
    const BatchLP &lp = lps[ids[slot]];
    const int stage = lp.stage;
    if (stage == BS_DONE || stage == BS_HOST || stage == BS_COLD || stage == BS_DUAL) return;
#ifdef GOMILP_DEBUG
    if (lp.bt.fault && g == 1) return;   // test hook (diagnostic flavour only): a workgroup that never takes part -> the others give up (ST_XCHG_TIMEOUT)
#endif
    // the argument block, field by field (scalar loads): what the loop reads stays in registers, the pointers are fetched again for the write-back
    const int m = lp.bt.m, nn = lp.bt.nn, kmax = lp.bt.kmax;
    const unsigned int ldt = (unsigned int)lp.bt.ldt;
    const double tol = lp.bt.tol, guard = lp.bt.guard, cguard = lp.bt.cguard;
    const int forced_q = lp.bt.forced_q, forced_p = lp.bt.forced_p, forced_nocommit = lp.bt.forced_nocommit, exact_once = lp.bt.exact_once;
    DevState *st = lp.bt.st;
    DevPivot *const trace = lp.bt.trace;
    const long long trace_cap = lp.bt.trace_cap;
    const int phase = lp.bt.phase;
    const int npiv = kmax >= 8 ? nb * 8 : kmax;
    int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wbase = __builtin_amdgcn_readfirstlane(tid & ~63);
    const double inf = __builtin_inf();
    const int done = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int R = 4 * ((m + 4 * G - 1) / (4 * G)), row0 = g * R;   // (the host launches this kernel only where R <= kResRMax)
    xpair *recs = xbase + (size_t)slot * kResSlotPairs, *rows = recs + kResRecPairs;
    unsigned int myxcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
    myxcc &= 0xFu;
    if (done) {
        if (g == 0 && tid == 0) st->kdone = 0;
        return;
    }
    const bool hasx = nn > XC;   // (the host admits nn <= XC + 1)
    // ---- state: the slab of the tableau, r and the nonbasic list by column, x_B and the basic list by row (lane l of every wave)
    rvec8 T[CJ][NV];
    double r[CJ], rX = inf;
    int nbv[CJ], nbX = 0;
    {
        const double *Tg = lp.bt.T, *rg = lp.bt.r;
        const int32_t *nbg = lp.bt.nonbasic;
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            r[s] = j < nn ? rg[j] : inf;   // padding never wins an argmin
            nbv[s] = j < nn ? nbg[j] : 0;
#pragma unroll
            for (int t = 0; t < NV; t++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int l = t * 8 + e, i = row0 + l;
                    T[s][t][e] = (l < R && i < m && (unsigned int)j < ldt) ? Tg[tile_off_g((unsigned int)i, (unsigned int)j, ldt)] : 0.0;
                }
            }
        }
        if (hasx) { rX = rg[XC]; nbX = nbg[XC]; }
        if (tid < RM) xcol[tid] = (hasx && tid < R && row0 + tid < m) ? Tg[tile_off_g((unsigned int)(row0 + tid), (unsigned int)XC, ldt)] : 0.0;
    }
    int irow = row0 + lane;
    bool rvalid = lane < R && irow < m;
    double xbl = rvalid ? lp.bt.xb[irow] : 0.0;
    int basl = rvalid ? lp.bt.basic[irow] : 0;
    long long trace_len = 0, npv = 0;
    if (g == 0 && tid == 0) { trace_len = st->trace_len; npv = st->pivots; }
    int kd = 0, status = ST_RUNNING, blands = 0;
    bool dead = false, fast = false, first = true;
    int selpar = 0;
    int ex = 0;   // exchanges of this launch
#ifdef GOMILP_DEBUG
    unsigned long long tacc[32] = {};
    unsigned long long tprev = 0;
#endif
    __syncthreads();   // (xcol)
    RES_STAMP(-1);

    // ---- the round: [select a column -> ratio test] -> exchange -> decide; a pivot ends a sequence of rounds
    int mode = (forced_q >= 0 && npiv > 0) ? RM_FORCED : RM_DANTZIG;
    int cand = -1;                    // Bland rule: the last candidate position tried
    bool bland = false;
    int q = 0, ent = 0;               // entering position / variable of the running pivot
    double rq = 0, dl = 0;            // its reduced cost; this lane's entry of its column
    const double *dcolp = &colbuf[0][0][0];
    unsigned int need = kResNone;     // RM_REPOST: the row the winner has to post
    int gwin = 0, lea = 0;
    double dpv = 1.0, xbp = 0;
    while (kd < npiv) {
        RES_STAMP(9);   // (what is left between two rounds: lists, trace, loop control)
        // (tid and what hangs on it are re-derived from an opaque copy every round: left loop-invariant, every per-lane predicate of the
        // loop — a 64-bit mask each — is hoisted in front of it, the scalar registers overflow and the body fills with spill moves)
        asm volatile("" : "+v"(tid));
        lane = tid & 63; wv = tid >> 6; irow = row0 + lane; rvalid = lane < R && irow < m;
        const bool free1 = (kd == 0 && exact_once);   // the first pivot decides on the r / x_B the host just refreshed
        // ---- this workgroup's bid
        double Lm = inf, Ldp = 0, Lxp = 0, Lru = inf, LdpB = 0, LxpB = 0;
        int Ll = -1, Llea = 0, Llb = -1, LleaB = 0;
        if (mode != RM_REPOST) {
            // ---- column selection: first index of the minimum of val over all columns (every workgroup computes the same).  Each wave
            // dumps the slab of ITS candidate column to LDS beside (value, index, r_j, variable); one barrier; everybody reads the winner's.
            // The buffers alternate: a wave can reach its next selection while a slower one still reads this one's.
            double val[CJ], valX = inf;
#pragma unroll
            for (int s = 0; s < CJ; s++) {
                const int j = tid + s * NT;
                if (mode == RM_DANTZIG) val[s] = r[s];                                  // simplex.go:247
                else if (mode == RM_FORCED) val[s] = (j == forced_q) ? 0.0 : inf;       // the host's choice
                else {                                                                   // replaceBland, simplex.go:351-353: candidates in position order
                    double rv = r[s];                                                    // with r_i <= -1e-14 after the 1e-13 rounding of :252-256
                    if (fabs(rv) < 1e-13) rv = 0;
                    val[s] = (j < nn && j > cand && !(rv > -1e-14)) ? 0.0 : inf;
                }
            }
            if (hasx) {
                if (mode == RM_DANTZIG) valX = rX;
                else if (mode == RM_FORCED) valX = (XC == forced_q) ? 0.0 : inf;
                else { double rv = rX; if (fabs(rv) < 1e-13) rv = 0; valX = (XC > cand && !(rv > -1e-14)) ? 0.0 : inf; }
            }
            selpar ^= 1;
            double wm = vmin_f64(val[0], val[1]);
            wm = wave_min_f64(wm);
            unsigned int wi = kResNone;
#pragma unroll
            for (int s = CJ - 1; s >= 0; s--) {
                const unsigned long long mask = __ballot(val[s] == wm);
                if (mask) wi = (unsigned int)(s * NT + wbase + __builtin_ctzll(mask));
            }
            RES_STAMP(10);   // own argmin
#pragma unroll
            for (int s = 0; s < CJ; s++) {
                if ((unsigned int)(tid + s * NT) == wi) { payR[selpar][wv] = r[s]; payN[selpar][wv] = nbv[s]; }
            }
            if (lane == 0) { redM[selpar][wv] = wm; redI[selpar][wv] = wi; }
            __syncthreads();
            RES_STAMP(1);   // barrier
            double fm;
            unsigned int fi;
            {
                const double x = lane < NW ? redM[selpar][lane & (NW - 1)] : inf;
                const unsigned int ii = lane < NW ? redI[selpar][lane & (NW - 1)] : kResNone;
                fm = readlane_f64(row_min_f64(x), 15);
                const unsigned int key = (x == fm) ? ii : kResNone;
                fi = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(key), 15);
            }
            // the column in LDS (index XC, behind every register column: it wins only with a strictly smaller value)
            const bool xwin = hasx && (valX < fm || (fm != fm && valX == valX));
            if (xwin) { fm = valX; fi = (unsigned int)XC; }
            const int ww = (int)((fi & (unsigned int)(NT - 1)) >> 6);
            rq = xwin ? rX : payR[selpar][ww];
            ent = xwin ? nbX : payN[selpar][ww];
            dcolp = xwin ? &xcol[0] : &colbuf[selpar][0][0];
            q = (int)fi;
            // the owner of column q hands its slab over (ONE lane writes: with every wave dumping its own candidate before the barrier the
            // four waves' single-lane stores queued behind each other in the LDS store path, 1270 cycles per selection)
            if (!xwin) {
#pragma unroll
                for (int s = 0; s < CJ; s++) {
                    if ((unsigned int)(tid + s * NT) == fi) {
                        rvec2 *dst = reinterpret_cast<rvec2 *>(&colbuf[selpar][0][0]);
#pragma unroll
                        for (int t = 0; t < NV; t++) {
                            if (t * 8 < R) {
#pragma unroll
                                for (int h = 0; h < 4; h++) dst[t * 4 + h] = rvec2{T[s][t][2 * h], T[s][t][2 * h + 1]};
                            }
                        }
                    }
                }
                RES_STAMP(0);   // column dump
                __syncthreads();
            }
            RES_STAMP(2);   // workgroup pick
            if (mode == RM_DANTZIG) {
                if (guard > 0 && !free1) {
                    // guard mode (BTArgs::guard; bt_kernels.hip bt_inner2_body): a minimum within the guard of the stop threshold, or two
                    // columns within the guard of each other, is decided by the rounding noise of the reference's fresh solve
                    double x2 = inf;
#pragma unroll
                    for (int s = 0; s < CJ; s++) x2 = vmin_f64(x2, (unsigned int)(tid + s * NT) == fi ? inf : r[s]);
                    x2 = wave_min_f64(x2);
                    __syncthreads();
                    if (lane == 0) red2[wv] = x2;
                    __syncthreads();
                    const double y2 = lane < NW ? red2[lane & (NW - 1)] : inf;
                    double r2 = readlane_f64(row_min_f64(y2), 15);
                    if (hasx && fi != (unsigned int)XC) r2 = fmin(r2, rX);
                    if (guard == inf || fabs(rq + tol) <= 1e-12 || (!(rq >= -tol) && r2 - rq <= guard * fmax(1.0, fabs(rq)))) { status = ST_NEED_EXACT; break; }
                }
                if (fi >= (unsigned int)nn) { dead = true; break; }   // every r_j is NaN (never: the host path reports what MinIdx makes of it)
                if (rq >= -tol) { status = ST_OPTIMAL; break; }       // simplex.go:248
            } else if (mode == RM_BLAND) {
                if (fm != 0.0) { status = ST_BLAND_FAILED; break; }   // candidates exhausted -> ErrBland (simplex.go:382)
            } else if (forced_nocommit) rq = 0.0;   // a set-up pivot leaves the reduced costs alone (they are rebuilt); a pivot the host decided on fresh solves is a pivot like any other
            // ---- ratio test of simplex.go:321-340 over this workgroup's rows, in every wave alike (lane l = local row l): d' = -d rounded
            // at 1e-13, move = x_B / |d'| where d' < 0, +Inf elsewhere.  A host-chosen leaving row: only its owner bids
            dl = (lane < RM) ? dcolp[lane < RM ? lane : 0] : 0.0;
            if (!rvalid) dl = 0.0;
            double dn = -dl;
            if (fabs(dn) < 1e-13) dn = 0;
            const double quot = div_pos(xbl, fabs(dn));   // == x_B / |d'| bit for bit; discarded where the reference does not divide
            double mv = (dn >= 0 || !rvalid) ? inf : quot;
            if (mode == RM_FORCED) mv = (rvalid && irow == forced_p) ? 0.0 : inf;
            RES_STAMP(11);   // column from LDS, quotients
            Lm = wave_min_f64(mv);
            const unsigned long long mask = __ballot(mv == Lm);
            const bool has = Lm < inf && mask != 0ull;   // (a NaN minimum: no candidate, like +Inf)
            Ll = has ? (int)__builtin_ctzll(mask) : -1;
            if (!has) Lm = inf;
            const int lr = Ll < 0 ? 0 : Ll;
            Ldp = readlane_f64(dl, lr);
            Lxp = readlane_f64(xbl, lr);
            Llea = __builtin_amdgcn_readlane(basl, lr);
            if (guard > 0) Lru = wave_min_f64(lane == Ll ? inf : mv);
            if (mode == RM_BLAND) {
                const unsigned long long mb = __ballot(rvalid && !(mv > 1e-12));   // simplex.go:368-379: rows with move <= blandZeroTol, in order
                if (mb) {
                    Llb = (int)__builtin_ctzll(mb);
                    LdpB = readlane_f64(dl, Llb);
                    LxpB = readlane_f64(xbl, Llb);
                    LleaB = __builtin_amdgcn_readlane(basl, Llb);
                }
            }
            RES_STAMP(12);   // wave minimum, first lane, its scalars
        } else if (g == gwin) Ll = (int)need - row0;   // second post: the winner's row, empty bids
        // ---- the exchange: this workgroup's candidate row (every wave its columns) and record (wave 0) out, the G records in
        ex += 1;
        const double seqd = seq0 + (double)ex;
        const int par = (int)(ex & 1);
        const bool blandx = mode == RM_BLAND;
        {
            const int prow = (blandx && Llb >= 0) ? Llb : Ll;
            if (prow >= 0) {   // one {sequence number, value} granule per column
                xpair *dst = rows + ((size_t)(par * kResGMax + g) * kResCols);
#pragma unroll
                for (int s = 0; s < CJ; s++) {
                    const int j = tid + s * NT;
                    if ((unsigned int)j < ldt) {
                        xpair v;
                        v.x = seqd;
                        v.y = res_rowval(T[s][0], T[s][1], T[s][2], T[s][3], T[s][4], prow);
                        xstore(dst + j, v, fast);
                    }
                }
                if (hasx && tid == 0) {
                    xpair v;
                    v.x = seqd;
                    v.y = xcol[prow];
                    xstore(dst + XC, v, fast);
                }
            }
        }
        RES_STAMP(13);   // candidate row out
        if (wv == 0 && lane < kXSlots) {
            const unsigned int gi = Ll >= 0 ? (unsigned int)(row0 + Ll) : kResNone;
            const unsigned int gb = Llb >= 0 ? (unsigned int)(row0 + Llb) : kResNone;
            xpair v;
            v.x = seqd;
            v.y = lane == 0 ? Lm : lane == 1 ? rpack(gi, ((unsigned int)Llea & 0xFFFFFFu) | (myxcc << 24)) : lane == 2 ? Ldp : lane == 3 ? Lxp : lane == 4 ? Lru
                  : lane == 5 ? rpack(gb, (unsigned int)LleaB & 0xFFFFFFu) : lane == 6 ? LdpB : LxpB;
            xstore(recs + ((size_t)(par * kResGMax + g) * kXSlots + lane), v, fast);
        }
        RES_STAMP(4);   // record out
        RES_COUNT(31, 1); RES_COUNT(28, fast ? 1 : 0);
        // lane l reads slots 2 (l >> 4) and 2 (l >> 4) + 1 of record l & 15: lanes 0..15 hold (min, index | variable) of record l,
        // lanes 16..31 (d_p, x_B[p]), lanes 32..47 (runner-up, Bland row | variable), lanes 48..63 (Bland d_p, x_B[p])
        const int rec = lane & 15, grp = lane >> 4;
        const bool act = rec < G;
        xpair got[2];
        {
            const xpair *base = recs + (size_t)(par * kResGMax + (act ? rec : 0)) * kXSlots + 2 * grp;
            int spins = 0;
            for (;;) {
                XLoad<2>::run(base, base + 1, got, fast);
                if (__all(!act || (got[0].x == seqd && got[1].x == seqd))) break;
                if (++spins > kResSpin) { dead = true; break; }
            }
            RES_COUNT(29, spins);
        }
        RES_STAMP(5);   // poll
        if (dead) break;
        const double v0 = got[0].y, v1 = got[1].y;
        const bool row0l = lane < 16 && act;
        if (first) {   // every workgroup on the same XCC: the same-XCD record accesses from here on (all see the same ids: all switch together)
            const unsigned int xl = (rhi(v1) >> 24) & 0xFu;
            fast = __all(!row0l || xl == myxcc);
            first = false;
        }
        unsigned int posted = kResNone;
        int p = 0;
        if (mode != RM_REPOST) {
            // lexicographic minimum (value, index) over the records = floats.MinIdx over the whole ratio vector
            const double xm = readlane_f64(row_min_f64(row0l ? v0 : inf), 0);
            const bool mine = row0l && v0 == xm;
            const unsigned int idxl = rlo(v1);
            const unsigned int kmin = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(mine ? idxl : kResNone), 0);
            const unsigned int mk = (unsigned int)(__ballot(mine && idxl == kmin) & 0xFFFFull);
            const int gw = mk ? __builtin_ctz(mk) : 0;
            const double Wm = mk ? xm : inf;
            const unsigned int Wi = mk ? kmin : kResNone;
            gwin = gw;
            p = (int)Wi;
            lea = (int)(__builtin_amdgcn_readlane((int)rhi(v1), gw) & 0xFFFFFF);
            dpv = readlane_f64(v0, 16 + gw);
            xbp = readlane_f64(v1, 16 + gw);
            const unsigned int bgw = (unsigned int)__builtin_amdgcn_readlane((int)rlo(v1), 32 + gw);
            posted = (blandx && bgw != kResNone) ? bgw : Wi;
            RES_STAMP(6);   // winner picked
            if (Wm == inf || Wi >= (unsigned int)m) { status = ST_UNBOUNDED; break; }   // simplex.go:328-330 (computeMove inside Bland: :356-360)
            if (mode == RM_DANTZIG) {
                if (guard > 0 && !free1) {
                    // degenerate (or nearly), two rows within the guard of each other, or a pivot element of rounding-noise size: decided on fresh solves
                    const double rug = readlane_f64(v0, 32 + gw);
                    const double mv2 = readlane_f64(row_min_f64(lane < 16 ? (lane == gw ? rug : (act ? v0 : inf)) : inf), 0);
                    if (Wm <= guard || mv2 - Wm <= guard * fmax(1.0, fabs(Wm)) || fabs(dpv) <= guard) { status = ST_NEED_EXACT; break; }
                }
                if (cguard > 0 && fabs(dpv) <= cguard && !free1) { status = ST_NEED_EXACT; break; }   // (BTArgs::cguard)
                if (Wm <= 0) {   // simplex.go:269 -> replaceBland
                    bland = true; blands++; cand = -1; mode = RM_BLAND;
                    continue;
                }
            } else if (mode == RM_BLAND) {
                if (!(fabs(Wm) > 1e-12)) {   // :362 fails: the first zero-level row over all workgroups (:368-379)
                    const bool row2l = grp == 2 && act;
                    const unsigned int bl = rlo(v1);
                    const unsigned int bmin = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(row2l ? bl : kResNone), 32);
                    const unsigned int mb = (unsigned int)((__ballot(row2l && bl == bmin) >> 32) & 0xFFFFull);
                    if (bmin == kResNone || !mb) { cand = q; continue; }   // no row: the next candidate column
                    const int gb = __builtin_ctz(mb);
                    gwin = gb; p = (int)bmin; posted = bmin;   // (a workgroup with a zero-level row posts that row)
                    lea = (int)(__builtin_amdgcn_readlane((int)rhi(v1), 32 + gb) & 0xFFFFFF);
                    dpv = readlane_f64(v0, 48 + gb);
                    xbp = readlane_f64(v1, 48 + gb);
                }
            } else if (Wi != (unsigned int)forced_p) { dead = true; break; }   // (a row outside the tableau: never ordered)
            if (posted != (unsigned int)p) {   // the Bland rule took another row of the winner than it had guessed (a negative ratio beside
                need = (unsigned int)p;        // zero-level rows): a second post — as a FULL exchange with empty bids, so that no workgroup
                mode = RM_REPOST;              // runs two exchanges ahead of one that still polls the records of this parity
                continue;
            }
        } else p = (int)need;
        // ---- row p of workgroup gwin for this thread's columns, then the rank-1 update in registers
        double vrow[CJ], vX = 0;
        {
            const xpair *src = rows + ((size_t)(par * kResGMax + gwin) * kResCols);
            const int j0 = tid, j1 = tid + NT;
            const bool in0 = (unsigned int)j0 < ldt, in1 = (unsigned int)j1 < ldt;
            xpair gr[2], gx[1];
            gx[0].x = seqd; gx[0].y = 0.0;
            int spins = 0;
            for (;;) {
                XLoad<2>::run(src + (in0 ? j0 : 0), src + (in1 ? j1 : 0), gr, fast);
                if (hasx) XLoad<1>::run(src + XC, gx, fast);   // (the column in LDS: one granule, the same for every lane)
                if (__all((!in0 || gr[0].x == seqd) && (!in1 || gr[1].x == seqd) && gx[0].x == seqd)) break;
                if (++spins > kResSpin) { dead = true; break; }
            }
            vrow[0] = in0 ? gr[0].y : 0.0;
            vrow[1] = in1 ? gr[1].y : 0.0;
            vX = gx[0].y;
        }
        RES_STAMP(7);   // the winner's row
        if (dead) break;
        const double rinv = 1.0 / dpv, nrinv = -rinv;
        const double mult = rq * rinv;
        const double theta = xbp * rinv;
        {
            const double u = (rvalid && irow == p) ? rinv - 1.0 : dl * nrinv;   // rows beyond m: d = 0
            if (rvalid) xbl = (irow == p) ? theta : __builtin_fma(-theta, dl, xbl);
            if (lane < RM) ubuf[wv][lane < RM ? lane : 0] = u;
            if (hasx && wv == 0 && lane < RM) xcol[lane < RM ? lane : 0] = __builtin_fma(u, (XC == q) ? dpv + 1.0 : vX, xcol[lane < RM ? lane : 0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        RES_STAMP(14);   // reciprocal, u, x_B
        double vp[CJ];
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
            const double v = vrow[s];
            // reduced costs (positional): r_j - (r_q / d_p) v_j; the leaving variable takes slot q
            if (j < nn) r[s] = (j == q) ? -mult : __builtin_fma(-mult, v, r[s]);
            vp[s] = (j == q) ? dpv + 1.0 : v;
        }
        if (hasx) rX = (XC == q) ? -mult : __builtin_fma(-mult, vX, rX);
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
        RES_STAMP(8);   // r, the rank-1 update in registers
        RES_COUNT(30, 1);
        // a host-chosen set-up pivot may leave the lists alone (forced_nocommit 1: the host uploads new ones) or exchange them without
        // being counted or traced as a pivot of the loop (2, 3)
        const bool was_forced = (kd == 0 && forced_q >= 0);
        const bool commit = !(was_forced && forced_nocommit);
        if (commit || (was_forced && forced_nocommit >= 2)) {
#pragma unroll
            for (int s = 0; s < CJ; s++) if (tid + s * NT == q) nbv[s] = lea;
            if (XC == q) nbX = lea;
            if (rvalid && irow == p) basl = ent;
        }
        if (commit && g == 0 && tid == 0) {   // simplex.go:280
            if (trace && trace_len < trace_cap) {
                DevPivot &tr = trace[trace_len];
                tr.phase = phase; tr.bland = bland ? 1 : 0; tr.min_idx = q; tr.replace = p; tr.entering = ent; tr.leaving = lea;
            }
            trace_len += 1;
            npv += 1;
        }
        kd += 1;
        if (was_forced && forced_nocommit == 3) { status = ST_FORCED_DONE; break; }   // batched schedule: this order runs once
        mode = RM_DANTZIG; bland = false;
    }
    if (dead) status = ST_XCHG_TIMEOUT;
#ifdef GOMILP_DEBUG
    if (g == 0 && lane == 0) {
        for (int sg = 0; sg < 28; sg++) atomicAdd(&g_res_stamps[wv * 32 + sg], tacc[sg]);
        if (wv == 0) for (int sg = 28; sg < 32; sg++) atomicAdd(&g_res_stamps[sg], tacc[sg]);
    }
#endif
    // ---- write back: the slab, x_B and the basic list of this workgroup's rows; workgroup 0: r, the nonbasic list, the state
    if (kd > 0 && !dead) {
        double *Tw = lp.bt.T, *rw = lp.bt.r;
        int32_t *nbw = lp.bt.nonbasic;
        __syncthreads();   // (xcol: wave 0's last update)
#pragma unroll
        for (int s = 0; s < CJ; s++) {
            const int j = tid + s * NT;
#pragma unroll
            for (int t = 0; t < NV; t++) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int l = t * 8 + e, i = row0 + l;
                    if (l < R && i < m && (unsigned int)j < ldt) Tw[tile_off_g((unsigned int)i, (unsigned int)j, ldt)] = T[s][t][e];
                }
            }
            if (g == 0 && (unsigned int)j < ldt) {
                rw[j] = j < nn ? r[s] : 0.0;
                if (j < nn) nbw[j] = nbv[s];
            }
        }
        if (hasx) {
            if (tid < R && row0 + tid < m) Tw[tile_off_g((unsigned int)(row0 + tid), (unsigned int)XC, ldt)] = xcol[tid];
            if (g == 0 && tid == 0) { rw[XC] = rX; nbw[XC] = nbX; }
        }
        if (wv == 0 && rvalid) { lp.bt.xb[irow] = xbl; lp.bt.basic[irow] = basl; }
    }
    if (g == 0 && tid == 0) {
        st->trace_len = trace_len;
        st->pivots = npv;
        st->kdone = 0;   // nothing is pending: the tableau in bt.T is current
        st->bland_steps += blands;
        if (status != ST_RUNNING) { st->done = 1; st->status = status; }
    }
}

#ifdef GOMILP_DEBUG
void res_stamps_read(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_res_stamps), sizeof(unsigned long long) * 128); }   // 4 waves x 32
#else
void res_stamps_read(unsigned long long *out) { for (int i = 0; i < 128; i++) out[i] = 0; }
#endif

// ---- host side ---------------------------------------------------------------------------------------------------------------------
size_t b_res_slot_bytes() { return kResSlotPairs * sizeof(xpair); }
int b_res_max_slots() { return 8; }
// workgroups per relaxation for a wave whose largest relaxation has m_max rows and nn_max nonbasic columns in a tableau of row length
// ldt_max (0: the kernel does not take this shape): 512 register columns + the one column in LDS (the artificial of Phase I)
int b_res_groups(int m_max, int nn_max, int ldt_max) {
    if (nn_max > kResCJ * kResNT + 1 || ldt_max > kResCJ * kResNT + 64 || (ldt_max & 63) != 0) return 0;
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
