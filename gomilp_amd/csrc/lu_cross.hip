// The panel of the compressed gonum-order LU (lu_compressed.hip) with its ROWS on the workgroups of one XCD (round 5, OPT-IN: knob lu_cross).
//
// k_luc_panel_slots runs a round's dense steps on ONE workgroup of sixteen waves: a dense step is the issue time of those waves on the four
// SIMDs of one CU (7.7 k cycles, DESIGN.md section 2.3 "Round 5").  Here G workgroups of four waves (one per SIMD) hold 256 rows each, one
// row per lane, the same NB register slots; everything a step decides is decided by every workgroup from the same values:
//   * the index maps (lpos / rowat / unit / ucol / active) and the slot tables are REPLICATED — every workgroup keeps all m rows' maps in LDS;
//     the bookkeeping runs are off the chain: a row knows the step that retires it, and the interchanges are replayed on the maps (from the
//     log of pivot rows, by every workgroup alike) only when a pivot search finds two rows with the same |a_ik|;
//   * the pivot search is local (four waves), then ONE exchange through the XCD's L2 (bt_loop.h: records of {sequence number, value}
//     slots, sequence-tagged so that no flag separates data from "ready"): the workgroup's candidate posts
//     {max |a_ik|, how many of its rows attain it, one of them, 1 / a_ik, the XCC id, that row's entries in the NB slots}; every wave polls
//     the G records; ONE row in the whole panel attaining the maximum is the pivot row — otherwise (dgetf2.go:38: the first maximum in
//     LAPACK's row order) the maps are brought up to date and a further exchange of logical positions decides — and the pivot row's entries
//     for the elimination came with the same load;
//   * the owner of the pivot row does the global bookkeeping (rowstep, pivrow, the control block's step list, the U row's stores).
// Same arithmetic, same step order, same round structure as the one-workgroup panel (k_luc_usolve / k_luc_trail follow unchanged): the
// schedules are compared bit for bit (tests/test_gpu_parity.py).  Waits are bounded: a workgroup that runs out of patience raises the
// control block's fault flag, every workgroup leaves, and the host repeats the factorization with the one-workgroup panel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bt_loop.h"
#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

namespace {

constexpr int kLxSlots = 24;          // slots per record: 3 header values + up to 16 + 5 spare
constexpr int kLxHeader = 16;         // doubles in front of the records: [0] = exchanges completed so far
constexpr int kLxSpinLimit = 400000;  // polls (~1 us each)

__device__ __forceinline__ void lx_load3(const xpair *p0, const xpair *p1, const xpair *p2, xpair (&v)[3]) {
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

__device__ __forceinline__ void lx_load3_fast(const xpair *p0, const xpair *p1, const xpair *p2, xpair (&v)[3]) {   // (never served by L1: the XCD's L2 is the coherence point of its CUs)
    asm volatile("global_load_dwordx4 %0, %3, off nt\n\tglobal_load_dwordx4 %1, %4, off nt\n\tglobal_load_dwordx4 %2, %5, off nt\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

}  // namespace

#ifdef GOMILP_DEBUG
__device__ unsigned long long g_lux_stamps[4 * 16];
#define LUX_STAMP(S)                                                                      \
    do {                                                                                  \
        unsigned long long t_;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if ((S) >= 0) tacc[(S) >= 0 ? (S) : 0] += t_ - tprev;                             \
        tprev = t_;                                                                       \
    } while (0)
#else
#define LUX_STAMP(S) do { } while (0)
#endif

template <int G, int NB, int SMAX>
__global__ __launch_bounds__(256) void k_luc_panel_x(LUArgs a, int32_t *__restrict__ pivrow, xpair *__restrict__ xrec) {
    constexpr int T = 256, NW = 4, MAXM = 256 * G;
    static_assert(NB == 16 && SMAX <= 32 && G * kLxSlots <= 192, "16 slots in one register tuple; a poll is three loads per lane");
    typedef unsigned short idx_t;
    typedef double vec __attribute__((ext_vector_type(NB)));
    constexpr idx_t NONE = 0xFFFF;
    if (blockIdx.x & 7) return;   // 8 G blocks are launched: blocks 0, 8, 16, ... land on one XCD (round-robin deal), the rest leave at once
    const int g = (int)blockIdx.x >> 3;
    __shared__ idx_t s_lpos[MAXM];
    __shared__ idx_t s_rowat[MAXM];
    __shared__ idx_t s_unit[MAXM];
    __shared__ idx_t s_ucol[MAXM];
    __shared__ unsigned char s_active[MAXM];
    __shared__ double redM[2][NW];
    __shared__ unsigned int redL[2][NW];
    __shared__ int s_slotcol[NB];
    __shared__ int s_nload, s_stop, s_limit, s_sigma;
    __shared__ __attribute__((aligned(16))) double s_post[NW][kLxSlots];   // a wave's candidate hands its record to the wave's lanes: one store instruction posts it
    if (a.ctl_base->fault) return;
    LUCtl *ctl = a.ctl;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = a.m;
    const size_t ldw = (size_t)a.ldw;
    const int k0 = a.ctl_prev->k_next;
    if (k0 >= m) {
        if (g == 0 && tid == 0) { ctl->nsteps = 0; ctl->ndrop = 0; ctl->nnext = 0; ctl->k_next = k0; ctl->k0 = k0; ctl->k1 = k0; ctl->rounds = a.ctl_prev->rounds; ctl->ksync = a.ctl_prev->ksync; }
        return;
    }
    for (int R = tid; R < MAXM; R += T) {
        const bool in = R < m;
        s_lpos[R] = (idx_t)(in ? a.lpos[R] : R);
        s_active[R] = (in && a.rowstep[R] < 0) ? 1 : 0;
        const int ur = (in && a.unit_row) ? a.unit_row[R] : -1;
        s_unit[R] = ur < 0 ? NONE : (idx_t)ur;
        s_ucol[R] = NONE;
    }
    __syncthreads();
    for (int R = tid; R < m; R += T) {
        s_rowat[s_lpos[R]] = (idx_t)R;
        if (s_unit[R] != NONE && R >= k0) s_ucol[s_unit[R]] = (idx_t)R;   // column R (still to come) is the unit vector of row s_unit[R]
    }
    for (;;) {   // (two unit columns with the same row — a singular basis —: the FIRST of them is the one whose step retires the row)
        __syncthreads();
        bool again = false;
        for (int R = tid; R < m; R += T)
            if (s_unit[R] != NONE && R >= k0 && R < (int)s_ucol[s_unit[R]]) { s_ucol[s_unit[R]] = (idx_t)R; again = true; }
        if (!__syncthreads_or(again)) break;
    }
    if (w == 0) {
        int n = 0;
        for (int base = k0; base < m && n < NB; base += 64) {
            const int k = base + lane;
            bool dense = false;
            if (k < m) {
                const idx_t ur = s_unit[k];
                dense = ur == NONE || !s_active[ur];
            }
            const unsigned long long mask = __ballot(dense);
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (dense && n + rank < NB) s_slotcol[n + rank] = k;
            n += __popcll(mask);
        }
        if (lane == 0) s_nload = n < NB ? n : NB;
    }
    __syncthreads();
    const int nload = s_nload;
    int myslotcol = (lane < NB && lane < nload) ? s_slotcol[lane < NB ? lane : 0] : 0x7FFFFFFF;
    int myslotin = -1;
    unsigned int live = nload >= 32 ? 0xFFFFFFFFu : ((1u << nload) - 1u);
    const int R = g * T + tid;   // this lane's row
    bool act = (R < m) && s_active[R < m ? R : 0];
    const idx_t uc0 = s_ucol[R < m ? R : 0];
    const int myucol = (act && uc0 != NONE) ? (int)uc0 : 0x7FFFFFFF;   // the step that retires this row unless a dense step takes it first
    vec v;
    {
        const double *src = a.W + (act ? R : 0);
#pragma unroll
        for (int c = 0; c < NB; c++) v[c] = (act && c < nload) ? src[(size_t)__builtin_amdgcn_readlane(myslotcol, c) * ldw] : 0.0;
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the column loads are retired in front of the loop (see k_luc_panel_slots)
    // exchanges: sequence numbers go on from launch to launch (header slot 0: written by workgroup 0 at the end of a launch)
    unsigned int myxcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(myxcc));
    myxcc &= 0xFu;
    bool fast = false;   // SAFE (sc1) accesses until an exchange has shown that all G workgroups sit on one XCD (slot 20 of a record: the XCC id)
    const double seq0 = xrec[0][0];
    xpair *recs = xrec + kLxHeader / 2;   // [2 parities][G][kLxSlots]
    int nx = 0;
#ifdef GOMILP_DEBUG
    unsigned long long tacc[16] = {}, tprev = 0;
#endif
    LUX_STAMP(-1);
    int kcur = k0, s = 0, k1 = m;
    int ksync = a.ctl_prev->ksync;   // the index maps (lpos / rowat) hold the interchanges of the steps < ksync: the rest waits in the log (pivrow)
    bool fault = false;
#pragma unroll 1
    for (;;) {
        if (w == 0) {
            const unsigned int key = (lane < NB && ((live >> lane) & 1u)) ? (unsigned int)myslotcol : 0x7FFFFFFFu;
            unsigned int mn = row_min_u32(key);
            mn = (unsigned int)__builtin_amdgcn_readlane((int)mn, 0);
            const bool listed = live != 0 && s < SMAX;
            const int limit = listed ? (int)mn : m;
            const unsigned long long hit = __ballot(key == mn && lane < NB);
            const int sigma = hit ? (int)__builtin_ctzll(hit) : 0;
            // run of bookkeeping steps [kcur, limit): only its END is needed on the chain — the first column whose step needs arithmetic.  A row
            // knows the step that retires it (myucol); LAPACK's logical row order — all the index maps are for — matters only when two rows tie in
            // a pivot search, and then every workgroup replays the interchanges since the last replay from the log of pivot rows (below).  With
            // ONE wave per SIMD the serial replay (1.8 k cycles per dense step) was on the chain: nobody to hide behind (section 2.3, "Round 5")
            int k = kcur;
            for (;;) {
                const int kk = k + lane;
                const idx_t ur = kk < limit ? s_unit[kk] : NONE;
                const bool triv = ur != NONE && s_active[ur] && (int)s_ucol[ur] == kk;   // (a second unit column of the same row finds it used: arithmetic)
                const unsigned long long nt = __ballot(!triv);
                const int cnt = nt ? (int)__builtin_ctzll(nt) : 64;
                k += cnt;
                if (cnt < 64) break;
            }
            if (lane == 0) { s_stop = k; s_limit = limit; s_sigma = sigma; }
        }
        LUX_STAMP(0);   // wave 0: next column + the run
        __syncthreads();
        LUX_STAMP(1);
        const int kstop = s_stop, limit = s_limit;
        const int sigma = __builtin_amdgcn_readfirstlane(s_sigma) & (NB - 1);
        // the rows of the run leave the active set on every workgroup's copy of the map (every column of the run is the unit column of an active row)
        for (int c = kcur + tid; c < kstop; c += T) s_active[s_unit[c]] = 0;
        // this lane's row retired by the run (the step of its unit column lies in it): its entries in the listed columns are final U entries
        if (act && myucol >= kcur && myucol < kstop) {
            const int kt = myucol;
            a.rowstep[R] = kt; pivrow[kt] = R;
            double *dst = a.W + R;
#pragma unroll
            for (int c = 0; c < NB; c++)
                if ((live >> c) & 1u) dst[(size_t)__builtin_amdgcn_readlane(myslotcol, c) * ldw] = v[c];
            act = false;
        }
        if (kstop < limit || live == 0 || s >= SMAX) { k1 = kstop; break; }
        const int k = limit;
        LUX_STAMP(2);
        // ---- dense step k on slot sigma: this workgroup's candidate
        const double x = v[sigma];
        const double xm = act ? -fabs(x) : __builtin_inf();
        // one exchange: the wave that holds the workgroup's candidate hands the record {h0, h1, row, 1 / a_ik, XCC id | the row's entries in the NB
        // slots} to its lanes (LDS, no barrier: one wave), which post it with ONE store instruction; every wave then polls the G records —
        // lane l reads the slots (l & 7), + 8, + 16 of record l >> 3
        xpair rv[3];
        auto exchange = [&](bool poster, double h0, double h1) {
            const double seq = seq0 + (double)(nx + 1);
            xpair *mine = recs + ((size_t)((nx + 1) & 1) * G + g) * kLxSlots;
            if (__any(poster)) {   // (uniform per wave)
                double *sp = s_post[w];
                if (poster) {
                    sp[0] = h0; sp[1] = h1; sp[2] = (double)R;
                    sp[3] = act ? 1.0 / x : 0.0;   // dgetf2.go:54-56 scales by the reciprocal
#pragma unroll
                    for (int c = 0; c < NB; c++) sp[4 + c] = v[c];
                    sp[20] = (double)myxcc;
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the candidate's LDS writes have landed (same wave: in order)
                if (lane < 21) xstore(mine + lane, xpair{seq, sp[lane]}, fast);
            }
            const xpair *base = recs + (size_t)((nx + 1) & 1) * G * kLxSlots;
            const int g2 = lane >> 3, j0 = lane & 7;
            const bool mineok = g2 < G;
            const xpair *q0 = base + (size_t)(mineok ? g2 : 0) * kLxSlots + j0;
            for (int it = 0;; it++) {
                if (fast) lx_load3_fast(q0, q0 + 8, q0 + 16, rv); else lx_load3(q0, q0 + 8, q0 + 16, rv);
                const bool ok = !mineok || (rv[0][0] == seq && rv[1][0] == seq && (j0 + 16 > 20 || rv[2][0] == seq));   // the slots a record uses: 0 .. 20
                if (__all(ok)) break;
                if (it >= kLxSpinLimit) { fault = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            nx++;
        };
        const bool head = (lane & 7) == 0 && (lane >> 3) < G;   // a record's header sits in the lanes 8 g2 .. 8 g2 + 4
        // ---- this workgroup's candidate: its maximum, how many of its rows attain it, and one of them (the smallest physical row)
        const double wm = wave_min_f64(xm);
        const bool hitw = act && xm == wm;
        const unsigned int cntw = (unsigned int)__popcll(__ballot(hitw));
        unsigned int lk = hitw ? (unsigned int)R : 0xFFFFFFFFu;
        lk = row_min_u32(lk);
        lk = min(min((unsigned int)__builtin_amdgcn_readlane((int)lk, 15), (unsigned int)__builtin_amdgcn_readlane((int)lk, 31)),
                 min((unsigned int)__builtin_amdgcn_readlane((int)lk, 47), (unsigned int)__builtin_amdgcn_readlane((int)lk, 63)));
        double *rm = redM[s & 1];
        unsigned int *rl = redL[s & 1];
        if (lane == 0) { rm[w] = wm; rl[w] = (min(cntw, 2u) << 16) | (lk & 0xFFFFu); }
        LUX_STAMP(3);   // own search
        __syncthreads();
        LUX_STAMP(4);
        const double bx = lane < NW ? rm[lane] : __builtin_inf();
        const double bml = readlane_f64(row_min_f64(bx), 15);
        const unsigned int infol = (lane < NW && bx == bml) ? rl[lane] : 0u;
        const unsigned int cntl = (unsigned int)__popcll(__ballot((infol >> 16) >= 1u)) + (unsigned int)__popcll(__ballot((infol >> 16) >= 2u));   // (0 without an active row)
        const unsigned int rowl = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32((lane < NW && (infol >> 16) >= 1u) ? (infol & 0xFFFFu) : 0xFFFFFFFFu), 15);
        exchange(cntl ? (act && (unsigned int)R == rowl) : (tid == 0), cntl ? bml : __builtin_inf(), (double)min(cntl, 2u));
        if (fault) break;
        LUX_STAMP(5);   // local pick + post + poll
        if (!fast) {   // every record carries its workgroup's XCC id: all equal -> the XCD's L2 is the coherence point, plain stores / nt loads from here on
            const double xc = __shfl(rv[2][1], (lane & ~7) + 4);   // slot 20 = 4 + 16
            fast = __all(!head || xc == (double)myxcc);
        }
        const double wmL = head ? rv[0][1] : __builtin_inf();
        const double bm = wave_min_f64(wmL);
        const double cnL = __shfl(rv[0][1], (lane & ~7) + 1);
        const unsigned long long at1 = __ballot(head && wmL == bm && cnL >= 1.0), at2 = __ballot(head && wmL == bm && cnL >= 2.0);
        int gw;
        if (__popcll(at1) == 1 && at2 == 0) {
            gw = (int)__builtin_ctzll(at1) >> 3;   // ONE row in the whole panel attains the maximum: the pivot row, whatever its logical position
        } else {
            // several rows with the same |a_ik| (or none): dgetf2.go:38 takes the first in LAPACK's logical row order.  (1) an exchange behind
            // which every workgroup's stores to the log have landed; (2) every workgroup replays the interchanges of the steps [ksync, k) on its
            // copy of the maps — the pivot row of step j is pivrow[j], whoever performed it (dlaswp.go); (3) the positions decide: a second
            // exchange of {smallest logical position among the workgroup's tied rows, ...}
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            exchange(tid == 0, 0.0, 0.0);
            if (fault) break;
            if (w == 0) {
                for (int j0 = ksync; j0 < k; j0 += 64) {
                    const int jj = j0 + lane;
                    const int pj = jj < k ? __hip_atomic_load(&pivrow[jj], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
                    const int nrep = min(64, k - j0);
                    for (int t = 0; t < nrep; t++) {
                        const int Pj = __builtin_amdgcn_readlane(pj, t);
                        if (lane == 0) {
                            const idx_t jq = s_lpos[Pj], Q = s_rowat[j0 + t];   // dlaswp.go: the row at logical j trades places with the pivot row
                            s_lpos[Q] = jq; s_rowat[jq] = Q;
                            s_lpos[Pj] = (idx_t)(j0 + t); s_rowat[j0 + t] = (idx_t)Pj;
                        }
                    }
                }
            }
            ksync = k;
            __syncthreads();
            const bool tied = act && xm == bm;
            unsigned int lp = tied ? (unsigned int)s_lpos[R] : 0xFFFFFFFFu;
            const unsigned int lpmine = lp;
            lp = row_min_u32(lp);
            lp = min(min((unsigned int)__builtin_amdgcn_readlane((int)lp, 15), (unsigned int)__builtin_amdgcn_readlane((int)lp, 31)),
                     min((unsigned int)__builtin_amdgcn_readlane((int)lp, 47), (unsigned int)__builtin_amdgcn_readlane((int)lp, 63)));
            unsigned int *rt = redL[(s & 1) ^ 1];
            if (lane == 0) rt[w] = lp;
            __syncthreads();
            const unsigned int lpl = (unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(lane < NW ? rt[lane] : 0xFFFFFFFFu), 15);
            const bool any_l = lpl != 0xFFFFFFFFu;
            exchange(any_l ? (tied && lpmine == lpl) : (tid == 0), any_l ? (double)lpl : __builtin_inf(), 0.0);
            if (fault) break;
            const double lpL = head ? rv[0][1] : __builtin_inf();
            const double blp = wave_min_f64(lpL);
            const unsigned long long won = __ballot(head && lpL == blp);
            gw = won ? ((int)__builtin_ctzll(won) >> 3) : 0;
        }
        const int P = (int)readlane_f64(rv[0][1], 8 * gw + 2);
        const double rinv = readlane_f64(rv[0][1], 8 * gw + 3);
        // lane c: the pivot row's value in slot c = record slot 4 + c: lane 8 gw + ((4 + c) & 7), register (4 + c) >> 3
        double prl;
        {
            const int c = lane & (NB - 1);
            const int src = 8 * gw + ((4 + c) & 7), rr = (4 + c) >> 3;
            const double b0 = __shfl(rv[0][1], src), b1 = __shfl(rv[1][1], src), b2 = __shfl(rv[2][1], src);
            prl = rr == 0 ? b0 : (rr == 1 ? b1 : b2);
        }
        LUX_STAMP(6);   // pick
        const bool owner = act && R == P;
        // the pivot row leaves the active set on every workgroup's copy of the map; the interchange itself (dlaswp.go) waits in the log
        if (tid == 0 && P >= 0 && P < MAXM) s_active[P] = 0;
        if (owner) {
            act = false;
            a.rowstep[P] = k; pivrow[k] = P;
            if (a.dense_flag) a.dense_flag[k] = 1;
            ctl->steps[s] = k; ctl->prow[s] = P;
            double *dst = a.W + R;   // the pivot row's U entries (slot sigma holds column k)
#pragma unroll
            for (int cc = 0; cc < NB; cc++)
                if ((live >> cc) & 1u) dst[(size_t)__builtin_amdgcn_readlane(myslotcol, cc) * ldw] = v[cc];
        }
        const idx_t uc = (P >= 0 && P < MAXM) ? s_ucol[P] : NONE;
        const int k2 = (uc != NONE && (int)uc > k) ? (int)uc : -1;   // taking row P makes its unit column (if still to come) dense
        LUX_STAMP(7);   // pick + bookkeeping
        const double piv = readlane_f64(prl, sigma);
        const bool singular = (piv == 0);  // dgetf2.go:48-49
        if (singular && owner) a.st->lu_singular = 1;
        double *wcol = a.W + (size_t)k * ldw;
        double *lcol = a.Lp + (size_t)s * ldw;
        double nl = 0.0;
        if (act) {
            const double l = singular ? x : __dmul_rn(x, rinv);
            wcol[R] = l;
            nl = singular ? 0.0 : -l;
            lcol[R] = nl;
        } else if (R < a.ldw) lcol[R] = 0.0;
        const unsigned int others = live & ~(1u << sigma);
        if (!singular && __any(act)) {
            const double prz = (lane < NB && ((others >> (lane & 31)) & 1u)) ? prl : 0.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const double pc = readlane_f64(prz, c);
                v[c] = __dadd_rn(__dmul_rn(nl, pc), v[c]);
            }
        }
        if (k2 >= 0) {
            const double vn = (act && !singular) ? __dadd_rn(__dmul_rn(nl, 1.0), 0.0) : 0.0;
            v[sigma] = vn;
            if (lane == sigma) { myslotcol = k2; myslotin = k; }
        } else live = others;
        s++;
        kcur = k + 1;
        LUX_STAMP(8);   // elimination
    }
#ifdef GOMILP_DEBUG
    if (g == 0 && lane == 0) {
        for (int sg = 0; sg < 9; sg++) atomicAdd(&g_lux_stamps[w * 16 + sg], tacc[sg]);
        if (w == 0) atomicAdd(&g_lux_stamps[15], (unsigned long long)s);
    }
#endif
    if (fault) {
        if (lane == 0) a.ctl_base->fault = 1;
        return;
    }
    if (g != 0) return;
    for (int R2 = tid; R2 < m; R2 += T) a.lpos[R2] = s_lpos[R2];
    int ndl = 0;
    if (w == 0) {
        if (lane == 0) ctl->nnext = 0;
        const bool on = lane < NB && ((live >> lane) & 1u);
        const unsigned long long msk = __ballot(on);
        ndl = __popcll(msk);
        if (on) {
            const int at = __popcll(msk & ((1ull << lane) - 1ull));
            ctl->dropcol[at] = myslotcol; ctl->dropin[at] = myslotin; ctl->dropout[at] = k1;
        }
    }
    if (tid == 0) {
        ctl->k0 = k0; ctl->k1 = k1; ctl->k_next = k1; ctl->nsteps = s; ctl->ndrop = ndl;
        ctl->rounds = a.ctl_prev->rounds + 1; ctl->ksync = ksync;
        xrec[0] = xpair{seq0 + (double)nx, 0.0};
    }
}

// once every step is done a row's logical position is the step that took it — what the solves and the host read (the panel's maps hold the
// interchanges up to its last replay only)
__global__ void k_luc_lpos_final(LUArgs a) {
    const int R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R < a.m) a.lpos[R] = a.rowstep[R];
}
void launch_luc_lpos_final(const LUArgs &a, hipStream_t s) { hipLaunchKernelGGL(k_luc_lpos_final, dim3((a.m + 255) / 256), dim3(256), 0, s, a); }

// ---- host side
size_t luc_cross_doubles() { return (size_t)kLxHeader + 2 * 2 * 8 * kLxSlots; }   // header + two parities of eight records (xpairs = 2 doubles)
int luc_cross_groups(int m, int want) {   // workgroups for a basis of m rows (0: not this schedule)
    if (want <= 0 || m < 64) return 0;
    const int need = (m + 255) / 256;
    if (need > 8) return 0;
    return need <= 2 ? 2 : (need <= 4 ? 4 : 8);
}
template <int G>
static void luc_cross_panel(const LUArgs &a, int32_t *pivrow, double *xrec, hipStream_t s) {
    hipLaunchKernelGGL((k_luc_panel_x<G, 16, 32>), dim3(8 * G), dim3(256), 0, s, a, pivrow, reinterpret_cast<xpair *>(xrec));
}
void launch_luc_cross_panel(const LUArgs &a, int32_t *pivrow, double *xrec, int G, hipStream_t s) {
    if (G == 2) luc_cross_panel<2>(a, pivrow, xrec, s);
    else if (G == 4) luc_cross_panel<4>(a, pivrow, xrec, s);
    else luc_cross_panel<8>(a, pivrow, xrec, s);
}
#ifdef GOMILP_DEBUG
void lux_stamps_read(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lux_stamps), sizeof(unsigned long long) * 64); }
#endif

}  // namespace gomilp
