// Compressed gonum-order LU for the final basis solve (gfx950) — the default schedule.
//
// Arithmetic: identical, element by element, to lu_kernels.hip and to lapack/gonum/dgetrf.go:29-70 / dgetf2.go:30-69
// (a_ij = (-l_ik)*u_kj + a_ij in ascending k as a rounded multiply then a rounded add, l_ik = a_ik*(1/a_kk), pivot =
// first max |a_ik| in LAPACK's logical row order, zero multipliers skipped like blas/gonum dgemm.go / dtrsm).
//
// Schedule: most columns of a B&B basis are slack (unit) columns.  A unit column e_r whose row r has not been a pivot
// row yet makes its elimination step pure bookkeeping (pivot exactly 1, multipliers exactly 0, one row interchange);
// only the other steps ("dense" steps) do arithmetic.  The blocked schedule of lu_kernels.hip still spends one
// register column and two workgroup barriers on every step.  Here a ROUND is
//   k_luc_panel_slots
//                 ONE workgroup: registers hold the next NB columns that are KNOWN to be dense (non-unit, or unit with
//                 a used row), wherever they are, each in a fixed slot; between two of them wave 0 replays the run of
//                 bookkeeping steps on LDS-resident index maps (lpos / rowat / active).  A unit column that BECOMES dense
//                 inside the round (a dense step took its row: the column is -l of that step) takes the slot the step's own
//                 column leaves.  (Rounds 1-3 kept the columns sorted in registers and shifted them every step: retired in
//                 round 4, see the comment at the kernel.)
//   k_luc_usolve  finishes the dense pivot rows right of the round (columns >= k1),
//   k_luc_trail   applies the round's dense steps to every other row right of the round: all of them for rows that
//                 are still active, the steps before its own for a row retired by a bookkeeping step of the round.
// Rounds are data dependent, so the kernels take their step range from a device control block (LUCtl) and the host
// enqueues rounds in batches until k_next == m.
// m = 2048 metric basis: 2048 steps, 384 dense -> 23 rounds instead of 128 panels.
// Two schedules of the same rounds (knob lu_blocked): the PLAIN one launches the three kernels above one after the other; the
// LOOK-AHEAD one (default beyond 768 rows, luc_role below) is ONE launch per round — workgroup 0 the panel of round r, the other
// workgroups the U-solve and update of round r-1 beside it, the columns the panel loads first.  Small bases (<= 128 rows) bring
// everything the host needs home in one block (k_luc_pack_small).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

// ---- slot form of the panel (round 4; default) -------------------------------------------------------------------------------
// Same arithmetic, same step order as the panel of rounds 1-3, another register discipline.  That panel kept its NB columns SORTED in registers: every
// dense step shifts all of them by one (the update folded into the shift behind 64-bit selects), and a column that becomes dense
// inside the round is inserted at its sorted position through a select chain over all NB registers — ~390 VALU instructions per
// wave and step, 4 waves per SIMD: the panel was issue-bound on ONE CU (4.7 us per dense step at 2048 rows, 20 % of the metric
// solve).  Here a column lives in a fixed SLOT for as long as it is listed:
//   * the step's column is the live slot with the smallest column index (wave 0: one DPP minimum over <= 32 lanes); its register is
//     read with a UNIFORM dynamic index (v_movrels: the slot number is the same in every lane);
//   * only the live slots are updated (uniform branch per slot: 1 LDS broadcast read + 2 VALU per row), nothing is shifted;
//   * the slot of the step's own column is free afterwards, so a unit column that just became dense ALWAYS finds a slot (no
//     dropping), and a round goes on for up to SMAX dense steps instead of ending when the first NB columns are used up;
//   * columns still live when the round ends (step cap, or the run met a dense column that is not listed) are handed to the
//     trailing kernels through the record the old panel used for dropped columns: rows that left while the column was listed hold
//     final values in it, all other rows still hold the original.
// Rows of retired rows keep whatever the later updates make of them (never read again): no predication on the update itself.
#ifdef GOMILP_DEBUG
// diagnostic flavour: cycles (s_memtime) per segment of a dense step, summed over the launches of a process by wave 0 .. 3:
// [wave][segment], segment 15 = dense steps; gomilp_debug_luc_stamps() hands them out
__device__ unsigned long long g_luc_stamps[4 * 16];
#define LUC_STAMP(S)                                                                      \
    do {                                                                                  \
        unsigned long long t_;                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
        __builtin_amdgcn_sched_barrier(0);                                                \
        if ((S) >= 0) tacc[(S) >= 0 ? (S) : 0] += t_ - tprev;                             \
        tprev = t_;                                                                       \
    } while (0)
#else
#define LUC_STAMP(S) do { } while (0)
#endif
constexpr int kLucSpinLimit = 200000;   // polls (1.5 us each and more: seconds, against the milliseconds another kernel can hold the device)
__device__ __forceinline__ bool luc_spin(const uint32_t *p, uint32_t target) {
    for (int it = 0; it < kLucSpinLimit; it++) {
        const uint32_t v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}
__device__ __forceinline__ double luc_ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void luc_st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- the trailing update of a round, cell by cell (look-ahead schedule) ------------------------------------------------------------
// a[R][j] += sum_s Lp[s][R] * Up[s][j] in ascending s, zero multipliers skipped — the arithmetic of k_luc_trail below, with the
// round's panels read straight from L2 (no staging, no barrier): the body of every 256-thread group of the workgroups that run
// BESIDE the next round's panel (luc_role, phase T) — hidden behind the panel's steps, so the efficiency of these loads does not matter.
// 4 x 4 cells per thread: rows R0 + 4 tx .. (consecutive lanes walk down a column), columns by `colj`.
template <int NB, int UNR>
__device__ __forceinline__ void luc_trail_cells(const LUArgs &a, const LUCtl *__restrict__ c, const double *__restrict__ Lp,
                                                const double *__restrict__ Up, const int32_t *__restrict__ rowstep, int ns, int k0,
                                                const int (&colj)[4], int R0, int tx) {
    const int m = a.m;
    const size_t ldw = (size_t)a.ldw;
    const int Rb = R0 + tx * 4;
    int rs[4], cls[4];
    bool anyrow = false;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int R = Rb + rr;
        rs[rr] = R < m ? rowstep[R] : 0;
        cls[rr] = R < m ? (rs[rr] < 0 ? 1 : (rs[rr] >= k0 ? 2 : 0)) : 0;   // 1 active, 2 left during the round, 0 finished earlier
        anyrow = anyrow || cls[rr] > 0;
    }
    bool anycol = false;
#pragma unroll
    for (int cc = 0; cc < 4; cc++) anycol = anycol || colj[cc] >= 0;
    if (!anyrow || !anycol) return;
    int din[4] = {0, 0, 0, 0}, dout[4] = {0, 0, 0, 0};   // rows that left at steps [din, dout) are final in the column
    const int nd = c->ndrop;
    for (int t = 0; t < nd; t++) {
        const int dc = c->dropcol[t], di = c->dropin[t], dq = c->dropout[t];
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
            if (dc == colj[cc]) { din[cc] = di; dout[cc] = dq; }
    }
    double acc[4][4];   // [cc][rr]
    bool live[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            live[cc][rr] = cls[rr] > 0 && colj[cc] >= 0 && !(cls[rr] == 2 && rs[rr] >= din[cc] && rs[rr] < dout[cc]);
            acc[cc][rr] = live[cc][rr] ? a.W[(size_t)colj[cc] * ldw + Rb + rr] : 0.0;
        }
    }
#pragma unroll UNR
    for (int s = 0; s < ns; s++) {   // (UNR steps' loads in flight: 4 where the registers are there, 2 beside a 1024-thread panel)
        double l[4], u[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) l[rr] = (Rb + rr) < m ? Lp[(size_t)s * ldw + Rb + rr] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) u[cc] = colj[cc] >= 0 ? luc_ld_agent(&Up[(size_t)s * ldw + colj[cc]]) : 0.0;   // (written by phase S of this launch)
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const bool nz = l[rr] != 0;
#pragma unroll
            for (int cc = 0; cc < 4; cc++) acc[cc][rr] = nz ? __dadd_rn(__dmul_rn(l[rr], u[cc]), acc[cc][rr]) : acc[cc][rr];
        }
    }
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
            if (live[cc][rr]) a.W[(size_t)colj[cc] * ldw + Rb + rr] = acc[cc][rr];
    }
}

// ---- look-ahead schedule: ONE launch per round -----------------------------------------------------------------------------------
// Workgroup 0 is the panel of round r.  The other workgroups finish round r-1 beside it, from the record round r-1 left (control block,
// multiplier panel, rowstep snapshot: all by round parity, the running panel writes the other parity):
//   phase U  (workgroups 1 .. nrt, 64 rows each): U-solve + update of the columns panel r LOADS (ctl_prev->next) — the only part of
//            the update on the critical path; the panel sets up its index maps meanwhile and waits for cnt_u before it loads.
//   phase S  (the same workgroups): the U-solve of one 64-column tile of all other columns -> Up_prev, then cnt_s.
//   phase T  (every workgroup, behind cnt_s): the update of all other columns, hidden behind the panel's steps.  Not touched: the
//            columns of phase U, and the unit columns of rows that were still active when round r-1 ended — the running panel may take
//            such a row, build the column in registers and write it, while the round's U rows are exactly zero in it (the update
//            would add l * 0 to every cell).
// What crosses workgroups INSIDE the launch (the cells of phase U -> the panel's loads, Up_prev -> phase T) moves with agent-scope
// stores / loads and an arrival counter, as in the loop kernels (bt_loop.h); everything else crossed a launch boundary.  Waits are
// bounded: a workgroup that runs out of patience raises ctl_base->fault, every later launch returns at once, and the host repeats the
// factorization with the plain schedule.
template <int T, int NB>
__device__ void luc_role(const LUArgs &a) {
    static_assert(T % 256 == 0 && NB == 32, "groups of 256 threads; 32 steps per round");
    const LUCtl *__restrict__ c = a.ctl_prev;
    LUCtl *base = a.ctl_base;
    const int b = (int)blockIdx.x - 1;
    const int tid = threadIdx.x;
    const int m = a.m, nrt = (m + 63) / 64;
    const size_t ldw = (size_t)a.ldw;
    const int ns = c->nsteps, k0 = c->k0, k1 = c->k1, nn = c->nnext, nd = c->ndrop;
    const bool work = ns > 0 && k1 < m;
    // phases U and S share the staging area (S starts behind U's last barrier)
    __shared__ double s_ln[NB][NB + 1];
    __shared__ double s_xa[2 * NB * 33];      // U: X[s][33] (pivot rows x listed columns) + Us[s][33]; S: X[s][65]
    static_assert(2 * NB * 33 >= NB * 65, "staging area of phase S");
    __shared__ double s_ls[NB][64];
    __shared__ int Ps[NB], Ss[NB], Dc[NB], Di[NB], Do[NB], Nx[NB];
    __shared__ int s_go;
    const double *__restrict__ Lp = a.Lp_prev;
    if (b < nrt) {
        if (work) {
            if (tid < NB) {
                Ps[tid] = tid < ns ? c->prow[tid] : 0;
                Ss[tid] = tid < ns ? c->steps[tid] : 0x7fffffff;
                Dc[tid] = tid < nd ? c->dropcol[tid] : -1;
                Di[tid] = tid < nd ? c->dropin[tid] : 0;
                Do[tid] = tid < nd ? c->dropout[tid] : 0;
                Nx[tid] = tid < nn ? c->next[tid] : -1;
            }
            __syncthreads();
            for (int idx = tid; idx < NB * NB; idx += T) {
                const int s2 = idx / NB, t = idx % NB;
                s_ln[s2][t] = (s2 < ns && t < s2) ? Lp[(size_t)t * ldw + Ps[s2]] : 0.0;
            }
        }
        // ---- phase U
        // The U-solve of the listed columns reads the round's pivot rows in them, and the update WRITES those cells (a pivot row's
        // update is its U-solve): ONE workgroup solves, before anybody writes, and hands the U rows over through Up_prev (otherwise
        // unused in these columns) and cnt_x; a workgroup that starts late — other kernels on the device — finds everything it needs.
        bool ok = true;
        if (work && nn > 0) {
            double (*X)[33] = reinterpret_cast<double (*)[33]>(s_xa);
            double (*Us)[33] = reinterpret_cast<double (*)[33]>(s_xa + NB * 33);
            double *Upw = const_cast<double *>(a.Up_prev);
            const int R0 = b * 64;
            for (int idx = tid; idx < NB * 64; idx += T) {
                const int s2 = idx / 64, r = idx % 64;
                s_ls[s2][r] = (s2 < ns && R0 + r < m) ? Lp[(size_t)s2 * ldw + R0 + r] : 0.0;
            }
            if (b == 0) {
                for (int idx = tid; idx < NB * 32; idx += T) {
                    const int s2 = idx / 32, ci = idx % 32;
                    X[s2][ci] = (s2 < ns && ci < nn) ? a.W[(size_t)Nx[ci] * ldw + Ps[s2]] : 0.0;
                }
                __syncthreads();
                if (tid < nn) {   // the U-solve of k_luc_usolve, one listed column per thread
                    const int j = Nx[tid];
                    int din = 0, dout = 0;
                    for (int t = 0; t < nd; t++)
                        if (Dc[t] == j) { din = Di[t]; dout = Do[t]; }
                    double u[NB];
#pragma unroll
                    for (int s2 = 0; s2 < NB; s2++) {
                        double x = 0;
                        if (s2 < ns) {
                            x = X[s2][tid];
                            if (!(Ss[s2] >= din && Ss[s2] < dout)) {
#pragma unroll
                                for (int t = 0; t < s2; t++) {
                                    const double l = s_ln[s2][t];
                                    x = (l != 0) ? __dadd_rn(__dmul_rn(l, u[t]), x) : x;
                                }
                            }
                            luc_st_agent(Upw + (size_t)s2 * ldw + j, x);
                        }
                        u[s2] = x;
                        Us[s2][tid] = x;
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#ifdef GOMILP_DEBUG
                if (a.pad3) ok = false;   // fault injection: this workgroup never arrives, the others run out of patience
                else
#endif
                if (tid == 0) __hip_atomic_fetch_add(&base->cnt_x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (tid == 0) s_go = luc_spin(&base->cnt_x, (uint32_t)(a.round + 1)) ? 1 : 0;
                __syncthreads();
                ok = s_go != 0;
                if (ok) {
                    for (int idx = tid; idx < NB * 32; idx += T) {
                        const int s2 = idx / 32, ci = idx % 32;
                        Us[s2][ci] = (s2 < ns && ci < nn) ? luc_ld_agent(Upw + (size_t)s2 * ldw + Nx[ci]) : 0.0;
                    }
                } else if (tid == 0) base->fault = 1;
                __syncthreads();
            }
            if (ok && tid < 128) {   // 64 rows x 32 listed columns, 4 x 4 cells per thread
                const int tx = tid & 15, ty = tid >> 4;
                const int Rb = R0 + tx * 4;
                int rs[4], cls[4], colj[4], din[4] = {0, 0, 0, 0}, dout[4] = {0, 0, 0, 0};
#pragma unroll
                for (int rr = 0; rr < 4; rr++) {
                    const int R = Rb + rr;
                    rs[rr] = R < m ? a.rowsnap_prev[R] : 0;
                    cls[rr] = R < m ? (rs[rr] < 0 ? 1 : (rs[rr] >= k0 ? 2 : 0)) : 0;
                }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) colj[cc] = Nx[ty * 4 + cc];
                for (int t = 0; t < nd; t++) {
#pragma unroll
                    for (int cc = 0; cc < 4; cc++)
                        if (Dc[t] == colj[cc]) { din[cc] = Di[t]; dout[cc] = Do[t]; }
                }
                double acc[4][4];
                bool live[4][4];
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++) {
                        live[cc][rr] = cls[rr] > 0 && colj[cc] >= 0 && !(cls[rr] == 2 && rs[rr] >= din[cc] && rs[rr] < dout[cc]);
                        acc[cc][rr] = live[cc][rr] ? a.W[(size_t)colj[cc] * ldw + Rb + rr] : 0.0;
                    }
                }
#pragma unroll
                for (int s2 = 0; s2 < NB; s2++) {
                    if (s2 < ns) {
                        double l[4], u[4];
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) l[rr] = s_ls[s2][tx * 4 + rr];
#pragma unroll
                        for (int cc = 0; cc < 4; cc++) u[cc] = Us[s2][ty * 4 + cc];
#pragma unroll
                        for (int rr = 0; rr < 4; rr++) {
                            const bool nz = l[rr] != 0;
#pragma unroll
                            for (int cc = 0; cc < 4; cc++) acc[cc][rr] = nz ? __dadd_rn(__dmul_rn(l[rr], u[cc]), acc[cc][rr]) : acc[cc][rr];
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 4; cc++) {
#pragma unroll
                    for (int rr = 0; rr < 4; rr++)
                        if (live[cc][rr]) luc_st_agent(&a.W[(size_t)colj[cc] * ldw + Rb + rr], acc[cc][rr]);
                }
            }
        } else if (b == 0) {
            if (tid == 0) __hip_atomic_fetch_add(&base->cnt_x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (one arrival per launch, work or not)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's cells have landed
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&base->cnt_u, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- phase S: column tile b of the columns behind the round
        const int j0 = k1 + b * 64;
        if (work && j0 < m) {
            double (*X)[65] = reinterpret_cast<double (*)[65]>(s_xa);
            for (int idx = tid; idx < NB * 64; idx += T) {
                const int s2 = idx / 64, ci = idx % 64;
                const int j = j0 + ci;
                bool listed = false;   // (phase U's columns: not needed, and their cells are being written)
                for (int t = 0; t < nn; t++) listed = listed || Nx[t] == j;
                X[s2][ci] = (s2 < ns && j < m && !listed) ? a.W[(size_t)j * ldw + Ps[s2]] : 0.0;
            }
            __syncthreads();
            const int j = j0 + tid;
            bool mine = tid < 64 && j < m;
            if (mine)
                for (int t = 0; t < nn; t++) mine = mine && Nx[t] != j;   // (phase U's columns: their U rows in Up_prev are phase U's)
            if (mine) {
                int din = 0, dout = 0;
                for (int t = 0; t < nd; t++)
                    if (Dc[t] == j) { din = Di[t]; dout = Do[t]; }
                double u[NB];
#pragma unroll
                for (int s2 = 0; s2 < NB; s2++) {
                    if (s2 < ns) {
                        double x = X[s2][tid];
                        if (!(Ss[s2] >= din && Ss[s2] < dout)) {
#pragma unroll
                            for (int t = 0; t < s2; t++) {
                                const double l = s_ln[s2][t];
                                x = (l != 0) ? __dadd_rn(__dmul_rn(l, u[t]), x) : x;
                            }
                        }
                        u[s2] = x;
                        luc_st_agent(const_cast<double *>(a.Up_prev) + (size_t)s2 * ldw + j, x);
                    } else u[s2] = 0;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&base->cnt_s, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!work) return;
    // ---- phase T
    if (tid == 0) s_go = luc_spin(&base->cnt_s, (uint32_t)(a.round + 1) * (uint32_t)nrt) ? 1 : 0;
    __syncthreads();
    if (!s_go) {
        if (tid == 0) base->fault = 1;
        return;
    }
    constexpr int G = T / 256;
    const int g = tid >> 8, t256 = tid & 255;
    const int tx = t256 & 15, ty = t256 >> 4;
    const int nct = (m - k1 + 63) / 64;
    const int ngroups = ((int)gridDim.x - 1) * G;
    for (int tile = b * G + g; tile < nct * nrt; tile += ngroups) {
        const int ct = tile / nrt, rt = tile % nrt;   // neighbouring groups: the row tiles of one column tile
        int colj[4];
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            int j = k1 + ct * 64 + ty * 4 + cc;
            if (j >= m) j = -1;
            if (j >= 0) {
                const int ur = a.unit_row ? a.unit_row[j] : -1;
                if (ur >= 0 && a.rowsnap_prev[ur] < 0) j = -1;
            }
            colj[cc] = j;
        }
        for (int t = 0; t < nn; t++) {
            const int nj = c->next[t];
#pragma unroll
            for (int cc = 0; cc < 4; cc++)
                if (colj[cc] == nj) colj[cc] = -1;
        }
        luc_trail_cells<NB, (T > 512 ? 2 : 4)>(a, c, a.Lp_prev, a.Up_prev, a.rowsnap_prev, ns, k0, colj, rt * 64, tx);
    }
}

// panel shapes that carry the look-ahead schedule: groups of 256 threads for the update role, and not the 4-rows-per-thread shape of
// 1024 threads (bases beyond 2048 rows: its 128 registers per lane are all taken, the additions spilled inside the step loop)
template <int T, int RPT>
constexpr bool luc_look_ok() { return T % 256 == 0 && !(T == 1024 && RPT == 4); }
// LK: the instance that carries the look-ahead schedule (the update role beside the panel: more registers, 42 KB of LDS); the plain
// instance serves the small bases, dozens of which factor side by side in a wave
template <int T, int RPT, int NB, int SMAX, bool LK>
__global__ __launch_bounds__(T) void k_luc_panel_slots(LUArgs a, int32_t *__restrict__ pivrow) {
    constexpr int NW = T / 64;
    constexpr int MAXM = T * RPT;
    static_assert(NB <= 32 && SMAX <= 32 && NW <= 16, "slots / steps per round are recorded in 32-entry tables");
    typedef unsigned short idx_t;            // m <= 4096
    // register columns in tuples of at most 16 doubles: the widest the uniform dynamic index (v_movrels) reaches; wider vectors go to scratch
    constexpr int NH = (NB + 15) / 16, VW = NB / NH;
    static_assert(NH * VW == NB, "NB: 8, 16 or 32");
    typedef double vec __attribute__((ext_vector_type(VW)));
    constexpr idx_t NONE = 0xFFFF;
    __shared__ idx_t s_lpos[MAXM];    // logical position of physical row R
    __shared__ idx_t s_rowat[MAXM];   // physical row at logical position
    __shared__ idx_t s_unit[MAXM];    // unit_row per column (NONE: not a unit column)
    __shared__ idx_t s_ucol[MAXM];    // inverse: the unit column of a row (NONE: none)
    __shared__ unsigned char s_active[MAXM];
    __shared__ double prow[2][NB];    // the pivot row's values by slot
    __shared__ double s_rinv[2];
    __shared__ double redM[2][16];
    __shared__ unsigned int redL[2][16];
    __shared__ int s_slotcol[NB];     // columns listed at the start of the round (afterwards the slot tables live in registers)
    __shared__ int s_nload, s_stop, s_limit, s_sigma, s_ins;
    if (LK && a.look && a.ctl_base->fault) return;   // (a wait of an earlier launch gave up: the host repeats the factorization)
    if (blockIdx.x > 0) {   // look-ahead schedule: the previous round's update beside this round's panel
        if constexpr (LK) luc_role<T, SMAX>(a);
        return;
    }
    LUCtl *ctl = a.ctl;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = a.m;
    const size_t ldw = (size_t)a.ldw;
    const int k0 = a.ctl_prev->k_next;
    if (k0 >= m) {
        if (tid == 0) { ctl->nsteps = 0; ctl->ndrop = 0; ctl->nnext = 0; ctl->k_next = k0; ctl->k0 = k0; ctl->k1 = k0; ctl->rounds = a.ctl_prev->rounds; }
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
        if (s_unit[R] != NONE) s_ucol[s_unit[R]] = (idx_t)R;   // column R is the unit vector of row s_unit[R]
    }
    if (w == 0) {
        // the first NB columns >= k0 that are dense for sure
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
    // the slot tables live in registers, replicated in every wave (lane c = slot c; every change is uniform): no LDS round trip per slot
    int myslotcol = (lane < NB && lane < nload) ? s_slotcol[lane < NB ? lane : 0] : 0x7FFFFFFF;
    int myslotin = -1;
    unsigned int live = nload >= 32 ? 0xFFFFFFFFu : ((1u << nload) - 1u);   // uniform: every thread keeps the same copy
    const bool lk = LK && a.look;
    if (lk) {
        // the columns this panel loads are brought up to date by phase U of this launch (luc_role): wait for its workgroups
        if (tid == 0) s_stop = luc_spin(&a.ctl_base->cnt_u, (uint32_t)(a.round + 1) * (uint32_t)((m + 63) / 64)) ? 1 : 0;
        __syncthreads();
        if (!s_stop) {
            if (tid == 0) {
                a.ctl_base->fault = 1;
                ctl->nsteps = 0; ctl->ndrop = 0; ctl->nnext = 0; ctl->k_next = k0; ctl->k0 = k0; ctl->k1 = k0; ctl->rounds = a.ctl_prev->rounds;
            }
            return;
        }
    }
    vec v[RPT][NH];
    int Rr[RPT];
    bool act[RPT];
#pragma unroll
    for (int r = 0; r < RPT; r++) {
        Rr[r] = tid + r * T;
        act[r] = (Rr[r] < m) && s_active[Rr[r] < m ? Rr[r] : 0];
        const double *src = a.W + (act[r] ? Rr[r] : 0);
        if (lk) {   // (uniform) the cells come from phase U of this launch: agent scope
#pragma unroll
            for (int c = 0; c < NB; c++) v[r][c / VW][c % VW] = (act[r] && c < nload) ? luc_ld_agent(&src[(size_t)__builtin_amdgcn_readlane(myslotcol, c) * ldw]) : 0.0;
        } else {
#pragma unroll
            for (int c = 0; c < NB; c++) v[r][c / VW][c % VW] = (act[r] && c < nload) ? src[(size_t)__builtin_amdgcn_readlane(myslotcol, c) * ldw] : 0.0;
        }
    }
    __syncthreads();   // every thread has taken its rows' `act` from s_active before wave 0's first run clears entries
    // The column loads above are the only global loads of this kernel.  Retire them HERE, with an instruction the compiler's wait
    // pass sees (inline asm would be opaque to it): otherwise it must assume, on every path of the loop below, that a register
    // column may still be in flight, and guards each use with s_waitcnt vmcnt(0) — which on gfx9 also waits for every STORE issued
    // so far: the step's multiplier / U-row stores (fire-and-forget by design) landed on the critical path, several times per step.
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
#ifdef GOMILP_DEBUG
    unsigned long long tacc[16] = {}, tprev = 0;
#endif
    LUC_STAMP(-1);
    int kcur = k0, s = 0, k1 = m;
#pragma unroll 1
    for (;;) {
        if (w == 0) {
            // the next listed column: smallest column index among the live slots
            const unsigned int key = (lane < NB && ((live >> lane) & 1u)) ? (unsigned int)myslotcol : 0x7FFFFFFFu;
            unsigned int mn = row_min_u32(key);
            mn = min((unsigned int)__builtin_amdgcn_readlane((int)mn, 0), (unsigned int)__builtin_amdgcn_readlane((int)mn, 16));
            const bool listed = live != 0 && s < SMAX;
            const int limit = listed ? (int)mn : m;
            const unsigned long long hit = __ballot(key == mn && lane < NB);
            const int sigma = hit ? (int)__builtin_ctzll(hit) : 0;
            // run of bookkeeping steps [kcur, limit) (see k_luc_panel)
            int k = kcur;
            for (;;) {
                const int kk = k + lane;
                const idx_t ur = kk < limit ? s_unit[kk] : NONE;
                const bool triv = ur != NONE && s_active[ur];
                const unsigned long long nt = __ballot(!triv);
                const int cnt = nt ? (int)__builtin_ctzll(nt) : 64;
                for (int j = 0; j < cnt; j++) {
                    const int urj = __builtin_amdgcn_readlane((int)ur, j);
                    if (lane == 0) {
                        const idx_t jp = s_lpos[urj], Q = s_rowat[k + j];
                        s_lpos[Q] = jp; s_rowat[jp] = Q;
                        s_lpos[urj] = (idx_t)(k + j); s_rowat[k + j] = (idx_t)urj;
                        s_active[urj] = 0;
                    }
                }
                k += cnt;
                if (cnt < 64) break;   // the limit or a step that needs arithmetic
            }
            if (lane == 0) { s_stop = k; s_limit = limit; s_sigma = sigma; }
        }
        LUC_STAMP(0);   // wave 0: next column + the run of bookkeeping steps
        __syncthreads();
        LUC_STAMP(1);   // barrier 1
        const int kstop = s_stop, limit = s_limit;
        const int sigma = __builtin_amdgcn_readfirstlane(s_sigma);
        // rows retired by the run become U rows: their entries in the listed columns are final
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            if (act[r] && !s_active[Rr[r]]) {
                const int kt = s_lpos[Rr[r]];
                a.rowstep[Rr[r]] = kt; pivrow[kt] = Rr[r];
                double *dst = a.W + Rr[r];
#pragma unroll
                for (int c = 0; c < NB; c++)
                    if ((live >> c) & 1u) dst[(size_t)__builtin_amdgcn_readlane(myslotcol, c) * ldw] = v[r][c / VW][c % VW];
                act[r] = false;
            }
        }
        if (kstop < limit || live == 0 || s >= SMAX) { k1 = kstop; break; }
        const int k = limit;
        LUC_STAMP(2);   // retire
        // ---- dense step k on slot sigma: first maximum of |a_ik| in LAPACK row order (as k_luc_panel)
        double x[RPT];
        double xm = __builtin_inf();
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            if constexpr (NH == 1) x[r] = v[r][0][sigma & (VW - 1)];   // uniform index
            else {   // (masked: an index beyond the tuple, even on a path not taken, would address a foreign register; the empty asm keeps the
                     // uniform branch a branch — if-converted, it becomes a 32-instruction select of whole tuples in front of the index)
                if (sigma < VW) { asm volatile(""); x[r] = v[r][0][sigma & (VW - 1)]; }
                else { asm volatile(""); x[r] = v[r][NH - 1][sigma & (VW - 1)]; }
            }
            if (act[r]) xm = vmin_f64(xm, -fabs(x[r]));
        }
        const double wm = wave_min_f64(xm);
        unsigned int lk = 0xFFFFFFFFu;
        int lpr[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            lpr[r] = -1;
            if (act[r] && -fabs(x[r]) == wm) { lpr[r] = s_lpos[Rr[r]]; lk = min(lk, (unsigned int)lpr[r]); }
        }
        lk = row_min_u32(lk);
        lk = min(min((unsigned int)__builtin_amdgcn_readlane((int)lk, 15), (unsigned int)__builtin_amdgcn_readlane((int)lk, 31)),
                 min((unsigned int)__builtin_amdgcn_readlane((int)lk, 47), (unsigned int)__builtin_amdgcn_readlane((int)lk, 63)));
        double *rm = redM[s & 1];
        unsigned int *rl = redL[s & 1];
        if (lane == 0) { rm[w] = wm; rl[w] = lk; }
        LUC_STAMP(3);   // own pivot search
        __syncthreads();
        LUC_STAMP(4);   // barrier 2
        const double bx = lane < NW ? rm[lane] : __builtin_inf();
        const double bm = readlane_f64(row_min_f64(bx), 15);
        const unsigned int bk = (lane < NW && bx == bm) ? rl[lane] : 0xFFFFFFFFu;
        const int jp = (int)(unsigned int)__builtin_amdgcn_readlane((int)row_min_u32(bk), 15);
        double *pr = prow[s & 1];
        bool owner[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            owner[r] = act[r] && lpr[r] == jp;
            if (owner[r]) {
                const int P = Rr[r];
#pragma unroll
                for (int cc = 0; cc < NB; cc++) pr[cc] = v[r][cc / VW][cc % VW];
                s_rinv[s & 1] = 1.0 / x[r];   // dgetf2.go:54-56 scales by the reciprocal
                act[r] = false;
                s_active[P] = 0;
                a.rowstep[P] = k; pivrow[k] = P;
                if (a.dense_flag) a.dense_flag[k] = 1;
                ctl->steps[s] = k; ctl->prow[s] = P;
                const idx_t Q = s_rowat[k];   // dlaswp.go: the row at logical k moves to jp
                s_lpos[Q] = (idx_t)jp; s_rowat[jp] = Q;
                s_lpos[P] = (idx_t)k; s_rowat[k] = (idx_t)P;
                // taking row P makes the unit column of P (if it is still to come) dense from this step on
                const idx_t uc = s_ucol[P];
                s_ins = (uc != NONE && (int)uc > k) ? (int)uc : -1;
            }
        }
        LUC_STAMP(5);   // workgroup pivot + publish
        __syncthreads();
        LUC_STAMP(6);   // barrier 3
        // the pivot row's U entries go to W behind the barrier, off everybody's critical path (slot sigma holds column k)
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            if (owner[r]) {
                double *dst = a.W + Rr[r];
#pragma unroll
                for (int cc = 0; cc < NB; cc++)
                    if ((live >> cc) & 1u) dst[(size_t)__builtin_amdgcn_readlane(myslotcol, cc) * ldw] = v[r][cc / VW][cc % VW];
            }
        }
        const double prl = pr[lane & (NB - 1)];   // lane c holds the pivot row's value in slot c: ONE LDS round trip, then scalar broadcasts
        const double piv = readlane_f64(prl, sigma);
        const bool singular = (piv == 0);  // dgetf2.go:48-49: no scaling, the rank-1 update is a no-op
        if (singular && tid == 0) a.st->lu_singular = 1;
        const double rinv = s_rinv[s & 1];
        const int k2 = s_ins;
        double *wcol = a.W + (size_t)k * ldw;      // column k of L\U
        double *lcol = a.Lp + (size_t)s * ldw;     // compact panel: -l (0 for rows that are not active)
        double nl[RPT];
#pragma unroll
        for (int r = 0; r < RPT; r++) {
            nl[r] = 0.0;
            if (act[r]) {
                const double l = singular ? x[r] : __dmul_rn(x[r], rinv);
                wcol[Rr[r]] = l;
                nl[r] = singular ? 0.0 : -l;
                lcol[Rr[r]] = nl[r];
            } else if (Rr[r] < a.ldw) lcol[Rr[r]] = 0.0;
        }
        const unsigned int others = live & ~(1u << sigma);
        bool any = false;
#pragma unroll
        for (int r = 0; r < RPT; r++) any = any || act[r];
        if (!singular && __any(any)) {   // (a wave whose rows have all left skips the update; Dger does not skip zero multipliers)
            // branch-free over the slots: a slot that is not listed takes (-l) * 0 (its register holds nothing anybody reads) — a
            // uniform branch per slot came out as two taken branches per LIVE slot (bodies moved out of line): ~2500 cycles at 32 slots
            const double prz = (lane < NB && ((others >> (lane & 31)) & 1u)) ? prl : 0.0;
#pragma unroll
            for (int c = 0; c < NB; c++) {
                const double pc = readlane_f64(prz, c);
#pragma unroll
                for (int r = 0; r < RPT; r++) v[r][c / VW][c % VW] = __dadd_rn(__dmul_rn(nl[r], pc), v[r][c / VW][c % VW]);
            }
        }
        if (k2 >= 0) {
            // the unit column of the pivot row is e_P: 0 + (-l) * 1 for the active rows, from this step on; it takes the slot this
            // step's column leaves
#pragma unroll
            for (int r = 0; r < RPT; r++) {
                const double vn = (act[r] && !singular) ? __dadd_rn(__dmul_rn(nl[r], 1.0), 0.0) : 0.0;
                if constexpr (NH == 1) v[r][0][sigma & (VW - 1)] = vn;
                else {
                    if (sigma < VW) { asm volatile(""); v[r][0][sigma & (VW - 1)] = vn; }
                    else { asm volatile(""); v[r][NH - 1][sigma & (VW - 1)] = vn; }
                }
            }
            if (lane == sigma) { myslotcol = k2; myslotin = k; }
        } else live = others;
        s++;
        kcur = k + 1;
        LUC_STAMP(7);   // elimination
    }
#ifdef GOMILP_DEBUG
    if (lane == 0 && w < 4) {
        for (int sg = 0; sg < 8; sg++) atomicAdd(&g_luc_stamps[w * 16 + sg], tacc[sg]);
        if (w == 0) atomicAdd(&g_luc_stamps[15], (unsigned long long)s);
    }
#endif
    for (int R = tid; R < m; R += T) a.lpos[R] = s_lpos[R];
    if (LK && a.look) {
        // rowstep as this round leaves it (the last writes to it are behind the loop's closing barrier), for the update workgroups that
        // run beside the next panel
        for (int R = tid; R < m; R += T) a.rowsnap[R] = s_active[R] ? -1 : a.rowstep[R];
    }
    // columns still listed: rows that left at steps [joined, k1) hold final values in them (retire / the pivot rows' stores)
    int ndl = 0;
    if (w == 0) {
        if (LK && a.look) {
            // the columns the next round's panel will load: its own scan (above), run on the state this round leaves
            int n = 0;
            for (int base = k1; base < m && n < NB; base += 64) {
                const int k = base + lane;
                bool dense = false;
                if (k < m) {
                    const idx_t ur = s_unit[k];
                    dense = ur == NONE || !s_active[ur];
                }
                const unsigned long long mask = __ballot(dense);
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                if (dense && n + rank < NB) ctl->next[n + rank] = k;
                n += __popcll(mask);
            }
            if (lane == 0) ctl->nnext = n < NB ? n : NB;
        } else if (lane == 0) ctl->nnext = 0;
        const bool on = lane < NB && ((live >> lane) & 1u);
        const unsigned long long msk = __ballot(on);
        ndl = __popcll(msk);
        if (on) {
            const int at = __popcll(msk & ((1ull << lane) - 1ull));
            ctl->dropcol[at] = myslotcol; ctl->dropin[at] = myslotin; ctl->dropout[at] = k1;
        }
    }
    if (tid == 0) {
        const int nd = ndl;
        // (table above): rows that left at steps [joined, k1) hold final values in them (retire / the pivot rows' stores)
        ctl->k0 = k0; ctl->k1 = k1; ctl->k_next = k1; ctl->nsteps = s; ctl->ndrop = nd;
        ctl->rounds = a.ctl_prev->rounds + 1;
    }
}

// dense pivot rows of the round, columns j >= k1:  u_s = a[P_s] + sum_{t<s} (-l[P_s][t]) * u_t  (ascending t, the
// Dtrsm of dgetrf.go:57-60) into the compact panel Up; the trailing kernel writes them back into W together with all
// other rows.  In a column the panel dropped, a pivot row that left while the column was listed is final already.
template <int NB>
__global__ __launch_bounds__(256) void k_luc_usolve(LUArgs a) {
    const LUCtl *ctl = a.ctl;
    const int ns = ctl->nsteps, k1 = ctl->k1;
    const int j0 = k1 + (int)blockIdx.x * 64;
    if (ns == 0 || j0 >= a.m) return;
    __shared__ double Ln[NB][NB + 1];
    __shared__ double X[NB][64 + 1];
    __shared__ int Ps[NB], Ss[NB], Dc[NB], Di[NB], Do[NB];
    const int tid = threadIdx.x;
    const size_t ldw = (size_t)a.ldw;
    const int nd = ctl->ndrop;
    if (tid < NB) {
        Ps[tid] = tid < ns ? ctl->prow[tid] : 0;
        Ss[tid] = tid < ns ? ctl->steps[tid] : 0x7fffffff;
        Dc[tid] = tid < nd ? ctl->dropcol[tid] : -1;
        Di[tid] = tid < nd ? ctl->dropin[tid] : 0;
        Do[tid] = tid < nd ? ctl->dropout[tid] : 0;
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int s = idx / NB, t = idx % NB;
        Ln[s][t] = (s < ns && t < s) ? a.Lp[(size_t)t * ldw + Ps[s]] : 0.0;
    }
    for (int idx = tid; idx < NB * 64; idx += 256) {
        const int s = idx / 64, c = idx % 64;
        const int j = j0 + c;
        X[s][c] = (s < ns && j < a.m) ? a.W[(size_t)j * ldw + Ps[s]] : 0.0;
    }
    __syncthreads();
    const int j = j0 + tid;
    if (tid >= 64 || j >= a.m) return;
    int din = 0, dout = 0;   // window of steps whose pivot rows are final in this column (empty: not a dropped column)
    for (int t = 0; t < nd; t++)
        if (Dc[t] == j) { din = Di[t]; dout = Do[t]; }
    double u[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < ns) {
            double x = X[s][tid];
            if (!(Ss[s] >= din && Ss[s] < dout)) {
#pragma unroll
                for (int t = 0; t < s; t++) {
                    const double l = Ln[s][t];
                    x = (l != 0) ? __dadd_rn(__dmul_rn(l, u[t]), x) : x;
                }
            }
            u[s] = x;
            a.Up[(size_t)s * ldw + j] = x;
        } else {
            u[s] = 0;
        }
    }
}

// every row that took part in the round, columns j >= k1: a[R][j] += sum_s Lp[s][R] * Up[s][j] in ascending s (the
// Dgemm of dgetrf.go:62-66; zero multipliers skipped).  Lp is zero from the step at which a row left the active set,
// so rows still active take all steps, a row retired by a bookkeeping step takes the steps before it and the pivot row
// of dense step s takes steps < s — which IS its U-solve recurrence.  In a column the panel dropped, the rows that left
// while it was listed hold final values already (retire / publish wrote them).
template <int NB>
__global__ __launch_bounds__(256) void k_luc_trail(LUArgs a) {
    const LUCtl *ctl = a.ctl;
    const int ns = ctl->nsteps, k0 = ctl->k0, k1 = ctl->k1;
    const int j0 = k1 + blockIdx.x * 64;
    if (ns == 0 || j0 >= a.m) return;
    __shared__ double Ls[NB][64];
    __shared__ double Us[NB][64];
    __shared__ unsigned char rowcls[64];
    __shared__ int rstep[64], cdin[64], cdout[64];
    __shared__ int Dc[NB], Di[NB], Do[NB];
    const int tid = threadIdx.x;
    const int R0 = blockIdx.y * 64;
    const size_t ldw = (size_t)a.ldw;
    // the tile's cells first: their loads overlap the staging of the panels (this kernel is a chain of dependent
    // memory round trips, not bandwidth)
    const int tx = tid & 15, ty = tid >> 4;   // rows tx*4.., columns ty*4..: consecutive lanes walk down a column
    double acc[4][4];   // [cc][rr]
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        const int c = ty * 4 + cc;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = tx * 4 + rr;
            acc[cc][rr] = ((j0 + c) < a.m && (R0 + r) < a.m) ? a.W[(size_t)(j0 + c) * ldw + R0 + r] : 0.0;
        }
    }
    const int nd = ctl->ndrop;
    if (tid < NB) {
        Dc[tid] = tid < nd ? ctl->dropcol[tid] : -1;
        Di[tid] = tid < nd ? ctl->dropin[tid] : 0;
        Do[tid] = tid < nd ? ctl->dropout[tid] : 0;
    }
    for (int idx = tid; idx < NB * 64; idx += 256) {
        const int s = idx / 64, c = idx % 64;
        Ls[s][c] = (s < ns && R0 + c < a.m) ? a.Lp[(size_t)s * ldw + R0 + c] : 0.0;
        Us[s][c] = (s < ns && j0 + c < a.m) ? a.Up[(size_t)s * ldw + j0 + c] : 0.0;
    }
    __syncthreads();
    int cls = 0;
    if (tid < 64) {
        const int R = R0 + tid;
        int rs = -1;
        if (R < a.m) {
            rs = a.rowstep[R];
            cls = rs < 0 ? 1 : (rs >= k0 ? 2 : 0);   // 1 active, 2 left during this round, 0 finished earlier
        }
        rowcls[tid] = (unsigned char)cls;
        rstep[tid] = rs;
    } else if (tid < 128) {
        const int j = j0 + (tid - 64);
        int din = 0, dout = 0;   // rows that left at steps [din, dout) are final in this column
        for (int t = 0; t < nd; t++)
            if (Dc[t] == j) { din = Di[t]; dout = Do[t]; }
        cdin[tid - 64] = din; cdout[tid - 64] = dout;
    }
    if (!__syncthreads_or(cls > 0)) return;
    bool live[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        const int c = ty * 4 + cc;
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = tx * 4 + rr;
            live[cc][rr] = rowcls[r] > 0 && (j0 + c) < a.m && !(rowcls[r] == 2 && rstep[r] >= cdin[c] && rstep[r] < cdout[c]);
        }
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < ns) {
            double l[4], u[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) l[rr] = Ls[s][tx * 4 + rr];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) u[cc] = Us[s][ty * 4 + cc];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const bool nz = l[rr] != 0;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[cc][rr] = nz ? __dadd_rn(__dmul_rn(l[rr], u[cc]), acc[cc][rr]) : acc[cc][rr];
            }
        }
    }
#pragma unroll
    for (int cc = 0; cc < 4; cc++) {
        const int c = ty * 4 + cc;
#pragma unroll
        for (int rr = 0; rr < 4; rr++)
            if (live[cc][rr]) a.W[(size_t)(j0 + c) * ldw + R0 + tx * 4 + rr] = acc[cc][rr];
    }
}

__global__ void k_luc_init(LUArgs a) {
    const int R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R < a.m) {
        a.lpos[R] = R; a.rowstep[R] = -1;
        if (a.dense_flag) a.dense_flag[R] = 0;
    }
    if (R == 0) {
        a.st->lu_singular = 0;   // (the host resets its copy; small bases skip the upload of the state block)
        for (int t = 0; t < 2; t++) {   // (two: by round parity)
            LUCtl *c = a.ctl + t;
            c->k_next = 0; c->k0 = 0; c->k1 = 0; c->nsteps = 0; c->ndrop = 0; c->rounds = 0; c->nnext = 0; c->ksync = 0;
            c->cnt_x = 0; c->cnt_u = 0; c->cnt_s = 0; c->fault = 0;
        }
    }
}

// Wc[k][R] = At[basic[k]][R]: the basis, column-major (rows of At are columns of A)
__global__ __launch_bounds__(256) void k_luc_gather(const double *__restrict__ At, int ld, int m, const int32_t *__restrict__ basic,
                                                    double *__restrict__ W, int ldw) {
    const int k = blockIdx.x;
    const double *src = At + (size_t)basic[k] * ld;
    double *dst = W + (size_t)k * ldw;
    for (int R = threadIdx.x; R < ldw; R += 256) dst[R] = R < m ? src[R] : 0.0;
}

// Wd[R*nd + t] = W(R, dlist[t]) (32x32 tiles through LDS: both sides coalesced) and diag[R] = W(R, lpos[R])
__global__ __launch_bounds__(256) void k_luc_pack(LUArgs a, const int32_t *__restrict__ dlist, int nd, double *__restrict__ Wd,
                                                  double *__restrict__ diag) {
    __shared__ double tile[32][33];
    const int t0 = blockIdx.x * 32, R0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const size_t ldw = (size_t)a.ldw;
    for (int tt = ty; tt < 32; tt += 8) {
        const int t = t0 + tt, R = R0 + tx;
        tile[tt][tx] = (t < nd && R < a.m) ? a.W[(size_t)dlist[t] * ldw + R] : 0.0;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        const int R = R0 + rr, t = t0 + tx;
        if (R < a.m && t < nd) Wd[(size_t)R * nd + t] = tile[tx][rr];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int R = R0 + threadIdx.x;
        if (R < a.m) diag[R] = a.W[(size_t)a.lpos[R] * ldw + R];
    }
}


// ---- Dgetrs split (large bases): the two triangular solves only couple the nd "dense" positions with each other; every
// other row is a chain of its own once the solution at the dense positions is known.  The host solves the nd x nd
// coupled part (k_luc_pack_dense hands it over), k_luc_solve_rows runs all remaining rows, one thread per physical row
// (consecutive threads read consecutive elements of a column of the column-major L\U), in gonum's operation order:
// ascending k, zero multipliers skipped, b_i = (-a_ik)*b_k + b_i as a rounded multiply and a rounded add, then
// b_i *= 1/u_ii (level3double.go:75-118).
__global__ __launch_bounds__(256) void k_luc_pack_dense(LUArgs a, const int32_t *__restrict__ dlist, int nd,
                                                        const int32_t *__restrict__ pivrow, double *__restrict__ Wdd,
                                                        double *__restrict__ diag) {
    // Wdd[s*nd + t] = W(pivrow[dlist[s]], dlist[t]); diag[R] = W(R, lpos[R]) for every row (LU.Det() needs all of them)
    const size_t ldw = (size_t)a.ldw;
    const int s = blockIdx.x;
    if (s < nd) {
        const int R = pivrow[dlist[s]];
        for (int t = threadIdx.x; t < nd; t += 256) Wdd[(size_t)s * nd + t] = a.W[(size_t)dlist[t] * ldw + R];
    }
    for (int R = blockIdx.x * 256 + threadIdx.x; R < a.m; R += gridDim.x * 256) diag[R] = a.W[(size_t)a.lpos[R] * ldw + R];
}

__global__ __launch_bounds__(256) void k_luc_solve_rows(LUArgs a, const int32_t *__restrict__ dlist, int nd,
                                                        const double *__restrict__ b, const double *__restrict__ xdL,
                                                        const double *__restrict__ xdU, double *__restrict__ x) {
    const int R = blockIdx.x * 256 + threadIdx.x;
    if (R >= a.m) return;
    const int i = a.lpos[R];                       // logical position (= elimination step) of this physical row
    if (a.dense_flag[i]) return;                   // coupled position: solved by the host
    const size_t ldw = (size_t)a.ldw;
    int lo = 0, hi = nd;                           // cnt = number of dense positions below i
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dlist[mid] < i) lo = mid + 1; else hi = mid; }
    const int cnt = lo;
    double acc = b[R];                             // Dlaswp: b in logical order is b[physical row]
    // Dtrsm(Left, Lower, NoTrans, Unit) over t < cnt, then Dtrsm(Left, Upper, NoTrans, NonUnit) over t >= cnt: one
    // ascending sweep; 8 independent loads in flight per trip (a plain loop waits a memory round trip per term)
    for (int t0 = 0; t0 < nd; t0 += 8) {
        double va[8], xk[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int t = t0 + u;
            va[u] = t < nd ? a.W[(size_t)dlist[t < nd ? t : 0] * ldw + R] : 0.0;
            xk[u] = t < nd ? (t < cnt ? xdL[t] : xdU[t]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (va[u] != 0) acc = __dadd_rn(__dmul_rn(-va[u], xk[u]), acc);
    }
    const double tinv = 1.0 / a.W[(size_t)i * ldw + R];
    x[i] = __dmul_rn(acc, tinv);
}

// slot form: up to kLucSlotSteps dense steps per round whatever the number of register slots
constexpr int kLucSlotSteps = 32;
// `base`: a.ctl / a.Lp / a.Up / a.rowsnap of parity 0 (the other parity behind it); round_base: rounds enqueued for this
// factorization so far (the parity goes on across the host's batches)
template <int T, int RPT, int NB>
static void luc_rounds_slots(const LUArgs &base, int32_t *pivrow, int nrounds, int round_base, hipStream_t s) {
    const int m = base.m;
    const int nt = (m + 63) / 64;
    if (!base.look) {
        LUArgs a = base;
        a.ctl_prev = a.ctl; a.Lp_prev = a.Lp; a.Up_prev = a.Up; a.rowsnap_prev = a.rowsnap;
        for (int r = 0; r < nrounds; r++) {
            hipLaunchKernelGGL((k_luc_panel_slots<T, RPT, NB, kLucSlotSteps, false>), dim3(1), dim3(T), 0, s, a, pivrow);
            hipLaunchKernelGGL((k_luc_usolve<kLucSlotSteps>), dim3(nt), dim3(256), 0, s, a);
            hipLaunchKernelGGL((k_luc_trail<kLucSlotSteps>), dim3(nt, nt), dim3(256), 0, s, a);
        }
        return;
    }
    // look-ahead: one launch per round = { panel r | U-solve and update of round r-1 } (luc_role)
    const size_t pstride = (size_t)kLucSlotSteps * (size_t)base.ldw;
    const int rest = std::max(nt, std::min(255, (nt * nt + std::max(1, T / 256) - 1) / std::max(1, T / 256)));   // (at least the nt workgroups of phases U / S)
    for (int r = 0; r < nrounds; r++) {
        const int p = (round_base + r) & 1;
        LUArgs a = base;
        a.ctl = base.ctl + p; a.ctl_prev = base.ctl + (p ^ 1);
        a.Lp = base.Lp + p * pstride; a.Lp_prev = base.Lp + (p ^ 1) * pstride;
        a.Up = base.Up + p * pstride; a.Up_prev = base.Up + (p ^ 1) * pstride;
        a.rowsnap = base.rowsnap + (size_t)p * m; a.rowsnap_prev = base.rowsnap + (size_t)(p ^ 1) * m;
        if (!luc_look_ok<T, RPT>()) {   // a panel shape without the update role: the control blocks still alternate (the host reads by parity)
            a.look = 0;
            hipLaunchKernelGGL((k_luc_panel_slots<T, RPT, NB, kLucSlotSteps, false>), dim3(1), dim3(T), 0, s, a, pivrow);
            hipLaunchKernelGGL((k_luc_usolve<kLucSlotSteps>), dim3(nt), dim3(256), 0, s, a);
            hipLaunchKernelGGL((k_luc_trail<kLucSlotSteps>), dim3(nt, nt), dim3(256), 0, s, a);
            continue;
        }
        a.ctl_base = base.ctl; a.round = round_base + r;
        if constexpr (luc_look_ok<T, RPT>()) hipLaunchKernelGGL((k_luc_panel_slots<T, RPT, NB, kLucSlotSteps, true>), dim3(1 + rest), dim3(T), 0, s, a, pivrow);
    }
}

// the rows of a round's panel on G workgroups of one XCD (lu_cross.hip, knob lu_cross): the plain schedule with that panel
void launch_luc_cross_panel(const LUArgs &a, int32_t *pivrow, double *xrec, int G, hipStream_t s);
int launch_luc_rounds_cross(const LUArgs &base, int32_t *pivrow, int nrounds, double *xrec, int G, hipStream_t s) {
    const int nt = (base.m + 63) / 64;
    LUArgs a = base;
    a.look = 0;
    a.ctl_prev = a.ctl; a.Lp_prev = a.Lp; a.Up_prev = a.Up; a.rowsnap_prev = a.rowsnap;
    for (int r = 0; r < nrounds; r++) {
        launch_luc_cross_panel(a, pivrow, xrec, G, s);
        hipLaunchKernelGGL((k_luc_usolve<kLucSlotSteps>), dim3(nt), dim3(256), 0, s, a);
        hipLaunchKernelGGL((k_luc_trail<kLucSlotSteps>), dim3(nt, nt), dim3(256), 0, s, a);
    }
    return 3 * nrounds;
}

bool lu_compressed_supported(int m) { return m <= 4096; }
// dense steps a round can take (the host sizes its batches of rounds with it)
int lu_compressed_nb(int, bool) { return kLucSlotSteps; }

void launch_luc_init(const LUArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_luc_init, dim3((a.m + 255) / 256), dim3(256), 0, s, a);
}
void launch_luc_gather(const double *At, int ld, int m, const int32_t *basic, double *W, int ldw, hipStream_t s) {
    hipLaunchKernelGGL(k_luc_gather, dim3(m), dim3(256), 0, s, At, ld, m, basic, W, ldw);
}
void launch_luc_pack_dense(const LUArgs &a, const int32_t *dlist, int nd, const int32_t *pivrow, double *Wdd, double *diag, hipStream_t s) {
    const int grid = std::max(nd, (a.m + 255) / 256);
    hipLaunchKernelGGL(k_luc_pack_dense, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, a, dlist, nd, pivrow, Wdd, diag);
}
void launch_luc_solve_rows(const LUArgs &a, const int32_t *dlist, int nd, const double *b, const double *xdL, const double *xdU,
                           double *x, hipStream_t s) {
    hipLaunchKernelGGL(k_luc_solve_rows, dim3((a.m + 255) / 256), dim3(256), 0, s, a, dlist, nd, b, xdL, xdU, x);
}
// Small bases: everything the host needs after a factorization in ONE block (one copy instead of six: a 5-row relaxation of a small
// tree is made of such copies).  Layout: m x m doubles W(R, t) row-major | m doubles diag | int32 lpos[m] | int32 dense_flag[m] |
// LUCtl[2] | int32 lu_singular.
__global__ __launch_bounds__(256) void k_luc_pack_small(LUArgs a, double *__restrict__ out) {
    const int m = a.m;
    const size_t ldw = (size_t)a.ldw;
    const int tid = blockIdx.x * 256 + threadIdx.x, nth = gridDim.x * 256;
    for (int idx = tid; idx < m * m; idx += nth) {
        const int R = idx / m, t = idx % m;
        out[idx] = a.W[(size_t)t * ldw + R];
    }
    double *diag = out + (size_t)m * m;
    for (int R = tid; R < m; R += nth) diag[R] = a.W[(size_t)a.lpos[R] * ldw + R];
    int32_t *io = reinterpret_cast<int32_t *>(diag + m);
    for (int R = tid; R < m; R += nth) { io[R] = a.lpos[R]; io[m + R] = a.dense_flag ? a.dense_flag[R] : 1; }
    int32_t *cb = io + 2 * m;
    const int32_t *src = reinterpret_cast<const int32_t *>(a.ctl_base);
    constexpr int NC = (int)(2 * sizeof(LUCtl) / sizeof(int32_t));
    for (int t = tid; t < NC; t += nth) cb[t] = src[t];
    if (tid == 0) cb[NC] = a.st->lu_singular;
}
size_t luc_pack_small_bytes(int m) { return ((size_t)m * m + m) * sizeof(double) + (size_t)(2 * m + 1) * sizeof(int32_t) + 2 * sizeof(LUCtl); }
void launch_luc_pack_small(const LUArgs &a, double *out, hipStream_t s) {
    const int grid = std::max(1, std::min(64, (a.m * a.m + 255) / 256));
    hipLaunchKernelGGL(k_luc_pack_small, dim3(grid), dim3(256), 0, s, a, out);
}
void launch_luc_pack(const LUArgs &a, const int32_t *dlist, int nd, double *Wd, double *diag, hipStream_t s) {
    dim3 grid((nd + 31) / 32 > 0 ? (nd + 31) / 32 : 1, (a.m + 31) / 32);
    hipLaunchKernelGGL(k_luc_pack, grid, dim3(256), 0, s, a, dlist, nd, Wd, diag);
}

// enqueue `nrounds` rounds; returns the number of kernel launches
int launch_luc_rounds(const LUArgs &a, int32_t *pivrow, int nrounds, int round_base, hipStream_t s) {
    const int m = a.m;
    {
        // (threads x rows per thread x register slots: 128 VGPRs per thread at 1024 threads hold 2 x 16 or 4 x 8 columns)
#ifdef GOMILP_DEBUG
        static const int forced = [] { const char *e = getenv("GOMILP_LUC_SLOTS"); return e ? atoi(e) : -1; }();   // developer knob (diagnostic flavour): panel shape
        if (forced >= 0) {
            bool done = true;
            switch (forced) {
                case 10: if (m <= 2048) luc_rounds_slots<256, 8, 8>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 11: if (m <= 2048) luc_rounds_slots<512, 4, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 12: if (m <= 2048) luc_rounds_slots<1024, 2, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 13: if (m <= 2048) luc_rounds_slots<512, 4, 8>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 20: if (m <= 1024) luc_rounds_slots<512, 2, 32>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 21: if (m <= 1024) luc_rounds_slots<256, 4, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 22: if (m <= 1024) luc_rounds_slots<512, 2, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 23: if (m <= 1024) luc_rounds_slots<1024, 1, 32>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 30: if (m <= 512) luc_rounds_slots<256, 2, 32>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 31: if (m <= 512) luc_rounds_slots<256, 2, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                case 32: if (m <= 512) luc_rounds_slots<128, 4, 16>(a, pivrow, nrounds, round_base, s); else done = false; break;
                default: done = false;
            }
            if (done) return 3 * nrounds;
        }
#endif
        if (m <= 512) luc_rounds_slots<512, 1, 32>(a, pivrow, nrounds, round_base, s);
        else if (m <= 1024) luc_rounds_slots<512, 2, 16>(a, pivrow, nrounds, round_base, s);   // (16 slots: C2 1.20 ms / 12 rounds against 1.34 ms / 7 rounds with 32: the branch-free slot loop costs per slot)
        else if (m <= 2048) luc_rounds_slots<1024, 2, 16>(a, pivrow, nrounds, round_base, s);
        else luc_rounds_slots<1024, 4, 8>(a, pivrow, nrounds, round_base, s);
        return 3 * nrounds;
    }
    return 0;
}

#ifdef GOMILP_DEBUG
void luc_stamps_read(unsigned long long *out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_luc_stamps), sizeof(unsigned long long) * 64); }
#endif

}  // namespace gomilp
