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
//   k_luc_panel   ONE workgroup: registers hold the next NB columns that are KNOWN to be dense (non-unit, or unit with
//                 a used row), wherever they are; between two of them a single thread replays the run of bookkeeping
//                 steps on LDS-resident index maps (lpos / rowat / active).  A unit column that BECOMES dense inside
//                 the round (a dense step took its row) is not in registers: the round ends in front of it ("cut") and
//                 the not yet eliminated register columns are dropped — W still holds their untouched originals.
//   k_luc_usolve  finishes the dense pivot rows right of the round (columns >= k1),
//   k_luc_trail   applies the round's dense steps to every other row right of the round: all of them for rows that
//                 are still active, the steps before its own for a row retired by a bookkeeping step of the round.
// Rounds are data dependent, so the kernels take their step range from a device control block (LUCtl) and the host
// enqueues rounds in batches until k_next == m.
// m = 2048 metric basis: 2048 steps, 384 dense -> ~30 rounds instead of 128 panels.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

template <int NB>
struct CPanelRow {
    typedef double vec __attribute__((ext_vector_type(NB)));
    vec v;
    int R, lp;
    bool act;
};

template <int T, int RPT, int NB>
__global__ __launch_bounds__(T) void k_luc_panel(LUArgs a, int32_t *__restrict__ pivrow) {
    constexpr int NW = T / 64;
    constexpr int MAXM = T * RPT;
    __shared__ int s_lpos[MAXM];    // logical position of physical row R
    __shared__ int s_rowat[MAXM];   // physical row at logical position
    __shared__ int s_unit[MAXM];    // unit_row per column
    __shared__ unsigned char s_active[MAXM];
    __shared__ double prow[2][NB];
    __shared__ BtCand sm2[2 * 16];
    __shared__ int s_cols[NB];
    __shared__ int s_ncols, s_stop;
    LUCtl *ctl = a.ctl;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m = a.m;
    const int k0 = ctl->k_next;
    if (k0 >= m) {
        if (tid == 0) ctl->nsteps = 0;
        return;
    }
    for (int R = tid; R < MAXM; R += T) {
        const bool in = R < m;
        s_lpos[R] = in ? a.lpos[R] : R;
        s_active[R] = (in && a.rowstep[R] < 0) ? 1 : 0;
        s_unit[R] = (in && a.unit_row) ? a.unit_row[R] : -1;
    }
    __syncthreads();
    for (int R = tid; R < m; R += T) s_rowat[s_lpos[R]] = R;
    if (w == 0) {
        // the first NB columns >= k0 that are dense for sure
        int n = 0;
        for (int base = k0; base < m && n < NB; base += 64) {
            const int k = base + lane;
            bool dense = false;
            if (k < m) {
                const int ur = s_unit[k];
                dense = ur < 0 || !s_active[ur];
            }
            const unsigned long long mask = __ballot(dense);
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            if (dense && n + rank < NB) s_cols[n + rank] = k;
            n += __popcll(mask);
        }
        if (lane == 0) s_ncols = n < NB ? n : NB;
    }
    __syncthreads();
    const int ncols = s_ncols;
    CPanelRow<NB> rows[RPT];
#define GOMILP_FOR_ROWS(F)                              \
    do {                                                \
        _Pragma("unroll") for (int rr_ = 0; rr_ < RPT; rr_++) F(rows[rr_], rr_); \
    } while (0)
    auto load_row = [&](CPanelRow<NB> &row, int r) {
        row.R = tid + r * T;
        row.act = (row.R < m) && s_active[row.R < m ? row.R : 0];
        row.lp = 0;
        const double *src = a.W + (size_t)(row.act ? row.R : 0) * a.ldw;
#pragma unroll
        for (int c = 0; c < NB; c++) row.v[c] = (row.act && c < ncols) ? src[s_cols[c]] : 0.0;
    };
    GOMILP_FOR_ROWS(load_row);
    if (tid < NB) ctl->cols[tid] = tid < ncols ? s_cols[tid] : -1;
    __syncthreads();   // every thread has taken its rows' `act` from s_active before thread 0's first run clears entries
    int kcur = k0, s = 0, k1 = m;
#pragma unroll 1
    for (;;) {
        const int limit = (s < ncols) ? s_cols[s] : m;
        if (tid == 0) {
            // run of bookkeeping steps [kcur, limit): Idamax finds the 1 in row ur, dlaswp exchanges logical
            // positions k and lpos[ur]; nothing else happens (the step's multipliers are exactly 0)
            int k = kcur;
            while (k < limit) {
                const int ur = s_unit[k];
                if (ur < 0 || !s_active[ur]) break;
                const int jp = s_lpos[ur], Q = s_rowat[k];
                s_lpos[Q] = jp; s_rowat[jp] = Q;
                s_lpos[ur] = k; s_rowat[k] = ur;
                s_active[ur] = 0;
                k++;
            }
            s_stop = k;
        }
        __syncthreads();
        const int kstop = s_stop;
        // rows retired by the run become U rows: their entries in the register columns are final
        auto retire = [&](CPanelRow<NB> &row, int) {
            if (!row.act || s_active[row.R]) return;
            const int kt = s_lpos[row.R];
            a.rowstep[row.R] = kt; pivrow[kt] = row.R;
            double *dst = a.W + (size_t)row.R * a.ldw;
#pragma unroll
            for (int c = 0; c < NB; c++)
                if (s + c < ncols) dst[s_cols[s + c]] = row.v[c];
            row.act = false;
        };
        GOMILP_FOR_ROWS(retire);
        if (kstop < limit || s >= ncols) { k1 = kstop; break; }
        const int k = limit;
        // ---- dense step k: pivot = first max |a_ik| in logical row order (idamax over the permuted column)
        BtCand c;
        c.k = ~0ull; c.i = 0xFFFFFFFFu;
        auto cand = [&](CPanelRow<NB> &row, int) {
            if (!row.act) return;
            row.lp = s_lpos[row.R];
            BtCand b;
            b.k = ordkey(-fabs(row.v[0])); b.i = (unsigned int)row.lp;
            bt_take(c, b);
        };
        GOMILP_FOR_ROWS(cand);
        bt_block_argmin<NW>(c, sm2 + 16 * (s & 1));
        const int jp = (int)c.i;
        double *pr = prow[s & 1];
        auto publish = [&](CPanelRow<NB> &row, int) {
            if (!row.act || row.lp != jp) return;
            const int P = row.R;
            double *dst = a.W + (size_t)P * a.ldw;
#pragma unroll
            for (int cc = 0; cc < NB; cc++) {
                pr[cc] = row.v[cc];
                if (s + cc < ncols) dst[s_cols[s + cc]] = row.v[cc];
            }
            row.act = false;
            s_active[P] = 0;
            a.rowstep[P] = k; pivrow[k] = P;
            if (a.dense_flag) a.dense_flag[k] = 1;
            ctl->steps[s] = k; ctl->prow[s] = P;
            const int Q = s_rowat[k];   // dlaswp.go: the row at logical k moves to jp
            s_lpos[Q] = jp; s_rowat[jp] = Q;
            s_lpos[P] = k; s_rowat[k] = P;
        };
        GOMILP_FOR_ROWS(publish);
        __syncthreads();
        const double piv = pr[0];
        const bool singular = (piv == 0);  // dgetf2.go:48-49: no scaling, the rank-1 update is a no-op
        if (singular && tid == 0) a.st->lu_singular = 1;
        const double rinv = 1.0 / piv;
        auto elim = [&](CPanelRow<NB> &row, int) {
            if (!row.act) return;
            const double l = singular ? row.v[0] : __dmul_rn(row.v[0], rinv);
            a.W[(size_t)row.R * a.ldw + k] = l;
            const double nl = -l;
            const bool skip = singular;   // Dger (dgetf2.go:60-66) does not skip zero multipliers
#pragma unroll
            for (int cc = 1; cc < NB; cc++) row.v[cc - 1] = skip ? row.v[cc] : __dadd_rn(__dmul_rn(nl, pr[cc]), row.v[cc]);
            row.v[NB - 1] = 0.0;
        };
        GOMILP_FOR_ROWS(elim);
        s++;
        kcur = k + 1;
    }
    for (int R = tid; R < m; R += T) a.lpos[R] = s_lpos[R];
    if (tid == 0) {
        ctl->k0 = k0; ctl->k1 = k1; ctl->k_next = k1; ctl->nsteps = s; ctl->ncols = ncols;
        ctl->rounds += 1;
    }
#undef GOMILP_FOR_ROWS
}

// dense pivot rows of the round, columns j >= k1:  u_s = a[P_s] + sum_{t<s} (-l[P_s][t]) * u_t  (ascending t, the
// Dtrsm of dgetrf.go:57-60).  Register columns of the panel (ctl->cols) are final already: the panel wrote them.
template <int NB>
__global__ __launch_bounds__(256) void k_luc_usolve(LUArgs a) {
    const LUCtl *ctl = a.ctl;
    const int ns = ctl->nsteps, k1 = ctl->k1;
    if (ns == 0 || k1 + (int)blockIdx.x * 256 >= a.m) return;
    __shared__ double Ln[NB][NB + 1];
    __shared__ int Ps[NB], Cs[NB];
    if (threadIdx.x < NB) {
        Ps[threadIdx.x] = threadIdx.x < ns ? ctl->prow[threadIdx.x] : 0;
        Cs[threadIdx.x] = threadIdx.x < ctl->ncols ? ctl->cols[threadIdx.x] : -1;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        const int s = idx / NB, t = idx % NB;
        Ln[s][t] = (s < ns && t < s) ? -a.W[(size_t)Ps[s] * a.ldw + Cs[t]] : 0.0;
    }
    __syncthreads();
    const int j = k1 + blockIdx.x * 256 + threadIdx.x;
    if (j >= a.m) return;
#pragma unroll
    for (int t = 0; t < NB; t++)
        if (Cs[t] == j) return;
    double u[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < ns) {
            double *cell = a.W + (size_t)Ps[s] * a.ldw + j;
            double x = *cell;
#pragma unroll
            for (int t = 0; t < s; t++) {
                const double l = Ln[s][t];
                x = (l != 0) ? __dadd_rn(__dmul_rn(l, u[t]), x) : x;
            }
            u[s] = x;
            *cell = x;
        } else {
            u[s] = 0;
        }
    }
}

// every other row, columns j >= k1: a[R][j] += sum_s (-l[R][s]) * u_s[j] over the round's dense steps in ascending s
// (the Dgemm of dgetrf.go:62-66).  Rows still active take all steps; a row retired by a bookkeeping step of the round
// takes the steps before its own and already holds final values in the panel's register columns.
template <int NB>
__global__ __launch_bounds__(256) void k_luc_trail(LUArgs a) {
    const LUCtl *ctl = a.ctl;
    const int ns = ctl->nsteps, k0 = ctl->k0, k1 = ctl->k1;
    const int j0 = k1 + blockIdx.x * 64;
    if (ns == 0 || j0 >= a.m) return;
    __shared__ double Ls[64][NB + 1];
    __shared__ double Us[NB][64];
    __shared__ int pre[64];
    __shared__ unsigned char retired[64], inlist[64];
    __shared__ int Ss[NB], Ps[NB], Cs[NB];
    const int tid = threadIdx.x;
    const int R0 = blockIdx.y * 64;
    if (tid < NB) {
        Ss[tid] = tid < ns ? ctl->steps[tid] : 0x7fffffff;
        Ps[tid] = tid < ns ? ctl->prow[tid] : 0;
        Cs[tid] = tid < ctl->ncols ? ctl->cols[tid] : -1;
    }
    __syncthreads();
    int p = 0;
    if (tid < 64) {
        const int R = R0 + tid;
        bool ret = false;
        if (R < a.m) {
            const int rs = a.rowstep[R];
            if (rs < 0) {
                p = ns;
            } else if (rs >= k0) {
                int cnt = 0;
                bool dense = false;
                for (int s = 0; s < ns; s++) { cnt += (Ss[s] < rs) ? 1 : 0; dense |= (Ss[s] == rs); }
                if (!dense) { p = cnt; ret = true; }
            }
        }
        pre[tid] = p;
        retired[tid] = ret ? 1 : 0;
    } else if (tid < 128) {
        const int j = j0 + (tid - 64);
        bool il = false;
        for (int t = 0; t < NB; t++) il |= (Cs[t] == j);
        inlist[tid - 64] = il ? 1 : 0;
    }
    if (!__syncthreads_or(p > 0)) return;
    for (int idx = tid; idx < 64 * NB; idx += 256) {
        const int r = idx / NB, s = idx % NB;
        Ls[r][s] = (s < pre[r]) ? -a.W[(size_t)(R0 + r) * a.ldw + Ss[s]] : 0.0;
    }
    for (int idx = tid; idx < NB * 64; idx += 256) {
        const int s = idx / 64, c = idx % 64;
        const int j = j0 + c;
        Us[s][c] = (s < ns && j < a.m) ? a.W[(size_t)Ps[s] * a.ldw + j] : 0.0;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    double acc[4][4];
    bool live[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = ty * 4 + rr;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int c = tx * 4 + cc;
            live[rr][cc] = pre[r] > 0 && (j0 + c) < a.m && !(retired[r] && inlist[c]);
            acc[rr][cc] = live[rr][cc] ? a.W[(size_t)(R0 + r) * a.ldw + j0 + c] : 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < ns) {
            double l[4], u[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) l[rr] = Ls[ty * 4 + rr][s];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) u[cc] = Us[s][tx * 4 + cc];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) {
                const bool nz = l[rr] != 0;
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[rr][cc] = nz ? __dadd_rn(__dmul_rn(l[rr], u[cc]), acc[rr][cc]) : acc[rr][cc];
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = ty * 4 + rr;
#pragma unroll
        for (int cc = 0; cc < 4; cc++)
            if (live[rr][cc]) a.W[(size_t)(R0 + r) * a.ldw + j0 + tx * 4 + cc] = acc[rr][cc];
    }
}

__global__ void k_luc_init(LUArgs a) {
    const int R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R < a.m) {
        a.lpos[R] = R; a.rowstep[R] = -1;
        if (a.dense_flag) a.dense_flag[R] = 0;
    }
    if (R == 0) {
        LUCtl *c = a.ctl;
        c->k_next = 0; c->k0 = 0; c->k1 = 0; c->nsteps = 0; c->ncols = 0; c->rounds = 0;
    }
}

template <int T, int RPT, int NB>
static void luc_rounds(const LUArgs &a, int32_t *pivrow, int nrounds, hipStream_t s) {
    const int m = a.m;
    for (int r = 0; r < nrounds; r++) {
        hipLaunchKernelGGL((k_luc_panel<T, RPT, NB>), dim3(1), dim3(T), 0, s, a, pivrow);
        hipLaunchKernelGGL((k_luc_usolve<NB>), dim3((m + 255) / 256), dim3(256), 0, s, a);
        hipLaunchKernelGGL((k_luc_trail<NB>), dim3((m + 63) / 64, (m + 63) / 64), dim3(256), 0, s, a);
    }
}

bool lu_compressed_supported(int m) { return m <= 4096; }
static int luc_cfg(int m) {
    if (const char *e = getenv("GOMILP_LUC_CFG")) return atoi(e);
    return m <= 512 ? 0 : (m <= 1024 ? 1 : (m <= 2048 ? 2 : 3));
}
int lu_compressed_nb(int m) { const int c = luc_cfg(m); return c <= 1 ? 32 : (c == 2 ? 16 : 8); }

void launch_luc_init(const LUArgs &a, hipStream_t s) {
    hipLaunchKernelGGL(k_luc_init, dim3((a.m + 255) / 256), dim3(256), 0, s, a);
}

// enqueue `nrounds` rounds; returns the number of kernel launches
int launch_luc_rounds(const LUArgs &a, int32_t *pivrow, int nrounds, hipStream_t s) {
    const int m = a.m;
    const int c = luc_cfg(m);
    if (c == 0) luc_rounds<512, 1, 32>(a, pivrow, nrounds, s);
    else if (c == 1) luc_rounds<1024, 1, 32>(a, pivrow, nrounds, s);
    else if (c == 2) luc_rounds<1024, 2, 16>(a, pivrow, nrounds, s);
    else luc_rounds<1024, 4, 8>(a, pivrow, nrounds, s);
    return 3 * nrounds;
}

}  // namespace gomilp
