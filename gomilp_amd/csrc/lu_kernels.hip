// Blocked gonum-order LU for the final basis solve (gfx950).
//
// Same arithmetic, element by element, as lapack/gonum/dgetrf.go:29-70 (panel Dgetf2, Dtrsm for the U block,
// Dgemm for the trailing matrix): every entry receives its rank-1 contributions in ascending k as a rounded
// multiply followed by a rounded add, a_ij = (-l_ik)*u_kj + a_ij, multipliers are a_ik*(1/a_kk), the pivot is the
// first maximum |a_ik| in LAPACK's logical row order.  Only the schedule differs from the reference:
//   k_lu_panel   one workgroup keeps the nb panel columns of all remaining rows in REGISTERS and runs the nb
//                elimination steps with two workgroup barriers each (no kernel launch per column);
//   k_lu_usolve  finishes the nb pivot rows to the right of the panel (the Dtrsm of dgetrf.go:57-60);
//   k_lu_trail   rank-nb update of the remaining rows, nb sequential multiply-adds per element held in registers
//                (the Dgemm of dgetrf.go:62-66), 64x64 tiles, operands staged in LDS.
// Rows never move: lpos[R] is the logical position of physical row R, rowstep[R] the step at which it became
// a pivot row, pivrow[k] the physical row of step k.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

__device__ __forceinline__ void lu_take3(unsigned long long &k, unsigned int &l, unsigned int &r, unsigned long long k2,
                                         unsigned int l2, unsigned int r2) {
    if (k2 < k || (k2 == k && l2 < l)) { k = k2; l = l2; r = r2; }
}

// One row of the panel held in registers.  The rows of a thread are separate named objects (not an
// array) and the column vector is an ext_vector: after full unrolling every access is a constant index,
// so nothing is demoted to scratch.
template <int NB>
struct PanelRow {
    typedef double vec __attribute__((ext_vector_type(NB)));
    vec v;
    int lp, R;
    bool act, was;
};

template <int T, int RPT, int NB>
__global__ __launch_bounds__(T) void k_lu_panel(LUArgs a, int k0, int nb, int32_t *__restrict__ pivrow) {
    // The elimination loop is ROLLED (a fully unrolled panel is >100 KB of straight-line code and runs at
    // instruction-fetch speed): after every step the row registers are shifted left by one column, so the
    // column being eliminated always sits in v[0] and the loop body has constant register indices.
    constexpr int NW = T / 64;
    __shared__ double prow[2][NB];
    __shared__ unsigned long long sk[NW];
    __shared__ unsigned int sl[NW], sr[NW];
    __shared__ unsigned char s_active[T * RPT];  // row still available as a pivot row
    __shared__ int s_jp[2];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    PanelRow<NB> rows[RPT];
#define GOMILP_FOR_ROWS(F)                              \
    do {                                                \
        _Pragma("unroll") for (int rr_ = 0; rr_ < RPT; rr_++) F(rows[rr_], rr_); \
    } while (0)
    auto load_row = [&](PanelRow<NB> &row, int r) {
        row.R = tid + r * T;
        row.act = row.was = (row.R < a.m) && (a.rowstep[row.R < a.m ? row.R : 0] < 0);
        row.lp = row.act ? a.lpos[row.R] : 0x7fffffff;
        const double *src = a.W + (size_t)(row.act ? row.R : 0) * a.ldw + k0;
#pragma unroll
        for (int c = 0; c < NB; c++) row.v[c] = (row.act && c < nb) ? src[c] : 0.0;
    };
    GOMILP_FOR_ROWS(load_row);
    auto mark_row = [&](PanelRow<NB> &row, int) { s_active[row.R] = row.act ? 1 : 0; };
    GOMILP_FOR_ROWS(mark_row);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < nb; s++) {
        const int k = k0 + s;
        // Unit-column fast path.  If column k of ab is the unit vector e_r and row r has not been a pivot row yet, no
        // earlier step can have touched the column (fill-in in column k only comes through row r), so Idamax picks
        // row r (pivot exactly 1), every multiplier is exactly 0 and the step is pure bookkeeping: no reduction, no
        // elimination arithmetic.  Most basis columns of a B&B relaxation are slack columns, so most steps are trivial.
        // (s_active[ur] is only cleared between this step's two barriers, i.e. after every thread took this test.)
        const int ur = a.unit_row ? a.unit_row[k] : -1;
        const bool trivial = (ur >= 0) && (s_active[ur] != 0);
        if (tid == 0 && a.dense_flag) a.dense_flag[k] = trivial ? 0 : 1;
        int P, jp;
        if (!trivial) {
            unsigned long long bk = ~0ull;
            unsigned int bl = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
            auto cand = [&](PanelRow<NB> &row, int) {
                if (row.act) lu_take3(bk, bl, br, ordkey(-fabs(row.v[0])), (unsigned int)row.lp, (unsigned int)row.R);
            };
            GOMILP_FOR_ROWS(cand);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                unsigned long long k2 = __shfl_xor(bk, o, 64);
                unsigned int l2 = __shfl_xor(bl, o, 64), r2x = __shfl_xor(br, o, 64);
                lu_take3(bk, bl, br, k2, l2, r2x);
            }
            if (lane == 0) { sk[w] = bk; sl[w] = bl; sr[w] = br; }
            __syncthreads();
            bk = sk[0]; bl = sl[0]; br = sr[0];
#pragma unroll
            for (int t = 1; t < NW; t++) lu_take3(bk, bl, br, sk[t], sl[t], sr[t]);
            P = (int)br; jp = (int)bl;
        } else {
            auto tell = [&](PanelRow<NB> &row, int) { if (row.act && row.R == ur) s_jp[s & 1] = row.lp; };
            GOMILP_FOR_ROWS(tell);
            __syncthreads();
            P = ur; jp = s_jp[s & 1];
        }
        double *pr = prow[s & 1];
        auto publish = [&](PanelRow<NB> &row, int) {
            if (!row.act) return;
            if (row.R == P) {
                // this row becomes U row k: its remaining panel entries are final
                double *dst = a.W + (size_t)P * a.ldw + k;
#pragma unroll
                for (int c = 0; c < NB; c++) { pr[c] = row.v[c]; if (c < nb - s) dst[c] = row.v[c]; }
                row.act = false; row.lp = k;
                s_active[P] = 0;
                a.rowstep[P] = k; pivrow[k] = P;
            } else if (row.lp == k) {
                row.lp = jp;  // the row that sat at logical k moves to jp (dlaswp.go)
            }
        };
        GOMILP_FOR_ROWS(publish);
        __syncthreads();
        const double piv = pr[0];
        const bool singular = (!trivial) && (piv == 0);  // dgetf2.go:48-49: no scaling, the rank-1 update is a no-op
        if (singular && tid == 0) a.st->lu_singular = 1;
        const bool skip = trivial || singular;
        const double rinv = 1.0 / piv;
        auto elim = [&](PanelRow<NB> &row, int) {
            if (!row.act) return;
            const double l = skip ? row.v[0] : __dmul_rn(row.v[0], rinv);
            if (!trivial) a.W[(size_t)row.R * a.ldw + k] = l;  // trivial: the entry is the 0 already in W
            const double nl = -l;
#pragma unroll
            for (int c = 1; c < NB; c++) row.v[c - 1] = skip ? row.v[c] : __dadd_rn(__dmul_rn(nl, pr[c]), row.v[c]);
            row.v[NB - 1] = 0.0;
        };
        GOMILP_FOR_ROWS(elim);
    }
    auto store_row = [&](PanelRow<NB> &row, int) {
        if (row.was) a.lpos[row.R] = row.lp;
    };
    GOMILP_FOR_ROWS(store_row);
#undef GOMILP_FOR_ROWS
}

// U rows of the panel, columns to the right: u_s = a[P_s] + sum_{s'<s} (-l[P_s][s']) * u_{s'}   (ascending s')
template <int NB>
__global__ __launch_bounds__(256) void k_lu_usolve(LUArgs a, int k0, int nb, const int32_t *__restrict__ pivrow) {
    __shared__ double Ln[NB][NB + 1];
    __shared__ int Ps[NB];
    for (int idx = threadIdx.x; idx < NB * NB; idx += 256) {
        const int s = idx / NB, t = idx % NB;
        Ln[s][t] = (s < nb && t < s) ? -a.W[(size_t)pivrow[k0 + s] * a.ldw + k0 + t] : 0.0;
    }
    if (threadIdx.x < NB) Ps[threadIdx.x] = threadIdx.x < nb ? pivrow[k0 + threadIdx.x] : 0;
    __syncthreads();
    const int j = k0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (j >= a.m) return;
    double u[NB];
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < nb) {
            double *cell = a.W + (size_t)Ps[s] * a.ldw + j;
            double x = *cell;
#pragma unroll
            for (int t = 0; t < s; t++) x = __dadd_rn(__dmul_rn(Ln[s][t], u[t]), x);
            u[s] = x;
            *cell = x;
        } else {
            u[s] = 0;
        }
    }
}

// trailing rows: a[R][j] += sum_s (-l[R][s]) * u_s[j] in ascending s.  64x64 tile, 4x4 per thread.
template <int NB>
__global__ __launch_bounds__(256) void k_lu_trail(LUArgs a, int k0, int nb, const int32_t *__restrict__ pivrow) {
    __shared__ double Ls[64][NB + 1];
    __shared__ double Us[NB][64];
    __shared__ int actrow[64];
    const int R0 = blockIdx.y * 64, j0 = k0 + nb + blockIdx.x * 64;
    if (threadIdx.x < 64) {
        const int R = R0 + threadIdx.x;
        actrow[threadIdx.x] = (R < a.m) && (a.rowstep[R] < 0);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 64 * NB; idx += 256) {
        const int r = idx / NB, s = idx % NB;
        Ls[r][s] = (actrow[r] && s < nb) ? -a.W[(size_t)(R0 + r) * a.ldw + k0 + s] : 0.0;
    }
    for (int idx = threadIdx.x; idx < NB * 64; idx += 256) {
        const int s = idx / 64, c = idx % 64;
        const int j = j0 + c;
        Us[s][c] = (s < nb && j < a.m) ? a.W[(size_t)pivrow[k0 + s] * a.ldw + j] : 0.0;
    }
    __syncthreads();
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
    double acc[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = ty * 4 + rr;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int j = j0 + tx * 4 + cc;
            acc[rr][cc] = (actrow[r] && j < a.m) ? a.W[(size_t)(R0 + r) * a.ldw + j] : 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < NB; s++) {
        if (s < nb) {
            double l[4], u[4];
#pragma unroll
            for (int rr = 0; rr < 4; rr++) l[rr] = Ls[ty * 4 + rr][s];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) u[cc] = Us[s][tx * 4 + cc];
#pragma unroll
            for (int rr = 0; rr < 4; rr++)
#pragma unroll
                for (int cc = 0; cc < 4; cc++) acc[rr][cc] = __dadd_rn(__dmul_rn(l[rr], u[cc]), acc[rr][cc]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < 4; rr++) {
        const int r = ty * 4 + rr;
        if (!actrow[r]) continue;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const int j = j0 + tx * 4 + cc;
            if (j < a.m) a.W[(size_t)(R0 + r) * a.ldw + j] = acc[rr][cc];
        }
    }
}

// Wd[R*nd + t] = W[R][dlist[t]] (the columns whose elimination step did arithmetic) and diag[R] = W[R][lpos[R]]:
// everything the host triangular solves need, m*(nd+1) doubles instead of m*m
__global__ void k_lu_pack(LUArgs a, const int32_t *__restrict__ dlist, int nd, double *__restrict__ Wd, double *__restrict__ diag) {
    const int R = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const double *row = a.W + (size_t)R * a.ldw;
    if (t < nd) Wd[(size_t)R * nd + t] = row[dlist[t]];
    if (blockIdx.x == 0 && threadIdx.x == 0) diag[R] = row[a.lpos[R]];
}

__global__ void k_lu_blocked_init(LUArgs a) {
    const int R = blockIdx.x * blockDim.x + threadIdx.x;
    if (R < a.m) { a.lpos[R] = R; a.rowstep[R] = -1; }
}

template <int T, int RPT, int NB>
static void lu_blocked_t(const LUArgs &a, int32_t *pivrow, hipStream_t s) {
    const int m = a.m;
    hipLaunchKernelGGL(k_lu_blocked_init, dim3((m + 255) / 256), dim3(256), 0, s, a);
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int nb = (m - k0 < NB) ? (m - k0) : NB;
        hipLaunchKernelGGL((k_lu_panel<T, RPT, NB>), dim3(1), dim3(T), 0, s, a, k0, nb, pivrow);
        const int rem = m - k0 - nb;
        if (rem > 0) {
            hipLaunchKernelGGL((k_lu_usolve<NB>), dim3((rem + 255) / 256), dim3(256), 0, s, a, k0, nb, pivrow);
            hipLaunchKernelGGL((k_lu_trail<NB>), dim3((rem + 63) / 64, (m + 63) / 64), dim3(256), 0, s, a, k0, nb, pivrow);
        }
    }
}

bool lu_blocked_supported(int m) { return m <= 4096; }

void launch_lu_pack(const LUArgs &a, const int32_t *dlist, int nd, double *Wd, double *diag, hipStream_t s) {
    dim3 grid((nd + 255) / 256 > 0 ? (nd + 255) / 256 : 1, a.m);
    hipLaunchKernelGGL(k_lu_pack, grid, dim3(256), 0, s, a, dlist, nd, Wd, diag);
}

// returns the number of kernel launches enqueued
int launch_lu_blocked(const LUArgs &a, int32_t *pivrow, hipStream_t s) {
    const int m = a.m;
    // measured at m = 2048 (gfx950): <1024,2,16> 80 us per 16-column panel, <256,8,16> 190 us: the per-step
    // reduce/publish/shift overhead is issue-bound, more resident waves hide it better than fewer, fatter ones
    if (m <= 512) { lu_blocked_t<512, 1, 32>(a, pivrow, s); return 1 + 3 * ((m + 31) / 32); }
    if (m <= 1024) { lu_blocked_t<1024, 1, 32>(a, pivrow, s); return 1 + 3 * ((m + 31) / 32); }
    if (m <= 2048) { lu_blocked_t<1024, 2, 16>(a, pivrow, s); return 1 + 3 * ((m + 15) / 16); }
    lu_blocked_t<1024, 4, 8>(a, pivrow, s);
    return 1 + 3 * ((m + 7) / 8);
}

}  // namespace gomilp
