// findLinearlyIndependent (simplex.go:611-637) on the device, for starting bases that are not slack bases (equality rows).
//
// The reference walks the columns of A from the last to the first and keeps a column when mat.Cond(columns so far + candidate,
// 1) <= 1e12 — a fresh factorisation per candidate, O(m^4).  engine_general.cpp carries ONE Householder QR along on the host
// (O(m^2 n)), but every candidate still streams the reflectors and R^-1 (2 x 8 m^2 bytes) through one core: 46 ms at 600 rows,
// 180 ms at 1000, an order of magnitude more than all pivots of the solve.  Here Q^T is kept EXPLICITLY in HBM, so a candidate
// costs a matrix-vector product instead of a chain of k dependent reflector applications, and every step is a grid-wide kernel:
//   k_gs_w       w = Q^T a                              (one wave per row)
//   k_gs_t       t = R^-1 w_top                         (one wave per row of the triangle)
//   k_gs_decide  ONE workgroup: |w_bottom|, the new diagonal entry beta, the 1-norms of R' and R'^-1 exactly as the host form
//                (cond = max(|R|_1, |r'|_1) max(|R^-1|_1, |r'^-1|_1)), accept = !(cond > 1e12) (a NaN passes, like the
//                reference); on acceptance the new column of R^-1, the reflector v and the index list
//   k_gs_y       y = v^T Q^T                            (accepted candidates only)
//   k_gs_rank1   Q^T -= (2 / v^T v) v y                 (accepted candidates only)
// No host round trip per candidate: the decisions live in a device state block; the host enqueues candidates in chunks and
// looks at the state once per chunk.  The scan stops when m - 1 columns are accepted: the last column makes the matrix square,
// where the reference measures kappa_1 of the matrix itself through its LU — the host does that with the inversion it needs
// for B^-1 anyway (engine_general.cpp: general_finish_last_column).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

namespace {

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off);
    return x;
}

}  // namespace

// Q^T = I, state = empty
__global__ __launch_bounds__(256) void k_gs_init(double *QT, int ldq, int m, GsState *st) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < ldq; c += 256) QT[(size_t)r * ldq + c] = (c == r) ? 1.0 : 0.0;
    if (r == 0 && threadIdx.x == 0) {
        st->k = 0; st->accept = 0; st->kacc = 0; st->done = 0; st->stop_col = -1; st->scanned = 0;
        st->nR = 0; st->nRinv = 0; st->vv = 0;
    }
}

// state after a leading run of `s0` accepted UNIT columns (decided on the host in O(1) each: engine.cpp): Q^T is the signed
// permutation the Householder steps of those columns produce (row i = sgn[i] * e_perm[i]^T, exactly what the rank-1 updates
// would have left: they only move and negate 0 / +-1 entries), R = R^-1 = diag(beta), |R|_1 = |R^-1|_1 = 1
__global__ __launch_bounds__(256) void k_gs_init_perm(double *QT, double *Rinv, int ldq, int m, const int32_t *__restrict__ perm, const double *__restrict__ sgn,
                                                      const double *__restrict__ beta, int s0, GsState *st) {
    const int r = blockIdx.x;
    const int pc = perm[r];
    const double sv = sgn[r];
    for (int c = threadIdx.x; c < ldq; c += 256) QT[(size_t)r * ldq + c] = (c == pc) ? sv : 0.0;
    if (r < s0 && threadIdx.x == 0) Rinv[(size_t)r * ldq + r] = 1.0 / beta[r];
    if (r == 0 && threadIdx.x == 0) {
        st->k = s0; st->accept = 0; st->kacc = 0; st->done = (s0 >= m - 1) ? 1 : 0; st->stop_col = -1; st->scanned = 0;
        st->nR = s0 ? 1.0 : 0.0; st->nRinv = s0 ? 1.0 : 0.0; st->vv = 0;
    }
}

__global__ __launch_bounds__(256) void k_gs_w(const double *__restrict__ acol, const double *__restrict__ QT, int ldq, int m, double *__restrict__ w, const GsState *st, int force) {
    if (st->done && !force) return;
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= m) return;
    const double *row = QT + (size_t)r * ldq;
    double s0 = 0, s1 = 0;
    int c = lane;
    for (; c + 64 < m; c += 128) { s0 += row[c] * acol[c]; s1 += row[c + 64] * acol[c + 64]; }
    if (c < m) s0 += row[c] * acol[c];
    const double s = wave_sum_f64(s0 + s1);
    if (lane == 0) w[r] = s;
}

__global__ __launch_bounds__(256) void k_gs_t(const double *__restrict__ Rinv, int ldq, const double *__restrict__ w, double *__restrict__ t, const GsState *st, int force) {
    if (st->done && !force) return;
    const int k = st->k;
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= k) return;
    const double *row = Rinv + (size_t)r * ldq;
    double s0 = 0;
    for (int c = r + lane; c < k; c += 64) s0 += row[c] * w[c];
    const double s = wave_sum_f64(s0);
    if (lane == 0) t[r] = s;
}

__global__ __launch_bounds__(1024) void k_gs_decide(double *__restrict__ Rinv, int ldq, int m, double *__restrict__ w, const double *__restrict__ t, int cand, int32_t *__restrict__ idxs,
                                                   GsState *st, int last) {
    // last = 1: the square step (k = m - 1): the column is taken tentatively, the host judges kappa_1 of the matrix itself
    if (st->done && !last) {   // the scan is over: the later candidates of the chunk do nothing
        if (threadIdx.x == 0) st->accept = 0;
        return;
    }
    __shared__ double red[3][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int k = st->k;
    // scale = max |w[r]|, r >= k
    double mx = 0;
    for (int r = k + tid; r < m; r += 1024) mx = fmax(mx, fabs(w[r]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
    if (lane == 0) red[0][wv] = mx;
    __syncthreads();
    mx = 0;
    for (int i = 0; i < 16; i++) mx = fmax(mx, red[0][i]);
    __syncthreads();
    const double inv_s = (mx > 0 && mx < __builtin_inf()) ? 1.0 / mx : 0.0;
    double ss = 0, cs = 0, csi = 0;
    for (int r = k + tid; r < m; r += 1024) { const double x = w[r] * inv_s; ss += x * x; }
    for (int r = tid; r < k; r += 1024) { cs += fabs(w[r]); csi += fabs(t[r]); }
    ss = wave_sum_f64(ss); cs = wave_sum_f64(cs); csi = wave_sum_f64(csi);
    if (lane == 0) { red[0][wv] = ss; red[1][wv] = cs; red[2][wv] = csi; }
    __syncthreads();
    ss = cs = csi = 0;
    for (int i = 0; i < 16; i++) { ss += red[0][i]; cs += red[1][i]; csi += red[2][i]; }
    double nrm = (mx > 0) ? mx * sqrt(ss) : 0.0;
    if (mx != mx || mx == __builtin_inf()) nrm = mx;   // NaN / Inf in the candidate: propagate
    const double alpha = w[k];
    const double beta = alpha >= 0 ? -nrm : nrm;
    const double nR = st->nR, nRinv = st->nRinv;
    cs += fabs(beta);                                   // |r'|_1 : new column of R
    csi = beta != 0 ? (csi / fabs(beta) + fabs(1.0 / beta)) : __builtin_inf();   // |r'^-1|_1 : new column of R^-1 = (-R^-1 w_top / beta, 1 / beta)
    bool accept;
    if (k == 0 || last) accept = true;                  // simplex.go:624-629: the first column is always kept
    else {
        double cond;
        if (beta == 0 || !(nRinv < __builtin_inf()) || !(csi < __builtin_inf())) cond = __builtin_inf();   // (NaN norms count as not finite, like std::isfinite)
        else cond = fmax(nR, cs) * fmax(nRinv, csi);
        accept = !(cond > 1e12);                        // :630 (a NaN condition number passes, as in the reference)
    }
    __syncthreads();
    if (accept) {
        const double ninvb = beta != 0 ? -1.0 / beta : __builtin_inf();
        for (int r = tid; r < k; r += 1024) Rinv[(size_t)r * ldq + k] = t[r] * ninvb;
        // reflector v = (0, ..., 0, alpha - beta, w[k+1..m)) in place of w
        double vv = 0;
        for (int r = tid; r < m; r += 1024) {
            const double v = r < k ? 0.0 : (r == k ? alpha - beta : w[r]);
            w[r] = v;
            vv += v * v;
        }
        vv = wave_sum_f64(vv);
        if (lane == 0) red[0][wv] = vv;
        __syncthreads();
        if (tid == 0) {
            vv = 0;
            for (int i = 0; i < 16; i++) vv += red[0][i];
            Rinv[(size_t)k * ldq + k] = beta != 0 ? 1.0 / beta : __builtin_inf();
            st->vv = vv; st->kacc = k; st->accept = 1;
            st->nR = k == 0 ? cs : fmax(nR, cs);
            st->nRinv = k == 0 ? csi : fmax(nRinv, csi);
            idxs[k] = cand;
            st->k = k + 1;
            if (k + 1 >= m - 1 && !last) { st->done = 1; st->stop_col = cand - 1; }   // the last column: the square step
            if (last) st->beta_last = beta;
        }
    } else if (tid == 0) st->accept = 0;
    if (tid == 0) st->scanned = st->scanned + 1;
}

// partial y: ypart[by][c] = sum over the rows r = kacc + 4 by + q, step 4 * gridDim.y, of v[r] Q^T[r][c]; workgroup = 64 columns x
// 4 interleaved row classes, grid.y row slices (a fixed summation tree: the result does not depend on scheduling)
constexpr int kGsSlices = 16;
__global__ __launch_bounds__(256) void k_gs_y(const double *__restrict__ QT, int ldq, int m, const double *__restrict__ v, double *__restrict__ ypart, const GsState *st, int force) {
    if (!st->accept) return;
    __shared__ double part[4][64];
    const int k = st->kacc;
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s0 = 0, s1 = 0;
    if (c < m) {
        const int step = 4 * (int)gridDim.y;
        int r = k + 4 * (int)blockIdx.y + q;
        for (; r + step < m; r += 2 * step) { s0 += v[r] * QT[(size_t)r * ldq + c]; s1 += v[r + step] * QT[(size_t)(r + step) * ldq + c]; }
        if (r < m) s0 += v[r] * QT[(size_t)r * ldq + c];
    }
    part[q][lane] = s0 + s1;
    __syncthreads();
    if (q == 0 && c < m) ypart[(size_t)blockIdx.y * ldq + c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

__global__ __launch_bounds__(256) void k_gs_rank1(double *__restrict__ QT, int ldq, int m, const double *__restrict__ v, const double *__restrict__ ypart, GsState *st, int force) {
    if (!st->accept) return;
    const int k = st->kacc;
    const double vv = st->vv;
    if (!(vv > 0)) return;                               // zero reflector: H = I
    const double f = 2.0 / vv;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    double y = 0;
#pragma unroll
    for (int sl = 0; sl < kGsSlices; sl++) y += ypart[(size_t)sl * ldq + c];
    const double yc = y * f;
    for (int r = k + blockIdx.y; r < m; r += gridDim.y) QT[(size_t)r * ldq + c] -= v[r] * yc;
}

// B^-1 = R^-1 Q^T after the last column (upper triangular times full): C[i][j] = sum_{p >= i} Rinv[i][p] QT[p][j]; 32 x 32 tiles
__global__ __launch_bounds__(256) void k_gs_binv(const double *__restrict__ Rinv, const double *__restrict__ QT, int ldq, int m, double *__restrict__ C) {
    __shared__ double sa[32][33], sb[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads, 4 rows each
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    double acc[4] = {0, 0, 0, 0};
    for (int p0 = i0 & ~31; p0 < m; p0 += 32) {               // Rinv[i][p] = 0 for p < i
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int r = ty + 8 * u;
            sa[r][tx] = (i0 + r < m && p0 + tx < m) ? Rinv[(size_t)(i0 + r) * ldq + p0 + tx] : 0.0;
            sb[r][tx] = (p0 + r < m && j0 + tx < m) ? QT[(size_t)(p0 + r) * ldq + j0 + tx] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < 32; pp++) {
            const double b = sb[pp][tx];
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] += sa[ty + 8 * u][pp] * b;
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
        if (i0 + ty + 8 * u < m && j0 + tx < m) C[(size_t)(i0 + ty + 8 * u) * ldq + j0 + tx] = acc[u];
}

// the square step's two 1-norms without a trip of the matrices to the host: out[sl][c] = sum over the rows r = sl, sl + 8, ... of
// |C[r][c]| (8 row slices: the host adds them in a fixed order), and colsum[p] = sum_r |A[r][idx(p)]| for the m basis columns (rows
// of At; position m - 1 is the candidate)
__global__ __launch_bounds__(256) void k_gs_norms(const double *__restrict__ C, int ldq, int m, double *__restrict__ out, const double *__restrict__ At, int ld,
                                                  const int32_t *__restrict__ idxs, int cand, double *__restrict__ colsum) {
    const int c = blockIdx.x * 256 + threadIdx.x, sl = blockIdx.y;
    if (c < m) {
        double s0 = 0, s1 = 0;
        int r = sl;
        for (; r + 8 < m; r += 16) { s0 += fabs(C[(size_t)r * ldq + c]); s1 += fabs(C[(size_t)(r + 8) * ldq + c]); }
        if (r < m) s0 += fabs(C[(size_t)r * ldq + c]);
        out[(size_t)sl * ldq + c] = s0 + s1;
    }
    // basis columns: one wave per position, the workgroups of slice 0 .. 7 share them
    const int lane = threadIdx.x & 63;
    for (int p = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6); p < m; p += gridDim.x * gridDim.y * 4) {
        const double *col = At + (size_t)(p < m - 1 ? idxs[p] : cand) * ld;
        double sc = 0;
        for (int r = lane; r < m; r += 64) sc += fabs(col[r]);
        sc = wave_sum_f64(sc);
        if (lane == 0) colsum[p] = sc;
    }
}
void launch_gs_norms(const double *C, int ldq, int m, double *out, const double *At, int ld, const int32_t *idxs, int cand, double *colsum, hipStream_t s) {
    hipLaunchKernelGGL(k_gs_norms, dim3((m + 255) / 256, 8), dim3(256), 0, s, C, ldq, m, out, At, ld, idxs, cand, colsum);
}

// Phase-I artificial column of a start from a searched basis (simplex.go:536-542): a_{n+1} = b - sum_{i != minidx} a_{basic_i}, every
// element taking its subtractions in ascending i as `-1 * a + art` (floats.Sub per column: the host loop's operation order — which walked
// a row-major A with stride n, 0.6 ms at 1000 rows); one thread per row, the basis columns are rows of At
__global__ __launch_bounds__(256) void k_gs_art(const double *__restrict__ At, int ld, int m, const int32_t *__restrict__ basic, int minidx, const double *__restrict__ b,
                                                double *__restrict__ art) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= ld) return;
    double acc = k < m ? b[k] : 0.0;
    if (k < m) {
        for (int i = 0; i < m; i++) {
            if (i == minidx) continue;
            acc = __dadd_rn(__dmul_rn(-1.0, At[(size_t)basic[i] * ld + k]), acc);
        }
    }
    art[k] = acc;
}
void launch_gs_art(const double *At, int ld, int m, const int32_t *basic, int minidx, const double *b, double *art, hipStream_t s) {
    hipLaunchKernelGGL(k_gs_art, dim3((ld + 255) / 256), dim3(256), 0, s, At, ld, m, basic, minidx, b, art);
}

void launch_gs_init(double *QT, int ldq, int m, GsState *st, hipStream_t s) { hipLaunchKernelGGL(k_gs_init, dim3(m), dim3(256), 0, s, QT, ldq, m, st); }
void launch_gs_init_perm(double *QT, double *Rinv, int ldq, int m, const int32_t *perm, const double *sgn, const double *beta, int s0, GsState *st, hipStream_t s) {
    hipLaunchKernelGGL(k_gs_init_perm, dim3(m), dim3(256), 0, s, QT, Rinv, ldq, m, perm, sgn, beta, s0, st);
}
// one candidate column: 5 launches
void launch_gs_candidate(const double *acol, double *QT, double *Rinv, int ldq, int m, double *w, double *t, double *ypart, int cand, int32_t *idxs, GsState *st, hipStream_t s,
                         int last) {
    hipLaunchKernelGGL(k_gs_w, dim3((m + 3) / 4), dim3(256), 0, s, acol, QT, ldq, m, w, st, last);
    hipLaunchKernelGGL(k_gs_t, dim3((m + 3) / 4), dim3(256), 0, s, Rinv, ldq, w, t, st, last);
    hipLaunchKernelGGL(k_gs_decide, dim3(1), dim3(1024), 0, s, Rinv, ldq, m, w, t, cand, idxs, st, last);
    hipLaunchKernelGGL(k_gs_y, dim3((m + 63) / 64, kGsSlices), dim3(256), 0, s, QT, ldq, m, w, ypart, st, last);
    hipLaunchKernelGGL(k_gs_rank1, dim3((m + 255) / 256, 32), dim3(256), 0, s, QT, ldq, m, w, ypart, st, last);
}
int gs_scratch_rows() { return 2 + kGsSlices; }   // w, t, the y partials: rows of ldq doubles
void launch_gs_binv(const double *Rinv, const double *QT, int ldq, int m, double *C, hipStream_t s) {
    hipLaunchKernelGGL(k_gs_binv, dim3((m + 31) / 32, (m + 31) / 32), dim3(256), 0, s, Rinv, QT, ldq, m, C);
}

}  // namespace gomilp
