// findLinearlyIndependent (simplex.go:611-637) on the device, BLOCKED: NB candidate columns per four launches (round 5).
//
// general_kernels.hip spends five launches on every candidate, each of them a grid-wide dependency of the next (w needs all of
// Q^T, the decision all of w, the rank-1 update the decision): at 1000 rows 21 of the 28 ms of a solve were launch gaps.  Here
// the same decisions — kappa_1 of the triangular factor from R^-1, `!(cond > 1e12)`, simplex.go:630 — are taken for NB
// candidates (the columns col0, col0 - 1, ...: the scan runs from the last column to the first) behind ONE pass over Q^T:
//   k_gsb_w       W[j] = Q^T a_j for the NB candidates of the block      (one wave per two rows of Q^T, 2 NB sums per lane
//                                                                          folded across the wave in 2 NB exchanges)
//   k_gsb_t       T0[j] = R^-1 W[j] over the k0 columns accepted BEFORE the block   (the same routine over R^-1's triangle)
//   k_gsb_decide  ONE workgroup walks the candidates in scan order with all NB vectors in registers: candidate j is first
//                 brought up to date with the reflectors accepted earlier IN the block (w_j' -= (2 / v^T v)(v^T w_j') v, applied to
//                 every later candidate the moment a column is accepted), its t = R^-1 w_top is T0 plus the block's own new
//                 columns of R^-1 (registers), the norms and the decision are those of k_gs_decide; two workgroup
//                 reductions per candidate (scale + the values everybody needs | the sums)
//   k_gsb_apply   Q^T <- H_last ... H_first Q^T for the reflectors the block accepted: a workgroup holds a strip of TCOLS columns
//                 of Q^T with all its rows in registers and applies the reflectors one after the other (y = v^T Q^T is a column
//                 sum inside the workgroup)
// The arithmetic is the engine's own (a threshold on a condition number, not values the reference defines bit by bit); what must
// agree, and is tested against the oracle's restatement of the reference and against the per-candidate form, is the list of
// accepted columns.  The square step (the last column) stays with general_kernels.hip / the host (engine.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

namespace {

constexpr int kGbDecT = 512;    // threads of the decide workgroup (two waves per SIMD: 256 registers per lane)
constexpr int kGbAppT = 1024;   // threads of an apply workgroup

// v[0..N) per lane -> every lane l holds the wave's total of v[l & (N - 1)]: a fixed tree of N - 1 + log2(64 / N) exchanges
// instead of 6 N (step s pairs the lanes that differ in bit s: one keeps the even entries of the pair, the other the odd ones)
template <int N, bool MAX = false>
__device__ __forceinline__ double wave_fold(double (&v)[N], int lane) {
    static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "power of two");
    int s = 0;
#pragma unroll
    for (int half = N / 2; half >= 1; half >>= 1, s++) {
        const bool up = (lane >> s) & 1;
#pragma unroll
        for (int i = 0; i < half; i++) {
            const double lo = v[2 * i], hi = v[2 * i + 1];
            const double send = up ? lo : hi, keep = up ? hi : lo;
            const double got = __shfl_xor(send, 1 << s);
            v[i] = MAX ? fmax(keep, got) : keep + got;
        }
    }
    double x = v[0];
#pragma unroll
    for (int b = N; b < 64; b <<= 1) {
        const double got = __shfl_xor(x, b);
        x = MAX ? fmax(x, got) : x + got;
    }
    return x;
}

// out[j][r] = sum_{c = cs(r)}^{c1 - 1} M[r][c] X_j[c] for the two rows r0, r0 + 1 of a wave and j < nx <= NB;
// X_j = xbase + j * xstride; cs(r) = tri ? r : 0 (the triangle of R^-1: entries left of the diagonal are never formed)
template <int NB>
__device__ __forceinline__ void gsb_rows_dot(const double *__restrict__ M, int ldq, int r0, int nrows, int c1, bool tri, const double *__restrict__ xbase, long xstride, int nx,
                                             double *__restrict__ out, int lane) {
    double acc[2 * NB];
#pragma unroll
    for (int i = 0; i < 2 * NB; i++) acc[i] = 0.0;
    const bool two = r0 + 1 < nrows;
    const double *m0 = M + (size_t)r0 * ldq, *m1 = M + (size_t)(two ? r0 + 1 : r0) * ldq;
    const int cs0 = tri ? r0 : 0, cs1 = tri ? r0 + 1 : 0;
    const double *xj[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) xj[j] = xbase + (long)(j < nx ? j : nx - 1) * xstride;
    for (int c = cs0 + lane; c < c1; c += 64) {
        const double q0 = m0[c];
        const double q1 = (two && c >= cs1) ? m1[c] : 0.0;
        const bool ok1 = two && c >= cs1;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const double x = xj[j][c];
            acc[j] = __builtin_fma(q0, x, acc[j]);
            acc[NB + j] = ok1 ? __builtin_fma(q1, x, acc[NB + j]) : acc[NB + j];
        }
    }
    const double tot = wave_fold<2 * NB>(acc, lane);
    if (lane < 2 * NB) {
        const int j = lane & (NB - 1), rr = lane / NB;
        if (j < nx && (rr == 0 || two)) out[(size_t)j * ldq + r0 + rr] = tot;
    }
}

}  // namespace

// W[j] = Q^T a_j, j < ncand: candidate j is column col0 - j of A = row col0 - j of At
template <int NB>
__global__ __launch_bounds__(256) void k_gsb_w(const double *__restrict__ At, int ld, int col0, int ncand, const double *__restrict__ QT, int ldq, int m, double *__restrict__ Wb,
                                               const GsState *st) {
    if (st->done) return;
    const int lane = threadIdx.x & 63, r0 = 2 * ((int)blockIdx.x * 4 + ((int)threadIdx.x >> 6));
    if (r0 >= m) return;
    gsb_rows_dot<NB>(QT, ldq, r0, m, m, false, At + (size_t)col0 * ld, -(long)ld, ncand, Wb, lane);
}

// T0[j][r] = sum_{r <= c < k0} R^-1[r][c] W[j][c], r < k0 = the columns accepted before this block
template <int NB>
__global__ __launch_bounds__(256) void k_gsb_t(const double *__restrict__ Rinv, int ldq, int ncand, const double *__restrict__ Wb, double *__restrict__ T0, const GsState *st) {
    if (st->done) return;
    const int k0 = st->k;
    const int lane = threadIdx.x & 63, r0 = 2 * ((int)blockIdx.x * 4 + ((int)threadIdx.x >> 6));
    if (r0 >= k0) return;
    gsb_rows_dot<NB>(Rinv, ldq, r0, k0, k0, true, Wb, (long)ldq, ncand, T0, lane);
}

// The candidates of a block, one after the other (see the head of the file).  Thread tid holds the rows tid + 512 rr.  Per row a
// thread keeps, for every candidate j, ONE value Y[j]: T0[j][r] if r < k0 (the candidate's own entries there are needed for
// |r'|_1 only: their absolute sum A0[j] is taken once, in front of the loop; no reflector of the block touches these rows), the
// running w_j[r] otherwise; and X[i] = the new column of R^-1 of the block's i-th accepted column (position k0 + i) in its row.
template <int RPT, int NB>
__global__ __launch_bounds__(kGbDecT) void k_gsb_decide(double *__restrict__ Rinv, int ldq, int m, const double *__restrict__ Wb, const double *__restrict__ T0, double *__restrict__ V,
                                                       GsBlock *__restrict__ blk, int col0, int ncand, int32_t *__restrict__ idxs, GsState *st) {
    constexpr int T = kGbDecT, NW = T / 64;
    static_assert(NB <= 16 && NW == 8, "the folds below: 16 values, 8 waves");
    __shared__ double redm[2][NW];        // scale
    __shared__ double red1[2][4][NW];     // ss, |w_top|, |t|, sum of squares below the diagonal
    __shared__ double red2[2][16][NW];    // A0 in front of the loop; the dot products of the new reflector with the later candidates
    __shared__ double bc[2][2 * 16];      // w_j at the block's own positions k0 + i | w_j'[k] for j' >= j
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (st->done) {
        if (tid == 0) { blk->nacc = 0; blk->k0 = st->k; }
        return;
    }
    const int k0 = st->k;
    int k = k0, nacc = 0, scanned = 0, done = 0, stop_col = -1;
    double nR = st->nR, nRinv = st->nRinv;
    int row[RPT];
    double Y[RPT][NB], X[RPT][NB];
#pragma unroll
    for (int rr = 0; rr < RPT; rr++) {
        row[rr] = tid + T * rr;
#pragma unroll
        for (int j = 0; j < NB; j++) {
            Y[rr][j] = (row[rr] < m && j < ncand) ? Wb[(size_t)j * ldq + row[rr]] : 0.0;
            X[rr][j] = 0.0;
        }
    }
    // A0[j] = sum_{r < k0} |w_j[r]|; afterwards those rows hold T0
    double a0lane;
    {
        double a[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            a[j] = 0.0;
            if (j < NB) {
#pragma unroll
                for (int rr = 0; rr < RPT; rr++)
                    if (row[rr] < k0) a[j] += fabs(Y[rr][j]);
            }
        }
        const double t = wave_fold<16>(a, lane);
        if (lane < 16) red2[1][lane][wv] = t;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
            if (row[rr] < k0) {
#pragma unroll
                for (int j = 0; j < NB; j++) Y[rr][j] = j < ncand ? T0[(size_t)j * ldq + row[rr]] : 0.0;
            }
        }
        __syncthreads();
        double x = red2[1][lane & 15][(lane >> 4) & 3] + red2[1][lane & 15][((lane >> 4) & 3) + 4];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        a0lane = x;   // lane l: A0[l & 15]
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        if (j < ncand && !done) {   // (uniform)
            const int cand = col0 - j;
            const int buf = j & 1;
            // ---- round A: scale = max |w[r]|, r >= k; the entries of w_j (and of the later candidates) everybody needs
            double mxl = 0.0;
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
                const int r = row[rr];
                if (r >= k && r < m) mxl = fmax(mxl, fabs(Y[rr][j]));
                if (r >= k0 && r < k) bc[buf][r - k0] = Y[rr][j];
                if (r == k) {
#pragma unroll
                    for (int j2 = j; j2 < NB; j2++) bc[buf][16 + j2] = Y[rr][j2];
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mxl = fmax(mxl, __shfl_xor(mxl, off));
            if (lane == 0) redm[buf][wv] = mxl;
            __syncthreads();
            double mx = 0.0;   // (fmax drops a NaN, as in k_gs_decide: the sums below carry it into the decision)
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) mx = fmax(mx, redm[buf][w2]);
            const double alpha = bc[buf][16 + j];
            double wk[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) wk[i] = (i < j && i < nacc) ? bc[buf][i] : 0.0;
            const double inv_s = (mx > 0 && mx < __builtin_inf()) ? 1.0 / mx : 0.0;
            // ---- t = R^-1 w_top in the rows above k; the sums
            double tt[RPT];
            double g1[4] = {0.0, 0.0, 0.0, 0.0};   // ss, cs, csi, sq
            double dl[16];
#pragma unroll
            for (int i = 0; i < 16; i++) dl[i] = 0.0;
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
                const int r = row[rr];
                const double y = Y[rr][j];
                tt[rr] = 0.0;
                if (r < k) {
                    double t = (r < k0) ? y : 0.0;
#pragma unroll
                    for (int i = 0; i < NB; i++)
                        if (i < j) t = __builtin_fma(X[rr][i], wk[i], t);   // (columns not accepted yet: X = 0, wk = 0)
                    tt[rr] = t;
                    g1[2] += fabs(t);
                    if (r >= k0) g1[1] += fabs(y);
                } else if (r < m) {
                    const double x = y * inv_s;
                    g1[0] = __builtin_fma(x, x, g1[0]);
                    if (r > k) {
                        g1[3] = __builtin_fma(y, y, g1[3]);
#pragma unroll
                        for (int j2 = j + 1; j2 < NB; j2++) dl[j2] = __builtin_fma(y, Y[rr][j2], dl[j2]);
                    }
                }
            }
            {
                const double t1 = wave_fold<4>(g1, lane);
                if (lane < 4) red1[buf][lane][wv] = t1;
            }
            const double t2 = wave_fold<16>(dl, lane);
            if (lane < 16) red2[buf][lane][wv] = t2;
            __syncthreads();
            double s1 = red1[buf][lane & 3][(lane >> 2) & 7];
            s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 8); s1 += __shfl_xor(s1, 16);
            double s2 = red2[buf][lane & 15][(lane >> 4) & 3] + red2[buf][lane & 15][((lane >> 4) & 3) + 4];
            s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
            const double ss = readlane_f64(s1, 0), csB = readlane_f64(s1, 1), csiB = readlane_f64(s1, 2), sq = readlane_f64(s1, 3);
            // ---- the decision (k_gs_decide)
            double nrm = (mx > 0) ? mx * sqrt(ss) : 0.0;
            if (mx != mx || mx == __builtin_inf()) nrm = mx;   // Inf in the candidate: propagate
            const double beta = alpha >= 0 ? -nrm : nrm;
            const double cs = readlane_f64(a0lane, j) + csB + fabs(beta);                                   // |r'|_1
            const double csi = beta != 0 ? (csiB / fabs(beta) + fabs(1.0 / beta)) : __builtin_inf();        // |r'^-1|_1
            bool accept;
            if (k == 0) accept = true;   // simplex.go:624-629
            else {
                double cond;
                if (beta == 0 || !(nRinv < __builtin_inf()) || !(csi < __builtin_inf())) cond = __builtin_inf();
                else cond = fmax(nR, cs) * fmax(nRinv, csi);
                accept = !(cond > 1e12);   // :630 (a NaN passes, as in the reference)
            }
            if (accept) {
                const double ninvb = beta != 0 ? -1.0 / beta : __builtin_inf();
                const double dinv = beta != 0 ? 1.0 / beta : __builtin_inf();
                const double vk = alpha - beta;
                const double vv = __builtin_fma(vk, vk, sq);
                const bool refl = vv > 0;
                const double f = refl ? 2.0 / vv : 0.0;
                double g[NB];
#pragma unroll
                for (int j2 = 0; j2 < NB; j2++) g[j2] = 0.0;
#pragma unroll
                for (int j2 = j + 1; j2 < NB; j2++) g[j2] = __builtin_fma(vk, bc[buf][16 + j2], readlane_f64(s2, j2)) * f;
#pragma unroll
                for (int rr = 0; rr < RPT; rr++) {
                    const int r = row[rr];
                    if (r >= m) continue;
                    double xn = 0.0, v = 0.0;
                    if (r < k) { xn = tt[rr] * ninvb; Rinv[(size_t)r * ldq + k] = xn; }
                    else if (r == k) { xn = dinv; Rinv[(size_t)r * ldq + k] = xn; v = vk; }
                    else v = Y[rr][j];
#pragma unroll
                    for (int i = 0; i < NB; i++)
                        if (i <= j && i == nacc) X[rr][i] = xn;
                    V[(size_t)nacc * ldq + r] = v;
                    if (refl && r >= k) {
#pragma unroll
                        for (int j2 = j + 1; j2 < NB; j2++) Y[rr][j2] = __builtin_fma(-v, g[j2], Y[rr][j2]);
                    }
                }
                if (tid == 0) { idxs[k] = cand; blk->vv[nacc] = vv; }
                nR = k == 0 ? cs : fmax(nR, cs);
                nRinv = k == 0 ? csi : fmax(nRinv, csi);
                k++; nacc++;
                if (k >= m - 1) { done = 1; stop_col = cand - 1; }   // the last column: the square step
            }
            scanned++;
        }
    }
    if (tid == 0) {
        blk->nacc = nacc; blk->k0 = k0;
        st->k = k; st->nR = nR; st->nRinv = nRinv; st->scanned = st->scanned + scanned; st->accept = 0;
        if (done) { st->done = 1; st->stop_col = stop_col; }
    }
}

// Q^T <- H_{nacc-1} ... H_0 Q^T, H_i = I - (2 / v_i^T v_i) v_i v_i^T: a strip of TCOLS columns per workgroup, thread (g, c) holds the
// rows g, g + RG, ... of column c (consecutive groups read consecutive entries of a reflector out of LDS)
template <int TCOLS>
__global__ __launch_bounds__(kGbAppT) void k_gsb_apply(double *__restrict__ QT, int ldq, int m, const double *__restrict__ V, const GsBlock *__restrict__ blk) {
    constexpr int T = kGbAppT, RG = T / TCOLS, RT = 32, MP = RG * RT;
    extern __shared__ __attribute__((aligned(16))) double gsb_lds[];   // sv[nacc][MP], part[2][16][TCOLS]
    const int nacc = blk->nacc;
    if (nacc <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = tid % TCOLS, g = tid / TCOLS;
    const int col = (int)blockIdx.x * TCOLS + c;
    double *sv = gsb_lds;
    double *part = gsb_lds + (size_t)nacc * MP;
    for (int idx = tid; idx < nacc * MP; idx += T) {
        const int i = idx / MP, r = idx % MP;
        sv[idx] = r < m ? V[(size_t)i * ldq + r] : 0.0;
    }
    double q[RT];
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int r = t * RG + g;
        q[t] = (r < m && col < m) ? QT[(size_t)r * ldq + col] : 0.0;
    }
    __syncthreads();
    for (int i = 0; i < nacc; i++) {
        const double vv = blk->vv[i];
        if (!(vv > 0)) continue;   // zero reflector: H = I (uniform)
        const double f = 2.0 / vv;
        const double *v = sv + (size_t)i * MP + g;
        double y0 = 0.0, y1 = 0.0;
#pragma unroll
        for (int t = 0; t < RT; t += 2) {
            y0 = __builtin_fma(v[t * RG], q[t], y0);
            y1 = __builtin_fma(v[(t + 1) * RG], q[t + 1], y1);
        }
        double y = y0 + y1;
#pragma unroll
        for (int b = TCOLS; b < 64; b <<= 1) y += __shfl_xor(y, b);
        double *pp = part + (size_t)(i & 1) * 16 * TCOLS;
        if (lane < TCOLS) pp[wv * TCOLS + lane] = y;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < 16; w2++) tot += pp[w2 * TCOLS + c];
        const double yc = tot * f;
#pragma unroll
        for (int t = 0; t < RT; t++) q[t] = __builtin_fma(-v[t * RG], yc, q[t]);
    }
#pragma unroll
    for (int t = 0; t < RT; t++) {
        const int r = t * RG + g;
        if (r < m && col < m) QT[(size_t)r * ldq + col] = q[t];
    }
}

// ---- host side
// candidates per block for a basis of m rows (the decide workgroup keeps 2 x RPT x NB doubles per lane), 0: too large for this form
int gs_block_width(int m) { return m <= 1024 ? 16 : (m <= 2048 ? 8 : (m <= 4096 ? 4 : 0)); }
int gs_block_scratch_rows() { return 3 * 16 + 1; }   // W, T0, V (16 rows of ldq doubles each at most) + the block record

template <int RPT, int NB, int TCOLS>
static void launch_gs_block_t(const double *At, int ld, int col0, int ncand, double *QT, double *Rinv, int ldq, int m, double *scratch, int32_t *idxs, GsState *st, hipStream_t s) {
    double *Wb = scratch, *T0 = scratch + (size_t)16 * ldq, *V = scratch + (size_t)32 * ldq;
    GsBlock *blk = reinterpret_cast<GsBlock *>(scratch + (size_t)48 * ldq);
    const int rows_per_wg = 8;
    hipLaunchKernelGGL((k_gsb_w<NB>), dim3((m + rows_per_wg - 1) / rows_per_wg), dim3(256), 0, s, At, ld, col0, ncand, QT, ldq, m, Wb, st);
    hipLaunchKernelGGL((k_gsb_t<NB>), dim3((m + rows_per_wg - 1) / rows_per_wg), dim3(256), 0, s, Rinv, ldq, ncand, Wb, T0, st);
    hipLaunchKernelGGL((k_gsb_decide<RPT, NB>), dim3(1), dim3(kGbDecT), 0, s, Rinv, ldq, m, Wb, T0, V, blk, col0, ncand, idxs, st);
    constexpr int MP = (kGbAppT / TCOLS) * 32;
    const int lds = (NB * MP + 2 * 16 * TCOLS) * (int)sizeof(double);
    lds_attr_once(reinterpret_cast<const void *>(&k_gsb_apply<TCOLS>), lds);
    hipLaunchKernelGGL((k_gsb_apply<TCOLS>), dim3((m + TCOLS - 1) / TCOLS), dim3(kGbAppT), lds, s, QT, ldq, m, V, blk);
}

// one block of candidates (columns col0, col0 - 1, ..., ncand <= gs_block_width(m) of them): 4 launches
void launch_gs_block(const double *At, int ld, int col0, int ncand, double *QT, double *Rinv, int ldq, int m, double *scratch, int32_t *idxs, GsState *st, hipStream_t s) {
    if (m <= 1024) launch_gs_block_t<2, 16, 32>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
    else if (m <= 2048) launch_gs_block_t<4, 8, 16>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
    else launch_gs_block_t<8, 4, 8>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
}

}  // namespace gomilp
