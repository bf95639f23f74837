// findLinearlyIndependent (simplex.go:611-637) on the device, BLOCKED: NB candidate columns per four launches (round 5).
//
// general_kernels.hip spends five launches on every candidate, each of them a grid-wide dependency of the next (w needs all of
// Q^T, the decision all of w, the rank-1 update the decision).  Here the same decisions — kappa_1 of the triangular factor from
// R^-1, `!(cond > 1e12)`, simplex.go:630 — are taken for NB candidates (the columns col0, col0 - 1, ...: the scan runs from the
// last column to the first) behind ONE pass over Q^T:
//   k_gsb_mv      W[j] = Q^T a_j for the NB candidates of the block: tiles of 16 rows x 16 candidates on the matrix cores
//   k_gsb_mv      T0[j] = R^-1 W[j] over the k0 columns accepted BEFORE the block (the same kernel over R^-1's triangle)
//   k_gsb_decide  ONE workgroup walks the candidates in scan order with all NB vectors in registers: a candidate has been brought
//                 up to date with the reflectors accepted earlier IN the block (w_j' -= (2 / v^T v)(v^T w_j') v, applied to every
//                 later candidate the moment a column is accepted), its t = R^-1 w_top is T0 plus the block's own new columns of
//                 R^-1 (registers), the norms and the decision are those of k_gs_decide; two workgroup reductions per
//                 candidate (scale + the entries everybody needs | the sums)
//   k_gsb_apply   Q^T <- H_last ... H_first Q^T for the reflectors the block accepted: a workgroup holds a strip of TCOLS columns
//                 of Q^T — the rows from k0 on: a reflector is zero above its own position — in registers and applies the
//                 reflectors one after the other (y = v^T Q^T is a column sum inside the workgroup)
// The arithmetic is the engine's own (a threshold on a condition number, not values the reference defines bit by bit); what must
// agree, and is tested against the CPU restatement of the reference (tests/) and against the per-candidate form, is the list of
// accepted columns.  The square step (the last column) stays with general_kernels.hip / the host (engine.cpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

namespace {

constexpr int kGbDecT = 512;   // threads of the decide workgroup: one wave per SIMD (512 registers per lane) — the walk is a chain of ~1000
                               // instructions per candidate, most of them the same in every wave (folds, the decision, the slot moves)
constexpr int kGbAppT = 512;   // threads of an apply workgroup

// v[0..N) per lane -> every lane l holds the wave's total of v[l & (N - 1)]: a fixed tree of N - 1 + log2(64 / N) exchanges
// instead of 6 N (step s pairs the lanes that differ in bit s: one keeps the even entries of the pair, the other the odd ones)
template <int N>
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
            v[i] = keep + __shfl_xor(send, 1 << s);
        }
    }
    double x = v[0];
#pragma unroll
    for (int b = N; b < 64; b <<= 1) x += __shfl_xor(x, b);
    return x;
}

}  // namespace

// out[j][r] = sum_c M[r][c] X_j[c], j < nx <= 16 — the two products of a block: W = Q^T [a_0 .. a_nx) (tri = 0: all m rows and
// columns; X_j = column col0 - j of A = row col0 - j of At) and T0 = R^-1 W over the k0 columns accepted before the block
// (tri = 1: rows and columns < k0 = st->k, the triangle c >= r; entries left of the diagonal are never formed and still zero).
// On the matrix cores: a tile of 16 rows x 16 candidates IS the C/D operand of v_mfma_f64_16x16x4_f64 (lane l, register v <-> row
// (l >> 4) + 4 v, candidate l & 15); A = M[row l & 15][k], B = X_{l & 15}[k] with k = l >> 4 inside a group of four columns.  A lane
// loads FOUR consecutive columns of its row / its candidate at once (32 bytes) and the four MFMAs of an iteration take them in turn:
// the k index of MFMA t is then column c + 4 (l >> 4) + t for A and B alike, so 16 columns pass per iteration.  A workgroup = one
// row tile x 8 column ranges (one per wave); the 8 partial tiles meet in LDS in a fixed order.  (The first form — lane = row, the
// candidates' entries as scalar operands, 16 multiply-adds per column and lane — took 32 us per product on 16 CUs.)
typedef double gsb_d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_gsb_mv(const double *__restrict__ M, int ldq, int m, int tri, const double *__restrict__ xbase, long xstride, int nx, double *__restrict__ out,
                                                const GsState *st) {
    if (st->done) return;
    const int lim = tri ? st->k : m;   // rows and columns
    const int r0 = (int)blockIdx.x * 16;
    if (r0 >= lim) return;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, kq = lane >> 4;
    const int cbase = tri ? r0 : 0;   // (a multiple of 16)
    const int chunk = (((lim - cbase + 7) >> 3) + 15) & ~15;
    const int c_lo = cbase + w * chunk, c_hi = min(c_lo + chunk, lim);
    const bool rowok = r0 + i < lim, candok = i < nx;
    const double *mrow = M + (size_t)(rowok ? r0 + i : r0) * ldq;
    const double *xrow = xbase + (long)(candok ? i : 0) * xstride;
    gsb_d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int c = c_lo; c < c_hi; c += 16) {
        const int cc = c + 4 * kq;
        double a[4], b[4];
        if (cc + 4 <= c_hi) {
            const double2 a0 = *reinterpret_cast<const double2 *>(mrow + cc), a1 = *reinterpret_cast<const double2 *>(mrow + cc + 2);
            const double2 b0 = *reinterpret_cast<const double2 *>(xrow + cc), b1 = *reinterpret_cast<const double2 *>(xrow + cc + 2);
            a[0] = a0.x; a[1] = a0.y; a[2] = a1.x; a[3] = a1.y;
            b[0] = b0.x; b[1] = b0.y; b[2] = b1.x; b[3] = b1.y;
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                a[t] = cc + t < c_hi ? mrow[cc + t] : 0.0;
                b[t] = cc + t < c_hi ? xrow[cc + t] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            if (!rowok) a[t] = 0.0;
            if (!candok) b[t] = 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
        }
    }
    __shared__ double part[8][4][64];
#pragma unroll
    for (int v = 0; v < 4; v++) part[w][v][lane] = acc[v];
    __syncthreads();
    if (tid < 256) {
        const int v = tid >> 6;
        const double s = ((part[0][v][lane] + part[1][v][lane]) + (part[2][v][lane] + part[3][v][lane])) +
                         ((part[4][v][lane] + part[5][v][lane]) + (part[6][v][lane] + part[7][v][lane]));
        const int r = r0 + kq + 4 * v;
        if (r < lim && candok) out[(size_t)i * ldq + r] = s;
    }
}

// The candidates of a block, one after the other (see the head of the file).  Thread tid holds the rows tid + T rr.  Per row a
// thread keeps, for every candidate still to come, ONE value: T0[j][r] if r < k0 (the candidate's own entries there are needed for
// |r'|_1 only: their absolute sum A0[j] is taken once, in front of the loop; no reflector of the block touches these rows), the
// running w_j[r] otherwise; and X[i] = the new column of R^-1 of the block's i-th accepted column (position k0 + i) in its row.
// The candidate in turn always sits in slot 0 (the slots move up by one behind every candidate): ONE loop body for all of them — the
// unrolled form, NB bodies each run once per launch, ran at the speed of its instruction fetches (6.3 us per candidate).
template <int T, int RPT, int NB>
__global__ __launch_bounds__(T) void k_gsb_decide(double *__restrict__ Rinv, int ldq, int m, const double *__restrict__ Wb, const double *__restrict__ T0, double *__restrict__ V,
                                                       GsBlock *__restrict__ blk, int col0, int ncand, int32_t *__restrict__ idxs, GsState *st) {
    constexpr int NW = T / 64;
    static_assert(NB <= 16 && (NW == 8 || NW == 4), "the folds below: 16 values, 4 or 8 waves");
    __shared__ double redm[2][NW];        // scale
    __shared__ double red1[2][4][NW];     // ss, |w_top|, |t|, sum of squares below the diagonal
    __shared__ double red2[2][16][NW];    // A0 in front of the loop; the dot products of the new reflector with the later candidates
    __shared__ double bc[2][2 * 16];      // w at the block's own positions k0 + i | the entries in row k of the candidate and the later ones
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (st->done) {
        if (tid == 0) { blk->nacc = 0; blk->k0 = st->k; }
        return;
    }
    if (tid < 2 * 2 * 16) (&bc[0][0])[tid] = 0.0;   // (entries of positions the block has not accepted yet stay zero)
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
        double x = red2[1][lane & 15][(lane >> 4) & 3];
        if constexpr (NW == 8) x += red2[1][lane & 15][((lane >> 4) & 3) + 4];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        a0lane = x;   // lane l: A0[l & 15]
    }
#pragma unroll 1
    for (int jj = 0; jj < ncand; jj++) {
        if (done) break;
        const int cand = col0 - jj;
        const int buf = jj & 1;
        // ---- round A: scale = max |w[r]|, r >= k; the entries of this candidate (and of the later ones) everybody needs
        // (the row loops below are straight-line: a row's class — above k / row k / below — enters as a select, not as a branch;
        // with branches the body was ~3000 instructions, most of them mask bookkeeping and moves)
        double mxl = 0.0;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
            const int r = row[rr];
            const double ay = fabs(Y[rr][0]);
            mxl = fmax(mxl, (r >= k && r < m) ? ay : 0.0);
            if (r >= k0 && r < k) bc[buf][r - k0] = Y[rr][0];
            if (r == k) {
#pragma unroll
                for (int s = 0; s < NB; s++) bc[buf][16 + s] = Y[rr][s];
            }
        }
        mxl = -wave_min_f64(-mxl);   // (DPP: no LDS round trips on the chain)
        if (lane == 0) redm[buf][wv] = mxl;
        __syncthreads();
        double mx = 0.0;   // (fmax drops a NaN, as in k_gs_decide: the sums below carry it into the decision)
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) mx = fmax(mx, redm[buf][w2]);
        const double alpha = bc[buf][16];
        double wk[NB];
#pragma unroll
        for (int i = 0; i < NB; i++) wk[i] = bc[buf][i];   // (0 for i >= nacc)
        const double inv_s = (mx > 0 && mx < __builtin_inf()) ? 1.0 / mx : 0.0;
        // ---- t = R^-1 w_top in the rows above k (X = 0 in every other row: t = 0 there); the sums
        double tt[RPT];
        double g1[4] = {0.0, 0.0, 0.0, 0.0};   // ss, cs, csi, sq
        double dl[16];
#pragma unroll
        for (int i = 0; i < 16; i++) dl[i] = 0.0;
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
            const int r = row[rr];
            const double y = Y[rr][0];
            double t = (r < k0) ? y : 0.0;
#pragma unroll
            for (int i = 0; i < NB; i++) t = __builtin_fma(X[rr][i], wk[i], t);   // (columns not accepted yet: X = 0, wk = 0)
            tt[rr] = t;
            g1[2] += fabs(t);
            g1[1] += (r >= k0 && r < k) ? fabs(y) : 0.0;
            const double x = (r >= k && r < m) ? y * inv_s : 0.0;
            g1[0] = __builtin_fma(x, x, g1[0]);
            const double yb = (r > k && r < m) ? y : 0.0;
            g1[3] = __builtin_fma(yb, yb, g1[3]);
#pragma unroll
            for (int s = 1; s < NB; s++) dl[s] = __builtin_fma(yb, Y[rr][s], dl[s]);
        }
        {
            const double t1 = wave_fold<4>(g1, lane);
            if (lane < 4) red1[buf][lane][wv] = t1;
        }
        const double t2 = wave_fold<16>(dl, lane);
        if (lane < 16) red2[buf][lane][wv] = t2;
        __syncthreads();
        double s1 = red1[buf][lane & 3][(lane >> 2) & (NW - 1)];
        s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 8);
        if constexpr (NW == 8) s1 += __shfl_xor(s1, 16);
        double s2 = red2[buf][lane & 15][(lane >> 4) & 3];
        if constexpr (NW == 8) s2 += red2[buf][lane & 15][((lane >> 4) & 3) + 4];
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        const double ss = readlane_f64(s1, 0), csB = readlane_f64(s1, 1), csiB = readlane_f64(s1, 2), sq = readlane_f64(s1, 3);
        // ---- the decision (k_gs_decide)
        double nrm = (mx > 0) ? mx * sqrt(ss) : 0.0;
        if (mx != mx || mx == __builtin_inf()) nrm = mx;   // Inf in the candidate: propagate
        const double beta = alpha >= 0 ? -nrm : nrm;
        const double cs = __shfl(a0lane, jj) + csB + fabs(beta);                                        // |r'|_1
        const double dinv = beta != 0 ? 1.0 / beta : __builtin_inf();
        const double csi = beta != 0 ? (csiB + 1.0) * fabs(dinv) : __builtin_inf();                    // |r'^-1|_1 = (|t|_1 + 1) / |beta|
        bool accept;
        if (k == 0) accept = true;   // simplex.go:624-629
        else {
            double cond;
            if (beta == 0 || !(nRinv < __builtin_inf()) || !(csi < __builtin_inf())) cond = __builtin_inf();
            else cond = fmax(nR, cs) * fmax(nRinv, csi);
            accept = !(cond > 1e12);   // :630 (a NaN passes, as in the reference)
        }
        if (accept) {
            const double ninvb = beta != 0 ? -dinv : __builtin_inf();
            const double vk = alpha - beta;
            const double vv = __builtin_fma(vk, vk, sq);
            const bool refl = vv > 0;
            const double f = refl ? 2.0 / vv : 0.0;
            // lane l: g[l & 15] = (2 / v^T v) v^T w of the candidate in slot l & 15 (0 for slot 0 and when H = I)
            const double glane = ((lane & 15) >= 1 && (lane & 15) < NB && refl) ? __builtin_fma(vk, bc[buf][16 + (lane & 15)], s2) * f : 0.0;
            double g[NB], e[NB];
#pragma unroll
            for (int s = 0; s < NB; s++) {
                g[s] = readlane_f64(glane, s);
                e[s] = s == nacc ? 1.0 : 0.0;   // (uniform) the slot of R^-1's new column
            }
#pragma unroll
            for (int rr = 0; rr < RPT; rr++) {
                const int r = row[rr];
                const bool top = r < k, piv = r == k, bot = r > k && r < m;
                const double xn = top ? tt[rr] * ninvb : (piv ? dinv : 0.0);
                const double v = piv ? vk : (bot ? Y[rr][0] : 0.0);
                if (top || piv) Rinv[(size_t)r * ldq + k] = xn;
                if (r < m) V[(size_t)nacc * ldq + r] = v;
#pragma unroll
                for (int i = 0; i < NB; i++) X[rr][i] = __builtin_fma(xn, e[i], X[rr][i]);   // (slot nacc held 0; xn = 0 below row k)
#pragma unroll
                for (int s = 1; s < NB; s++) Y[rr][s] = __builtin_fma(-v, g[s], Y[rr][s]);   // (v = 0 above row k)
            }
            if (tid == 0) { idxs[k] = cand; blk->vv[nacc] = vv; }
            nR = k == 0 ? cs : fmax(nR, cs);
            nRinv = k == 0 ? csi : fmax(nRinv, csi);
            k++; nacc++;
            if (k >= m - 1) { done = 1; stop_col = cand - 1; }   // the last column: the square step
        }
        scanned++;
        // the next candidate moves into slot 0
#pragma unroll
        for (int rr = 0; rr < RPT; rr++) {
#pragma unroll
            for (int s = 0; s + 1 < NB; s++) Y[rr][s] = Y[rr][s + 1];
            Y[rr][NB - 1] = 0.0;
        }
    }
    if (tid == 0) {
        blk->nacc = nacc; blk->k0 = k0;
        st->k = k; st->nR = nR; st->nRinv = nRinv; st->scanned = st->scanned + scanned; st->accept = 0;
        if (done) { st->done = 1; st->stop_col = stop_col; }
    }
}

// Q^T <- H_{nacc-1} ... H_0 Q^T, H_i = I - (2 / v_i^T v_i) v_i v_i^T, rows k0 .. m - 1 (a reflector is zero above its position k0 + i,
// the rows above k0 are final): a strip of TCOLS columns per workgroup, thread (g, c) holds the rows k0 + g, k0 + g + RG, ... of
// column c (consecutive groups read consecutive entries of a reflector out of LDS) and a reflector's entries in its rows in registers.
// RTU = row slots in use, rounded up to a multiple of 8: the loops over them carry no test (a uniform test per slot — 64 scalar
// branches per reflector — cost as much as the arithmetic)
template <int TCOLS, int RTU>
__device__ __forceinline__ void gsb_apply_body(double *__restrict__ QT, int ldq, int m, const double *__restrict__ V, const GsBlock *__restrict__ blk, double *gsb_lds) {
    constexpr int T = kGbAppT, NW = T / 64, RG = T / TCOLS, MP = RG * 32, MU = RG * RTU;
    const int nacc = blk->nacc, k0 = blk->k0;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int c = tid % TCOLS, g = tid / TCOLS;
    const int col = (int)blockIdx.x * TCOLS + c;
    const int nrel = m - k0;                   // rows that take part (<= MU)
    double *sv = gsb_lds;
    double *part = gsb_lds + (size_t)nacc * MP;
    for (int idx = tid; idx < nacc * MU; idx += T) {
        const int i = idx / MU, rel = idx % MU;
        sv[(size_t)i * MP + rel] = rel < nrel ? V[(size_t)i * ldq + k0 + rel] : 0.0;
    }
    double q[RTU];
#pragma unroll
    for (int t = 0; t < RTU; t++) {
        const int rel = t * RG + g;
        q[t] = (rel < nrel && col < m) ? QT[(size_t)(k0 + rel) * ldq + col] : 0.0;
    }
    __syncthreads();
    for (int i = 0; i < nacc; i++) {
        const double vv = blk->vv[i];
        if (!(vv > 0)) continue;   // zero reflector: H = I (uniform)
        const double f = 2.0 / vv;
        const double *v = sv + (size_t)i * MP + g;
        double vr[RTU];
        double y0 = 0.0, y1 = 0.0;
#pragma unroll
        for (int t = 0; t < RTU; t += 2) {
            vr[t] = v[t * RG];
            vr[t + 1] = v[(t + 1) * RG];
            y0 = __builtin_fma(vr[t], q[t], y0);
            y1 = __builtin_fma(vr[t + 1], q[t + 1], y1);
        }
        double y = y0 + y1;
#pragma unroll
        for (int b = TCOLS; b < 64; b <<= 1) y += __shfl_xor(y, b);
        double *pp = part + (size_t)(i & 1) * NW * TCOLS;
        if (lane < TCOLS) pp[wv * TCOLS + lane] = y;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) tot += pp[w2 * TCOLS + c];
        const double yc = tot * f;
#pragma unroll
        for (int t = 0; t < RTU; t++) q[t] = __builtin_fma(-vr[t], yc, q[t]);
    }
#pragma unroll
    for (int t = 0; t < RTU; t++) {
        const int rel = t * RG + g;
        if (rel < nrel && col < m) QT[(size_t)(k0 + rel) * ldq + col] = q[t];
    }
}
template <int TCOLS>
__global__ __launch_bounds__(kGbAppT) void k_gsb_apply(double *__restrict__ QT, int ldq, int m, const double *__restrict__ V, const GsBlock *__restrict__ blk) {
    extern __shared__ __attribute__((aligned(16))) double gsb_lds[];   // sv[nacc][MP], part[2][NW][TCOLS]
    if (blk->nacc <= 0) return;
    constexpr int RG = kGbAppT / TCOLS;
    const int nt = (m - blk->k0 + RG - 1) / RG;   // (uniform) row slots in use, <= 32
    if (nt <= 8) gsb_apply_body<TCOLS, 8>(QT, ldq, m, V, blk, gsb_lds);
    else if (nt <= 16) gsb_apply_body<TCOLS, 16>(QT, ldq, m, V, blk, gsb_lds);
    else if (nt <= 24) gsb_apply_body<TCOLS, 24>(QT, ldq, m, V, blk, gsb_lds);
    else gsb_apply_body<TCOLS, 32>(QT, ldq, m, V, blk, gsb_lds);
}

// ---- host side
// candidates per block for a basis of m rows (the decide workgroup keeps 2 x RPT x NB doubles per lane), 0: too large for this form
int gs_block_width(int m) { return m <= 1024 ? 16 : (m <= 2048 ? 8 : (m <= 4096 ? 4 : 0)); }
int gs_block_scratch_rows() { return 3 * 16 + 1; }   // W, T0, V (16 rows of ldq doubles each at most) + the block record

template <int RPT, int NB, int TCOLS>
static void launch_gs_block_t(const double *At, int ld, int col0, int ncand, double *QT, double *Rinv, int ldq, int m, double *scratch, int32_t *idxs, GsState *st, hipStream_t s) {
    double *Wb = scratch, *T0 = scratch + (size_t)16 * ldq, *V = scratch + (size_t)32 * ldq;
    GsBlock *blk = reinterpret_cast<GsBlock *>(scratch + (size_t)48 * ldq);
    const int rtiles = (m + 15) / 16;
    hipLaunchKernelGGL(k_gsb_mv, dim3(rtiles), dim3(512), 0, s, QT, ldq, m, 0, At + (size_t)col0 * ld, -(long)ld, ncand, Wb, st);
    hipLaunchKernelGGL(k_gsb_mv, dim3(rtiles), dim3(512), 0, s, Rinv, ldq, m, 1, Wb, (long)ldq, ncand, T0, st);
    hipLaunchKernelGGL((k_gsb_decide<kGbDecT, (512 / kGbDecT) * RPT, NB>), dim3(1), dim3(kGbDecT), 0, s, Rinv, ldq, m, Wb, T0, V, blk, col0, ncand, idxs, st);
    constexpr int MP = (kGbAppT / TCOLS) * 32;
    static_assert(MP >= 512 * RPT, "the strip holds every row of the largest basis of its class");
    const int lds = (NB * MP + 2 * (kGbAppT / 64) * TCOLS) * (int)sizeof(double);
    lds_attr_once(reinterpret_cast<const void *>(&k_gsb_apply<TCOLS>), lds);
    hipLaunchKernelGGL((k_gsb_apply<TCOLS>), dim3((m + TCOLS - 1) / TCOLS), dim3(kGbAppT), lds, s, QT, ldq, m, V, blk);
}

// one block of candidates (columns col0, col0 - 1, ..., ncand <= gs_block_width(m) of them): 4 launches
void launch_gs_block(const double *At, int ld, int col0, int ncand, double *QT, double *Rinv, int ldq, int m, double *scratch, int32_t *idxs, GsState *st, hipStream_t s) {
    if (m <= 1024) launch_gs_block_t<2, 16, 16>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
    else if (m <= 2048) launch_gs_block_t<4, 8, 8>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
    else launch_gs_block_t<8, 4, 4>(At, ld, col0, ncand, QT, Rinv, ldq, m, scratch, idxs, st, s);
}

}  // namespace gomilp
