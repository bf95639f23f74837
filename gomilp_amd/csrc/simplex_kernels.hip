// gfx950 (CDNA4, wave64) kernels of the revised-simplex inner loop that replaces the per-pivot
// work of gonum's lp.Simplex (reference: vendor/gonum.org/v1/gonum/optimize/convex/lp/simplex.go:233-293).
//
// Data layout in HBM (all fp64, row-major, rows padded to `ld` doubles with zeros):
//   At    (ncols x ld)  row j = column j of the standard-form A  -> pricing and FTRAN read contiguous rows
//   Binv  (m x ld)      explicit basis inverse, two copies (ping-pong target of the rank-1 update)
// Every hot kernel is a streaming kernel: one 64-lane wave per matrix row, 16-byte loads per lane
// (1 KiB per wave instruction), the shared vector staged once per workgroup in LDS, and the
// arg-reductions (entering column, leaving row) fused into the epilogue as first-index argmins.
// Kernels communicate only across kernel boundaries; nothing depends on dispatch order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

// ------------------------------------------------------------------------------------------------
// K1  pricing:  r[pos] = cost[j] - At[j,:].y   (simplex.go:242-243), fused first-index argmin (:247)
//     algorithmic traffic: m*(n-m)*8 bytes read (A_N once)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_price(LPArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    DevState *st = a.st;
    if (st->done) return;
    if (st->max_pivots > 0 && st->pivots >= st->max_pivots) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_MAX_PIVOTS; }
        return;
    }
    const int ld2 = a.ld >> 1;
    stage_vec(svec, a.y, ld2);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    unsigned long long bk = ~0ull;
    unsigned int bi = 0xFFFFFFFFu;
    for (int pos = wave; pos < a.nn; pos += nwaves) {
        const int j = a.nonbasic[pos];
        const double dot = wave_dot_row(a.At + (size_t)j * a.ld, svec, ld2, lane);
        const double r = a.cost[j] - dot;
        if (lane == 0) a.rvec[pos] = r;
        amin_take(bk, bi, ordkey(r), (unsigned int)pos);
    }
    block_argmin(bk, bi, sk, si);
    if (threadIdx.x == 0) { a.pk_price[blockIdx.x] = bk; a.pi_price[blockIdx.x] = bi; }
}

// ------------------------------------------------------------------------------------------------
// K2  FTRAN + ratio test:  d' = Binv a_q ; move_i = x_B[i]/|d_i| for d_i = -d'_i < 0 (simplex.go:306-342),
//     fused first-index argmin of move (:268).   traffic: m*m*8 bytes read (B^-1 once)
//     forced_pos >= 0: entering position given (Bland / setup); forced_var >= 0: entering variable id given.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_ftran(LPArgs a, int nparts_price, int forced_pos, int forced_var) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    DevState *st = a.st;
    if (st->done) return;
    int q = forced_pos;
    int var;
    if (forced_var >= 0) {
        var = forced_var;
    } else {
        if (q < 0) {
            q = (int)reduce_partials(a.pk_price, a.pi_price, nparts_price, sk, si, nullptr);
            const double rq = a.rvec[q];
            if (rq >= -a.tol) {  // simplex.go:248 — optimal
                if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_OPTIMAL; st->q = q; st->rq = rq; }
                return;
            }
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->q = q; st->rq = rq; }
        } else if (blockIdx.x == 0 && threadIdx.x == 0) {
            st->q = q; st->rq = a.rvec[q];
        }
        var = a.nonbasic[q];
    }
    const int ld2 = a.ld >> 1;
    stage_vec(svec, a.At + (size_t)var * a.ld, ld2);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    unsigned long long bk = ~0ull;
    unsigned int bi = 0xFFFFFFFFu;
    for (int i = wave; i < a.m; i += nwaves) {
        const double dp = wave_dot_row(a.binv_cur + (size_t)i * a.ld, svec, ld2, lane);
        double d = -dp;                       // simplex.go:319
        if (fabs(d) < 1e-13) d = 0;           // dRoundTol, :321-325
        const double mv = (d >= 0) ? __builtin_inf() : a.xb[i] / fabs(d);  // :334-340
        if (lane == 0) { a.dvec[i] = dp; a.move[i] = mv; }
        amin_take(bk, bi, ordkey(mv), (unsigned int)i);
    }
    block_argmin(bk, bi, sk, si);
    if (threadIdx.x == 0) { a.pk_ratio[blockIdx.x] = bk; a.pi_ratio[blockIdx.x] = bi; }
}

// ------------------------------------------------------------------------------------------------
// K3  basis change: rank-1 update of B^-1 (ping-pong), x_B, y, index swap (simplex.go:280-292 without
//     the three fresh LU factorizations).   traffic: m*m*8 read + m*m*8 written
//     forced_p >= 0: leaving position given.  no_swap: setup pivot (indices managed by the host).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_update(LPArgs a, int nparts_ratio, int forced_p, int no_swap, int bland) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    DevState *st = a.st;
    if (st->done) return;
    int p = forced_p;
    if (p < 0) {
        p = (int)reduce_partials(a.pk_ratio, a.pi_ratio, nparts_ratio, sk, si, nullptr);
        const double mv = a.move[p];
        if (mv == __builtin_inf()) {  // no d_i < 0: unbounded (simplex.go:328-330)
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_UNBOUNDED; st->p = p; st->mv = mv; }
            return;
        }
        if (mv <= 0) {  // degenerate step -> Bland rule (simplex.go:269)
            if (blockIdx.x == 0 && threadIdx.x == 0) { st->done = 1; st->status = ST_NEED_BLAND; st->p = p; st->mv = mv; }
            return;
        }
    }
    const double dpv = a.dvec[p];
    const int ld2 = a.ld >> 1;
    stage_vec(svec, a.binv_cur + (size_t)p * a.ld, ld2);  // old row p
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int i = wave; i < a.m; i += nwaves) {
        const double2 *src = reinterpret_cast<const double2 *>(a.binv_cur + (size_t)i * a.ld);
        double2 *dst = reinterpret_cast<double2 *>(a.binv_next + (size_t)i * a.ld);
        if (i == p) {
            for (int c = lane; c < ld2; c += 64) {
                double2 v = svec[c];
                v.x = v.x / dpv; v.y = v.y / dpv;
                dst[c] = v;
            }
        } else {
            const double f = a.dvec[i] / dpv;
            for (int c = lane; c < ld2; c += 64) {
                double2 v = src[c];
                const double2 rp = svec[c];
                v.x = v.x - f * rp.x; v.y = v.y - f * rp.y;
                dst[c] = v;
            }
        }
    }
    if (blockIdx.x == 0) {
        // O(m) vector updates by workgroup 0 (nobody else touches xb / y in this kernel)
        const double theta = a.xb[p] / dpv;
        const double rq = no_swap ? 0.0 : st->rq;
        const double alpha = rq / dpv;
        __syncthreads();
        for (int i = threadIdx.x; i < a.m; i += kBlock) a.xb[i] = (i == p) ? theta : a.xb[i] - theta * a.dvec[i];
        const double *rowp = reinterpret_cast<const double *>(svec);
        for (int j = threadIdx.x; j < a.ld; j += kBlock) a.y[j] = a.y[j] + alpha * rowp[j];
        if (threadIdx.x == 0) {
            const int q = st->q;
            st->p = p; st->dp = dpv; st->mv = a.move[p];
            if (!no_swap) {
                const int ent = a.nonbasic[q], lea = a.basic[p];
                a.basic[p] = ent; a.nonbasic[q] = lea;  // simplex.go:280
                if (a.trace && st->trace_len < a.trace_cap) {
                    DevPivot &t = a.trace[st->trace_len];
                    t.phase = a.phase; t.bland = bland; t.min_idx = q; t.replace = p; t.entering = ent; t.leaving = lea;
                }
                st->trace_len += 1;
                st->pivots += 1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// set-up / refresh kernels (outside the per-pivot path)
// ------------------------------------------------------------------------------------------------

// At[j*ld + i] = A[i*lda + j]   (32x32 LDS tile transpose; block = 32x8)
__global__ void k_transpose_in(const double *__restrict__ A, int64_t lda, int m, int n, double *__restrict__ At, int ld) {
    __shared__ double tile[32][33];
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + threadIdx.x;
        if (i < m && j < n) tile[r][threadIdx.x] = A[(size_t)i * lda + j];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int j = j0 + r, i = i0 + threadIdx.x;
        if (i < m && j < n) At[(size_t)j * ld + i] = tile[threadIdx.x][r];
    }
}

// per column j of A (row of At): nnz, row of the last non-zero, all non-zeros equal to 1 ; and mark non-empty rows
// range (optional): [0] = bits of max |a_ij|, [1] = bits of the smallest non-zero |a_ij| (positive doubles order like their bit patterns)
__global__ __launch_bounds__(kBlock) void k_col_stats(const double *__restrict__ At, int ld, int m, int n, int32_t *nnz,
                                                      int32_t *lastrow, int32_t *allone, int32_t *rowflag, unsigned long long *range) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int j = wave; j < n; j += nwaves) {
        const double *row = At + (size_t)j * ld;
        int cnt = 0, last = -1, one = 1;
        unsigned long long hi = 0ull, lo = ~0ull;
        for (int i = lane; i < m; i += 64) {
            const double v = row[i];
            if (v != 0) {
                cnt++; last = i; if (v != 1.0) one = 0; rowflag[i] = 1;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(fabs(v));
                hi = bits > hi ? bits : hi; lo = bits < lo ? bits : lo;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            cnt += __shfl_xor(cnt, o, 64);
            last = max(last, __shfl_xor(last, o, 64));
            one &= __shfl_xor(one, o, 64);
            const unsigned long long h2 = __shfl_xor(hi, o, 64), l2 = __shfl_xor(lo, o, 64);
            hi = h2 > hi ? h2 : hi; lo = l2 < lo ? l2 : lo;
        }
        if (lane == 0) {
            nnz[j] = cnt; lastrow[j] = last; allone[j] = one;
            if (range && cnt) { atomicMax(range, hi); atomicMin(range + 1, lo); }
        }
    }
}

// Child relaxation of a B&B node assembled in HBM from the resident root (subproblem.go:81-139, :245-255):
//   child A = [[A0, 0], [G#, I_K]],  G# row k = sign_k * e_{var_k}.  In the transposed layout every root column keeps
//   its m0 entries and gains K entries; K unit columns are appended; the last row is the (zeroed) artificial slot.
__global__ __launch_bounds__(kBlock) void k_child_assemble(const double *__restrict__ At0, int ld0, int m0, int n0,
                                                           double *__restrict__ At1, int ld1, int K,
                                                           const int32_t *__restrict__ var, const double *__restrict__ sign) {
    const int j = blockIdx.x;  // child column
    double *dst = At1 + (size_t)j * ld1;
    if (j < n0) {
        const double *src = At0 + (size_t)j * ld0;
        for (int i = threadIdx.x; i < ld1; i += kBlock) {
            double v = 0.0;
            if (i < m0) v = src[i];
            else if (i < m0 + K && var[i - m0] == j) v = sign[i - m0];
            dst[i] = v;
        }
    } else {
        const int unit = (j < n0 + K) ? m0 + (j - n0) : -1;
        for (int i = threadIdx.x; i < ld1; i += kBlock) dst[i] = (i == unit) ? 1.0 : 0.0;
    }
}

// Binv = permutation: Binv[pos, rho[pos]] = 1 (buffer pre-zeroed)
__global__ void k_set_binv_perm(double *binv, int ld, int m, const int32_t *rho) {
    const int pos = blockIdx.x * blockDim.x + threadIdx.x;
    if (pos < m) binv[(size_t)pos * ld + rho[pos]] = 1.0;
}

// out[i] = M[i,:].vec   (x_B = Binv b refresh)
__global__ __launch_bounds__(kBlock) void k_matvec_rows(const double *__restrict__ M, int ld, int m,
                                                        const double *__restrict__ vec, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double2 svec[];
    const int ld2 = ld >> 1;
    stage_vec(svec, vec, ld2);
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    for (int i = wave; i < m; i += nwaves) {
        const double d = wave_dot_row(M + (size_t)i * ld, svec, ld2, lane);
        if (lane == 0) out[i] = d;
    }
}

// y[j] = sum_i cost[basic[i]] * Binv[i,j]   (y = B^-T c_B, simplex.go:236) — column-parallel, row-chunked;
// partial sums in chunk-major scratch, then reduced in fixed order (deterministic).
__global__ __launch_bounds__(kBlock) void k_y_partial(const double *__restrict__ binv, int ld, int m,
                                                      const double *__restrict__ cost, const int32_t *__restrict__ basic,
                                                      double *__restrict__ scratch, int rows_per_chunk) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int chunk = blockIdx.y;
    const int i0 = chunk * rows_per_chunk, i1 = min(m, i0 + rows_per_chunk);
    if (j >= ld) return;
    double acc = 0;
    for (int i = i0; i < i1; i++) {
        const double cb = cost[basic[i]];
        if (cb != 0) acc += cb * binv[(size_t)i * ld + j];
    }
    scratch[(size_t)chunk * ld + j] = acc;
}
__global__ void k_y_reduce(const double *__restrict__ scratch, int ld, int nchunks, double *__restrict__ y) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    double acc = 0;
    for (int c = 0; c < nchunks; c++) acc += scratch[(size_t)c * ld + j];
    y[j] = acc;
}

// ------------------------------------------------------------------------------------------------
// Final basis solve: LU with partial pivoting in gonum's operation order, so that x_B = ab^-1 b is
// bit-identical to the reference's last `xbVec.SolveVec(ab, bVec)` (simplex.go:288-292 ->
// lapack/gonum/dgetf2.go:30-69 / dgetrf.go:29-70):
//   pivot = first max |a_ik| over the remaining rows in LOGICAL row order (Idamax, level1double.go:121-165)
//   l_ik  = a_ik * (1/a_kk)                      (dgetf2.go:56: Dscal by the reciprocal)
//   a_ij  = (-l_ik) * a_kj + a_ij                (Dger/Dgemm: rounded multiply THEN rounded add, no FMA)
// Rows are never moved: lpos[R] is the logical position of physical row R (row interchanges of
// dlaswp.go become index updates), rowstep[R] >= 0 marks rows already used as pivot rows.
// One launch per column; the argmax for column k+1 is fused into step k.
// ------------------------------------------------------------------------------------------------

// W[i*ldw + pos] = At[basic[pos]*ld + i]   (ab of simplex.go:144: column pos = column basic[pos] of A)
__global__ void k_gather_w(const double *__restrict__ At, int ld, int m, const int32_t *__restrict__ basic,
                           double *__restrict__ W, int ldw) {
    __shared__ double tile[32][33];
    const int p0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int pos = p0 + r, i = i0 + threadIdx.x;
        if (pos < m && i < m) tile[r][threadIdx.x] = At[(size_t)basic[pos] * ld + i];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, pos = p0 + threadIdx.x;
        if (pos < m && i < m) W[(size_t)i * ldw + pos] = tile[threadIdx.x][r];
    }
}

__device__ __forceinline__ void lu_take(unsigned long long &k, unsigned int &l, unsigned int &r, unsigned long long k2,
                                        unsigned int l2, unsigned int r2) {
    if (k2 < k || (k2 == k && l2 < l)) { k = k2; l = l2; r = r2; }
}

__device__ __forceinline__ void lu_block_reduce(unsigned long long &k, unsigned int &l, unsigned int &r,
                                                unsigned long long *sk, unsigned int *sl, unsigned int *sr) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long k2 = __shfl_xor(k, o, 64);
        unsigned int l2 = __shfl_xor(l, o, 64), r2 = __shfl_xor(r, o, 64);
        lu_take(k, l, r, k2, l2, r2);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sk[w] = k; sl[w] = l; sr[w] = r; }
    __syncthreads();
    k = sk[0]; l = sl[0]; r = sr[0];
#pragma unroll
    for (int t = 1; t < kWavesPerBlock; t++) lu_take(k, l, r, sk[t], sl[t], sr[t]);
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void k_lu_init(LUArgs a) {
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int sl[kWavesPerBlock], sr[kWavesPerBlock];
    unsigned long long bk = ~0ull;
    unsigned int bl = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
    for (int R = blockIdx.x * kBlock + threadIdx.x; R < a.m; R += gridDim.x * kBlock) {
        a.lpos[R] = R;
        a.rowstep[R] = -1;
        lu_take(bk, bl, br, ordkey(-fabs(a.W[(size_t)R * a.ldw])), (unsigned int)R, (unsigned int)R);
    }
    lu_block_reduce(bk, bl, br, sk, sl, sr);
    if (threadIdx.x == 0) { a.pk[0][blockIdx.x] = bk; a.pl[0][blockIdx.x] = bl; a.pr[0][blockIdx.x] = br; }
}

__global__ __launch_bounds__(kBlock) void k_lu_step(LUArgs a, int k, int nparts) {
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int sl[kWavesPerBlock], sr[kWavesPerBlock];
    const int par = k & 1;
    // pivot of column k from the partials written by the previous step
    unsigned long long bk = ~0ull;
    unsigned int bl = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
    for (int t = threadIdx.x; t < nparts; t += kBlock) lu_take(bk, bl, br, a.pk[par][t], a.pl[par][t], a.pr[par][t]);
    lu_block_reduce(bk, bl, br, sk, sl, sr);
    const int P = (int)br;    // physical pivot row
    const int jp = (int)bl;   // its logical position before the interchange
    const double piv = a.W[(size_t)P * a.ldw + k];
    const bool singular = (piv == 0);  // dgetf2.go:48-49: ok = false, no scaling, the rank-1 update is a no-op
    if (singular && blockIdx.x == 0 && threadIdx.x == 0) a.st->lu_singular = 1;
    const double rinv = 1.0 / piv;
    const double *__restrict__ prow = a.W + (size_t)P * a.ldw;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    unsigned long long nk = ~0ull;
    unsigned int nl = 0xFFFFFFFFu, nr = 0xFFFFFFFFu;
    for (int R = wave; R < a.m; R += nwaves) {
        if (a.rowstep[R] >= 0) continue;
        if (R == P) {
            if (lane == 0) { a.rowstep[R] = k; a.lpos[R] = k; }
            continue;
        }
        int lp = a.lpos[R];
        if (lp == k) { lp = jp; if (lane == 0) a.lpos[R] = jp; }  // the row that sat at logical k moves to jp (dlaswp)
        double *row = a.W + (size_t)R * a.ldw;
        if (!singular) {
            const double l = __dmul_rn(row[k], rinv);
            const double nlv = -l;
            for (int j = k + 1 + lane; j < a.m; j += 64) row[j] = __dadd_rn(__dmul_rn(nlv, prow[j]), row[j]);
            if (lane == 0) row[k] = l;
        }
        if (k + 1 < a.m) {
            // lane 0 wrote row[k+1] in its first iteration above (j = k+1+0); re-read through the same lane
            double v = 0;
            if (lane == 0) v = row[k + 1];
            v = __shfl(v, 0, 64);
            lu_take(nk, nl, nr, ordkey(-fabs(v)), (unsigned int)lp, (unsigned int)R);
        }
    }
    lu_block_reduce(nk, nl, nr, sk, sl, sr);
    if (threadIdx.x == 0) { a.pk[par ^ 1][blockIdx.x] = nk; a.pl[par ^ 1][blockIdx.x] = nl; a.pr[par ^ 1][blockIdx.x] = nr; }
}

// ------------------------------------------------------------------------------------------------
// host-callable launch wrappers (the engine is plain C++; only this file is device code)
// ------------------------------------------------------------------------------------------------

static inline int grid_for_rows(int rows) {
    int g = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > kMaxPartials) g = kMaxPartials;
    if (g < 1) g = 1;
    return g;
}

int launch_price(const LPArgs &a, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int g = grid_for_rows(a.nn);
    hipExtLaunchKernelGGL(k_price, dim3(g), dim3(kBlock), (size_t)a.ld * sizeof(double), s, e0, e1, 0, a);
    return g;
}
int launch_ftran(const LPArgs &a, int nparts_price, int forced_pos, int forced_var, hipStream_t s, hipEvent_t e0,
                 hipEvent_t e1) {
    const int g = grid_for_rows(a.m);
    hipExtLaunchKernelGGL(k_ftran, dim3(g), dim3(kBlock), (size_t)a.ld * sizeof(double), s, e0, e1, 0, a, nparts_price,
                          forced_pos, forced_var);
    return g;
}
void launch_update(const LPArgs &a, int nparts_ratio, int forced_p, int no_swap, int bland, hipStream_t s, hipEvent_t e0,
                   hipEvent_t e1) {
    const int g = grid_for_rows(a.m);
    hipExtLaunchKernelGGL(k_update, dim3(g), dim3(kBlock), (size_t)a.ld * sizeof(double), s, e0, e1, 0, a, nparts_ratio,
                          forced_p, no_swap, bland);
}
void launch_transpose_in(const double *A, int64_t lda, int m, int n, double *At, int ld, hipStream_t s) {
    dim3 grid((n + 31) / 32, (m + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(k_transpose_in, grid, block, 0, s, A, lda, m, n, At, ld);
}
void launch_col_stats(const double *At, int ld, int m, int n, int32_t *nnz, int32_t *lastrow, int32_t *allone,
                      int32_t *rowflag, unsigned long long *range, hipStream_t s) {
    hipLaunchKernelGGL(k_col_stats, dim3(grid_for_rows(n)), dim3(kBlock), 0, s, At, ld, m, n, nnz, lastrow, allone, rowflag, range);
}
void launch_child_assemble(const double *At0, int ld0, int m0, int n0, double *At1, int ld1, int K, const int32_t *var,
                           const double *sign, hipStream_t s) {
    hipLaunchKernelGGL(k_child_assemble, dim3(n0 + K + 1), dim3(kBlock), 0, s, At0, ld0, m0, n0, At1, ld1, K, var, sign);
}
void launch_set_binv_perm(double *binv, int ld, int m, const int32_t *rho, hipStream_t s) {
    hipLaunchKernelGGL(k_set_binv_perm, dim3((m + 255) / 256), dim3(256), 0, s, binv, ld, m, rho);
}
void launch_matvec_rows(const double *M, int ld, int m, const double *vec, double *out, hipStream_t s) {
    hipLaunchKernelGGL(k_matvec_rows, dim3(grid_for_rows(m)), dim3(kBlock), (size_t)ld * sizeof(double), s, M, ld, m, vec, out);
}
// scratch must hold y_chunks(m) * ld doubles
int y_chunks(int m) { int c = (m + 63) / 64; return c > 64 ? 64 : c; }
void launch_y_from_binv(const double *binv, int ld, int m, const double *cost, const int32_t *basic, double *scratch,
                        double *y, hipStream_t s) {
    const int nchunks = y_chunks(m);
    const int rpc = (m + nchunks - 1) / nchunks;
    dim3 grid((ld + kBlock - 1) / kBlock, nchunks);
    hipLaunchKernelGGL(k_y_partial, grid, dim3(kBlock), 0, s, binv, ld, m, cost, basic, scratch, rpc);
    hipLaunchKernelGGL(k_y_reduce, dim3((ld + 255) / 256), dim3(256), 0, s, scratch, ld, nchunks, y);
}
void launch_gather_w(const double *At, int ld, int m, const int32_t *basic, double *W, int ldw, hipStream_t s) {
    dim3 grid((m + 31) / 32, (m + 31) / 32), block(32, 8);
    hipLaunchKernelGGL(k_gather_w, grid, block, 0, s, At, ld, m, basic, W, ldw);
}
int lu_grid(int m) { return grid_for_rows(m); }
void launch_lu(const LUArgs &a, hipStream_t s) {
    const int g = lu_grid(a.m);
    hipLaunchKernelGGL(k_lu_init, dim3(g), dim3(kBlock), 0, s, a);
    for (int k = 0; k < a.m; k++) hipLaunchKernelGGL(k_lu_step, dim3(g), dim3(kBlock), 0, s, a, k, g);
}

}  // namespace gomilp
