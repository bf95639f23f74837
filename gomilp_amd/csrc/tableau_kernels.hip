// Single-kernel pivot on the explicit tableau T = B^-1 A_N (gfx950) — the fast path when n - m < 2m.
//
// The reference prices with fresh duals every pivot (simplex.go:236-247).  Keeping T = B^-1 A_N (m x (n-m), one
// column per nonbasic POSITION, same positional order as nonBasicIdx) turns a whole pivot into ONE streaming pass:
//
//   prologue  reduce the ratio-test partials of the current pivot -> leaving row p (simplex.go:268);
//             stage row p of T in LDS; update the reduced costs r <- r - (r_q/d_p) * T[p,:] (r_q slot: -r_q/d_p);
//             first-index argmin of the new r -> next entering position (simplex.go:247) or OPTIMAL (:248)
//   pass      T[i,:] <- T[i,:] - (d_i/d_p) T[p,:]  (row p: / d_p; column q becomes the leaving variable's column),
//             written to the other copy of T; the lane that owns the next entering column hands d'_i = T'[i, q']
//             to the ratio test (simplex.go:306-342) — FTRAN and pricing cost no extra pass
//   epilogue  per-workgroup first-index argmin of the ratios, winner carries (d'_i, basic[i], x_B[i])
//
// HBM traffic per pivot: 16*m*(n-m) bytes (read + write T) — at n = 2m that is half of what the
// pricing + FTRAN + update formulation (8*[m(n-m) + 3m^2]) moves, in one launch instead of two or three.
// Launch t reads the "next pivot" slot (t & 1) of DevState and writes slot ((t+1) & 1); a launch only turns
// later launches into no-ops (stop_at), so nothing depends on when a workgroup of the same launch starts.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

struct ArgMinT {
    unsigned long long k;
    unsigned int i, u;
    double d, x;
};
__device__ __forceinline__ void amint_take(ArgMinT &a, const ArgMinT &b) {
    if (b.k < a.k || (b.k == a.k && b.i < a.i)) a = b;
}
__device__ __forceinline__ void block_argmint(ArgMinT &a, ArgMinT *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMinT b;
        b.k = __shfl_xor(a.k, o, 64); b.i = __shfl_xor(a.i, o, 64); b.u = __shfl_xor(a.u, o, 64);
        b.d = __shfl_xor(a.d, o, 64); b.x = __shfl_xor(a.x, o, 64);
        amint_take(a, b);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = a;
    __syncthreads();
    a = sm[0];
#pragma unroll
    for (int t = 1; t < kWavesPerBlock; t++) amint_take(a, sm[t]);
    __syncthreads();
}

// U = number of 1-KiB row chunks (64 lanes x 16 bytes) loaded back to back before the first use.
// flags: bit0 pending (a pivot is waiting to be applied), bit1 forced (skip the unbounded / degenerate tests:
// Bland step or set-up pivot chosen by the host), bit2 no_commit (set-up pivot: indices managed by the host),
// bit3 bland (trace only)
template <int U>
__global__ __launch_bounds__(kBlock) void k_tableau_pivot(TabArgs a, int flags, int nparts, long long t) {
    extern __shared__ __attribute__((aligned(16))) double2 srow[];  // old row p
    __shared__ ArgMinT sm[kWavesPerBlock];
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    DevState *st = a.st;
    const bool pending = flags & 1, forced = flags & 2, no_commit = flags & 4;
    const int par = (int)(t & 1);
    const int ld2 = a.ldt >> 1;
    const int nch = ld2 >> 6;  // 1-KiB chunks per row
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * kWavesPerBlock;
    // ---- round 1: state + partials
    const long long stop_at = st->stop_at;
    const int q_cur = st->nq[par];
    const double rq_cur = st->nrq[par];
    const int ent_cur = st->nent[par];
    ArgMinT best;
    best.k = ~0ull; best.i = 0xFFFFFFFFu; best.u = 0; best.d = 0; best.x = 0;
    if (pending) {
        for (int c = threadIdx.x; c < nparts; c += kBlock) {
            ArgMinT b;
            b.k = a.pk_ratio[c]; b.i = a.pi_ratio[c]; b.u = a.pb_ratio[c]; b.d = a.pd_ratio[c]; b.x = a.px_ratio[c];
            amint_take(best, b);
        }
    }
    if (t >= stop_at) return;
    int p = -1, lea = -1;
    double dpv = 1, theta = 0, mult = 0;
    if (pending) {
        // ---- round 2: leaving row of the current pivot
        block_argmint(best, sm);
        p = (int)best.i;
        const double mv = orddecode(best.k);
        if (!forced) {
            if (mv == __builtin_inf()) {  // simplex.go:328-330
                if (blockIdx.x == 0 && threadIdx.x == 0) { st->stop_at = t + 1; st->done = 1; st->status = ST_UNBOUNDED; st->p = p; st->mv = mv; }
                return;
            }
            if (mv <= 0) {  // simplex.go:269 -> Bland
                if (blockIdx.x == 0 && threadIdx.x == 0) { st->stop_at = t + 1; st->done = 1; st->status = ST_NEED_BLAND; st->p = p; st->mv = mv; }
                return;
            }
        }
        dpv = best.d; lea = (int)best.u; theta = best.x / dpv; mult = rq_cur / dpv;
    }
    // ---- round 3: row p -> LDS, new reduced costs, next entering position
    unsigned long long bk = ~0ull;
    unsigned int bi = 0xFFFFFFFFu;
    {
        const double2 *rp2 = reinterpret_cast<const double2 *>(a.T_cur + (size_t)(pending ? p : 0) * a.ldt);
        const double2 *ri2 = reinterpret_cast<const double2 *>(a.r_in);
        double2 *ro2 = reinterpret_cast<double2 *>(a.r_out);
        for (int c = threadIdx.x; c < ld2; c += kBlock) {
            double2 rr = ri2[c];
            if (pending) {
                const double2 w = rp2[c];
                srow[c] = w;
                rr.x = rr.x - mult * w.x;
                rr.y = rr.y - mult * w.y;
                if (2 * c == q_cur) rr.x = -mult;      // the leaving variable takes position q: r = -r_q/d_p
                if (2 * c + 1 == q_cur) rr.y = -mult;
                if (blockIdx.x == 0) ro2[c] = rr;
            }
            if (2 * c < a.nn) amin_take(bk, bi, ordkey(rr.x), (unsigned int)(2 * c));
            if (2 * c + 1 < a.nn) amin_take(bk, bi, ordkey(rr.y), (unsigned int)(2 * c + 1));
        }
    }
    block_argmin(bk, bi, sk, si);  // contains the barrier that publishes srow
    const int q_next = (int)bi;
    const double rq_next = orddecode(bk);
    const bool optimal = (rq_next >= -a.tol);  // simplex.go:248
    if (!pending && optimal) {
        if (blockIdx.x == 0 && threadIdx.x == 0) { st->stop_at = t + 1; st->done = 1; st->status = ST_OPTIMAL; st->q = q_next; st->rq = rq_next; }
        return;
    }
    // ---- streaming pass
    const int qn_chunk = (q_next >> 1) >> 6, qn_lane = (q_next >> 1) & 63, qn_half = q_next & 1;
    const int qc_chunk = (q_cur >> 1) >> 6, qc_lane = (q_cur >> 1) & 63, qc_half = q_cur & 1;
    ArgMinT mine;
    mine.k = ~0ull; mine.i = 0xFFFFFFFFu; mine.u = 0; mine.d = 0; mine.x = 0;
    for (int i = wave; i < a.m; i += nwaves) {
        const double2 *src = reinterpret_cast<const double2 *>(a.T_cur + (size_t)i * a.ldt);
        double2 *dst = reinterpret_cast<double2 *>(a.T_next + (size_t)i * a.ldt);
        const double di = pending ? a.dvec[i] : 0.0;
        const double f = di / dpv;
        const bool is_p = (i == p);
        double dn = 0;
        for (int c0 = 0; c0 < nch; c0 += U) {
            // scalars, not double2: a select between .x and .y of a vector becomes a dynamic extractelement,
            // which demotes the whole array to scratch
            double vx[U], vy[U];
#pragma unroll
            for (int k = 0; k < U; k++) {  // nch is a multiple of U (launch wrapper)
                const double2 ld = src[lane + 64 * (c0 + k)];
                vx[k] = ld.x; vy[k] = ld.y;
            }
#pragma unroll
            for (int k = 0; k < U; k++) {
                const int ch = c0 + k;
                if (pending) {
                    const double2 rp = srow[lane + 64 * ch];
                    if (is_p) { vx[k] = vx[k] / dpv; vy[k] = vy[k] / dpv; }
                    else { vx[k] = vx[k] - f * rp.x; vy[k] = vy[k] - f * rp.y; }
                    if (ch == qc_chunk && lane == qc_lane) {  // column q now belongs to the leaving variable (eta column)
                        const double e = is_p ? 1.0 / dpv : -f;
                        vx[k] = qc_half ? vx[k] : e;
                        vy[k] = qc_half ? e : vy[k];
                    }
                    double2 outv;
                    outv.x = vx[k]; outv.y = vy[k];
                    dst[lane + 64 * ch] = outv;
                }
                if (ch == qn_chunk && lane == qn_lane) dn = qn_half ? vy[k] : vx[k];
            }
        }
        dn = __shfl(dn, qn_lane, 64);
        double xbi = a.xb[i];
        if (pending) {
            xbi = is_p ? theta : xbi - theta * di;
            if (lane == 0) a.xb[i] = xbi;
        }
        if (!optimal) {
            double d = -dn;                  // simplex.go:319
            if (fabs(d) < 1e-13) d = 0;      // :321-325
            const double mv = (d >= 0) ? __builtin_inf() : xbi / fabs(d);  // :334-340
            if (lane == 0) { a.dvec[i] = dn; a.move[i] = mv; }
            ArgMinT c;
            c.k = ordkey(mv); c.i = (unsigned int)i; c.d = dn; c.x = xbi;
            c.u = (unsigned int)((pending && is_p && !no_commit) ? ent_cur : a.basic[i]);
            amint_take(mine, c);
        }
    }
    if (!optimal) {
        block_argmint(mine, sm);
        if (threadIdx.x == 0) {
            a.pk_ratio[blockIdx.x] = mine.k; a.pi_ratio[blockIdx.x] = mine.i; a.pb_ratio[blockIdx.x] = mine.u;
            a.pd_ratio[blockIdx.x] = mine.d; a.px_ratio[blockIdx.x] = mine.x;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (pending && !no_commit) {  // simplex.go:280
            a.basic[p] = ent_cur;
            a.nonbasic[q_cur] = lea;
            if (a.trace && st->trace_len < a.trace_cap) {
                DevPivot &tr = a.trace[st->trace_len];
                tr.phase = a.phase; tr.bland = (flags & 8) ? 1 : 0; tr.min_idx = q_cur; tr.replace = p; tr.entering = ent_cur; tr.leaving = lea;
            }
            st->trace_len += 1;
            st->pivots += 1;
            st->p = p; st->dp = dpv; st->theta = theta;
        }
        st->nq[par ^ 1] = q_next;
        st->nrq[par ^ 1] = rq_next;
        // variable id at the next entering position AFTER this launch's swap
        st->nent[par ^ 1] = (pending && !no_commit && q_next == q_cur) ? lea : a.nonbasic[q_next];
        st->q = q_next; st->rq = rq_next;
        if (optimal) { st->stop_at = t + 1; st->done = 1; st->status = ST_OPTIMAL; }
    }
}

// ---- set-up kernels ---------------------------------------------------------------------------

// T[pos, jp] = At[var(jp)][rho[pos]]  (B^-1 = permutation of the slack basis: row pos of T is row rho[pos] of A_N)
// (tab_idx: element (i, j) of T in either layout — kernels_common.h)

__global__ void k_tab_gather(const double *__restrict__ At, int ld, int m, int nn, const int32_t *__restrict__ nonbasic,
                             const int32_t *__restrict__ rho, double *__restrict__ T, int ldt, int tiled) {
    __shared__ double tile[32][33];
    const int j0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    // the source rows are permuted by rho, so read one At row segment per (jp) and scatter through LDS
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int jp = j0 + r, pos = p0 + threadIdx.x;
        tile[r][threadIdx.x] = (jp < nn && pos < m) ? At[(size_t)nonbasic[jp] * ld + rho[pos]] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int pos = p0 + r, jp = j0 + threadIdx.x;
        // the padding columns nn..ldt (and, tiled, the padding rows m..m4) are zeroed here
        const int mrows = tiled ? ((m + 3) & ~3) : m;
        if (jp < ldt && pos < mrows) T[tab_idx(pos, jp, ldt, tiled)] = pos < m ? tile[threadIdx.x][r] : 0.0;
    }
}

// T[pos, jp] = sum_i Binv[pos, i] * At[var(jp)][i]  — T = B^-1 A_N for a general (non-slack) starting basis: B^-1 comes from
// the host (engine_general.cpp), the product runs here.  64 x 64 output tile per workgroup, 4 x 4 per thread, K chunks of 16
// through LDS; both operands are contiguous along i, so the staging loads are coalesced.
__global__ __launch_bounds__(kBlock) void k_tab_gemm(const double *__restrict__ Binv, int ldb, const double *__restrict__ At, int ld, int m, int nn,
                                                    const int32_t *__restrict__ nonbasic, double *__restrict__ T, int ldt, int tiled) {
    __shared__ double sa[16][65], sb[16][65];
    const int p0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // thread -> outputs (p0 + ty*4 + a, j0 + tx*4 + b)
    double acc[4][4] = {};
    const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;   // staging: row lr (0..63), k offset lk (0,4,8,12)
    const int arow = p0 + lr, bcol = j0 + lr;
    const double *ap = arow < m ? Binv + (size_t)arow * ldb : nullptr;
    const double *bp = bcol < nn ? At + (size_t)nonbasic[bcol] * ld : nullptr;
    for (int k0 = 0; k0 < m; k0 += 16) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int kk = k0 + lk + t;
            sa[lk + t][lr] = (ap && kk < m) ? ap[kk] : 0.0;
            sb[lk + t][lr] = (bp && kk < m) ? bp[kk] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; kk++) {
            double av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; a++) av[a] = sa[kk][ty * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; b++) bv[b] = sb[kk][tx * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] += av[a] * bv[b];
        }
        __syncthreads();
    }
    const int mrows = tiled ? ((m + 3) & ~3) : m;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int pos = p0 + ty * 4 + a, jp = j0 + tx * 4 + b;
            if (pos < mrows && jp < ldt) T[tab_idx(pos, jp, ldt, tiled)] = (pos < m && jp < nn) ? acc[a][b] : 0.0;
        }
}

// T_out[:, jp] = T_in[:, src[jp]]  (Phase I -> Phase II: nonbasic list rebuilt in ascending variable order)
__global__ void k_tab_permute_cols(const double *__restrict__ Tin, int ld_in, double *__restrict__ Tout, int ld_out, int m,
                                   int nn_out, const int32_t *__restrict__ srcpos, int tiled) {
    const int i = blockIdx.y;   // tiled: the padding rows m..m4 are written too (zeros)
    const int jp = blockIdx.x * blockDim.x + threadIdx.x;
    if (jp < ld_out) Tout[tab_idx(i, jp, ld_out, tiled)] = (jp < nn_out && i < m) ? Tin[tab_idx(i, srcpos[jp], ld_in, tiled)] : 0.0;
}

// r[jp] = cost[nonbasic[jp]] - sum_i cost[basic[i]] * T[i, jp]   (row-chunked, fixed-order reduction)
__global__ __launch_bounds__(kBlock) void k_tab_r_partial(const double *__restrict__ T, int ldt, int m, int nn,
                                                          const double *__restrict__ cost, const int32_t *__restrict__ basic,
                                                          double *__restrict__ scratch, int rows_per_chunk, int tiled) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int chunk = blockIdx.y;
    const int i0 = chunk * rows_per_chunk, i1 = min(m, i0 + rows_per_chunk);
    if (j >= ldt) return;
    double acc = 0;
    for (int i = i0; i < i1; i++) {
        const double cb = cost[basic[i]];
        if (cb != 0) acc += cb * T[tab_idx(i, j, ldt, tiled)];
    }
    scratch[(size_t)chunk * ldt + j] = acc;
}
__global__ void k_tab_r_reduce(const double *__restrict__ scratch, int ldt, int nn, int nchunks, const double *__restrict__ cost,
                               const int32_t *__restrict__ nonbasic, double *__restrict__ r) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldt) return;
    double acc = 0;
    for (int c = 0; c < nchunks; c++) acc += scratch[(size_t)c * ldt + j];
    r[j] = (j < nn) ? cost[nonbasic[j]] - acc : 0.0;
}

// column jp of T -> dvec, ratio vector (computeMove for a Bland candidate, simplex.go:306-342)
__global__ void k_tab_column(const double *__restrict__ T, int ldt, int m, int jp, const double *__restrict__ xb,
                             double *__restrict__ dvec, double *__restrict__ move, int tiled) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double dn = T[tab_idx(i, jp, ldt, tiled)];
    double d = -dn;
    if (fabs(d) < 1e-13) d = 0;
    dvec[i] = dn;
    move[i] = (d >= 0) ? __builtin_inf() : xb[i] / fabs(d);
}

// ---- launch wrappers ---------------------------------------------------------------------------

static inline int grid_rows_t(int rows) {
    int g = (rows + kWavesPerBlock - 1) / kWavesPerBlock;
    if (g > kMaxPartials) g = kMaxPartials;
    return g < 1 ? 1 : g;
}

int launch_tableau_pivot(const TabArgs &a, int flags, int nparts, long long t, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const int g = grid_rows_t(a.m);
    const int nch = a.ldt / 128;  // ldt is a multiple of 512 doubles (tab_ld), so U = 4 always divides nch
    const size_t lds = (size_t)a.ldt * sizeof(double);
    if (nch % 16 == 0)
        hipExtLaunchKernelGGL((k_tableau_pivot<16>), dim3(g), dim3(kBlock), lds, s, e0, e1, 0, a, flags, nparts, t);
    else if (nch % 8 == 0)
        hipExtLaunchKernelGGL((k_tableau_pivot<8>), dim3(g), dim3(kBlock), lds, s, e0, e1, 0, a, flags, nparts, t);
    else
        hipExtLaunchKernelGGL((k_tableau_pivot<4>), dim3(g), dim3(kBlock), lds, s, e0, e1, 0, a, flags, nparts, t);
    return g;
}
void launch_tab_gather(const double *At, int ld, int m, int nn, const int32_t *nonbasic, const int32_t *rho, double *T, int ldt,
                       bool tiled, hipStream_t s) {
    dim3 grid((m + 3 + 31) / 32, (ldt + 31) / 32), block(32, 8);   // all ldt columns (+ pad rows): the kernel zero-fills the padding
    hipLaunchKernelGGL(k_tab_gather, grid, block, 0, s, At, ld, m, nn, nonbasic, rho, T, ldt, tiled ? 1 : 0);
}
void launch_tab_gemm(const double *Binv, int ldb, const double *At, int ld, int m, int nn, const int32_t *nonbasic, double *T, int ldt, bool tiled, hipStream_t s) {
    dim3 grid((((m + 3) & ~3) + 63) / 64, (ldt + 63) / 64);
    hipLaunchKernelGGL(k_tab_gemm, grid, dim3(kBlock), 0, s, Binv, ldb, At, ld, m, nn, nonbasic, T, ldt, tiled ? 1 : 0);
}
int tab_ld(int nn) { return ((nn + 511) / 512) * 512; }
void launch_tab_permute_cols(const double *Tin, int ld_in, double *Tout, int ld_out, int m, int nn_out, const int32_t *srcpos,
                             bool tiled, hipStream_t s) {
    dim3 grid((ld_out + 255) / 256, tiled ? ((m + 3) & ~3) : m);
    hipLaunchKernelGGL(k_tab_permute_cols, grid, dim3(256), 0, s, Tin, ld_in, Tout, ld_out, m, nn_out, srcpos, tiled ? 1 : 0);
}
int tab_r_chunks(int m) { int c = (m + 63) / 64; return c > 64 ? 64 : c; }
void launch_tab_r(const double *T, int ldt, int m, int nn, const double *cost, const int32_t *basic, const int32_t *nonbasic,
                  double *scratch, double *r, bool tiled, hipStream_t s) {
    const int nchunks = tab_r_chunks(m);
    const int rpc = (m + nchunks - 1) / nchunks;
    dim3 grid((ldt + kBlock - 1) / kBlock, nchunks);
    hipLaunchKernelGGL(k_tab_r_partial, grid, dim3(kBlock), 0, s, T, ldt, m, nn, cost, basic, scratch, rpc, tiled ? 1 : 0);
    hipLaunchKernelGGL(k_tab_r_reduce, dim3((ldt + 255) / 256), dim3(256), 0, s, scratch, ldt, nn, nchunks, cost, nonbasic, r);
}
// out[jp] = T[row][jp] and out[ldt + jp] = max_i |T[i][jp]| for every nonbasic position: what the exchange of a
// zero-level artificial (simplex.go:581-606) needs to rank ALL candidate columns in one pass instead of one
// column fetch + host round trip per candidate
__global__ void k_tab_row_colmax(const double *__restrict__ T, int ldt, int m, int nn, int row, double *__restrict__ out, int tiled) {
    const int jp = blockIdx.x * blockDim.x + threadIdx.x;
    if (jp >= nn) return;
    double mx = 0;
    for (int i = 0; i < m; i++) mx = fmax(mx, fabs(T[tab_idx(i, jp, ldt, tiled)]));
    out[jp] = T[tab_idx(row, jp, ldt, tiled)];
    out[ldt + jp] = mx;
}
void launch_tab_row_colmax(const double *T, int ldt, int m, int nn, int row, double *out, bool tiled, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_row_colmax, dim3((nn + 255) / 256), dim3(256), 0, s, T, ldt, m, nn, row, out, tiled ? 1 : 0);
}

// Reduced costs the way the reference forms them every pivot (simplex.go:242-243): data = an^T y by gonum's Dgemv(Trans)
// (blas/gonum/level2double.go:99-106: for i < m, if y_i != 0: data += y_i * an[i, :], a rounded multiply and a rounded add per
// element, ascending i), then r = (-1 * data) + cn (floats.SubTo).  One thread per nonbasic position; padding columns get 0.
__global__ __launch_bounds__(256) void k_exact_r(const double *__restrict__ At, int ld, int m, int nn, const int32_t *__restrict__ nonbasic,
                                                 const double *__restrict__ y, const double *__restrict__ cost, double *__restrict__ r, int ldt) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= ldt) return;
    if (j >= nn) { r[j] = 0.0; return; }
    const int var = nonbasic[j];
    const double *col = At + (size_t)var * ld;
    double data = 0.0;
    for (int i = 0; i < m; i++) {
        const double yi = y[i];
        if (yi != 0) data = __dadd_rn(__dmul_rn(yi, col[i]), data);
    }
    r[j] = __dadd_rn(__dmul_rn(-1.0, data), cost[var]);
}
void launch_exact_r(const double *At, int ld, int m, int nn, const int32_t *nonbasic, const double *y, const double *cost, double *r, int ldt, hipStream_t s) {
    hipLaunchKernelGGL(k_exact_r, dim3((unsigned int)((ldt + 255) / 256)), dim3(256), 0, s, At, ld, m, nn, nonbasic, y, cost, r, ldt);
}

// ---- condition numbers of the current basis from the tableau (gonum's LU.Solve guards, mat/lu.go:301,321) -------------------------
// A standard form that starts from its slack basis carries B^-1 inside the tableau: the columns of B^-1 are the tableau columns of the m
// identity (slack) variables [slack0, nvar) — column j of T for one that is nonbasic at position j, the unit vector e_p for one that is
// basic at position p.  So |B^-1|_1 and |B^-1|_inf are a column-sum / row-sum pass over T, and |B|_1, |B|_inf one over the m basic columns
// of A: the EXACT kappa_1 and kappa_inf of any basis, at any size, for two passes of the tableau's size (the reference has gonum's Dgecon
// estimate of the same numbers).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// grid (ceil(ldt / 64), ceil(m / 64)), block (64, 4): thread (x, y) = column x of the block, rows y * 16 .. y * 16 + 15
__global__ __launch_bounds__(256) void k_cond_tab(const double *__restrict__ T, int ldt, int m, int nn, const int32_t *__restrict__ nonbasic, int slack0,
                                                  int nvar, double *__restrict__ colsum, double *__restrict__ rowsum, int tiled) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i0 = blockIdx.y * 64 + threadIdx.y * 16;
    bool on = false;
    if (j < nn) { const int var = nonbasic[j]; on = var >= slack0 && var < nvar; }
    double cs = 0.0;
    for (int r = 0; r < 16; r++) {
        const int i = i0 + r;
        const double v = (on && i < m) ? fabs(T[tab_idx(i, j, ldt, tiled)]) : 0.0;
        cs += v;
        const double rs = wave_sum_f64(v);   // the 64 lanes of a wave are 64 columns of row i
        if (threadIdx.x == 0 && i < m) atomicAdd(rowsum + i, rs);
    }
    if (on) atomicAdd(colsum + j, cs);
}
// grid (ceil(m / 64), ceil(m / 64)), block (64, 4): thread (x, y) = row x of the block's row chunk, basic positions y * 16 .. + 15 of its
// column chunk; At is column-major A, so the lanes of a wave read 64 consecutive doubles
__global__ __launch_bounds__(256) void k_cond_basis(const double *__restrict__ At, int ld, int m, const int32_t *__restrict__ basic, double *__restrict__ colsumB,
                                                    double *__restrict__ rowsumB) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int p0 = blockIdx.y * 64 + threadIdx.y * 16;
    double rs = 0.0;
    for (int r = 0; r < 16; r++) {
        const int p = p0 + r;
        const double v = (p < m && i < m) ? fabs(At[(size_t)basic[p] * ld + i]) : 0.0;
        rs += v;
        const double cs = wave_sum_f64(v);
        if (threadIdx.x == 0 && p < m) atomicAdd(colsumB + p, cs);
    }
    if (i < m) atomicAdd(rowsumB + i, rs);
}
// one workgroup: out = {|B|_1, |B^-1|_1, |B|_inf, |B^-1|_inf}; a NaN anywhere makes the norm NaN (mat.Norm / mat.Cond propagate it)
__global__ __launch_bounds__(1024) void k_cond_finish(const double *__restrict__ colsum, const double *__restrict__ rowsum, const double *__restrict__ colsumB,
                                                      const double *__restrict__ rowsumB, int m, int nn, const int32_t *__restrict__ basic, int slack0, int nvar,
                                                      double *__restrict__ out) {
    __shared__ double red[4][16];
    __shared__ int s_nan;
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    double b1 = 0, i1 = 0, binf = 0, iinf = 0;
    bool bad = false;
    for (int j = threadIdx.x; j < nn; j += 1024) { const double v = colsum[j]; bad |= v != v; i1 = fmax(i1, v); }
    for (int p = threadIdx.x; p < m; p += 1024) {
        const bool slack = basic[p] >= slack0 && basic[p] < nvar;
        const double c = colsumB[p], r = rowsumB[p], q = rowsum[p] + (slack ? 1.0 : 0.0);
        bad |= c != c || r != r || q != q;
        b1 = fmax(b1, c); binf = fmax(binf, r); iinf = fmax(iinf, q);
        if (slack) i1 = fmax(i1, 1.0);
    }
    if (bad) s_nan = 1;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        b1 = fmax(b1, __shfl_xor(b1, o)); i1 = fmax(i1, __shfl_xor(i1, o)); binf = fmax(binf, __shfl_xor(binf, o)); iinf = fmax(iinf, __shfl_xor(iinf, o));
    }
    if (lane == 0) { red[0][wv] = b1; red[1][wv] = i1; red[2][wv] = binf; red[3][wv] = iinf; }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0;
        for (int w2 = 0; w2 < 16; w2++) v = fmax(v, red[threadIdx.x][w2]);
        out[threadIdx.x] = s_nan ? __builtin_nan("") : v;
    }
}
// scratch: colsum[ldt] | rowsum[ld] | colsumB[ld] | rowsumB[ld] | out[4] (zeroed here)
void launch_cond_check(const double *T, int ldt, int m, int nn, const int32_t *nonbasic, const int32_t *basic, const double *At, int ld, int slack0, int nvar,
                       double *scratch, bool tiled, hipStream_t s) {
    double *colsum = scratch, *rowsum = colsum + ldt, *colsumB = rowsum + ld, *rowsumB = colsumB + ld, *out = rowsumB + ld;
    hipMemsetAsync(scratch, 0, ((size_t)ldt + 3 * (size_t)ld + 4) * sizeof(double), s);
    hipLaunchKernelGGL(k_cond_tab, dim3((unsigned int)((nn + 63) / 64), (unsigned int)((m + 63) / 64)), dim3(64, 4), 0, s, T, ldt, m, nn, nonbasic, slack0, nvar, colsum, rowsum,
                       tiled ? 1 : 0);
    hipLaunchKernelGGL(k_cond_basis, dim3((unsigned int)((m + 63) / 64), (unsigned int)((m + 63) / 64)), dim3(64, 4), 0, s, At, ld, m, basic, colsumB, rowsumB);
    hipLaunchKernelGGL(k_cond_finish, dim3(1), dim3(1024), 0, s, colsum, rowsum, colsumB, rowsumB, m, nn, basic, slack0, nvar, out);
}

void launch_tab_column(const double *T, int ldt, int m, int jp, const double *xb, double *dvec, double *move, bool tiled, hipStream_t s) {
    hipLaunchKernelGGL(k_tab_column, dim3((m + 255) / 256), dim3(256), 0, s, T, ldt, m, jp, xb, dvec, move, tiled ? 1 : 0);
}

}  // namespace gomilp
