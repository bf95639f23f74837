// Device-batched relaxations (gfx950): the set-up, phase-change and reduced-cost kernels that let a whole wave of
// branch-and-bound children (reference: the solveWorker pool of tree.go:98-100,196-205, one lp.Simplex per node,
// subproblem.go:141-159) advance through ONE fixed schedule of launches — grid.x / grid.z = relaxation, per-relaxation
// argument block (BatchLP) in HBM, finished relaxations leave at once.  The pivots themselves are the block kernels of
// bt_kernels.hip (k_bt_inner2_batch / k_bt_update_tiled_batch: same device code as the single-relaxation path).
//
// What the host does per relaxation on the single path (engine_tableau.cpp: Phase-I set-up, the forced pivot, the
// Phase I -> II column permutation, the reduced-cost rebuilds) happens here on the device:
//   k_b_setup     x_B, index lists, feasibility of the slack basis, artificial column, forced-pivot order (simplex.go:492-556)
//   k_b_gather    T = B^-1 A_N straight from the ROOT's columns + the child's branch rows (no per-child copy of A)
//   k_b_ctrl      the stage machine: after the forced pivot, at the end of Phase I (infeasible / Phase II / hand to the
//                 host) and at the end of Phase II; rebuilds the ascending nonbasic list of simplex.go:174-184
//   k_b_permute   T columns follow the rebuilt list
//   k_b_tab_r     reduced costs r = c_N - c_B^T T in the fixed chunk order of the single path (same bits)
// Anything outside the common path (a zero-level artificial that needs the exchange of simplex.go:581-606, the guard band
// around phaseIZeroTol, a zero artificial column) sets BS_HOST: the host solves that relaxation through the
// single-relaxation engine (the path the parity tests pin pivot by pivot).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "device_types.h"
#include "kernels_common.h"
#include "batch_dev.h"

namespace gomilp {

__device__ __forceinline__ void b_reset_state(DevState *st) {
    st->done = 0; st->status = ST_RUNNING; st->pivots = 0; st->kdone = 0; st->bland_steps = 0; st->trace_len = 0;
    st->max_pivots = 0; st->lu_singular = 0;
    st->kdone2[0] = st->kdone2[1] = 0; st->loop_blocks = 0; st->dead1 = 0;
}

// ---- set-up: slack basis of the child, feasibility, Phase-I order ---------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_b_setup(BatchLP *__restrict__ lps) {
    __shared__ unsigned long long sk[kWavesPerBlock];
    __shared__ unsigned int si[kWavesPerBlock];
    __shared__ int s_bad;
    BatchLP &lp = lps[blockIdx.x];
    if (lp.warm) return;   // (k_b_setup_warm)
    const int tid = threadIdx.x, m = lp.m, n = lp.n;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    unsigned long long k = ~0ull;
    unsigned int idx = 0xFFFFFFFFu;
    int bad = 0;
    for (int pos = tid; pos < lp.ldu; pos += kBlock) {
        double x = 0.0;
        if (pos < m && lp.gen) {
            // basis of the child = its K branch slacks (the descending scan of findLinearlyIndependent meets them first:
            // positions 0 .. K-1, slack of branch row K-1-pos) + the root's initial basis; B'^-1 b' by block elimination
            if (pos < lp.K) {
                const int kk = lp.K - 1 - pos, pv = lp.gposvar0[lp.var[kk]];
                x = lp.rhs[kk] - lp.sign[kk] * (pv >= 0 ? lp.gxb0[pv] : 0.0);
                lp.basic[pos] = lp.n0 + kk;
            } else {
                x = lp.gxb0[pos - lp.K];
                lp.basic[pos] = lp.gbasic0[pos - lp.K];
            }
            amin_take(k, idx, ordkey(x), (unsigned int)pos);
            if (x < -1e-13) bad = 1;
        } else if (pos < m) {
            x = b_rhs(lp, b_rho(lp, pos));   // ab = permutation: x_B = ab^-1 b exactly (initializeFromBasic, simplex.go:447-471)
            lp.basic[pos] = n - 1 - pos;      // descending scan of findLinearlyIndependent (simplex.go:618-635)
            amin_take(k, idx, ordkey(x), (unsigned int)pos);
            if (x < -1e-13) bad = 1;          // initPosTol
        }
        lp.xb[pos] = x;
    }
    if (bad) s_bad = 1;
    block_argmin(k, idx, sk, si);   // floats.MinIdx (simplex.go:532); barriers inside
    const int infeasible = s_bad;
    const int minidx = idx == 0xFFFFFFFFu ? 0 : (int)idx;
    const int nn2 = n - m;
    for (int jp = tid; jp < nn2; jp += kBlock) lp.nonbasic[jp] = lp.gen ? lp.gnonbasic0[jp] : jp;   // ascending ids not in the basis (simplex.go:174-184)
    int art_zero = 0;
    if (infeasible && lp.gen) {
        // the artificial column in tableau space, by basis POSITION: B^-1 (b - sum_{i != minidx} a_basic_i) = x_B - 1 + e_minidx
        int nz = 0;
        for (int pos = tid; pos < lp.ldu; pos += kBlock) {
            double v = 0.0;
            if (pos < m) v = (pos == minidx) ? lp.xb[pos] : lp.xb[pos] - 1.0;
            lp.art[pos] = v;
            if (v != 0) nz = 1;
        }
        art_zero = !__syncthreads_or(nz);
    } else if (infeasible) {
        // a_{n+1} = b - sum_{i != minidx} a_{basic_i}: unit columns, one exact "- 1" per row (floats.Sub, simplex.go:536-542)
        const int rmin = b_rho(lp, minidx);
        int nz = 0;
        for (int r = tid; r < lp.ldu; r += kBlock) {
            double v = 0.0;
            if (r < m) v = (r == rmin) ? b_rhs(lp, r) : -1 * 1.0 + b_rhs(lp, r);
            lp.art[r] = v;
            if (v != 0) nz = 1;
        }
        art_zero = !__syncthreads_or(nz);
    }
    if (tid == 0) {
        BTArgs &a = lp.bt;
        a.m = m; a.ldu = lp.ldu; a.T = lp.T[0]; a.U = lp.U; a.V = lp.V; a.r = lp.R; a.xb = lp.xb;
        a.basic = lp.basic; a.nonbasic = lp.nonbasic; a.st = lp.st; a.trace = nullptr; a.trace_cap = 0;
        a.nt_force = 0; a.tiled = 1; a.old_only = 0; a.stamps = nullptr;
        lp.tcur = 0; lp.do_permute = 0; lp.do_r = 0; lp.wrapped = 0; lp.piv1 = lp.piv2 = lp.bland = 0;
        lp.status = 0;
        b_reset_state(lp.st);
        lp.st->tsel2[0] = lp.st->tsel2[1] = 0;
        if (!infeasible) {
            lp.phase1_used = 0;
            a.nn = nn2; a.ldt = b_ldt(nn2); a.phase = 2; a.tol = lp.tol_user; a.kmax = 0;   // no pivot in the forced round
            a.forced_q = a.forced_p = -1; a.forced_nocommit = 0;
            lp.stage = BS_P2_START;
        } else {
            lp.phase1_used = 1;
            const int nn1 = nn2 + 1;
            lp.nonbasic[nn2] = n;   // the artificial is the last nonbasic column; one forced pivot brings it to position minidx
            a.nn = nn1; a.ldt = b_ldt(nn1); a.phase = 1; a.tol = 1e-10; a.kmax = 1;
            a.forced_q = nn1 - 1; a.forced_p = minidx; a.forced_nocommit = 2;   // lists exchanged on the device, uncounted
            lp.art_pos = minidx;
            lp.stage = art_zero ? BS_HOST : BS_FORCED;   // zero artificial column: verifyInputs of the recursive call fails (host path reports it)
        }
    }
}

// ---- warm start: the child = its parent's final basis + the slack of its ONE new branch row (the last of its list) -----------------------
// Parent state (WarmStore, engine_batch.cpp): tableau T_p (wm rows, 4x4 tiles, this child's ldt), updated x_B, positional lists, map
// variable -> position (>= 0 basic, -1 - position nonbasic).  New row:  sign * x_var + s = rhs  with x_var = x_B[pv] - T_p[pv, :] x_N
// (var basic at pv) or x_var = the nonbasic variable itself — so the slack is basic at position 0 with
//   x_B[0] = rhs - sign * x_B_p[pv]   (rhs when var is nonbasic),   T[0, :] = -sign * T_p[pv, :]   (sign * e_jq),
// the parent's rows behind it.  The reduced costs of the parent's optimum stay valid (the slack costs nothing): the basis is dual
// feasible, and primal infeasible at most in row 0 — the dual simplex repairs that (k_bt_inner2_dual_batch), or finds the row without
// a negative entry: infeasible.
__global__ __launch_bounds__(kBlock) void k_b_setup_warm(BatchLP *__restrict__ lps) {
    BatchLP &lp = lps[blockIdx.x];
    if (!lp.warm) return;
    const int tid = threadIdx.x, m = lp.m, n = lp.n, wm = lp.wm;
    const int kk = lp.K - 1;
    const int var = lp.var[kk];
    const double sign = lp.sign[kk], rhs = lp.rhs[kk];
    const int pv = lp.wposvar[var];
    const int nn2 = n - m;
    for (int pos = tid; pos < lp.ldu; pos += kBlock) {
        double x = 0.0;
        if (pos == 0) { x = rhs - sign * (pv >= 0 ? lp.wxb[pv] : 0.0); lp.basic[0] = n - 1; }
        else if (pos < m) { x = lp.wxb[pos - 1]; lp.basic[pos] = lp.wbasic[pos - 1]; }
        lp.xb[pos] = x;
    }
    for (int jp = tid; jp < nn2; jp += kBlock) lp.nonbasic[jp] = lp.wnonbasic[jp];
    if (tid == 0) {
        BTArgs &a = lp.bt;
        a.m = m; a.ldu = lp.ldu; a.T = lp.T[0]; a.U = lp.U; a.V = lp.V; a.r = lp.R; a.xb = lp.xb;
        a.basic = lp.basic; a.nonbasic = lp.nonbasic; a.st = lp.st; a.trace = nullptr; a.trace_cap = 0;
        a.nt_force = 0; a.tiled = 1; a.old_only = 0; a.stamps = nullptr;
        lp.tcur = 0; lp.do_permute = 0; lp.do_r = 0; lp.wrapped = 0; lp.piv1 = lp.piv2 = lp.bland = 0; lp.pivd = 0;
        lp.status = 0; lp.phase1_used = 0;
        b_reset_state(lp.st);
        lp.st->tsel2[0] = lp.st->tsel2[1] = 0;
        a.nn = nn2; a.ldt = b_ldt(nn2); a.phase = 2; a.tol = lp.tol_user; a.kmax = 0;   // no pivot before the reduced costs exist
        a.forced_q = a.forced_p = -1; a.forced_nocommit = 0;
        lp.stage = wm + 1 == m ? BS_DUAL_START : BS_HOST;
    }
}

// mode 0: every relaxation's whole tableau.
// mode 1 (with k_b_gather mode 2 behind the set-up block): a relaxation that starts with the forced Phase-I pivot (BS_FORCED) gets only
//   the 32 x 32 blocks that hold the pivot's row and column — all the block kernel reads for that one pivot;
// mode 2: the whole tableau of those relaxations WITH the pivot's rank-1 term applied, T' = T + u v'^T, in the arithmetic of
//   k_bt_update_tiled_batch (a rounded multiply, a rounded add, then the + 0 of its seven empty terms) — the tableau is written once
//   instead of written, read and written again: on a wide frontier wave, where Phase I is most of the work, 4.9 GB instead of 14.7 GB
//   (2048 children).  Everybody else was gathered in full by mode 1 and has no term to apply (kmax = 0 in the set-up block).
// mode 3 (virtual tableau, BatchLP::virt == 1: the set-up pivot and the first block ran on computed entries): the tableau of every relaxation
//   that is still alive, ONCE: T0, the set-up pivot's term (U / V row 8) as mode 2 applies it, then the first block's kdone terms (rows 0 .. 7)
//   in the arithmetic of k_bt_update_tiled_batch<8> (k ascending, a rounded multiply and a rounded add per term, empty terms as + 0 * 0) — the
//   bits the materialised path would hold behind its first update.  Relaxations whose Phase I ended inside the block with the artificial
//   above the zero tolerance are skipped by the update kernel's own test: on a B&B frontier that is nearly all of a wide wave.
constexpr int kGatherTiles = 8;   // 32 x 32 blocks per workgroup, side by side in a row of blocks (one block per workgroup: 554 k workgroups of ~1 us for a 2048-wide wave)
__global__ void k_b_gather(const BatchLP *__restrict__ lps, int nlp, int mode) {
    __shared__ double tile[32][33];
    // (grid.z may be smaller than the wave: a slice walks the relaxations z, z + gridDim.z, ... — mode 3 visits a wide wave of which a few
    // per cent are alive, and one workgroup per relaxation and tile that only finds out that it has nothing to do was most of its time)
    for (int zi = blockIdx.z; zi < nlp; zi += gridDim.z) {
    const BatchLP &lp = lps[zi];
    if (lp.stage == BS_HOST || lp.stage == BS_DONE || lp.stage == BS_COLD) continue;
    const int m = lp.m, nn = lp.bt.nn, ldt = lp.bt.ldt;
    const int m4 = (m + 3) & ~3;
    const int p0 = blockIdx.x * 32;
    if (p0 >= m4) continue;
    if (mode == 3 && (lp.virt != 1 || lp.st->dead1)) continue;   // (dead1: the block kernel's verdict — the test of k_bt_update_tiled_batch, taken once per relaxation)
    const bool forced = lp.stage == BS_FORCED;
    if (mode == 2 && !forced) continue;
    const int fp = lp.bt.forced_p, fq = lp.bt.forced_q;
    const bool rowhit = fp >= p0 && fp < p0 + 32;
    const bool term = mode == 2 && lp.st->kdone > 0;   // (uniform) the pivot ran: its term, row 0 of U / V
    const double *U = lp.bt.U, *V = lp.bt.V;
    const int kd3 = mode == 3 ? lp.st->kdone : 0;       // mode 3: terms of the first block
    const int ldu3 = lp.bt.ldu;
    double *T = lp.T[0];
    // mode 3: the nine u entries of this thread's row (the same for every column it visits)
    double u3[9];
    if (mode == 3) {
        const int pos = p0 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 8; k++) u3[k] = (k < kd3 && pos < m) ? U[(size_t)k * ldu3 + pos] : 0.0;
        u3[8] = (lp.virt_t0 && pos < m) ? U[(size_t)8 * ldu3 + pos] : 0.0;
    }
    const int t = threadIdx.y * 32 + threadIdx.x;
    const int til = t >> 2, rit = t & 3;
    const int tr = til >> 3, tc = til & 7;
    for (int g = 0; g < kGatherTiles; g++) {
        const int j0 = ((int)blockIdx.y * kGatherTiles + g) * 32;
        if (j0 >= ldt) break;
        if (mode == 1 && forced && !rowhit && !(fq >= j0 && fq < j0 + 32)) continue;   // (uniform)
        for (int rr = threadIdx.y; rr < 32; rr += 8) {
            double v = b_entry(lp, p0 + threadIdx.x, j0 + rr, nn);
            if (mode == 3) {
                const int jp = j0 + rr;
                if (lp.virt_t0) {
                    const double vv = jp < ldt ? V[(size_t)8 * ldt + jp] : 0.0;
                    v = __dadd_rn(v, __dmul_rn(u3[8], vv));
                    v = __dadd_rn(v, 0.0);
                }
                if (kd3 > 0) {
                    double vk[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) vk[k] = (k < kd3 && jp < ldt) ? V[(size_t)k * ldt + jp] : 0.0;   // (all eight loads in flight together)
#pragma unroll
                    for (int k = 0; k < 8; k++) v = __dadd_rn(v, __dmul_rn(u3[k], vk[k]));
                }
            }
            if (term) {
                const int pos = p0 + threadIdx.x, jp = j0 + rr;
                const double u = pos < m ? U[pos] : 0.0;       // (rows beyond m: zero, as the update kernel stages them)
                const double vv = jp < ldt ? V[jp] : 0.0;
                v = __dadd_rn(v, __dmul_rn(u, vv));
                v = __dadd_rn(v, 0.0);                        // the empty terms k = 1 .. 7 of the update kernel: t + 0 * 0
            }
            tile[rr][threadIdx.x] = v;
        }
        __syncthreads();
        // the 32 x 32 block is 8 x 8 tiles of the 4x4-tiled tableau: a thread stores one row of a tile (32 contiguous bytes), four neighbours
        // a whole tile, a wave two tile rows of 1 KB each — full lines (row by row, 32 lanes wrote 32-byte pieces 128 bytes apart: 2.9 TB/s
        // for the 4.9 GB of a 2048-wide wave).  ldt is a multiple of 4, m4 too: a tile is inside or outside as a whole.
        const int pos = p0 + tr * 4 + rit, jp = j0 + tc * 4;
        if (jp < ldt && pos < m4) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 lo, hi;
            lo.x = tile[tc * 4 + 0][tr * 4 + rit]; lo.y = tile[tc * 4 + 1][tr * 4 + rit];
            hi.x = tile[tc * 4 + 2][tr * 4 + rit]; hi.y = tile[tc * 4 + 3][tr * 4 + rit];
            d2 *dst = reinterpret_cast<d2 *>(T + tab_idx(pos, jp, ldt, 1));   // (16-byte aligned: 4 doubles of a tile row)
            dst[0] = lo; dst[1] = hi;   // padding rows / columns are zeros
        }
        __syncthreads();   // (the block's values are out of LDS before the next one comes in)
    }
    }
}

// ---- the tableaus that survive a virtual first block, written once (the job of k_b_gather mode 3, one 4x4 tile per thread) -------------
// k_b_gather moves 32 x 32 blocks through LDS, four entries per thread and two barriers per block: built for the plain T0 of mode 0 / 1, it wrote the
// term-laden tableaus of mode 3 at 0.8 TB/s — 18 dependent term loads per entry.  Here a thread owns a whole tile (one 128-byte line of the
// 4x4-tiled layout): the nine v' entries of its four columns stay in registers for the four rows, the nine u entries of a row are the same
// address for every thread of a tile row (broadcast loads), and the finished tile leaves as four 32-byte stores — no LDS, no barrier.
// Arithmetic per entry exactly as mode 3 / the update kernel: T0, (+ u0 v0' rounded twice, + 0), then k = 0 .. 7 ascending, a rounded multiply
// and a rounded add each.  grid.x: 256-tile pieces of a tableau, grid.y: slices of the wave (a slice walks z, z + gridDim.y, ...)
// (The first form walked the whole wave in every workgroup — grid.y slices over all nlp relaxations, 87 % of them dead: 2.96 ms of an
// 8192-wide wave, 0.39 of a 2048-wide one, most of it workgroups that only looked.  Now k_b_virt_alive lists the survivors first and the
// workgroups of k_b_write_virt stride over (survivor, 256-tile piece) items.)
__global__ __launch_bounds__(1024) void k_b_virt_alive(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, int *__restrict__ list, int *__restrict__ nalive) {
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int n = *count;
    for (int k = threadIdx.x; k < n; k += 1024) {
        const int li = ids[k];
        const BatchLP &lp = lps[li];
        if (lp.stage == BS_HOST || lp.stage == BS_DONE || lp.stage == BS_COLD || lp.virt != 1 || lp.st->dead1) continue;
        list[atomicAdd(&s_n, 1)] = li;
    }
    __syncthreads();
    if (threadIdx.x == 0) *nalive = s_n;
}
// Eight threads per 4x4 tile, 16 bytes each: a wave's store is eight whole 128-byte lines (one tile per thread — eight 16-byte stores into a
// line of its own — ran at 1.0 TB/s).  An item is a strip of 32 tile columns of one survivor, walked from the first tile row to the last:
// the nine v' pairs of a thread's two columns stay in registers, a tile row costs it nine u loads (shared by the lanes of that row), two
// entries of T0 and one store — the index arithmetic and the v' loads per ENTRY were what the flat form (a thread = two entries of
// any tile) spent its time on: 2.1 ms for the 2.6 GB of an 8192-wide wave's survivors.
__global__ __launch_bounds__(256) void k_b_write_virt(const BatchLP *__restrict__ lps, const int *__restrict__ list, const int *__restrict__ nalive, int strips) {
    // few survivors: a strip is cut into row segments so that every workgroup of the grid has work
    const int na = *nalive;
    const int segs = na * strips >= (int)gridDim.x ? 1 : min(16, ((int)gridDim.x + na * strips - 1) / max(1, na * strips));
    const int total = na * strips * segs;
    const int sub = threadIdx.x & 7, rr = sub >> 1, ch = (sub & 1) * 2;
    typedef double d2 __attribute__((ext_vector_type(2)));
    for (int w0 = blockIdx.x; w0 < total; w0 += gridDim.x) {
        const int w = w0 / segs, seg = w0 % segs;
        const BatchLP &lp = lps[list[w / strips]];
        const int m = lp.m, nn = lp.bt.nn, ldt = lp.bt.ldt, ldu = lp.bt.ldu;
        const int ntr = (m + 3) >> 2, ntc = ldt >> 2;
        const int J = (w % strips) * 32 + (threadIdx.x >> 3);
        if (J >= ntc) continue;
        const int kd = lp.st->kdone;
        const bool t0 = lp.virt_t0 != 0;
        const double *U = lp.bt.U, *V = lp.bt.V;
        const int jc = 4 * J + ch;
        d2 vk[9];
#pragma unroll
        for (int k = 0; k < 9; k++) vk[k] = (k < 8 ? k < kd : t0) ? *reinterpret_cast<const d2 *>(V + (size_t)k * ldt + jc) : d2{0.0, 0.0};
        double *dst = lp.T[0] + (size_t)J * 16 + rr * 4 + ch;
        const int Ia = (int)((long)ntr * seg / segs), Ib = (int)((long)ntr * (seg + 1) / segs);
#pragma unroll 4
        for (int I = Ia; I < Ib; I++) {
            const int pos = 4 * I + rr;
            double uk[9];
#pragma unroll
            for (int k = 0; k < 9; k++) uk[k] = ((k < 8 ? k < kd : t0) && pos < m) ? U[(size_t)k * ldu + pos] : 0.0;
            double e[2];
#pragma unroll
            for (int c = 0; c < 2; c++) {
                double v = b_entry<true>(lp, pos, jc + c, nn);
                if (t0) {
                    v = __dadd_rn(v, __dmul_rn(uk[8], vk[8][c]));
                    v = __dadd_rn(v, 0.0);
                }
                if (kd > 0) {
#pragma unroll
                    for (int k = 0; k < 8; k++) v = __dadd_rn(v, __dmul_rn(uk[k], vk[k][c]));
                }
                e[c] = v;
            }
            *reinterpret_cast<d2 *>(dst + (size_t)I * ntc * 16) = d2{e[0], e[1]};
        }
    }
}
// ids / count: the active list the first block ran on; list / nalive: scratch (the list and the count the coming control step will write)
void launch_b_write_virt(const BatchLP *lps, const int *ids, const int *count, int *list, int *nalive, int bound, int m_max, int ldt_max, int ncu, hipStream_t s) {
    const int strips = (ldt_max / 4 + 31) / 32;
    hipLaunchKernelGGL(k_b_virt_alive, dim3(1), dim3(1024), 0, s, lps, ids, count, list, nalive);
    const long want = (long)bound * strips;
    const int grid = (int)std::max<long>(1, std::min<long>(want, (long)std::max(ncu, 1) * 8));
    hipLaunchKernelGGL(k_b_write_virt, dim3(grid), dim3(256), 0, s, lps, list, nalive, strips);
}

// ---- the stage machine ----------------------------------------------------------------------------------------------
// dynamic LDS: 2 * (n_max + 2) ints (flags, old positions)
// loop_par >= 0: the blocks of this superstep ran in the persistent loop kernel (launch parity loop_par): it left the buffer that holds
// the tableau in DevState::tsel2 (bt_kernels.hip k_b_loop)
__global__ __launch_bounds__(kBlock) void k_b_ctrl(BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count, BatchOut *__restrict__ outs, int loop_par) {
    extern __shared__ __attribute__((aligned(16))) int sh_ctrl[];
    __shared__ int s_scan[kBlock];
    __shared__ double s_red[kBlock];
    __shared__ int s_added;
    if ((int)blockIdx.x >= *count) return;   // the grid is sized from an older (larger) count
    const int li = ids[blockIdx.x];
    BatchLP &lp = lps[li];
    const int tid = threadIdx.x;
    const int m = lp.m, n = lp.n;
    const int stage = lp.stage;
    DevState *st = lp.st;
    if (tid == 0) {
        lp.do_permute = 0; lp.do_r = 0; s_added = -1;
        if (loop_par >= 0) { lp.tcur = st->tsel2[(loop_par ^ 1) & 1] & 1; lp.bt.T = lp.T[lp.tcur]; }
        // virtual tableau: 2 (set-up pivot) -> 1 (first block) -> 0 (k_b_gather mode 3 has written out what is alive, in front of this step)
        if (lp.virt == 2) { lp.virt = 1; lp.virt_t0 = (stage == BS_FORCED && st->kdone > 0) ? 1 : 0; }
        else if (lp.virt == 1) lp.virt = 0;
    }
    __syncthreads();
    bool to_phase2 = false;   // uniform
    if (stage == BS_FORCED) {
        // the Phase-I starting vertex must be feasible: initializeFromBasic inside the recursive call panics otherwise
        // (simplex.go:155-158) — the host path reports that
        int bad = 0;
        for (int i = tid; i < m; i += kBlock) if (lp.xb[i] < -1e-13) bad = 1;
        bad = __syncthreads_or(bad);
        if (tid == 0) {
            if (bad) lp.stage = BS_HOST;
            else {
                BTArgs &a = lp.bt;
                a.forced_q = a.forced_p = -1; a.forced_nocommit = 0; a.kmax = lp.kblock;
                b_reset_state(st);
                lp.do_r = 1; lp.r_phase = 1; lp.stage = BS_P1;
            }
        }
    } else if (stage == BS_P2_START) {
        if (tid == 0) {
            lp.bt.kmax = lp.kblock;
            b_reset_state(st);
            lp.do_r = 1; lp.r_phase = 2; lp.stage = BS_P2;
        }
    } else if (stage == BS_DUAL_START) {
        // warm start: reduced costs exist now; is the new row violated at the parent's optimum?
        if (tid == 0) {
            b_reset_state(st);
            lp.bt.kmax = lp.kblock;
            lp.do_r = 1; lp.r_phase = 2;   // r = c_N - c_B^T T of the parent's basis (k_b_tab_r, behind this step)
            const bool violated = lp.xb[0] < -1e-13;   // (initPosTol: what initializeFromBasic calls infeasible, simplex.go:447-471)
            lp.bt.tol = violated ? 1e-13 : lp.tol_user;   // the dual loop's tolerance is the primal feasibility tolerance
            lp.stage = violated ? BS_DUAL : BS_P2;
        }
    } else if (stage == BS_DUAL) {
        if (tid == 0) {
            if (st->done) {
                const int status = st->status;
                lp.pivd += st->pivots;
                if (status == ST_OPTIMAL) {   // primal feasible again: the primal loop confirms the optimum (usually without a pivot)
                    b_reset_state(st);
                    lp.bt.tol = lp.tol_user;
                    lp.stage = BS_P2;
                } else if (status == ST_DUAL_INFEASIBLE) {
                    lp.status = 2;   // lp.ErrInfeasible
                    lp.stage = BS_DONE;
                } else lp.stage = BS_COLD;
            } else if (lp.pivd + st->pivots >= lp.dual_budget) {
                lp.pivd += st->pivots;
                lp.stage = BS_COLD;   // budget spent: the caller takes the cold path for this relaxation
            }
        }
    } else if (stage == BS_P1 && st->done) {
        const int status = st->status;
        if (status != ST_OPTIMAL && status != ST_UNBOUNDED && status != ST_BLAND_FAILED) {
            // not an outcome of the algorithm (an exchange of the multi-workgroup block kernel timed out: a workgroup was not
            // resident in time): the single-relaxation engine solves this relaxation instead of reporting a device error
            if (tid == 0) lp.stage = BS_HOST;
        } else if (status != ST_OPTIMAL) {   // simplex.go:557-559: any error of the recursive call comes back wrapped
            if (tid == 0) {
                lp.piv1 += st->pivots; lp.bland += st->bland_steps;
                lp.wrapped = status == ST_UNBOUNDED ? 4 : 1;
                lp.status = 9;   // GOMILP_ERR_PHASE1_WRAPPED
                lp.stage = BS_DONE;
            }
        } else {
            for (int i = tid; i < m; i += kBlock) if (lp.basic[i] == n) s_added = i;
            __syncthreads();
            const int added = s_added;
            const double xart = added >= 0 ? lp.xb[added] : 0.0;
            const double ax = fabs(xart);
            if (added >= 0 && ax > 1e-13 && ax < 1e-11) {
                // guard band around phaseIZeroTol: the engine takes the value from a fresh gonum-order solve there
                if (tid == 0) lp.stage = BS_HOST;
            } else if (ax > 1e-12) {
                if (tid == 0) {
                    lp.piv1 += st->pivots; lp.bland += st->bland_steps;
                    lp.status = 2;   // lp.ErrInfeasible, phaseIZeroTol (simplex.go:563-565)
                    lp.stage = BS_DONE;
                }
            } else if (added >= 0) {
                // ---- simplex.go:581-606: the artificial stayed basic at level zero: exchange it for the first nonbasic variable
                // (ascending id) whose pivot element is usable and whose basis is feasible — the tests of engine_tableau.cpp
                int *pos_of = sh_ctrl;
                const int nn1 = lp.bt.nn, ldt = lp.bt.ldt;
                const double *T = lp.bt.T;
                for (int j = tid; j <= n; j += kBlock) pos_of[j] = -1;
                __syncthreads();
                // (the artificial's row first, all positions side by side: a position whose element cannot pass the first test is not
                // listed — the scan below went through them one dependent load after the other, 160 us of a 2.6 ms wave of the C3 tree)
                for (int jp = tid; jp < nn1; jp += kBlock) {
                    const double dv = T[tab_idx(added, jp, ldt, 1)];
                    pos_of[lp.nonbasic[jp]] = (fabs(dv) > 1e-9) ? jp : -1;   // 1e-9 * max(1, column max) >= 1e-9
                }
                __syncthreads();
                int found = -1, tries = 0;
                for (int id = 0; id < n && found < 0 && tries < 512; id++) {
                    const int jp = pos_of[id];
                    if (jp < 0) continue;
                    const double dpv = T[tab_idx(added, jp, ldt, 1)];
                    if (!(fabs(dpv) > 1e-9)) continue;
                    tries++;
                    const double theta = xart / dpv;
                    double mx = 0;
                    int bad = 0;
                    for (int i = tid; i < m; i += kBlock) {
                        const double dv = T[tab_idx(i, jp, ldt, 1)];
                        mx = fmax(mx, fabs(dv));
                        const double v = (i == added) ? theta : lp.xb[i] - theta * dv;
                        if (v < -1e-13) bad = 1;
                    }
                    s_red[tid] = mx;
                    bad = __syncthreads_or(bad);
                    for (int off = kBlock / 2; off > 0; off >>= 1) {
                        if (tid < off) s_red[tid] = fmax(s_red[tid], s_red[tid + off]);
                        __syncthreads();
                    }
                    const double dmax = s_red[0];
                    __syncthreads();
                    if (!(fabs(dpv) > 1e-9 * fmax(1.0, dmax)) || bad) continue;
                    found = jp;
                }
                if (tid == 0) {
                    lp.piv1 += st->pivots; lp.bland += st->bland_steps;
                    if (found >= 0) {
                        BTArgs &a = lp.bt;
                        a.forced_q = found; a.forced_p = added; a.forced_nocommit = 3; a.kmax = 1;   // lists exchanged on the device, runs once
                        b_reset_state(st);
                        lp.stage = BS_EXCH;
                    } else if (tries >= 512) {
                        lp.stage = BS_HOST;
                    } else {
                        lp.status = 2;   // no column works: lp.ErrInfeasible (simplex.go:606)
                        lp.stage = BS_DONE;
                    }
                }
            } else {
                if (tid == 0) { lp.piv1 += st->pivots; lp.bland += st->bland_steps; }
                to_phase2 = true;
            }
        }
    } else if (stage == BS_EXCH && st->done) {
        to_phase2 = true;   // the forced exchange pivot has run (ST_FORCED_DONE): the artificial is nonbasic now
    } else if (stage == BS_P2 && st->done) {
        if (tid == 0) {
            const int status = st->status;
            if (status != ST_OPTIMAL && status != ST_UNBOUNDED && status != ST_BLAND_FAILED) {
                lp.stage = BS_HOST;   // (see the Phase-I case: a transient device condition, the worker path solves it)
            } else {
                lp.piv2 += st->pivots; lp.bland += st->bland_steps;
                lp.status = status == ST_OPTIMAL ? 0 : (status == ST_UNBOUNDED ? 4 : 1);
                lp.stage = BS_DONE;
            }
        }
    }
    if (to_phase2) {
        // ---- Phase I -> Phase II: nonbasic list rebuilt in ascending variable order (simplex.go:174-184), the
        // artificial (nonbasic now) dropped; srcpos[new position] = old position, T columns follow (k_b_permute)
        __syncthreads();
        int *flag = sh_ctrl, *pos_of = sh_ctrl + (n + 2);
        const int nn1 = lp.bt.nn;
        for (int j = tid; j <= n; j += kBlock) flag[j] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += kBlock) flag[lp.basic[i]] = 1;
        for (int jp = tid; jp < nn1; jp += kBlock) pos_of[lp.nonbasic[jp]] = jp;
        __syncthreads();
        const int chunk = (n + kBlock - 1) / kBlock;
        const int lo = min(n, tid * chunk), hi = min(n, lo + chunk);
        int cnt = 0;
        for (int j = lo; j < hi; j++) cnt += !flag[j];
        s_scan[tid] = cnt;
        __syncthreads();
        if (tid == 0) {   // exclusive scan over 256 counts
            int run = 0;
            for (int t = 0; t < kBlock; t++) { const int c = s_scan[t]; s_scan[t] = run; run += c; }
        }
        __syncthreads();
        int at = s_scan[tid];
        for (int j = lo; j < hi; j++)
            if (!flag[j]) { lp.nonbasic[at] = j; lp.srcpos[at] = pos_of[j]; at++; }
        if (tid == 0) {
            const int nn2 = n - m;
            BTArgs &a = lp.bt;
            lp.perm_ld_in = a.ldt; lp.perm_nn_out = nn2;
            lp.tcur ^= 1;
            a.T = lp.T[lp.tcur]; a.ldt = b_ldt(nn2); a.nn = nn2; a.phase = 2; a.tol = lp.tol_user; a.kmax = lp.kblock;
            a.forced_q = a.forced_p = -1; a.forced_nocommit = 0;
            b_reset_state(st);
            lp.do_permute = 1; lp.do_r = 1; lp.r_phase = 2;
            lp.stage = BS_P2;
        }
    }
    __syncthreads();
    if (tid == 0) {
        st->tsel2[0] = st->tsel2[1] = lp.tcur;   // the next loop launch (either parity) starts from the current buffer
        st->kdone2[0] = st->kdone2[1] = 0;
        BatchOut &o = outs[li];
        o.stage = lp.stage; o.status = lp.status; o.wrapped = lp.wrapped; o.phase1_used = lp.phase1_used;
        o.piv1 = lp.piv1; o.piv2 = lp.piv2; o.bland = lp.bland; o.pivd = lp.pivd; o.tcur = lp.tcur; o.pad = 0;
    }
}

// the active list for the launches of the next superstep: the still-active members of the previous list, in order
__global__ __launch_bounds__(kBlock) void k_b_compact(const BatchLP *__restrict__ lps, const int *__restrict__ ids_in, const int *__restrict__ count_in,
                                                      int *__restrict__ ids, int *__restrict__ count) {
    __shared__ int s_cnt[kBlock];
    const int tid = threadIdx.x;
    const int nin = *count_in;
    const int chunk = (nin + kBlock - 1) / kBlock;
    const int lo = min(nin, tid * chunk), hi = min(nin, lo + chunk);
    int cnt = 0;
    for (int i = lo; i < hi; i++) { const int sg = lps[ids_in[i]].stage; cnt += (sg != BS_DONE && sg != BS_HOST && sg != BS_COLD); }
    s_cnt[tid] = cnt;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < kBlock; t++) { const int c = s_cnt[t]; s_cnt[t] = run; run += c; }
        *count = run;
    }
    __syncthreads();
    int at = s_cnt[tid];
    for (int i = lo; i < hi; i++) { const int li = ids_in[i]; const int sg = lps[li].stage; if (sg != BS_DONE && sg != BS_HOST && sg != BS_COLD) ids[at++] = li; }
}
__global__ void k_b_init_ids(int *__restrict__ ids, int *__restrict__ count, int nlp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nlp) ids[i] = i;
    if (i == 0) *count = nlp;
}

// T_out[:, jp] = T_in[:, srcpos[jp]] (both 4x4-tiled); the control kernel already made T_out the current buffer.
// One workgroup = 16 rows of one relaxation (a launch without work orders is nlp * m/16 empty workgroups, not m * nlp).
__global__ __launch_bounds__(kBlock) void k_b_permute(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.y >= *count) return;
    const BatchLP &lp = lps[ids[blockIdx.y]];
    if (!lp.do_permute) return;
    const int m = lp.m, m4 = (m + 3) & ~3, ld_out = lp.bt.ldt;
    const int i0 = blockIdx.x * 16;
    if (i0 >= m4) return;
    const double *Tin = lp.T[lp.tcur ^ 1];
    double *Tout = lp.T[lp.tcur];
    const int nn_out = lp.perm_nn_out, ld_in = lp.perm_ld_in;
    for (int jp = threadIdx.x; jp < ld_out; jp += kBlock) {
        const int src = jp < nn_out ? lp.srcpos[jp] : 0;
#pragma unroll 4
        for (int i = i0; i < min(i0 + 16, m4); i++)
            Tout[tab_idx(i, jp, ld_out, 1)] = (jp < nn_out && i < m) ? Tin[tab_idx(i, src, ld_in, 1)] : 0.0;
    }
}

// r[jp] = cost[nonbasic[jp]] - sum_i cost[basic[i]] * T[i, jp]: row chunks, fixed-order reduction — the arithmetic of
// k_tab_r_partial / k_tab_r_reduce (tableau_kernels.hip), so the batched and the single path see the same bits
__host__ __device__ __forceinline__ int b_r_chunks(int m) { int c = (m + 63) / 64; return c > 64 ? 64 : c; }

__global__ __launch_bounds__(kBlock) void k_b_tab_r_partial(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.z >= *count) return;
    const BatchLP &lp = lps[ids[blockIdx.z]];
    if (!lp.do_r) return;
    const int m = lp.m, ldt = lp.bt.ldt, phase = lp.r_phase;
    const int nchunks = b_r_chunks(m), rpc = (m + nchunks - 1) / nchunks;
    const int chunk = blockIdx.y, j = blockIdx.x * kBlock + threadIdx.x;
    if (chunk >= nchunks || j >= ldt) return;
    // Phase I on a slack start: the artificial (cost 1) sits at its entering position, every other basic variable costs nothing — the sum
    // below is the one term 0 + 1 * T[art_pos, j] in the chunk of that row and zeros elsewhere (a wide wave spent more time finding that
    // out row by row than on the pivots of the block)
    if (phase == 1 && !lp.gen && !lp.warm && lp.art_pos >= 0 && lp.art_pos < m && lp.basic[lp.art_pos] == lp.n) {
        const int i = lp.art_pos;
        double acc1 = 0;
        if (i >= chunk * rpc && i < min(m, chunk * rpc + rpc)) {
            const bool virt1 = lp.virt > 0;
            const double t = virt1 ? b_virt_entry<true>(lp, i, j, lp.bt.nn, lp.virt_t0 ? lp.bt.U[(size_t)8 * lp.bt.ldu + i] : 0.0, lp.virt_t0 ? lp.bt.V[(size_t)8 * ldt + j] : 0.0)
                                   : lp.bt.T[tab_idx(i, j, ldt, 1)];
            acc1 += 1.0 * t;
        }
        lp.scratch[(size_t)chunk * ldt + j] = acc1;
        return;
    }
    const int i0 = chunk * rpc, i1 = min(m, i0 + rpc);
    const double *T = lp.bt.T;
    const bool virt = lp.virt > 0;   // no tableau in HBM yet: the entries are computed (the set-up pivot's term: U / V row 8)
    const double v0 = (virt && lp.virt_t0) ? lp.bt.V[(size_t)8 * ldt + j] : 0.0;
    double acc = 0;
    for (int i = i0; i < i1; i++) {
        const double cb = b_cost(lp, phase, lp.basic[i]);
        if (cb != 0) acc += cb * (virt ? b_virt_entry<true>(lp, i, j, lp.bt.nn, lp.virt_t0 ? lp.bt.U[(size_t)8 * lp.bt.ldu + i] : 0.0, v0) : T[tab_idx(i, j, ldt, 1)]);
    }
    lp.scratch[(size_t)chunk * ldt + j] = acc;
}
__global__ void k_b_tab_r_reduce(const BatchLP *__restrict__ lps, const int *__restrict__ ids, const int *__restrict__ count) {
    if ((int)blockIdx.z >= *count) return;
    const BatchLP &lp = lps[ids[blockIdx.z]];
    if (!lp.do_r) return;
    const int ldt = lp.bt.ldt, nn = lp.bt.nn, phase = lp.r_phase;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ldt) return;
    const int nchunks = b_r_chunks(lp.m);
    double acc = 0;
    for (int c = 0; c < nchunks; c++) acc += lp.scratch[(size_t)c * ldt + j];
    lp.R[j] = (j < nn) ? b_cost(lp, phase, lp.nonbasic[j]) - acc : 0.0;
}

// warm store: variable -> position map of a finished relaxation (>= 0: basic at that position; -1 - jp: nonbasic at jp)
__global__ void k_b_posvar(const int32_t *__restrict__ basic, int m, const int32_t *__restrict__ nonbasic, int nn, int32_t *__restrict__ posvar) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) posvar[basic[i]] = i;
    if (i < nn) posvar[nonbasic[i]] = -1 - i;
}
void launch_b_posvar(const int32_t *basic, int m, const int32_t *nonbasic, int nn, int32_t *posvar, hipStream_t s) {
    const int n = m > nn ? m : nn;
    hipLaunchKernelGGL(k_b_posvar, dim3((n + 255) / 256), dim3(256), 0, s, basic, m, nonbasic, nn, posvar);
}
void launch_b_setup_warm(BatchLP *lps, int nlp, hipStream_t s) { hipLaunchKernelGGL(k_b_setup_warm, dim3(nlp), dim3(kBlock), 0, s, lps); }

// ---- launch wrappers --------------------------------------------------------------------------------------------------
int batch_ldt(int nn) { return b_ldt(nn); }

void launch_b_setup(BatchLP *lps, int nlp, hipStream_t s) { hipLaunchKernelGGL(k_b_setup, dim3(nlp), dim3(kBlock), 0, s, lps); }
void launch_b_gather(const BatchLP *lps, int nlp, int m_max, int ldt_max, int mode, hipStream_t s) {
    dim3 grid(((m_max + 3) / 4 * 4 + 31) / 32, ((ldt_max + 31) / 32 + kGatherTiles - 1) / kGatherTiles, mode == 3 ? std::min(nlp, 256) : nlp), block(32, 8);
    hipLaunchKernelGGL(k_b_gather, grid, block, 0, s, lps, nlp, mode);
}
// ids_in / count_in: the active list the previous control step left (everybody at the start); bound >= *count_in on the host
void launch_b_ctrl(BatchLP *lps, const int *ids_in, const int *count_in, int bound, int n_max, BatchOut *outs, int *ids_out, int *count_out, int loop_par, hipStream_t s) {
    const size_t lds = (size_t)2 * (n_max + 2) * sizeof(int);
    hipLaunchKernelGGL(k_b_ctrl, dim3(bound), dim3(kBlock), lds, s, lps, ids_in, count_in, outs, loop_par);
    hipLaunchKernelGGL(k_b_compact, dim3(1), dim3(kBlock), 0, s, lps, ids_in, count_in, ids_out, count_out);
}
void launch_b_init_ids(int *ids, int *count, int nlp, hipStream_t s) {
    hipLaunchKernelGGL(k_b_init_ids, dim3((nlp + 255) / 256), dim3(256), 0, s, ids, count, nlp);
}
void launch_b_permute(const BatchLP *lps, const int *ids, const int *count, int bound, int m_max, int ldt_max, hipStream_t s) {
    dim3 grid(((m_max + 3) / 4 * 4 + 15) / 16, bound);
    hipLaunchKernelGGL(k_b_permute, grid, dim3(kBlock), 0, s, lps, ids, count);
}
void launch_b_tab_r(const BatchLP *lps, const int *ids, const int *count, int bound, int m_max, int ldt_max, hipStream_t s) {
    dim3 grid((ldt_max + kBlock - 1) / kBlock, b_r_chunks(m_max), bound);
    hipLaunchKernelGGL(k_b_tab_r_partial, grid, dim3(kBlock), 0, s, lps, ids, count);
    hipLaunchKernelGGL(k_b_tab_r_reduce, dim3((ldt_max + 255) / 256, 1, bound), dim3(256), 0, s, lps, ids, count);
}

}  // namespace gomilp
