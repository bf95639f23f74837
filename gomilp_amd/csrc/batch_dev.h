// Device-side description of a batched relaxation's STARTING tableau (shared by batch_kernels.hip and the virtual-tableau block kernel of
// bt_kernels.hip): the child's standard form A' = [[A0, 0], [G#, I_K]] (subproblem.go:81-139) is never materialised — an entry of
// T0 = B0^-1 A'_N0 is read off the root's resident columns / rows and the child's branch rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

__device__ __forceinline__ int b_rho(const BatchLP &lp, int pos) {   // row of the 1 in the slack column at basis position pos
    return pos < lp.K ? lp.m0 + lp.K - 1 - pos : lp.rho0[pos - lp.K];
}
__device__ __forceinline__ double b_rhs(const BatchLP &lp, int r) { return r < lp.m0 ? lp.b0[r] : lp.rhs[r - lp.m0]; }
// cost of variable j in the given phase: Phase I minimises the artificial (simplex.go:545-547), Phase II c' = [c0, 0]
__device__ __forceinline__ double b_cost(const BatchLP &lp, int phase, int j) {
    if (phase == 1) return j == lp.n ? 1.0 : 0.0;
    return j < lp.n0 ? lp.c0[j] : 0.0;
}
__host__ __device__ __forceinline__ int b_ldt(int nn) { return ((nn + 63) / 64) * 64; }

// ---- T[pos, jp] = A'[rho(pos)][var(jp)] in 4x4 tiles, A' = [[A0, 0], [G#, I_K]] never materialised ---------------------
// rowwise: the caller walks a ROW of the tableau (consecutive jp): the root part comes from the row-major copy of A where there is one
template <bool ROWWISE = false>
__device__ __forceinline__ double b_entry(const BatchLP &lp, int pos, int jp, int nn) {
    if (pos >= lp.m || jp >= nn) return 0.0;
    if (lp.warm) {   // row 0 = the new branch row in the parent's nonbasic terms, then the parent's rows (k_b_setup_warm)
        const int ldt = b_ldt(nn);
        if (pos > 0) return lp.wT[tab_idx(pos - 1, jp, ldt, 1)];
        const int kk = lp.K - 1, pv = lp.wposvar[lp.var[kk]];
        if (pv >= 0) return -lp.sign[kk] * lp.wT[tab_idx(pv, jp, ldt, 1)];
        return (-1 - pv) == jp ? lp.sign[kk] : 0.0;
    }
    if (lp.gen) {
        const int nn0 = lp.n0 - lp.m0;
        if (jp >= nn0) return lp.art[pos];   // the artificial (tableau space, by position: k_b_setup)
        if (pos >= lp.K) return lp.gT0[(size_t)(pos - lp.K) * lp.gldt + jp];
        // branch row kk: sign * x_var + s = rhs with x_var = x_B0[pv] - T0[pv, :] x_N (var basic at pv) or the nonbasic variable itself
        const int kk = lp.K - 1 - pos, pv = lp.gposvar0[lp.var[kk]];
        if (pv >= 0) return -lp.sign[kk] * lp.gT0[(size_t)pv * lp.gldt + jp];
        return (-1 - pv) == jp ? lp.sign[kk] : 0.0;
    }
    const int r = b_rho(lp, pos);
    const int nn2 = lp.n - lp.m;
    const int j = jp < nn2 ? jp : lp.n;   // slack start: the nonbasic list is 0 .. nn2-1 (+ the artificial)
    if (j == lp.n) return lp.art[r];
    if (j < lp.n0) {
        if (r < lp.m0) {
            if constexpr (ROWWISE) { if (lp.A0r) return lp.A0r[(size_t)r * lp.lda0r + j]; }
            return lp.At0[(size_t)j * lp.ld0 + r];
        }
        const int kk = r - lp.m0;
        return lp.var[kk] == j ? lp.sign[kk] : 0.0;   // G# row k = sign_k * e_{var_k} (subproblem.go:245-255)
    }
    return (r == lp.m0 + (j - lp.n0)) ? 1.0 : 0.0;    // (a branch slack can only be nonbasic here if the list said so)
}


// entry (pos, jp) of the VIRTUAL tableau of a relaxation with lp.virt > 0: T0, and — once the host-chosen Phase-I pivot has run (virt_t0) —
// its rank-1 term u0 v0'^T on top, in the arithmetic k_b_gather mode 2 materialises it with (a rounded multiply, a rounded add, then the + 0
// of the update kernel's seven empty terms): the same bits as a read of the materialised tableau
template <bool ROWWISE>
__device__ __forceinline__ double b_virt_entry(const BatchLP &lp, int pos, int jp, int nn, double u0, double v0) {
    double v = b_entry<ROWWISE>(lp, pos, jp, nn);
    if (lp.virt_t0) {
        v = __dadd_rn(v, __dmul_rn(u0, v0));
        v = __dadd_rn(v, 0.0);
    }
    return v;
}

}  // namespace gomilp
