// Device-side helpers shared by the kernel files (wave64 reductions, first-index argmin keys, row dot).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"

#include <mutex>
#include <set>
#include <utility>

namespace gomilp {

// Raise a kernel's dynamic-LDS limit once per (device, kernel): the attribute is per device, and launches come from many host
// threads (flat contexts, pool workers, the two batch schedules).
inline void lds_attr_once(const void *fn, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(mu);
    if (done.insert(std::make_pair(dev, fn)).second) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// ------------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------------

// Order-preserving map double -> u64 for floats.MinIdx semantics (floats/floats.go:458-474):
// NaN never wins (largest key), -0 == +0, ties resolved by the smaller index.
__device__ __forceinline__ unsigned long long ordkey(double v) {
    if (v != v) return ~0ull;
    v = v + 0.0;  // -0 -> +0
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// inverse of ordkey (exact; -0 comes back as +0, NaN as NaN)
__device__ __forceinline__ double orddecode(unsigned long long k) {
    if (k == ~0ull) return __builtin_nan("");
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ void amin_take(unsigned long long &k, unsigned int &i, unsigned long long k2,
                                          unsigned int i2) {
    if (k2 < k || (k2 == k && i2 < i)) { k = k2; i = i2; }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void wave_argmin(unsigned long long &k, unsigned int &i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long k2 = __shfl_xor(k, o, 64);
        unsigned int i2 = __shfl_xor(i, o, 64);
        amin_take(k, i, k2, i2);
    }
}

// argmin over the 4 waves of a 256-thread workgroup; result valid in every thread
__device__ __forceinline__ void block_argmin(unsigned long long &k, unsigned int &i, unsigned long long *sk,
                                             unsigned int *si) {
    wave_argmin(k, i);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sk[w] = k; si[w] = i; }
    __syncthreads();
    k = sk[0]; i = si[0];
#pragma unroll
    for (int t = 1; t < kWavesPerBlock; t++) amin_take(k, i, sk[t], si[t]);
    __syncthreads();
}

// reduce the per-workgroup partials of the previous kernel (every workgroup does it redundantly:
// <= 1024 entries out of L2, cheaper than another kernel boundary)
__device__ __forceinline__ unsigned int reduce_partials(const unsigned long long *pk, const unsigned int *pi,
                                                        int nparts, unsigned long long *sk, unsigned int *si,
                                                        unsigned long long *key_out) {
    unsigned long long k = ~0ull;
    unsigned int i = 0xFFFFFFFFu;
    for (int t = threadIdx.x; t < nparts; t += kBlock) amin_take(k, i, pk[t], pi[t]);
    block_argmin(k, i, sk, si);
    if (key_out) *key_out = k;
    return i;
}

// argmin with two payload words travelling with the winner (fused pipeline: saves the dependent
// global loads that would otherwise follow the reduction)
struct ArgMinP {
    unsigned long long k;
    unsigned int i;
    unsigned int u;  // payload: variable id
    double d;        // payload: d'_i
};
__device__ __forceinline__ void aminp_take(ArgMinP &a, const ArgMinP &b) {
    if (b.k < a.k || (b.k == a.k && b.i < a.i)) a = b;
}
__device__ __forceinline__ void wave_argminp(ArgMinP &a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMinP b;
        b.k = __shfl_xor(a.k, o, 64); b.i = __shfl_xor(a.i, o, 64); b.u = __shfl_xor(a.u, o, 64); b.d = __shfl_xor(a.d, o, 64);
        aminp_take(a, b);
    }
}
__device__ __forceinline__ void block_argminp(ArgMinP &a, ArgMinP *sm) {
    wave_argminp(a);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sm[w] = a;
    __syncthreads();
    a = sm[0];
#pragma unroll
    for (int t = 1; t < kWavesPerBlock; t++) aminp_take(a, sm[t]);
    __syncthreads();
}

// ---- (key, index) candidates and DPP reductions for the single-workgroup kernels (bt_kernels.hip, lu_compressed.hip)
struct BtCand {   // (sort key, index): payloads (pivot element, x_B[p]) are published through LDS by the owner of the
    unsigned long long k;   // winning row after the reduction, so the reduction moves 3 dwords instead of 5
    unsigned int i;
};
__device__ __forceinline__ void bt_take(BtCand &a, const BtCand &b) {
    if (b.k < a.k || (b.k == a.k && b.i < a.i)) a = b;
}
// ---- DPP reductions.  A ds_bpermute-based __shfl_xor butterfly costs ~400 cycles per round (5 dwords through the
// LDS crossbar); the two workgroup-wide argmins per pivot were 2/3 of the inner kernel's time.  DPP row shifts are
// plain VALU moves: 4 row_shr steps leave each 16-lane row's result in its last lane, v_readlane combines the rows.
template <int CTRL>
__device__ __forceinline__ unsigned int dpp_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);  // lanes without a source keep v
}
template <int CTRL>
__device__ __forceinline__ BtCand dpp_cand(const BtCand &a) {
    BtCand b;
    const unsigned int klo = dpp_u32<CTRL>((unsigned int)a.k), khi = dpp_u32<CTRL>((unsigned int)(a.k >> 32));
    b.k = ((unsigned long long)khi << 32) | klo;
    b.i = dpp_u32<CTRL>(a.i);
    return b;
}
__device__ __forceinline__ BtCand readlane_cand(const BtCand &a, int lane) {
    BtCand b;
    const unsigned int klo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)a.k, lane);
    const unsigned int khi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(a.k >> 32), lane);
    b.k = ((unsigned long long)khi << 32) | klo;
    b.i = (unsigned int)__builtin_amdgcn_readlane((int)a.i, lane);
    return b;
}
// reduce within each row of 16 lanes: afterwards lane 15 of every row holds that row's argmin
__device__ __forceinline__ void row_argmin(BtCand &a) {
    bt_take(a, dpp_cand<0x111>(a));  // row_shr:1
    bt_take(a, dpp_cand<0x112>(a));  // row_shr:2
    bt_take(a, dpp_cand<0x114>(a));  // row_shr:4
    bt_take(a, dpp_cand<0x118>(a));  // row_shr:8
}
// argmin over the whole 1024-thread workgroup; result (uniform) in every thread.  `sm` is double buffered by the
// caller (sm + 16*parity) so that one barrier per reduction is enough.
template <int NW>
__device__ __forceinline__ void bt_block_argmin(BtCand &a, BtCand *sm) {
    row_argmin(a);
    BtCand w = readlane_cand(a, 15);
    bt_take(w, readlane_cand(a, 31));
    bt_take(w, readlane_cand(a, 47));
    bt_take(w, readlane_cand(a, 63));
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[wv] = w;
    __syncthreads();
    BtCand c;
    c.k = ~0ull; c.i = 0xFFFFFFFFu;
    if (lane < NW) c = sm[lane];
    row_argmin(c);  // NW <= 16: one row
    a = readlane_cand(c, 15);
}

// ---- value reductions without index payloads (bt_kernels.hip k_bt_inner2, lu_compressed.hip k_luc_panel)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = dpp_u32<CTRL>((unsigned int)b), hi = dpp_u32<CTRL>((unsigned int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)b, lane);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// v_min_f64 directly: minnum semantics in hardware (a NaN operand loses); __builtin_fmin would add a v_max_f64 x,x
// canonicalisation per operand, doubling the instruction count of the reductions
__device__ __forceinline__ double vmin_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// min over each 16-lane row (NaN operands lose: minnum)
// rotations (row_ror) have a source in every lane, so the DPP move needs no `old` operand (one v_mov less per
// dword and step than row_shr with old = own value); afterwards EVERY lane of a row holds the row's minimum
template <int CTRL>
__device__ __forceinline__ unsigned int ror_u32(unsigned int v) {
    return (unsigned int)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double ror_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = ror_u32<CTRL>((unsigned int)b), hi = ror_u32<CTRL>((unsigned int)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double row_min_f64(double x) {
    x = vmin_f64(x, ror_f64<0x121>(x));  // row_ror:1
    x = vmin_f64(x, ror_f64<0x122>(x));  // row_ror:2
    x = vmin_f64(x, ror_f64<0x124>(x));  // row_ror:4
    x = vmin_f64(x, ror_f64<0x128>(x));  // row_ror:8
    return x;
}
// rows combined with the gfx9 broadcast forms of DPP instead of 4 x v_readlane pairs: row_bcast15 hands lane 15 of a row to
// the next row (row_mask 0xA: rows 1 and 3 take it), row_bcast31 hands lane 31 to rows 2 and 3 (row_mask 0xC); lane 63 then
// holds the minimum of the wave — 6 VALU + 2 v_readlane instead of 19 instructions
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double bcast_f64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned int lo = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)b, (int)(unsigned int)b, CTRL, ROWMASK, 0xF, false);
    const unsigned int hi = (unsigned int)__builtin_amdgcn_update_dpp((int)(unsigned int)(b >> 32), (int)(unsigned int)(b >> 32), CTRL, ROWMASK, 0xF, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_min_f64(double x) {
    x = row_min_f64(x);                               // every lane of a row holds the row's minimum
    x = vmin_f64(x, bcast_f64<0x142, 0xA>(x));        // row_bcast:15 -> rows 1, 3: min(r0, r1), min(r2, r3)
    x = vmin_f64(x, bcast_f64<0x143, 0xC>(x));        // row_bcast:31 -> rows 2, 3: row 3 = min of the wave
    return readlane_f64(x, 63);
}
__device__ __forceinline__ unsigned int row_min_u32(unsigned int x) {
    x = min(x, ror_u32<0x121>(x));
    x = min(x, ror_u32<0x122>(x));
    x = min(x, ror_u32<0x124>(x));
    x = min(x, ror_u32<0x128>(x));
    return x;
}

// x / d for d > 0 and operands in the normal range (|x| / d far from overflow and underflow: ratios of the ratio test):
// reciprocal + two Newton steps + one residual correction — the instruction sequence of the compiler's IEEE division without
// its scaling and fix-up stages (8 instead of 13 instructions), the same correctly rounded quotient where those stages are idle
__device__ __forceinline__ double div_pos(double x, double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    double q = x * r;
    const double rem = __builtin_fma(-d, q, x);
    return __builtin_fma(rem, r, q);
}

// dot of one padded row (ld doubles, 16-byte aligned) with the LDS-staged vector; result in all lanes
__device__ __forceinline__ double wave_dot_row(const double *__restrict__ row, const double2 *__restrict__ svec,
                                               int ld2, int lane) {
    const double2 *r2 = reinterpret_cast<const double2 *>(row);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int c = lane;
    for (; c + 192 < ld2; c += 256) {
        double2 v0 = r2[c], v1 = r2[c + 64], v2 = r2[c + 128], v3 = r2[c + 192];
        double2 s0 = svec[c], s1 = svec[c + 64], s2 = svec[c + 128], s3 = svec[c + 192];
        a0 += v0.x * s0.x + v0.y * s0.y;
        a1 += v1.x * s1.x + v1.y * s1.y;
        a2 += v2.x * s2.x + v2.y * s2.y;
        a3 += v3.x * s3.x + v3.y * s3.y;
    }
    for (; c < ld2; c += 64) {
        double2 v0 = r2[c];
        double2 s0 = svec[c];
        a0 += v0.x * s0.x + v0.y * s0.y;
    }
    return wave_sum((a0 + a1) + (a2 + a3));
}

// element (i, j) of the tableau T in either layout: row-major, or the 4x4 tiles of the blocked pipeline (bt_kernels.hip:
// tile (I, J) at ((I * ldt/4) + J) * 16, element (i&3)*4 + (j&3))
__device__ __forceinline__ size_t tab_idx(int i, int j, int ldt, int tiled) {
    return tiled ? ((size_t)(i >> 2) * (size_t)(ldt >> 2) + (size_t)(j >> 2)) * 16u + (size_t)(((i & 3) << 2) + (j & 3))
                 : (size_t)i * ldt + j;
}

__device__ __forceinline__ void stage_vec(double2 *__restrict__ svec, const double *__restrict__ src, int ld2) {
    const double2 *s2 = reinterpret_cast<const double2 *>(src);
    for (int c = threadIdx.x; c < ld2; c += kBlock) svec[c] = s2[c];
    __syncthreads();
}

}  // namespace gomilp
