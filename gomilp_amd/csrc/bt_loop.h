// Protocol of the persistent loop kernels (btg_kernels.hip k_bt_loop: one large relaxation; bt_kernels.hip k_b_loop: the long chains of
// a device-batched wave): exchange-buffer layout, agent-scope accesses, bounded waits on arrival counters, and the UPDATE ROLE — the
// workgroups that apply T_next = T + sum_k u_k v'_k^T of block t on the matrix cores beside the pivots of block t + 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_types.h"
#include "kernels_common.h"

namespace gomilp {

constexpr int kXSlots = 8;             // slots per record (one 128-byte line): min, first index, 3 scalars of the winner
constexpr int kXHeader = 16;           // doubles in front of the records: [0] = exchanges completed so far
constexpr int kXSpinLimit = 2000000;   // polls (~0.5 us each) before a launch gives up
// Persistent loop kernel (k_bt_loop): header [1 + 5 par + j] = arrivals expected before the first block of the launch with parity
// par on counter j below (written by workgroup 0 of the previous launch: a launch never rewrites what its own late starters
// still have to read); the counters sit on lines of their own.  Update arrivals are counted per block index mod 4: with ONE
// total a fast update workgroup's arrival for block t + 1 could stand in for a slow one's missing arrival for block t
// behind the records: blocks finished by the pivot workgroups (G arrivals per block), blocks applied by the update
// workgroups (one arrival per update workgroup and block)
constexpr int kXSync = kXHeader + 2 * 16 * kXSlots * 2;   // doubles in front of the counters (records of up to 16 workgroups, two parities)
constexpr int kXSyncDoubles = 16 * 5;   // the block counter and four update counters (blocks = j mod 4), a line each
constexpr int kLoopSpinLimit = 600000;   // polls (~1.5 us each) of a block / update counter before a workgroup gives up
constexpr int kBLoopSpinLimit = 100000;  // the same in the batched loop kernel (k_b_loop): giving up there costs one relaxation a re-solve on a worker — milliseconds, not a failed solve — so a wave that lost the device to somebody else should not sit out the best part of a second (round-4 advisory)

typedef double xpair __attribute__((ext_vector_type(2)));   // {sequence number, value}

__device__ __forceinline__ unsigned int tile_off_g(unsigned int i, unsigned int j, unsigned int ldt) {
    return ((i >> 2) * (ldt >> 2) + (j >> 2)) * 16u + ((i & 3u) << 2) + (j & 3u);
}
// Two forms of the record accesses.  SAFE: sc1 (agent scope: written through to / read from the memory side — correct
// wherever the workgroups run, but the round trip depends on which HBM stack the line lives in: 1950-2630 cycles per
// exchange over 16 placements in tools/xsync_bench.hip, which showed up as 68 vs 78 us per launch from one process to the
// next).  FAST, used once the workgroups have seen that they share one XCD: plain stores (write-through L1 -> that XCD's
// L2, the coherence point of its CUs) and nt loads (never served by L1) — 1860-1920 cycles at every placement.
// (the strings end in s_nop 1: the compiler pads nothing after an asm statement, and its next instruction may overwrite the data
// registers of a 16-byte store before the store has read them)
__device__ __forceinline__ void xstore(xpair *p, xpair v, bool fast) {
    if (fast) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <int H> struct XLoad;
template <> struct XLoad<1> {
    static __device__ __forceinline__ void run(const xpair *p, xpair (&v)[1], bool fast) {
        if (fast) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(p) : "memory");
    }
};
template <> struct XLoad<2> {   // two slots per lane (16 workgroups), one wait
    static __device__ __forceinline__ void run(const xpair *p0, const xpair *p1, xpair (&v)[2], bool fast) {
        if (fast) asm volatile("global_load_dwordx4 %0, %2, off nt\n\tglobal_load_dwordx4 %1, %3, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(p0), "v"(p1) : "memory");
        else asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]), "=&v"(v[1]) : "v"(p0), "v"(p1) : "memory");
    }
};
__device__ __forceinline__ double ld_agent(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(double *p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until a monotonic arrival counter has reached `target` (wrap-safe); false: no progress within the limit
__device__ __forceinline__ bool spin_counter(const unsigned int *p, unsigned int target, int sleep_ticks, int limit = kLoopSpinLimit) {
    for (int it = 0; it < limit; it++) {
        const unsigned int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) return true;
        if (sleep_ticks > 8) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

// ---- persistent loop kernel ----------------------------------------------------------------------------------------------------
// Update role: workgroup u of nupd applies T_next = T + sum_k u_k v'_k^T for block after block of the SAME launch, each as soon as
// the G pivot workgroups have arrived at the end of that block, on the matrix cores (a 16 x 16 block of the 4x4-tiled tableau is the
// C/D operand of v_mfma_f64_16x16x4_f64, two MFMAs for the 8 terms of a block: bt_kernels.hip k_bt_update_mfma16 has the layout).
// Everything that crosses workgroups inside the launch — the terms, the pivot count, both tableau buffers — moves with agent-scope
// accesses; a workgroup always owns the same units of the tableau, so it only ever re-reads what it wrote itself.
typedef double btg_d4 __attribute__((ext_vector_type(4)));
typedef double btg_d2 __attribute__((ext_vector_type(2)));
typedef unsigned int btg_u4 __attribute__((ext_vector_type(4)));
// 16-byte accesses: the column index n of an MFMA block may stand for any 16 tableau columns as long as the B operand uses the
// same map, so two blocks X / Y cover 32 columns INTERLEAVED (X: c0 + 2n, Y: c0 + 2n + 1): a lane's X and Y entries of a row are
// neighbours inside one 4x4 tile, one 16-byte agent-scope load / store instead of two of 8 bytes (8-byte agent-scope accesses run
// at 0.54-0.70 of the 16-byte rate).  One unit of a wave = 16 rows x 64 columns: eight 16-byte tableau loads per lane in flight.
// FLAT: the 16 x 64 wave units of a block are dealt over ALL waves of the update workgroups (small tableaus with few column units:
// the batched loop kernel); otherwise a workgroup takes a 16-row strip of NWV neighbouring column units (large tableaus: k_bt_loop)
template <int NT, int KB, bool FLAT = false>
__device__ __forceinline__ void bt_loop_update_role(const BTArgs &a, const int u, const int nupd, const int G) {
    constexpr int NWV = NT / 64;
    DevState *st = a.st;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (no look at `done` here: a launch behind the end of the loop is released by the pivot workgroups' arrival for block 0,
    // and a loop that ends in block 0 of THIS launch still has that block's terms to apply)
    __shared__ int s_go;
    const int sel0 = a.par ? st->tsel2[1] : st->tsel2[0];
    const unsigned int blk_base = (unsigned int)(unsigned long long)a.xbuf[1 + 5 * a.par];
    unsigned int *blk_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync), *upd_cnt = reinterpret_cast<unsigned int *>(a.xbuf + kXSync + 16);
    const int l15 = lane & 15, l4 = lane >> 4;
    const int ncp = a.ldt >> 6;                      // units of 64 columns per row strip (ldt is a multiple of 512)
    const int groups = (ncp + NWV - 1) / NWV, strips = (a.m + 15) >> 4, nunits = groups * strips;
    const int ntr = (a.m + 3) >> 2;                  // tile rows that exist
    const size_t trow = (size_t)(a.ldt >> 2) * 16;   // doubles per tile row
    for (int blk = 0; blk < a.nblocks; blk++) {
        if (tid == 0) s_go = spin_counter(blk_cnt, blk_base + (unsigned int)G * (unsigned int)(blk + 1), 32, FLAT ? kBLoopSpinLimit : kLoopSpinLimit) ? 1 : 0;   // (FLAT: the batched loop kernel)
        __syncthreads();
        if (!s_go) return;   // the pivot workgroups never arrived: give up (they report the failure, or nobody is left to)
        const int kd = __hip_atomic_load(&st->kdone2[blk & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int dn = __hip_atomic_load(&st->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kd > 0) {
            const double *src = ((sel0 ^ blk) & 1) ? a.Tbuf[1] : a.Tbuf[0];
            double *dst = ((sel0 ^ blk) & 1) ? a.Tbuf[0] : a.Tbuf[1];
            const int k0 = (blk & 1) * KB;
            const double *Ub = a.U + (size_t)k0 * a.ldu, *Vb = a.V + (size_t)k0 * a.ldt;
            const int tbytes = (int)((size_t)ntr * trow * 8);
            const auto rs_src = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(src), 0, tbytes, 0x00020000);
            const auto rs_dst = __builtin_amdgcn_make_buffer_rsrc(dst, 0, tbytes, 0x00020000);
            const auto rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(Vb), 0, (int)((size_t)KB * a.ldt * 8), 0x00020000);
            auto do_unit = [&](const int strip, const int cp) {
                const int c0 = cp * 64;
                const int row = strip * 16 + l15, rowc = row < a.m ? row : a.m - 1;
                // element (row mrow + 4 r of the strip, columns c0 + 32 x + 2 n, + 1): tile row strip * 4 + r, row l4 inside the tile
                const size_t boff = (size_t)(strip * 4) * trow + (size_t)((c0 + 2 * l15) >> 2) * 16 + l4 * 4 + ((2 * l15) & 3);
                // 16-byte agent-scope (sc1, aux 16) buffer loads / stores: instructions the compiler knows, so it counts them and
                // places the wait states around the matrix instructions itself (an inline-asm global_store_dwordx4 here lost
                // data: the next instruction may overwrite a wide store's data registers before the store has read them)
                constexpr int NS = KB / 4;   // matrix instructions (4 terms each) per 16 x 16 block
                btg_d2 c[8], bv[2 * NS];
                double av[NS];
#pragma unroll
                for (int x = 0; x < 2; x++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const size_t off = boff + (size_t)(strip * 4 + r < ntr ? r : 0) * trow + (size_t)x * 128;   // (32 columns = 8 tiles on)
                        c[x * 4 + r] = __builtin_bit_cast(btg_d2, __builtin_amdgcn_raw_buffer_load_b128(rs_src, (int)(off * 8), 0, 16));
                    }
                }
#pragma unroll
                for (int t = 0; t < 2 * NS; t++) {   // bv[x * NS + s2]: terms 4 s2 + l4 for the column half x
                    const size_t off = (size_t)((t % NS) * 4 + l4) * a.ldt + c0 + (t / NS) * 32 + 2 * l15;
                    bv[t] = __builtin_bit_cast(btg_d2, __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)(off * 8), 0, 16));
                }
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++) av[s2] = ld_agent(Ub + (size_t)(4 * s2 + l4) * a.ldu + rowc);
                // rows of U / V beyond the pivots of this block are stale, rows beyond m do not exist
#pragma unroll
                for (int s2 = 0; s2 < NS; s2++) {
                    const bool kon = 4 * s2 + l4 < kd;
                    if (!kon || row >= a.m) av[s2] = 0.0;
                    if (!kon) { bv[s2] = btg_d2{0.0, 0.0}; bv[NS + s2] = btg_d2{0.0, 0.0}; }
                }
                btg_d4 cx[2], cy[2];
#pragma unroll
                for (int x = 0; x < 2; x++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) { cx[x][r] = c[x * 4 + r][0]; cy[x][r] = c[x * 4 + r][1]; }
#pragma unroll
                    for (int s2 = 0; s2 < NS; s2++) {
                        cx[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[x * NS + s2][0], cx[x], 0, 0, 0);
                        cy[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s2], bv[x * NS + s2][1], cy[x], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int x = 0; x < 2; x++) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if (strip * 4 + r < ntr) {
                            const btg_d2 o = {cx[x][r], cy[x][r]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(btg_u4, o), rs_dst, (int)((boff + (size_t)r * trow + (size_t)x * 128) * 8), 0, 16);
                        }
                    }
                }
            };
            if constexpr (FLAT) {
                const int nwu = strips * ncp;
                for (int wu = u * NWV + wv; wu < nwu; wu += nupd * NWV) do_unit(wu / ncp, wu % ncp);
            } else {
                for (int unit = u; unit < nunits; unit += nupd) {
                    const int cp = (unit % groups) * NWV + wv;
                    if (cp < ncp) do_unit(unit / groups, cp);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's tableau stores have landed
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(upd_cnt + 16 * (blk & 3) * 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (unsigned ints: 32 per line)
        if (dn) return;
    }
}

}  // namespace gomilp
