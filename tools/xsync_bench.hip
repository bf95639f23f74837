// Micro-benchmark: cost of one cross-workgroup exchange round (every workgroup posts a 128-byte line, every workgroup reads
// all lines) between G workgroups of one launch, through L2 — the synchronisation a multi-workgroup block kernel needs twice
// per pivot.   hipcc --offload-arch=gfx950 -O3 -o /tmp/xsync tools/xsync_bench.hip && /tmp/xsync
// Variants: same XCD (blocks b with b % 8 == 0) or neighbouring blocks (different XCDs); cache policy of the loads / stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int POL> __device__ __forceinline__ double ld_pol(const double *p) {
    double v;
    if (POL == 0) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 1) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 3) asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 4) asm volatile("buffer_inv sc1\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 5) asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (POL == 6 || POL == 8 || POL == 9) asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else { unsigned long long z = 0, r; asm volatile("global_atomic_or_x2 %0, %1, %2, off sc0\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(z) : "memory"); v = __longlong_as_double((long long)r); }
    return v;
}
template <int POL> __device__ __forceinline__ void st_pol(double *p, double v) {
    if (POL == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else if (POL == 9) asm volatile("global_store_dwordx2 %0, %1, off nt" : : "v"(p), "v"(v) : "memory");
    else if (POL == 0 || POL >= 3) asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory");
    else if (POL == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
}

// lines: [2 parities][G][16 doubles]
template <int POL, bool ALLWAVES>
__global__ __launch_bounds__(512) void k_xsync(double *lines, int G, int stride, int rounds, long long *cycles, int *err, double *sink, int *bad, int *xcc) {
    if (blockIdx.x % stride != 0) return;
    const int g = blockIdx.x / stride;
    if (g >= G) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) { unsigned int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); xcc[g] = (int)x; }
    int nbad = 0;
    __shared__ double s_res[2][8 * 16];
    __shared__ int s_err[2];
    double acc = 0;
    long long t0 = 0;
    if (tid < 2) s_err[tid] = 0;
    __syncthreads();
    for (int r = 1; r <= rounds; r++) {
        if (r == 9) t0 = __builtin_readcyclecounter();
        double *mine = lines + ((size_t)(r & 1) * G + g) * 16;
        if (wave == 0 && lane < 16) {
            double v = (lane == 0 || lane == 15 || lane == 7 || lane == 8) ? (double)r : (double)(g * 1000 + lane) + (double)r * 0.5;
            st_pol<POL>(mine + lane, v);
        }
        const double *all = lines + (size_t)(r & 1) * G * 16;
        if (ALLWAVES || wave == 0) {
            int spins = 0;
            double v0 = 0, v1 = 0;
            for (;;) {
                bool ok = true;
                if (lane < G * 16) { v0 = ld_pol<POL>(all + lane); const int w = lane & 15; if ((w == 0 || w == 15 || w == 7 || w == 8) && v0 != (double)r) ok = false; }
                if (G * 16 > 64 && lane + 64 < G * 16) { v1 = ld_pol<POL>(all + lane + 64); const int w = lane & 15; if ((w == 0 || w == 15 || w == 7 || w == 8) && v1 != (double)r) ok = false; }
                if (__all(ok)) {
                    const int w = lane & 15; const bool sq = (w == 0 || w == 15 || w == 7 || w == 8);
                    if (lane < G * 16 && !sq && v0 != (double)((lane >> 4) * 1000 + w) + (double)r * 0.5) nbad++;
                    if (G * 16 > 64 && lane + 64 < G * 16 && !sq && v1 != (double)(((lane + 64) >> 4) * 1000 + w) + (double)r * 0.5) nbad++;
                    break;
                }
                if (++spins > 2000000) { if (lane == 0) { atomicExch(err, r); s_err[r & 1] = 1; } break; }
            }
            if (!ALLWAVES) { if (lane < G * 16) s_res[r & 1][lane] = v0; if (G * 16 > 64 && lane + 64 < G * 16) s_res[r & 1][lane + 64] = v1; }
            else acc += v0 + v1;
        }
        if (!ALLWAVES) { __syncthreads(); acc += s_res[r & 1][(tid & 15) + 16 * ((tid >> 4) % G)]; }
        if (ALLWAVES ? (*((volatile int *)err) != 0) : (s_err[r & 1] != 0)) break;
    }
    long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cycles[g] = t1 - t0;
    if (nbad) atomicAdd(bad, nbad);
    sink[blockIdx.x * 512 + tid] = acc;
}

template <int POL, bool ALLWAVES> void run(const char *name, int G, int stride, int rounds) {
    double *lines, *sink; long long *cycles; int *err, *bad, *xcc;
    const int nblocks = G * stride;
    CK(hipMalloc(&lines, 2 * 8 * 16 * sizeof(double))); CK(hipMemset(lines, 0, 2 * 8 * 16 * sizeof(double)));
    CK(hipMalloc(&sink, (size_t)nblocks * 512 * sizeof(double)));
    CK(hipMalloc(&cycles, 8 * sizeof(long long))); CK(hipMalloc(&err, sizeof(int))); CK(hipMemset(err, 0, sizeof(int))); CK(hipMalloc(&bad, sizeof(int))); CK(hipMemset(bad, 0, sizeof(int))); CK(hipMalloc(&xcc, 8 * sizeof(int))); CK(hipMemset(xcc, 0xff, 8 * sizeof(int)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_xsync<POL, ALLWAVES>), dim3(nblocks), dim3(512), 0, 0, lines, G, stride, rounds, cycles, err, sink, bad, xcc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h[8]; int herr; CK(hipMemcpy(h, cycles, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    int hbad, hx[8]; CK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost)); CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
    printf("xcc=[%d %d %d %d %d %d %d %d] bad=%d ", hx[0], hx[1], hx[2], hx[3], hx[4], hx[5], hx[6], hx[7], hbad);
    printf("%-28s G=%d stride=%d rounds=%d: %.3f us/round (event), %.0f ticks/round (readcyclecounter, wg0)%s\n", name, G, stride, rounds, ms * 1e3 / rounds,
           (double)h[0] / (rounds - 8), herr ? "  ** TIMED OUT **" : "");
    fflush(stdout);
    CK(hipFree(lines)); CK(hipFree(sink)); CK(hipFree(cycles)); CK(hipFree(err));
}

// placement test: the same exchange with its record lines at different offsets of one large allocation
template <int POL> void place(int G, int rounds) {
    const size_t span = (size_t)64 << 20;
    char *big; double *sink; long long *cycles; int *err, *bad, *xcc;
    CK(hipMalloc(&big, span)); CK(hipMemset(big, 0, span));
    CK(hipMalloc(&sink, (size_t)G * 8 * 512 * sizeof(double)));
    CK(hipMalloc(&cycles, 8 * sizeof(long long))); CK(hipMalloc(&err, sizeof(int))); CK(hipMalloc(&bad, sizeof(int))); CK(hipMalloc(&xcc, 8 * sizeof(int)));
    for (size_t off : {(size_t)0, (size_t)4096, (size_t)8192, (size_t)16384, (size_t)65536, (size_t)1 << 20, (size_t)2 << 20, (size_t)3 << 20, (size_t)5 << 20, (size_t)8 << 20,
                       (size_t)13 << 20, (size_t)21 << 20, (size_t)34 << 20, (size_t)55 << 20, ((size_t)55 << 20) + 4096 * 3, ((size_t)21 << 20) + 4096 * 7}) {
        CK(hipMemset(err, 0, sizeof(int))); CK(hipMemset(bad, 0, sizeof(int)));
        hipLaunchKernelGGL((k_xsync<POL, false>), dim3(G * 8), dim3(512), 0, 0, reinterpret_cast<double *>(big + off), G, 8, rounds, cycles, err, sink, bad, xcc);
        CK(hipDeviceSynchronize());
        long long h[8]; CK(hipMemcpy(h, cycles, sizeof(h), hipMemcpyDeviceToHost));
        printf("POL %d G=%d offset %9zu: %.0f ticks/round\n", POL, G, off, (double)h[0] / (rounds - 8));
        fflush(stdout);
    }
    CK(hipFree(big)); CK(hipFree(sink)); CK(hipFree(cycles)); CK(hipFree(err)); CK(hipFree(bad)); CK(hipFree(xcc));
}

int main(int argc, char **argv) {
    if (argc > 1) { place<1>(8, 20000); place<6>(8, 20000); return 0; }

    const int R = 20000;
    for (int G : {2, 4, 8}) {
        run<1, false>("sc1/sc1 sameXCD", G, 8, R);
        run<6, false>("ld_nt/st sameXCD", G, 8, R);
        run<8, false>("ld_nt/st_sc1 sameXCD", G, 8, R);
        run<9, false>("ld_nt/st_nt sameXCD", G, 8, R);
        run<1, false>("sc1/sc1 crossXCD", G, 1, R);
        run<6, false>("ld_nt/st crossXCD", G, 1, R);
        run<8, false>("ld_nt/st_sc1 crossXCD", G, 1, R);
        run<9, false>("ld_nt/st_nt crossXCD", G, 1, R);
    }
    return 0;
}
