// swaplat.hip -- what does a register/lane swap cost on gfx950?  Cycles (s_memtime) per v_permlane32_swap / v_permlane16_swap
// when every swap depends on the previous one (latency) and when eight independent pairs are in flight (throughput), next to a
// dependent v_fma_f64 and a ds_write_b64 + ds_read_b64 round trip; one wave per SIMD and two.
//   hipcc -O3 --offload-arch=gfx950 -o swaplat swaplat.hip && ./swaplat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int ITERS = 4096;

template <int MODE>
__global__ void __launch_bounds__(512) probe(unsigned long long* out, double seed) {
    __shared__ double lds[64 * 8 * 2];
    unsigned a[8], b[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + i + 1; }
    double x = seed + threadIdx.x, y = 1.0000001;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {            // dependent chain of permlane32 swaps (each feeds the next)
#pragma unroll
            for (int k = 0; k < 8; k++) { auto r = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false); a[0] = r[0] + 1; b[0] = r[1]; }
        } else if (MODE == 1) {     // eight independent pairs
#pragma unroll
            for (int k = 0; k < 8; k++) { auto r = __builtin_amdgcn_permlane32_swap(a[k], b[k], false, false); a[k] = r[0]; b[k] = r[1]; }
        } else if (MODE == 2) {     // dependent chain of permlane16 swaps
#pragma unroll
            for (int k = 0; k < 8; k++) { auto r = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false); a[0] = r[0] + 1; b[0] = r[1]; }
        } else if (MODE == 3) {     // dependent f64 FMAs
#pragma unroll
            for (int k = 0; k < 8; k++) x = __builtin_fma(x, y, 0.5);
        } else if (MODE == 4) {     // LDS round trip: write 8 bytes, wait, read another lane's
#pragma unroll
            for (int k = 0; k < 8; k++) {
                lds[threadIdx.x] = x;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                x = lds[threadIdx.x ^ 17] + 1.0;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    unsigned acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i] ^ b[i];
    if (threadIdx.x % 64 == 0) out[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
    if (acc == 0x12345 && x == 3.0) out[0] = 0;
}

template <int MODE>
static void run(const char* what, int waves_per_simd) {
    unsigned long long* d;
    const int threads = 256 * waves_per_simd;       // 4 or 8 waves on one CU's four SIMDs
    hipMalloc(&d, 256 * 8 * 8);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.5);
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, d, 1.5);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < 256; b++) for (int w = 0; w < threads / 64; w++) { sum += (double)h[b * 8 + w]; n++; }
    printf("  %-58s %d wave(s) per SIMD: %7.1f cycles per operation and wave\n", what, waves_per_simd, sum / n / (ITERS * 8.0));
    hipFree(d);
}

int main() {
    for (int w = 1; w <= 2; w++) {
        run<0>("v_permlane32_swap, each depending on the previous", w);
        run<1>("v_permlane32_swap, eight independent pairs", w);
        run<2>("v_permlane16_swap, each depending on the previous", w);
        run<3>("v_fma_f64, each depending on the previous", w);
        run<4>("ds_write_b64 + wait + ds_read_b64 + wait (round trip)", w);
    }
    return 0;
}
