// clockrate.hip -- effective shader clock over time, measured inside a kernel: s_memtime (shader cycles) against
// s_memrealtime (constant 100 MHz) per chunk of dependent FMAs, on every CU, (a) right after a busy period, (b) after
// the GPU sat nearly idle (one tiny workgroup spinning) for 10 ms.  Does the part run slower for tens of ms after idle
// although amdsmi reports an unchanged clock (profiles/r03_after_idle.txt)?
//   hipcc -O3 --offload-arch=gfx950 -o clockrate clockrate.hip && ./clockrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>

constexpr int CHUNKS = 64;

__global__ void __launch_bounds__(256) burn(unsigned long long* out, int iters, int record) {
    double x = threadIdx.x * 1e-3, y = 1.0000001;
    for (int c = 0; c < CHUNKS; c++) {
        unsigned long long t0, r0, t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
#pragma unroll 1
        for (int i = 0; i < iters; i++) { x = __builtin_fma(x, y, 0.5); x = __builtin_fma(x, y, -0.5); }
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        if (record && threadIdx.x == 0 && blockIdx.x < 256) {
            out[(blockIdx.x * CHUNKS + c) * 2] = t1 - t0;
            out[(blockIdx.x * CHUNKS + c) * 2 + 1] = r1 - r0;
        }
    }
    if (x == 1234.5) out[0] = 0;
}

static void report(const char* what, const std::vector<unsigned long long>& h) {
    printf("%s: effective shader clock per chunk, median over 256 workgroups (MHz):\n  ", what);
    for (int c = 0; c < CHUNKS; c += 4) {
        std::vector<double> f;
        for (int b = 0; b < 256; b++) {
            const double cyc = (double)h[(b * CHUNKS + c) * 2], ticks = (double)h[(b * CHUNKS + c) * 2 + 1];
            if (ticks > 0) f.push_back(cyc / ticks * 100.0);
        }
        std::sort(f.begin(), f.end());
        printf("%.0f ", f.empty() ? 0.0 : f[f.size() / 2]);
    }
    double tot_ticks = 0;
    for (int c = 0; c < CHUNKS; c++) tot_ticks += (double)h[c * 2 + 1];
    printf("\n  workgroup 0 took %.2f ms\n", tot_ticks / 1e5);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * CHUNKS * 2 * 8);
    std::vector<unsigned long long> h(256 * CHUNKS * 2);
    const int iters = 20000;                     // ~ 64 chunks x 0.3 ms = 20 ms per launch, 8 waves per CU
    for (int rep = 0; rep < 2; rep++) {
        // (a) busy: three launches back to back, the third recorded
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(burn, dim3(512), dim3(256), 0, 0, d, iters, i == 2);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        report("after a busy period", h);
        // (b) nearly idle for ~ 10 ms (one workgroup), then the recorded launch
        hipLaunchKernelGGL(burn, dim3(1), dim3(64), 0, 0, d, iters / 2, 0);
        hipLaunchKernelGGL(burn, dim3(512), dim3(256), 0, 0, d, iters, 1);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        report("after 10 ms of a nearly idle GPU", h);
    }
    return 0;
}
