// Instruction-throughput microbenchmarks for gfx950: how many cycles one SIMD needs per wave64
// instruction for the f64 VALU ops and LDS accesses the blind-rotation loop is made of, alone and
// mixed.  Build: hipcc -O2 --offload-arch=gfx950 ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define HIP_OK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 65536;
constexpr int UNROLL = 16;

enum Op { ADD_F64, MUL_F64, FMA_F64, RNDNE_F64, FLOOR_F64, LDEXP_F64, CVT_I32_F64, CVT_F64_I32, CVT_U32_F64,
          ADD_U32, LSHL_ADD_U64, MAD_U64_U32, CNDMASK, SUBB_PAIR,
          DS_WRITE_B64, DS_READ_B64, DS_WRITE_B128, DS_READ_B128, DS_WRITE2ST64, DS_RW_B64,
          MIX_FMA4_WRITE1, MIX_FMA8_WRITE1, MIX_FMA4_READ1, MIX_FMA4_RW,
          PERMLANE32_SWAP, PERMLANE16_SWAP, MOV_DPP_QUAD, MOV_DPP_ROW_SHR, DS_SWIZZLE, DS_BPERMUTE, ROLE_FMA_ONLY, ROLE_WRITE_ONLY, ROLE_SPLIT, N_OPS };
static const char* NAMES[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_rndne_f64", "v_floor_f64", "v_ldexp_f64",
    "v_cvt_i32_f64", "v_cvt_f64_i32", "v_cvt_u32_f64", "v_add_u32", "v_lshl_add_u64", "v_mad_u64_u32", "v_cndmask_b32",
    "v_sub_co+v_subb_co", "ds_write_b64", "ds_read_b64", "ds_write_b128", "ds_read_b128", "ds_write2st64_b64",
    "ds_write_b64+ds_read_b64", "4 fma : 1 ds_write_b64", "8 fma : 1 ds_write_b64", "4 fma : 1 ds_read_b64",
    "4 fma : 1 write + 1 read", "v_permlane32_swap_b32", "v_permlane16_swap_b32", "v_mov_b32_dpp quad_perm", "v_mov_b32_dpp row_shr:4",
    "ds_swizzle_b32", "ds_bpermute_b32", "role: even waves 16 fma, odd idle", "role: even idle, odd 4 ds_write_b64", "role: even 16 fma | odd 4 ds_write_b64"};

template <int OP>
__global__ void __launch_bounds__(1024) bench(unsigned long long* cycles, double* sink, int dummy) {
    extern __shared__ double lds[];
    double x[UNROLL];
    unsigned int u[UNROLL];
    unsigned long long w[UNROLL];
    const double c = 1.0000001 + dummy, d = 0.5 + dummy;
    for (int i = 0; i < UNROLL; ++i) { x[i] = threadIdx.x + i; u[i] = threadIdx.x * 3 + i; w[i] = u[i]; }
    // conflict-free LDS addresses: consecutive lanes -> consecutive 8-byte (or 16-byte) words
    const unsigned a8 = threadIdx.x * 8, a16 = threadIdx.x * 16;
    const unsigned span8 = blockDim.x * 8, span16 = blockDim.x * 16;
    typedef double d2_t __attribute__((ext_vector_type(2)));
    d2_t v2[4];
    for (int i = 0; i < 4; ++i) { v2[i].x = i; v2[i].y = i + 1; }
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
            if (OP == RNDNE_F64) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
            if (OP == FLOOR_F64) asm volatile("v_floor_f64 %0, %0" : "+v"(x[i]));
            if (OP == LDEXP_F64) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(dummy));
            if (OP == CVT_I32_F64) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(x[i]));
            if (OP == CVT_U32_F64) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(u[i]) : "v"(x[i]));
            if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[i]) : "v"(u[i]));
            if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(dummy));
            if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"(w[(i + 1) % UNROLL]));
            if (OP == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, s[4:5], %1, %2, %0" : "+v"(w[i]) : "v"(u[i]), "v"(dummy) : "s4", "s5");
            if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(dummy) : "vcc");
            if (OP == SUBB_PAIR) asm volatile("v_sub_co_u32 %0, vcc, %0, %2\n v_subb_co_u32 %1, vcc, %1, %2, vcc" : "+v"(u[i]), "+v"(u[(i + 1) % UNROLL]) : "v"(dummy) : "vcc");
            if (OP == DS_WRITE_B64) asm volatile("ds_write_b64 %0, %1" :: "v"(a8 + (i & 3) * span8), "v"(x[i]) : "memory");
            if (OP == DS_READ_B64) asm volatile("ds_read_b64 %0, %1" : "=v"(x[i]) : "v"(a8 + (i & 3) * span8) : "memory");
            if (OP == DS_WRITE_B128) asm volatile("ds_write_b128 %0, %1" :: "v"(a16 + (i & 1) * span16), "v"(v2[i & 3]) : "memory");
            if (OP == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(v2[i & 3]) : "v"(a16 + (i & 1) * span16) : "memory");
            if (OP == DS_WRITE2ST64) asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:8" :: "v"(a8 & 4095), "v"(x[i]), "v"(x[(i + 1) % UNROLL]) : "memory");
            if (OP == DS_RW_B64) {
                if (i & 1) asm volatile("ds_write_b64 %0, %1" :: "v"(a8 + (i & 3) * span8), "v"(x[i]) : "memory");
                else asm volatile("ds_read_b64 %0, %1" : "=v"(x[i]) : "v"(a8 + (i & 3) * span8) : "memory");
            }
            if (OP == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % UNROLL]));
            if (OP == PERMLANE16_SWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 1) % UNROLL]));
            if (OP == MOV_DPP_QUAD) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(u[i]) : "v"(u[(i + 3) % UNROLL]));
            if (OP == MOV_DPP_ROW_SHR) asm volatile("v_mov_b32_dpp %0, %1 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 3) % UNROLL]));
            if (OP == DS_SWIZZLE) asm volatile("ds_swizzle_b32 %0, %1 offset:swizzle(BITMASK_PERM, \"01pip\")" : "=v"(u[i]) : "v"(u[(i + 3) % UNROLL]) : "memory");
            if (OP == DS_BPERMUTE) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(u[i]) : "v"(a8 & 255), "v"(u[(i + 3) % UNROLL]) : "memory");
            if (OP == ROLE_FMA_ONLY || OP == ROLE_WRITE_ONLY || OP == ROLE_SPLIT) {
                const bool odd = (threadIdx.x >> 6) & 1;      // wave-uniform
                if (!odd && OP != ROLE_WRITE_ONLY) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
                if (odd && OP != ROLE_FMA_ONLY && (i & 3) == 3) asm volatile("ds_write_b64 %0, %1" :: "v"(a8 + ((i >> 2) & 3) * span8), "v"(x[i]) : "memory");
            }
            if (OP == MIX_FMA4_WRITE1 || OP == MIX_FMA8_WRITE1 || OP == MIX_FMA4_READ1 || OP == MIX_FMA4_RW) {
                const int period = (OP == MIX_FMA8_WRITE1) ? 8 : 4;
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "v"(d));
                if (i % period == period - 1) {
                    if (OP == MIX_FMA4_READ1)
                        asm volatile("ds_read_b64 %0, %1" : "=v"(w[i]) : "v"(a8 + (i & 3) * span8) : "memory");
                    else
                        asm volatile("ds_write_b64 %0, %1" :: "v"(a8 + ((i / period) & 3) * span8), "v"(x[(i + 5) % UNROLL]) : "memory");
                    if (OP == MIX_FMA4_RW)
                        asm volatile("ds_read_b64 %0, %1" : "=v"(w[i]) : "v"(a8 + ((i / period + 2) & 3) * span8) : "memory");
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < UNROLL; ++i) s += x[i] + u[i] + (double)w[i];
    for (int i = 0; i < 4; ++i) s += v2[i].x + v2[i].y;
    if (s == 12345.678) sink[0] = s + lds[threadIdx.x];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

__global__ void swap_semantics(unsigned* out) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 4 + 0] = r[0]; out[threadIdx.x * 4 + 1] = r[1];
    out[threadIdx.x * 4 + 2] = q[0]; out[threadIdx.x * 4 + 3] = q[1];
}

template <int OP>
void run(int waves_per_simd, unsigned long long* d_cycles, double* d_sink, int blocks) {
    const int threads = 256 * waves_per_simd;
    const size_t lds = (size_t)threads * 16 * 4;   // 4 slabs of 16 B per thread (<= 64 KB)
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
    bench<OP><<<blocks, threads, lds>>>(d_cycles, d_sink, 0);   // warm-up
    HIP_OK(hipEventRecord(e0));
    bench<OP><<<blocks, threads, lds>>>(d_cycles, d_sink, 0);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipDeviceSynchronize());
    float ms = 0; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    HIP_OK(hipMemcpy(h.data(), d_cycles, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double instr_per_wave = (double)ITERS * UNROLL;
    // s_memtime counts at a fixed 100 MHz on this part; the event time gives wall seconds: report both
    const double ns = ms * 1e6 / (instr_per_wave * waves_per_simd);
    printf("%-40s waves/SIMD=%d %9.3f ms -> %6.2f ns = %5.2f cyc@2.4GHz per wave-instr per SIMD (counter ticks/instr %.3f)\n",
           NAMES[OP], waves_per_simd, ms, ns, ns * 2.4, avg / (instr_per_wave * waves_per_simd));
}

template <int OP> void both(unsigned long long* c, double* s, int blocks) { run<OP>(1, c, s, blocks); run<OP>(2, c, s, blocks); run<OP>(4, c, s, blocks); fflush(stdout); }

template <int OP> struct All { static void go(unsigned long long* c, double* s, int b) { both<OP>(c, s, b); All<OP + 1>::go(c, s, b); } };
template <> struct All<N_OPS> { static void go(unsigned long long*, double*, int) {} };

int main() {
    hipDeviceProp_t prop; HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;   // one workgroup per CU
    printf("device %s, %d CUs, clock %d kHz; 1 WG per CU, each WG = 4 SIMDs x waves/SIMD; %d x %d instrs per wave\n",
           prop.name, blocks, prop.clockRate, ITERS, UNROLL);
    printf("(mixed rows count every fma as one slot; the LDS op rides along)\n");
    unsigned long long* d_cycles; double* d_sink;
    HIP_OK(hipMalloc(&d_cycles, blocks * 8)); HIP_OK(hipMalloc(&d_sink, 8));
    {
        unsigned* d; HIP_OK(hipMalloc(&d, 64 * 16));
        swap_semantics<<<1, 64>>>(d);
        unsigned h[256]; HIP_OK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        printf("lane: permlane32_swap(a=lane,b=100+lane) -> (a',b') | permlane16_swap -> (a',b')\n");
        for (int l = 0; l < 64; l += 1) printf("  %2d: (%3u,%3u) | (%3u,%3u)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    All<0>::go(d_cycles, d_sink, blocks);
    return 0;
}
