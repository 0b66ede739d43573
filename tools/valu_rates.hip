// GPU box: issue cost of the fp64 instructions the colour chain is made of, in SIMD cycles per wave-instruction
// (one workgroup of 4 waves per CU x 8 rounds, 8 independent chains per lane so that latency is hidden).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rates tools/valu_rates.hip ; run on the box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHAINS 8
#define ITERS 2048
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, int *iout) {
    double v[CHAINS];
    int iv[CHAINS];
    for (int c = 0; c < CHAINS; c++) { v[c] = seed + threadIdx.x * 1e-3 + c; iv[c] = threadIdx.x + c; }
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) v[c] = __builtin_fma(v[c], 1.0000001, 1e-9);
            if (OP == 1) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[c]));
            if (OP == 2) asm volatile("v_rsq_f64 %0, %0" : "+v"(v[c]));
            if (OP == 3) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v[c]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[c]) : "v"(v[c]));
            if (OP == 5) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(v[c]) : "v"(iv[c]));
            if (OP == 6) asm volatile("v_trunc_f64 %0, %0" : "+v"(v[c]));
            if (OP == 7) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(v[c]));
            if (OP == 8) asm volatile("v_add_f64 %0, %0, %0" : "+v"(v[c]));
            if (OP == 9) asm volatile("v_div_scale_f64 %0, vcc, %0, %0, %0" : "+v"(v[c]) : : "vcc");
            if (OP == 10) asm volatile("v_div_fixup_f64 %0, %0, %0, %0" : "+v"(v[c]));
            if (OP == 11) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(v[c]) : "v"(iv[c]));
            if (OP == 12) asm volatile("v_cmp_lt_f64 vcc, %0, %0" : : "v"(v[c]) : "vcc");
            if (OP == 13) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(iv[c]));
            if (OP == 14) asm volatile("v_div_fmas_f64 %0, %0, %0, %0" : "+v"(v[c]) : : "vcc");
            if (OP == 15) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[c]));
            if (OP == 16) asm volatile("v_and_b32 %0, %0, %0" : "+v"(iv[c]));
            if (OP == 17) asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(iv[c]) : : "vcc");
            if (OP == 18) asm volatile("v_max_f64 %0, %0, %0" : "+v"(v[c]));
        }
    }
    double s = 0; int is = 0;
    for (int c = 0; c < CHAINS; c++) { s += v[c]; is += iv[c]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    iout[blockIdx.x * blockDim.x + threadIdx.x] = is;
}
template <int OP>
static void run(const char *name, double *out, int *iout) {
    const int blocks = 256 * 8; // 8 workgroups per CU in turn: 4 waves per SIMD... one wave per SIMD at a time per WG
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iout);
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 1.5, iout);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    // wave-instructions per SIMD: blocks * 4 waves * ITERS * CHAINS / 1024 SIMDs
    const double per_simd = (double)blocks * 4 * ITERS * CHAINS / 1024.0;
    printf("%-18s %8.3f ms  %6.2f cycles per wave-instruction at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    double *out; int *iout;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    hipMalloc(&iout, 256 * 8 * 256 * sizeof(int));
    run<0>("v_fma_f64", out, iout); run<7>("v_mul_f64", out, iout); run<8>("v_add_f64", out, iout); run<18>("v_max_f64", out, iout);
    run<1>("v_rcp_f64", out, iout); run<2>("v_rsq_f64", out, iout); run<3>("v_sqrt_f64", out, iout);
    run<4>("v_cvt_i32_f64", out, iout); run<5>("v_cvt_f64_i32", out, iout); run<11>("v_cvt_f64_u32", out, iout);
    run<6>("v_trunc_f64", out, iout); run<9>("v_div_scale_f64", out, iout); run<10>("v_div_fixup_f64", out, iout);
    run<14>("v_div_fmas_f64", out, iout); run<15>("v_ldexp_f64", out, iout);
    run<12>("v_cmp_lt_f64", out, iout); run<13>("v_fma_f32", out, iout); run<16>("v_and_b32", out, iout); run<17>("v_cndmask_b32", out, iout);
    return 0;
}
