// VALU issue-rate microbenchmark for gfx950: how many wave64 VALU instructions per cycle per SIMD
// for the instruction kinds k_trace is made of.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {        // independent v_fma_f32
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            } else if (KIND == 1) { // v_mul + v_add (no contraction)
                x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b;
            } else if (KIND == 2) { // compare + select (v_cmp + v_cndmask)
                x0 = x0 > x1 ? x2 : x0; x1 = x1 > x2 ? x3 : x1; x2 = x2 > x3 ? x4 : x2; x3 = x3 > x4 ? x5 : x3;
                x4 = x4 > x5 ? x6 : x4; x5 = x5 > x6 ? x7 : x5; x6 = x6 > x7 ? x0 : x6; x7 = x7 > x0 ? x1 : x7;
            } else if (KIND == 4) { // v_pk_fma_f32: two f32 FMAs per lane per instruction
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, a2 = {a, a}, b2 = {b, b};
                p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
                p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
                p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
                p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
                x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
            } else if (KIND == 5) { // v_min_f32 / v_max_f32
                x0 = __builtin_fminf(x0, x1); x1 = __builtin_fmaxf(x1, x2); x2 = __builtin_fminf(x2, x3); x3 = __builtin_fmaxf(x3, x4);
                x4 = __builtin_fminf(x4, x5); x5 = __builtin_fmaxf(x5, x6); x6 = __builtin_fminf(x6, x7); x7 = __builtin_fmaxf(x7, x0);
            } else if (KIND == 6) { // v_mul_lo_u32
                unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1), u2 = __float_as_uint(x2), u3 = __float_as_uint(x3);
                unsigned u4 = __float_as_uint(x4), u5 = __float_as_uint(x5), u6 = __float_as_uint(x6), u7 = __float_as_uint(x7);
                u0 *= u1; u1 *= u2; u2 *= u3; u3 *= u4; u4 *= u5; u5 *= u6; u6 *= u7; u7 *= u0;
                x0 = __uint_as_float(u0); x1 = __uint_as_float(u1); x2 = __uint_as_float(u2); x3 = __uint_as_float(u3);
                x4 = __uint_as_float(u4); x5 = __uint_as_float(u5); x6 = __uint_as_float(u6); x7 = __uint_as_float(u7);
            } else {                // IEEE division
                x0 = x0 / a; x1 = x1 / a; x2 = x2 / a; x3 = x3 / a; x4 = x4 / a; x5 = x5 / a; x6 = x6 / a; x7 = x7 / a;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND>
void run(const char* name, int valu_per_iter, int waves_per_simd) {
    float* d; hipMalloc(&d, 1 << 24);
    int blocks = 256 * waves_per_simd;   // 256 CUs, 4 waves per block -> waves_per_simd per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, 16);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, N_IT);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)N_IT * valu_per_iter * waves_per_simd;   // per SIMD
    printf("%-24s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 64, w);
        run<1>("v_mul/v_add", 64, w);
        run<2>("v_cmp+v_cndmask (x2)", 128, w);
        run<3>("IEEE div (per div)", 64, w);
        run<4>("v_pk_fma_f32", 64, w);
        run<5>("v_min/v_max_f32", 64, w);
        run<6>("v_mul_lo_u32", 64, w);
    }
    return 0;
}
