// Scalar issue-rate microbenchmark for gfx950: cycles per wave SALU instruction per SIMD, alone and
// interleaved with VALU work of other waves / of the same wave.  k_trace issues 0.5 - 0.7 SALU
// instructions per VALU instruction (exec-mask bookkeeping of divergent regions), so whether the two
// pipelines overlap decides what is worth removing.   hipcc --offload-arch=gfx950 -O3 salu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096

// KIND 0: 64 SALU adds/xors per iteration; 1: 64 VALU fma; 2: 64 SALU + 64 VALU interleaved in one wave.
// Measured on MI355X (8 waves/SIMD): SALU 4.2 cycles per instruction per SIMD, v_fma_f32 2.6, an interleaved
// pair 4.9 - the pipelines overlap, and a scalar instruction costs MORE issue time than a plain f32 one.
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int n, int sk) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int s0 = sk, s1 = sk + 1, s2 = sk + 2, s3 = sk + 3;
    const bool scalar_wave = KIND == 0 || KIND == 2;
    const bool vector_wave = KIND == 1 || KIND == 2;
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (scalar_wave)
                asm volatile("s_add_u32 %0, %0, %4\n s_xor_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_xor_b32 %3, %3, %4"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(sk) : "scc");   // s_add/s_xor write SCC
            if (vector_wave) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(s0 + s1 + s2 + s3);
}

template <int KIND>
void run(const char* name, double insts_per_iter_per_wave, int waves_per_simd) {
    float* d; hipMalloc(&d, 1 << 24);
    int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, 16, 3);
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, 1.0001f, 0.5f, N_IT, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)N_IT * insts_per_iter_per_wave * waves_per_simd;
    printf("%-44s waves/SIMD=%d  %.3f ms  -> %.2f cycles @2.4GHz per counted instruction per SIMD\n", name, waves_per_simd, ms, ms * 1e6 / insts * 2.4);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("SALU only (per SALU)", 64, w);
        run<1>("VALU fma only (per VALU)", 64, w);
        run<2>("one wave: 64 SALU + 64 VALU (per pair)", 64, w);
    }
    return 0;
}
