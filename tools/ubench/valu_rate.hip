// Issue-rate microbenchmark: cycles per wave-instruction of v_exp_f32 / v_fma_f32 / v_pk_fma_f32 / v_cvt_pk_f16_f32 /
// v_max3_f32 with 1, 2 and 4 waves per SIMD (s_memtime around 4096 independent instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
    float a[8];
    f2 b[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i * 0.01f + threadIdx.x * 1e-6f; b[i] = {a[i], a[i] * 0.5f}; }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 512; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = __builtin_amdgcn_exp2f(a[i]) * 0.0f + a[i];      // exp + fma
            if (OP == 1) a[i] = fmaf(a[i], 0.999f, 0.001f);
            if (OP == 2) b[i] = __builtin_elementwise_fma(b[i], (f2){0.999f, 0.999f}, (f2){0.001f, 0.001f});
            if (OP == 3) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.001f);
            if (OP == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + b[i][0] + b[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const char* names[5] = {"v_exp_f32 + v_fma", "v_fma_f32", "v_pk_fma_f32", "v_mul + v_exp_f32", "v_rcp_f32"};
    for (int waves = 1; waves <= 4; waves *= 2) {
        for (int op = 0; op < 5; ++op) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                dim3 g(256), b(256 * waves);
                if (op == 0) hipLaunchKernelGGL(k<0>, g, b, 0, 0, out, cyc, 0.5f);
                if (op == 1) hipLaunchKernelGGL(k<1>, g, b, 0, 0, out, cyc, 0.5f);
                if (op == 2) hipLaunchKernelGGL(k<2>, g, b, 0, 0, out, cyc, 0.5f);
                if (op == 3) hipLaunchKernelGGL(k<3>, g, b, 0, 0, out, cyc, 0.5f);
                if (op == 4) hipLaunchKernelGGL(k<4>, g, b, 0, 0, out, cyc, 0.5f);
                hipDeviceSynchronize();
                hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            }
            printf("waves/SIMD %d  %-20s %6.2f cycles per wave-iteration-element (4096 elements per wave)\n", waves, names[op], (double)h / 4096.0);
        }
    }
    return 0;
}
