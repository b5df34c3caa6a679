// mfma_probe.hip -- issue rate of v_mfma_f32_16x16x4_f32 from ONE wavefront per SIMD: dependent chains and 2 .. 16 independent accumulators all issue
// every 32 cycles (155 TFLOP/s over the chip at 2.3 - 2.4 GHz): whatever a kernel loses against the MFMA peak is not the accumulator pattern.
//   hipcc --offload-arch=gfx950 -O3 -w -o tools/mfma_probe tools/mfma_probe.hip && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MF(a,b,c) __builtin_amdgcn_mfma_f32_16x16x4f32(a,b,c,0,0,0)
template <int NACC, int REPS>
__global__ void probe(float *out, long long *t, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0,0,0,0};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = a0 + i + threadIdx.x; b[i] = b0 - i; }
    long long t0 = clock64(); long long w0 = wall_clock64();
#pragma unroll 1
    for (int r = 0; r < REPS; ++r) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = MF(a[k], b[(k + i) & 7], acc[i]);
    }
    long long t1 = clock64(); long long w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = w1 - w0; }
}
template <int NACC>
void run(const char *name, int waves) {
    float *out; long long *t, th[2];
    hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&t, 16);
    constexpr int REPS = 4000;
    hipLaunchKernelGGL((probe<NACC, REPS>), dim3(256), dim3(64 * waves), 0, 0, out, t, 1.0f, 2.0f);
    hipLaunchKernelGGL((probe<NACC, REPS>), dim3(256), dim3(64 * waves), 0, 0, out, t, 1.0f, 2.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<NACC, REPS>), dim3(256), dim3(64 * waves), 0, 0, out, t, 1.0f, 2.0f);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("   kernel %.3f ms -> %.1f TFLOP/s\n", ms, 256.0 * waves * REPS * 8.0 * NACC * 2048.0 / (ms * 1e-3) / 1e12);
    hipMemcpy(th, t, 16, hipMemcpyDeviceToHost);
    printf("%s nacc=%d waves/WG=%d: %.1f clock64 ticks, %.2f ns per MFMA per wave (clock64 rate %.0f MHz)\n", name, NACC, waves, (double)th[0] / (REPS * 8.0 * NACC), (double)th[1] * 10.0 / (REPS * 8.0 * NACC), (double)th[0] / ((double)th[1] * 0.01));
}
int main() {
    run<1>("16x16x4", 4); run<2>("16x16x4", 4); run<4>("16x16x4", 4); run<8>("16x16x4", 4); run<16>("16x16x4", 4);
    run<4>("16x16x4", 8); run<16>("16x16x4", 8); run<4>("16x16x4", 16);
    return 0;
}
