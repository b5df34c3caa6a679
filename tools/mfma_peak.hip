// Calibration: what fp32 MFMA rate does an MI355X sustain with NO memory traffic at all?
// (roofline.peak in bench.py is the datasheet 157.3 TFLOP/s; this prints the attainable ceiling.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) spin(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 12345.f) out[0] = s;
}

template <int NACC>
void run(int blocks, int iters, const char *label) {
    float *out;
    (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(spin<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * NACC * 4096.0;
        printf("%s rep %d: %.3f ms  %.1f TFLOP/s\n", label, rep, ms, fl / (ms * 1e-3) / 1e12);
    }
    (void)hipFree(out);
}

int main() {
    run<8>(256, 20000, "1 wave/SIMD, 8 acc, ~27 ms");
    run<8>(512, 20000, "2 waves/SIMD, 8 acc");
    run<4>(1024, 20000, "4 waves/SIMD, 4 acc");
    run<8>(512, 200000, "2 waves/SIMD, 8 acc, long (~0.5 s)");
    return 0;
}
