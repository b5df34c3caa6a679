// fetch_calib.hip -- known-byte streams for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (VERDICT r05 item 6,
// MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane streaming read; other widths and
// WRITE_SIZE are uncalibrated).  Every kernel moves a buffer of 1 GiB (4 x the 256-MiB Infinity Cache) exactly once, in one of
// the access patterns the kernels of this repo use; tools/fetch_calib.py divides the known bytes by what the counters report.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fetch_calib tools/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d out_f -o f -- tools/fetch_calib ;  rocprofv3 --pmc WRITE_SIZE -d out_w -o w -- tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void calib_read_dword(const float *__restrict__ in, float *__restrict__ out, size_t n) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 12345.678f) out[0] = s;                    // never true for the zero-filled buffer: the loads stay, nothing is written
}
__global__ void calib_read_f4(const float4 *__restrict__ in, float *__restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = in[i];
        s += (v.x + v.y) + (v.z + v.w);
    }
    if (s == 12345.678f) out[0] = s;
}
__global__ void calib_write_dword(float *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}
__global__ void calib_write_f4(float4 *__restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void calib_write_f4_nt(float4 *__restrict__ out, size_t n4) {      // the activation cache's stores (st_stream4, csrc/tsc_model.hip)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float *p = reinterpret_cast<float *>(out + i);
        __builtin_nontemporal_store(1.f, p); __builtin_nontemporal_store(2.f, p + 1);
        __builtin_nontemporal_store(3.f, p + 2); __builtin_nontemporal_store(4.f, p + 3);
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 4, n4 = bytes / 16;
    float *a, *b, *o;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 256));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), blk(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_read_dword, grid, blk, 0, 0, a, o, n);
        hipLaunchKernelGGL(calib_read_f4, grid, blk, 0, 0, (const float4 *)b, o, n4);
        hipLaunchKernelGGL(calib_write_dword, grid, blk, 0, 0, a, n);
        hipLaunchKernelGGL(calib_write_f4, grid, blk, 0, 0, (float4 *)b, n4);
        hipLaunchKernelGGL(calib_write_f4_nt, grid, blk, 0, 0, (float4 *)a, n4);
        CK(hipDeviceSynchronize());
    }
    printf("{\"bytes_per_launch\": %zu}\n", bytes);
    return 0;
}
