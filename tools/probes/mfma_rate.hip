// How long does one v_mfma_f32_32x32x16_bf16 take?  One or two waves per SIMD issue N MFMAs back to back on 4 accumulators; prints ns and
// s_memtime ticks per MFMA and the implied TFLOP/s.  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o tools/probes/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void k(const float* in, float* out, unsigned long long* ticks, int n) {
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)in[threadIdx.x % 64 + i]; b[i] = (__bf16)in[64 + threadIdx.x % 64 + i]; }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
int main() {
    float *in, *out; unsigned long long* ticks;
    hipMalloc(&in, 4096); hipMalloc(&out, 2048 * 512 * 4); hipMalloc(&ticks, 2048 * 8);
    float h[1024]; for (int i = 0; i < 1024; i++) h[i] = (float)((i * 37) % 17) * 0.01f - 0.05f;
    hipMemcpy(in, h, 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) for (int blocks : {256, 1}) {
        const int n = 50000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, in, out, ticks, 1000);
        hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, in, out, ticks, n); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t; hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double per_simd = 4.0 * n * (threads / 256);          // MFMAs per SIMD
        printf("blocks %4d threads %3d: %.3f ms  %.1f ns / MFMA per SIMD  %.1f ticks / MFMA per SIMD  -> %.0f TFLOP/s for the launch\n", blocks, threads, ms,
               ms * 1e6 / per_simd, (double)t / per_simd, blocks * 4.0 * per_simd * 32768.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
