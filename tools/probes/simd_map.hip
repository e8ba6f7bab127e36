// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits [5:4] = SIMD_ID, [3:0] = WAVE_ID, [11:8] = CU_ID on gfx9)
// hipcc --offload-arch=gfx950 -O2 tools/probes/simd_map.hip -o tools/probes/simd_map && tools/probes/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512, 2) void probe(unsigned* out, int lds_words) {
    extern __shared__ unsigned lds[];
    if (threadIdx.x == 0) lds[lds_words - 1] = 1;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; const int nb = 16;
    hipMalloc(&d, nb * 8 * 4);
    for (int lds_kb : {0, 150}) {
        hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(probe, dim3(nb), dim3(512), lds_kb * 1024 + 16, 0, d, lds_kb * 256 + 4);
        unsigned h[nb * 8];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("dynamic LDS %d KiB\n", lds_kb);
        for (int b = 0; b < 4; b++) {
            printf("  wg %d (cu %2u se %u):", b, (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
            for (int w = 0; w < 8; w++) printf("  w%d->simd%u", w, (h[b * 8 + w] >> 4) & 3);
            printf("\n");
        }
    }
    return 0;
}
