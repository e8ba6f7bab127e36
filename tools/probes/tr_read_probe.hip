// Probe: semantics of ds_read_b64_tr_b16 (what the wgrad kernel's operand reads rely on).
// Expected (cdna_hip_programming.md T10): within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3 of
// a 4 x 16 block of 16-bit elements; lane i receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
    __shared__ short tile[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = in[i];
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, q = i >> 2, p = i & 3, g = l >> 4;
    v4s* a = (v4s*)&tile[(q + 4 * g) * 64 + 16 * g + 4 * p];     // group g: rows 4g..4g+3, columns 16g..16g+15
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)a);
    ((v4s*)out)[l] = r;
}
int main() {
    std::vector<short> h(64 * 64), o(256);
    for (int i = 0; i < 64 * 64; i++) h[i] = (short)i;
    short *d, *e;
    if (hipMalloc(&d, h.size() * 2) != hipSuccess || hipMalloc(&e, 512) != hipSuccess) return 2;
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    if (hipMemcpy(o.data(), e, 512, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int q = 0; q < 4; q++) {
        const int g = l >> 4, i = l & 15;
        const short want = (short)((4 * g + q) * 64 + 16 * g + i);
        if (o[l * 4 + q] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, q, o[l * 4 + q], want); bad++; }
    }
    printf("tr_read_probe: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK", bad);
    return bad ? 1 : 0;
}
