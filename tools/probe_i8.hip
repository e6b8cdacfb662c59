// probe of v_mfma_i32_16x16x64_i8 operand layout: which (row, k) does byte j of lane l supply?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    signed char a[16], b[16];
    for (int j = 0; j < 16; ++j) {
        const int kk = 16 * g + j;
        if (mode == 0) { a[j] = (signed char)(r + 1); b[j] = (kk == r) ? 1 : 0; }          // C[i][n] = (i+1) if layout (row = lane%16)
        else { a[j] = (kk == r) ? 1 : 0; b[j] = (signed char)(r + 1); }                     // C[i][n] = (n+1)
    }
    v4i av, bv, acc = {0, 0, 0, 0};
    __builtin_memcpy(&av, a, 16); __builtin_memcpy(&bv, b, 16);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = acc[e];
}
int main() {
    int* d; hipMalloc(&d, 64 * 4 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode);
        int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: C as [lane][e] assuming row=4g+e col=r:\n", mode);
        for (int g = 0; g < 4; ++g) for (int e = 0; e < 4; ++e) { printf("row %2d:", 4 * g + e); for (int r = 0; r < 16; ++r) printf(" %3d", h[(16 * g + r) * 4 + e]); printf("\n"); }
    }
    return 0;
}
