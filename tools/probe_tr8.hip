// Probe of ds_read_b64_tr_b8 (gfx950): which lane's address supplies which bytes, and which bytes every lane receives.
// build + run on the GPU box: hipcc --offload-arch=gfx950 tools/probe_tr8.hip -o /tmp/probe_tr8 && /tmp/probe_tr8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v2i32 lds_v2i32;
__global__ void k(uint32_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char img[64 * 32];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 32; i += 64) img[i] = (unsigned char)((i / 32) * 16 + (i % 32 & 15));   // [row][32 B]: value = row * 16 + column (columns 0..15 twice)
    __syncthreads();
    const int g = lane >> 4, idx = lane & 15;
    int off;
    if (mode == 0) off = (8 * g + (idx >> 1)) * 32 + (idx & 1) * 8;   // hypothesis: lane 2q + p -> row q, bytes 8p ..
    else off = (8 * g + (idx & 7)) * 32 + (idx >> 3) * 8;              // alternative: lane q + 8p
    const v2i32 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i32*)(img + off));
    out[2 * lane] = v[0]; out[2 * lane + 1] = v[1];
}
int main() {
    uint32_t* d; hipMalloc(&d, 512);
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode);
        uint32_t h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
        printf("mode %d (value = 16 * (row %% 16) + column)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int b = 0; b < 8; ++b) { unsigned v = (h[2 * l + b / 4] >> (8 * (b % 4))) & 0xff; printf(" r%02u.c%02u", v >> 4, v & 15); }
            printf("\n");
        }
    }
    return 0;
}
