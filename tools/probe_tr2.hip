// Probe 2: replicate gemm.hip's TN LDS image + tr_frag addressing; LDS short value = row*128+col.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ inline int tn_sw(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ inline int tn_off(int row, int chunk) { return row * 256 + ((chunk ^ tn_sw(row)) << 4); }
__global__ void k(short* out, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // fill: 64 rows x 128 cols, via 8-byte stores like the P path
    for (int i = threadIdx.x; i < 64 * 32; i += 64) {
        int row = i / 32, c4 = i % 32;
        s16x4 v;
        for (int j = 0; j < 4; ++j) v[j] = (short)(row * 128 + c4 * 4 + j);
        *reinterpret_cast<s16x4*>(smem + tn_off(row, c4 >> 1) + (c4 & 1) * 8) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x;
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int row = 0 + 8 * g + q;
    const int chunk = (col0 >> 3) + (pp >> 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + tn_off(row, chunk) + (pp & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + tn_off(row + 4, chunk) + (pp & 1) * 8));
    for (int j = 0; j < 4; ++j) { out[lane * 8 + j] = lo[j]; out[lane * 8 + 4 + j] = hi[j]; }
}
int main() {
    short* d; (void)hipMalloc(&d, 64 * 8 * 2);
    for (int col0 = 0; col0 <= 16; col0 += 16) {
        k<<<1, 64, 64 * 256>>>(d, col0);
        short h[512]; (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
        printf("col0=%d\n", col0);
        for (int l = 0; l < 64; l += 1) {
            printf("lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" (%2d,%3d)", h[l*8+j] / 128, h[l*8+j] % 128);
            printf("\n");
        }
    }
    return 0;
}
