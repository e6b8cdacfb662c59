// Probe: what does ds_read_b64_tr_b16 deliver?  LDS short[i] = i; lane l passes byte address 8*l
// (its own 4 consecutive shorts 4l..4l+3).  Prints, per lane, the 4 shorts received.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    k<<<1, 64>>>(d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    return 0;
}
