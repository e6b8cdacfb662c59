// Fixed cost of a one-round launch: an (almost) empty kernel at the step's launch geometries, timed back to back with HIP events.
// hipcc --offload-arch=gfx950 tools/probe_launch.hip -o /tmp/probe_launch && /tmp/probe_launch
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(float* out) {
    extern __shared__ char smem[];
    if (out && threadIdx.x == 0 && blockIdx.x == 0xffffff) out[0] = smem[0];
}
__global__ void k_store(float4* out, int n4_per_wg) {   // every workgroup writes n4_per_wg float4 (the partial tile of a weight gradient: 12,288)
    for (int i = threadIdx.x; i < n4_per_wg; i += blockDim.x) out[(size_t)blockIdx.x * n4_per_wg + i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
static float time_it(void (*launch)(), int n) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / n;
}
static int G, T, L; static float4* buf;
int main() {
    hipFuncSetAttribute((const void*)k_empty, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipMalloc(&buf, (size_t)256 * 12288 * 16);
    const int geo[][3] = {{252, 512, 160 * 1024}, {252, 512, 0}, {243, 512, 160 * 1024}, {1536, 512, 151296}, {485, 768, 151552}, {2048, 256, 0}, {252, 64, 0}};
    for (auto& g : geo) {
        G = g[0]; T = g[1]; L = g[2];
        float us = time_it([]() { k_empty<<<G, T, L, 0>>>(nullptr); }, 2000);
        printf("empty kernel  grid %5d x %4d threads, %6d B LDS: %.2f us per launch (back to back)\n", G, T, L, us);
    }
    G = 252;
    float us = time_it([]() { k_store<<<G, 512, 0, 0>>>(buf, 12288); }, 500);
    printf("store kernel  252 workgroups x 196 KB (49.5 MB): %.2f us per launch = %.2f TB/s\n", us, 252.0 * 12288 * 16 / us / 1e6);
    return 0;
}
