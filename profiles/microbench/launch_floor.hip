// What an (almost) empty launch of the small-conv shape costs on MI355X: 256 workgroups x 256 threads, back to back on one stream, as a function
// of the dynamic LDS size, the kernel-argument size and the register allocation — the floor under conv3x3_tile_kernel's per-launch fixed cost
// (round 5).   hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o bin/launch_floor && bin/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int w[136]; };                 // 544 bytes: sizeof(ConvArgs)
extern __shared__ unsigned char smem[];
__global__ __launch_bounds__(256, 1) void k_small(int* out, int v) { if (v == 12345) out[threadIdx.x] = smem[threadIdx.x]; }
__global__ __launch_bounds__(256, 1) void k_big(const Big a, int* out) { if (a.w[7] == 12345) out[threadIdx.x] = smem[threadIdx.x] + a.w[100]; }
// touches every argument line and stores one 16-byte vector per lane (what the real kernel cannot avoid)
__global__ __launch_bounds__(256, 1) void k_touch(const Big a, uint4* out) {
    int s = 0;
#pragma unroll
    for (int i = 0; i < 136; i += 16) s += a.w[i];
    out[blockIdx.x * 256 + threadIdx.x] = make_uint4(s, threadIdx.x, blockIdx.x, 0);
}
template <typename F> static float per_launch_us(F launch, int n = 2000) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / n;
}
int main() {
    int* out; hipMalloc(&out, 1 << 24);
    Big a{}; 
    hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int wgs : {8, 256, 512})
        for (size_t lds : {(size_t)0, (size_t)47 * 1024, (size_t)150 * 1024}) {
            const float t0 = per_launch_us([&] { hipLaunchKernelGGL(k_small, dim3(wgs), dim3(256), lds, 0, out, 0); });
            const float t1 = per_launch_us([&] { hipLaunchKernelGGL(k_big, dim3(wgs), dim3(256), lds, 0, a, out); });
            const float t2 = per_launch_us([&] { hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(256), lds, 0, a, (uint4*)out); });
            printf("%3d workgroups, %3zu KiB LDS: empty kernel %.2f us per launch, with 544-byte arguments %.2f us, reading them + one 16-byte store per lane %.2f us\n",
                   wgs, lds / 1024, t0, t1, t2);
        }
    return 0;
}
