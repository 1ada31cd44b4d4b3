// Which XCD does workgroup g of a 1-D launch run on?  Prints HW_REG_XCC_ID (s_getreg hwreg 20, low 4 bits) for the first 32 workgroups and
// whether "g & 7 fixes the XCD" (same g & 7 -> same XCC_ID, different g & 7 -> different XCC_ID) holds over a launch of 1024 workgroups.
//   hipcc --offload-arch=gfx950 -O2 xcc_map.hip -o /tmp/xcc_map && /tmp/xcc_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
}
int main() {
    const int n = 1024;
    unsigned* d;
    hipMalloc(&d, n * 4);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
        std::vector<unsigned> h(n);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        printf("raw XCC_ID register of workgroups 0..31:");
        for (int i = 0; i < 32; ++i) printf(" %x", h[i]);
        printf("\n");
        int of[8];
        bool ok = true;
        for (int i = 0; i < 8; ++i) of[i] = h[i] & 0xF;
        for (int i = 0; i < n; ++i) ok = ok && (int)(h[i] & 0xF) == of[i & 7];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < i; ++j) ok = ok && of[i] != of[j];
        printf("g & 7 fixes the XCD over %d workgroups: %s;  XCC_ID of g & 7 = 0..7:", n, ok ? "yes" : "NO");
        for (int i = 0; i < 8; ++i) printf(" %d", of[i]);
        printf("\n");
    }
    return 0;
}
