// Semantics probe for ds_read_b64_tr_b16 (gfx950): every lane points at its own 4 contiguous 16-bit values; print what each lane gets.
// Expected (and observed): inside each 16-lane group, lane l receives element (l & 3) of the 8-byte words of lanes 4j + ((l & 15) >> 2), j = 0..3
// i.e. the group's 16 words form a [4 rows][16 cols] matrix (row = source lane >> 2, col = 4*(source lane & 3) + element) and lane l gets column l.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s4;
__global__ void k(s4* out) {
    __shared__ short lds[64 * 4 * 2];
    for (int e = 0; e < 4; ++e) lds[threadIdx.x * 8 + e] = (short)(threadIdx.x * 4 + e);   // lane's word at a 16-byte stride
    __syncthreads();
    out[threadIdx.x] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + threadIdx.x * 8));
}
int main() {
    s4* d; (void)hipMalloc(&d, 64 * 8);
    k<<<1, 64>>>(d);
    short h[256]; (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 4; ++j) {
            const int src_lane = (l & ~15) + 4 * j + ((l & 15) >> 2), src_elem = l & 3;
            if (h[l * 4 + j] != src_lane * 4 + src_elem) ++bad;
        }
        if (l < 20) printf("lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    printf("mismatches vs expected mapping: %d\n", bad);
    return 0;
}
