// debug tool: occupancy of the conv kernel instantiations as HIP computes it
#include "../explorable-super-resolution_amd/csrc/esr_conv.hip"
#include <cstdio>
template <int NPL, int MT, int R> void q(size_t lds) {
    int nb = -1;
    auto k = conv3x3_kernel<NPL, MT, R>;
    hipError_t e0 = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, lds);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k);
    printf("NPL %d MT %d R %d lds %zu -> blocks/CU %d (err %d %d) regs %d static lds %zu maxdyn %d\n", NPL, MT, R, lds, nb, (int)e0, (int)e, fa.numRegs, fa.sharedSizeBytes, fa.maxDynamicSharedSizeBytes);
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s LDS/CU %zu maxSharedPerBlock %zu regsPerBlock %d\n", p.name, p.maxSharedMemoryPerMultiProcessor, p.sharedMemPerBlock, p.regsPerBlock);
    q<2,1,3>(52864); q<2,1,3>(40000); q<2,1,3>(30000); q<2,2,3>(71296); q<2,2,3>(60000); q<2,1,2>(40000); q<1,1,3>(26000);
    return 0;
}
