// Workgroups of one wavefront per CU as a function of their dynamic LDS size (hipOccupancyMaxActiveBlocksPerMultiprocessor
// on a trivial kernel): the allocation granule of the 160 KB.  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_lds_granule.hip -o /tmp/probe_lds
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {
    extern __shared__ double s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    out[threadIdx.x] = s[63 - threadIdx.x];
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    int last = -1;
    for (int bytes = 512; bytes <= 16384; bytes += 64) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes) != hipSuccess) { printf("error at %d\n", bytes); return 1; }
        if (n != last) { printf("%6d B -> %d workgroups per CU (x %d = %d B)\n", bytes, n, bytes, n * bytes); last = n; }
    }
    return 0;
}
