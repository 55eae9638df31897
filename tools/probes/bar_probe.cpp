// Probe: is device memory (hipMalloc) directly writable by the CPU on this box (large BAR), and how long does it
// take to put 22 KB there that way, against hipMemcpyAsync from pinned memory?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <xmmintrin.h>
__global__ void sum_kernel(const double* p, int n, double* out) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += p[i];
    for (int off = 32; off; off >>= 1) s += __shfl_down(s, off);
    if (threadIdx.x == 0) *out = s;
}
int main() {
    int large = -1;
    hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, 0);
    printf("IsLargeBar=%d\n", large);
    fflush(stdout);
    const int n = 2816;   // 22 KB
    double *d = nullptr, *h_out = nullptr, *h_pin = nullptr, *d2 = nullptr;
    hipMalloc(&d, n * 8);
    hipMalloc(&d2, n * 8);
    hipHostMalloc(&h_out, 8, hipHostMallocMapped);
    hipHostMalloc(&h_pin, n * 8, hipHostMallocDefault);
    double* d_out = nullptr;
    hipHostGetDevicePointer((void**)&d_out, h_out, 0);
    std::vector<double> src(n);
    for (int i = 0; i < n; ++i) src[i] = i * 0.5;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipPointerAttribute_t attr;
    hipPointerGetAttributes(&attr, d);
    printf("type=%d hostPointer=%p devicePointer=%p\n", (int)attr.type, attr.hostPointer, attr.devicePointer);
    fflush(stdout);
    if (large != 1) { printf("no large BAR: not trying\n"); return 0; }
    // direct CPU writes into device memory
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::memcpy(d, src.data(), n * 8);
        _mm_sfence();
        auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, d, n, d_out);
        hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        printf("BAR write %.2f us, kernel+sync %.2f us, sum=%.1f (want %.1f)\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count(),
               std::chrono::duration<double, std::micro>(t2 - t1).count(), *h_out, 0.5 * (double)n * (n - 1) / 2.0);
        src[0] += 1.0;
    }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::memcpy(h_pin, src.data(), n * 8);
        hipMemcpyAsync(d2, h_pin, n * 8, hipMemcpyHostToDevice, st);
        auto t1 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(64), 0, st, d2, n, d_out);
        hipStreamSynchronize(st);
        auto t2 = std::chrono::steady_clock::now();
        printf("pinned memcpyAsync enqueue %.2f us, kernel+sync %.2f us, sum=%.1f\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count(),
               std::chrono::duration<double, std::micro>(t2 - t1).count(), *h_out);
        src[0] += 1.0;
    }
    return 0;
}
