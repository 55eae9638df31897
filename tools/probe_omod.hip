// probe_omod.hip - does gfx950 apply the VOP3 output modifier (mul:2 / mul:4 / div:2) to FP64 results, and under
// which MODE-register settings (IEEE bit, FP64 denormal mode)?  With it one Newton step of 1/sqrt(x) is three
// instructions instead of four:  t = (x y) div:2;  r = fma(-t, y, 1.5);  y' = y r.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_omod.hip -o /tmp/probe_omod && /tmp/probe_omod
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE register: bits 4-5 FP32 denormals, 6-7 FP64/FP16 denormals, bit 9 IEEE
template <int IEEE, int DEN64>
__device__ __forceinline__ void set_mode() {
    if (IEEE) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 1");
    else asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
    if (DEN64) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 3");
    else asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 0");
}

__device__ __forceinline__ double mul_half(double a, double b) {
    double r;
    asm("v_mul_f64 %0, %1, %2 div:2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double mul_x4(double a, double b) {
    double r;
    asm("v_mul_f64 %0, %1, %2 mul:4" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double rsq_half(double a) {
    double r;
    asm("v_rsq_f64_e64 %0, %1 div:2" : "=v"(r) : "v"(a));
    return r;
}
__device__ __forceinline__ double fma_half(double a, double b, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3 div:2" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int IEEE, int DEN64>
__global__ void k_modes(const double* x, double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    unsigned mode_before, mode_in;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode_before));
    set_mode<IEEE, DEN64>();
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode_in));
    const double a = mul_half(v, 3.0);        // expect 1.5 v
    const double b = mul_x4(v, 3.0);          // expect 12 v
    const double c = rsq_half(v);             // expect rsq(v) / 2
    const double d = fma_half(v, 3.0, 1.0);   // expect (3 v + 1) / 2
    // the three-instruction Newton step
    const double y = __builtin_amdgcn_rsq(v);
    const double t = mul_half(v, y);
    const double r = __builtin_fma(-t, y, 1.5);
    const double y3 = y * r;
    set_mode<1, 1>();
    out[i * 8 + 0] = a; out[i * 8 + 1] = b; out[i * 8 + 2] = c; out[i * 8 + 3] = d; out[i * 8 + 4] = y3;
    out[i * 8 + 5] = (double)mode_before; out[i * 8 + 6] = (double)mode_in;
    const double t4 = v * y;
    const double e4 = __builtin_fma(-t4, y, 1.0);
    out[i * 8 + 7] = __builtin_fma(y * e4, 0.5, y);             // the four-instruction step
}

// issue cost: 8 independent chains of Newton steps, three- against four-instruction form
template <int FORM>
__global__ void k_rate(double* out, int iters) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + threadIdx.x * 1e-3 + 0.1 * k;
    if (FORM == 1) set_mode<0, 0>();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double x = a[k];
            const double y = __builtin_amdgcn_rsq(x);
            double yn;
            if (FORM == 0) {
                const double t = x * y;
                const double e = __builtin_fma(-t, y, 1.0);
                yn = __builtin_fma(y * e, 0.5, y);
            } else {
                const double t = mul_half(x, y);
                const double r = __builtin_fma(-t, y, 1.5);
                yn = y * r;
            }
            a[k] = __builtin_fma(yn, 0.25, 1.0);
        }
    }
    if (FORM == 1) set_mode<1, 1>();
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FORM>
double rate(double* d_out) {
    const int iters = 20000, blocks = 256 * 4 * 2, threads = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<FORM>, dim3(blocks), dim3(threads), 0, 0, d_out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<FORM>, dim3(blocks), dim3(threads), 0, 0, d_out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int IEEE, int DEN64>
int run(const double* dx, double* dy, const std::vector<double>& x, std::vector<double>& y, int n) {
    hipLaunchKernelGGL((k_modes<IEEE, DEN64>), dim3((n + 255) / 256), dim3(256), 0, 0, dx, dy, n);
    CHECK(hipMemcpy(y.data(), dy, (size_t)n * 64, hipMemcpyDeviceToHost));
    double ea = 0, eb = 0, ec = 0, ed = 0, e3 = 0, e4 = 0;
    for (int i = 0; i < n; ++i) {
        const double v = x[i];
        const long double ex = 1.0L / sqrtl((long double)v);
        ea = fmax(ea, fabs(y[i * 8 + 0] / (1.5 * v) - 1.0));
        eb = fmax(eb, fabs(y[i * 8 + 1] / (12.0 * v) - 1.0));
        ec = fmax(ec, fabs((double)(y[i * 8 + 2] / (0.5L * ex) - 1.0L)));
        ed = fmax(ed, fabs(y[i * 8 + 3] / ((3.0 * v + 1.0) * 0.5) - 1.0));
        e3 = fmax(e3, fabs((double)(y[i * 8 + 4] / ex - 1.0L)));
        e4 = fmax(e4, fabs((double)(y[i * 8 + 7] / ex - 1.0L)));
    }
    printf("IEEE=%d DEN64=%d  MODE before 0x%x inside 0x%x | rel err: mul div:2 %.2e  mul mul:4 %.2e  rsq div:2 %.2e  fma div:2 %.2e | "
           "newton3 %.3e (2^%.1f)  newton4 %.3e (2^%.1f)\n", IEEE, DEN64, (unsigned)y[5], (unsigned)y[6], ea, eb, ec, ed,
           e3, log2(e3), e4, log2(e4));
    return 0;
}

int main() {
    const int n = 1 << 18;
    std::vector<double> x(n), y((size_t)n * 8);
    std::mt19937_64 rng(1);
    std::uniform_real_distribution<double> mant(1.0, 4.0), ex(-40, 10);
    for (int i = 0; i < n; ++i) x[i] = mant(rng) * std::pow(2.0, std::floor(ex(rng)));
    double *dx, *dy;
    CHECK(hipMalloc(&dx, (size_t)n * 8)); CHECK(hipMalloc(&dy, (size_t)n * 64));
    CHECK(hipMemcpy(dx, x.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    if (run<1, 1>(dx, dy, x, y, n)) return 1;
    if (run<0, 1>(dx, dy, x, y, n)) return 1;
    if (run<1, 0>(dx, dy, x, y, n)) return 1;
    if (run<0, 0>(dx, dy, x, y, n)) return 1;
    double* d_out; CHECK(hipMalloc(&d_out, 2048 * 256 * 8));
    for (int rep = 0; rep < 2; ++rep) {
        const double m0 = rate<0>(d_out), m1 = rate<1>(d_out);
        printf("rate: newton4 chain %.3f ms, newton3 (omod) chain %.3f ms  ratio %.3f (instruction counts 7 : 6 incl. rsq as 4 -> %.3f)\n",
               m0, m1, m1 / m0, 9.0 / 10.0);
    }
    return 0;
}
