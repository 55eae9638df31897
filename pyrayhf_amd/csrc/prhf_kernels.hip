// prhf_kernels.hip - fused vertical-ionogram forward operator for gfx950 (MI355X, CDNA4).
//
// One wavefront (64 lanes) evaluates one (profile, frequency) pair - or one chunk of its
// stretched grid when few pairs are submitted.  A workgroup shares one profile: its
// bottomside columns are staged once into LDS as 64-byte nodes
//     {alt, den, d(den)/dz, |B|, d|B|/dz, psi, d(psi)/dz, f_N^2}
// so that every grid point costs four ds_read_b128 and no HBM traffic.  Per pair:
//   S3-S6  reflection height: lanes stride over the levels, first level with X (or X+Y) > 1
//          by ballot, running maximum below it by a wave max-reduce, np.interp semantics;
//   S7-S10 lanes stride over the n_points stretched altitudes: locate the segment through
//          an LDS hint table, interpolate den/|B|/psi linearly, Appleton-Hartree mu and mu';
//   S11    left-rectangle sum of mu'*dh with NaNs skipped, wave sum-reduce, 0 -> NaN, + min(alt).
// Stage names S0-S11 are SURVEY.md section 2.1; reference line numbers are PyRayHF/library.py.
//
// No MFMA: the work is elementwise float64 transcendental + reduction ("Roofline").

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "prhf_kernels.h"

namespace prhf {

namespace {

constexpr double kPlasma = 8.97866275;            // library.py:61
constexpr double kGyro = 2.799249247e10;          // library.py:64
constexpr double kBackoff = 1e-6;                 // library.py:378
constexpr double kDegToRad = 0.017453292519943295;  // numpy deg2rad multiplies by pi/180
constexpr double kUnmagTol = 1e-12;               // library.py:163

struct __attribute__((aligned(16))) Node {
    double alt, den, sden, b, sb, psi, spsi, pf2;
};
static_assert(sizeof(Node) == 64, "node must be 64 bytes");

__device__ __forceinline__ double qnan() { return __builtin_nan(""); }

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// ---------------------------------------------------------------------------------------
// Appleton-Hartree group index, reference operation order (library.py:194-256).
// SIGN = +1 ordinary, -1 extraordinary (library.py:221-224).
// ---------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ double group_index_faithful(double X, double Y, double psi_deg) {
#pragma clang fp contract(off)
    constexpr double sgn = (MODE == PRHF_KMODE_O) ? 1.0 : -1.0;
    const double r = psi_deg * kDegToRad;
    double s, c;
    sincos(r, &s, &c);
    const double YT = Y * s;                                   // :210
    const double YL = Y * c;                                   // :211
    const double Xm1 = 1.0 - X;                                // :214
    const double YT2 = YT * YT;
    const double YL2 = YL * YL;
    const double Xm12 = Xm1 * Xm1;
    const double alpha = 0.25 * (YT2 * YT2) + YL2 * Xm12;      // :217
    const double beta = sqrt(alpha);                           // :218
    const double D = (Xm1 - 0.5 * YT2) + sgn * beta;           // :229
    const double XXm1 = X * Xm1;
    const double q = XXm1 / D;
    double rad = 1.0 - q;                                      // :232
    if (rad < 0.0) rad = qnan();                               // :233
    double mu = sqrt(rad);
    if (mu > 1.0) mu = qnan();                                 // :238
    const double dbdX = ((-YL2) * Xm1) / beta;                 // :241
    const double dDdX = -1.0 + sgn * dbdX;                     // :242
    const double dadY = (YT2 * YT) * s + ((2.0 * YL) * Xm12) * c;   // :244-245
    const double dbdY = (0.5 * dadY) / beta;                   // :246
    const double dDdY = (-YT) * s + sgn * dbdY;                // :247
    const double two_mu = 2.0 * mu;
    const double dmudY = (XXm1 * dDdY) / (two_mu * (D * D));   // :250
    const double dmudX = (1.0 / (two_mu * D)) * (((2.0 * X) - 1.0) + q * dDdX);   // :251
    return mu - ((2.0 * X) * dmudX + Y * dmudY);               // :254
}

// Isotropic plasma (library.py:201-207): mu = sqrt(1-X) for X < 1, mu' = 1/mu.
__device__ __forceinline__ double group_index_unmagnetised(double X) {
    const double m2 = 1.0 - X;
    if (!(m2 > 0.0)) return qnan();
    const double mu = sqrt(m2);
    return 1.0 / mu;
}

// ---------------------------------------------------------------------------------------
// S3-S6: reflection height of one pair.  Returns false when the frequency escapes.
// np.interp(1.0, running_max, alt) semantics (library.py:388-407): j = last level whose
// running maximum is <= 1; exact hit returns alt[j]; otherwise linear between j and j+1.
// ---------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ bool reflection_height(const Node* __restrict__ nodes, int K, double f_hz,
                                                  double f2, int lane, double* h_out) {
#pragma clang fp contract(off)
    double lmax = -__builtin_inf();
    int kstar = K;
    double col_star = 0.0;
    for (int base = 0; base < K; base += 64) {
        const int k = base + lane;
        double col = -__builtin_inf();
        if (k < K) {
            col = nodes[k].pf2 / f2;                                        // :136 on (F,K)
            if (MODE == PRHF_KMODE_X) col = col + (kGyro * nodes[k].b) / f_hz;   // :157, :389
        }
        const unsigned long long hit = __ballot(col > 1.0);
        if (hit) {
            const int first = __ffsll((long long)hit) - 1;
            kstar = base + first;
            col_star = __shfl(col, first);
            if (lane < first) lmax = fmax(lmax, col);
            break;
        }
        lmax = fmax(lmax, col);
    }
    const double below = wave_max(lmax);        // running maximum at level kstar-1
    double h;
    if (kstar == K) {
        if (!(below >= 1.0)) return false;      // never reaches the cutoff (:399)
        h = nodes[K - 1].alt;                   // running max == 1 exactly at the top level
    } else if (kstar == 0) {
        h = nodes[0].alt;                       // already above cutoff at the bottom: left clamp
    } else {
        const int j = kstar - 1;
        const double aj = nodes[j].alt;
        if (below == 1.0) {
            h = aj;
        } else {
            const double slope = (nodes[j + 1].alt - aj) / (col_star - below);
            h = slope * (1.0 - below) + aj;
        }
    }
    *h_out = h - kBackoff;                      // :407
    return true;
}

struct BlockInfo {
    int K;            // bottomside levels (index of the density peak)
    int bad;          // PRHF_STATUS_* bits for this profile
    int unmag;        // isotropic branch
    double alt_min;   // min over the whole altitude column (:507)
    double a0;        // alt[0]
    double inv_w;     // hint buckets per km
};

constexpr int kHintBuckets = PRHF_HINT_BUCKETS;

// Stage one profile into LDS.  Every thread of the block calls this.
template <int THREADS>
__device__ __forceinline__ BlockInfo stage_profile(const double* __restrict__ den,
                                                   const double* __restrict__ bmag,
                                                   const double* __restrict__ bpsi,
                                                   const double* __restrict__ alt,
                                                   const double* __restrict__ freq, int n_freq,
                                                   int n_alt, Node* nodes, unsigned short* hint,
                                                   double* red) {
#pragma clang fp contract(off)
    constexpr int W = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // red layout: [0,W) peak value, [W,2W) peak index (as double), [2W,3W) alt min,
    //             [3W,4W) freq min, [4W,5W) |B| max, [5W,6W) negative-density flag
    // ---- phase 1: first-occurrence argmax of density, min altitude, min frequency ---------
    double bv = -__builtin_inf();
    int bi = 0x7fffffff;
    double amin = __builtin_inf();
    for (int i = tid; i < n_alt; i += THREADS) {
        const double v = den[i];
        if (v > bv) { bv = v; bi = i; }
        amin = fmin(amin, alt[i]);
    }
    double fm = __builtin_inf();
    for (int i = tid; i < n_freq; i += THREADS) fm = fmin(fm, fabs(freq[i]));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    amin = wave_min(amin);
    fm = wave_min(fm);
    if (lane == 0) {
        red[wave] = bv;
        red[W + wave] = (double)bi;
        red[2 * W + wave] = amin;
        red[3 * W + wave] = fm;
    }
    __syncthreads();
    bv = red[0];
    bi = (int)red[W];
    amin = red[2 * W];
    fm = red[3 * W];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        const double ov = red[w];
        const int oi = (int)red[W + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        amin = fmin(amin, red[2 * W + w]);
        fm = fmin(fm, red[3 * W + w]);
    }
    BlockInfo info;
    info.K = (bi == 0x7fffffff) ? 0 : bi;      // library.py:371-375: levels [0, argmax)
    info.alt_min = amin;
    info.bad = 0;
    info.unmag = 0;
    info.a0 = 0.0;
    info.inv_w = 0.0;
    const int K = info.K;
    if (K == 0) {
        info.bad = PRHF_STATUS_PEAK0;
        return info;
    }
    // ---- phase 2: nodes (values, np.interp slopes, f_N^2), |B| max, negative density ------
    double bmax = 0.0;
    int neg = 0;
    for (int k = tid; k < K; k += THREADS) {
        const double a = alt[k], d = den[k], b = bmag[k], p = bpsi[k];
        Node nd;
        nd.alt = a; nd.den = d; nd.b = b; nd.psi = p;
        if (k + 1 < K) {
            const double da = alt[k + 1] - a;
            nd.sden = (den[k + 1] - d) / da;       // numpy arr_interp: (dy[i+1]-dy[i])/(dx[i+1]-dx[i])
            nd.sb = (bmag[k + 1] - b) / da;
            nd.spsi = (bpsi[k + 1] - p) / da;
        } else {
            nd.sden = 0.0; nd.sb = 0.0; nd.spsi = 0.0;
        }
        const double fn = sqrt(d) * kPlasma;       // :96
        nd.pf2 = fn * fn;                          // :136 numerator
        nodes[k] = nd;
        bmax = fmax(bmax, fabs(b));
        neg |= (d < 0.0) ? 1 : 0;
    }
    bmax = wave_max(bmax);
    neg = __any(neg) ? 1 : 0;
    if (lane == 0) {
        red[4 * W + wave] = bmax;
        red[5 * W + wave] = (double)neg;
    }
    __syncthreads();
    bmax = red[4 * W];
    neg = (int)red[5 * W];
#pragma unroll
    for (int w = 1; w < W; ++w) {
        bmax = fmax(bmax, red[4 * W + w]);
        neg |= (int)red[5 * W + w];
    }
    if (neg) info.bad = PRHF_STATUS_NEGDEN;        // library.py:93-94
    // library.py:201: nanmax|Y| < y_tol over the call's whole (F, N) array.  |Y| is largest
    // at the lowest frequency and the strongest field; the node maximum bounds the sampled
    // maximum from above and equals it unless |B| < ~4e-18 T ("Deviations").
    info.unmag = ((kGyro * bmax) / (fm * 1e6) < kUnmagTol) ? 1 : 0;
    // ---- phase 3: hint table: hint[b] = last level with alt <= a0 + b*w ---------------------
    const double a0 = nodes[0].alt;
    const double span = nodes[K - 1].alt - a0;
    const double w = span / (double)kHintBuckets;
    info.a0 = a0;
    info.inv_w = (span > 0.0) ? (double)kHintBuckets / span : 0.0;
    for (int b = tid; b < kHintBuckets; b += THREADS) {
        const double t = a0 + (double)b * w;
        int lo = 0, hi = K - 1;                   // invariant: alt[lo] <= t (alt[0] = a0 <= t)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (nodes[mid].alt <= t) lo = mid; else hi = mid - 1;
        }
        hint[b] = (unsigned short)lo;
    }
    __syncthreads();
    return info;
}

// ---------------------------------------------------------------------------------------
// S7-S11 for grid points [i0, i1) of one pair; returns this wave's partial sum (all lanes).
// ---------------------------------------------------------------------------------------
template <int MODE, int TIER, bool UNMAG>
__device__ __forceinline__ double integrate_chunk(const Node* __restrict__ nodes,
                                                  const unsigned short* __restrict__ hint,
                                                  const BlockInfo& info, const double* __restrict__ mult,
                                                  int n_points, int i0, int i1, double f_hz, double f2,
                                                  double h_refl, int lane) {
#pragma clang fp contract(off)
    const int K = info.K;
    const double a0 = info.a0;
    const double span = h_refl - a0;               // :413 (critical_height - aalt[0])
    double acc = 0.0;
    for (int i = i0 + lane; i < i1; i += 64) {
        const double z = mult[i] * span + a0;      // :413
        double dh = kBackoff;                      // :415-416 last thickness
        if (i + 1 < n_points) dh = (mult[i + 1] * span + a0) - z;
        // segment of np.interp: alt[j] <= z < alt[j+1]
        int bucket = (int)((z - a0) * info.inv_w);
        bucket = bucket < 0 ? 0 : (bucket > kHintBuckets - 1 ? kHintBuckets - 1 : bucket);
        int j = hint[bucket];
        while (j > 0 && z < nodes[j].alt) --j;
        while (j + 1 < K && z >= nodes[j + 1].alt) ++j;
        const Node nd = nodes[j];
        double dz = z - nd.alt;
        if (dz < 0.0) dz = 0.0;                    // z below the first level: left value
        const double den = nd.sden * dz + nd.den;  // numpy arr_interp: slope*(x - xp[j]) + fp[j]
        double mup;
        if (UNMAG) {
            const double fn = sqrt(den) * kPlasma;
            mup = group_index_unmagnetised((fn * fn) / f2);
        } else {
            const double b = nd.sb * dz + nd.b;
            const double psi = nd.spsi * dz + nd.psi;
            const double fn = sqrt(den) * kPlasma;     // :96
            const double X = (fn * fn) / f2;           // :136
            const double Y = (kGyro * b) / f_hz;       // :157
            mup = group_index_faithful<MODE>(X, Y, psi);
        }
        const double term = mup * dh;              // :288
        if (term == term) acc = acc + term;        // nansum
    }
    return wave_sum(acc);
}

template <int MODE, int TIER, int THREADS>
__device__ __forceinline__ void run_items(const KArgs& a, const SegDev& sg, const Node* nodes,
                                          const unsigned short* hint, const BlockInfo& info,
                                          long long prof_local, int block_in_prof) {
    constexpr int W = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int F = (int)a.n_freq;
    const int C = sg.chunks;
    const long long T = (long long)F * C;
    const double* mult = a.mult + sg.mult_off;
    const long long pair_base = prof_local * F;
    for (long long t = (long long)block_in_prof * W + wave; t < T; t += (long long)sg.blocks_per_prof * W) {
        const int f = (int)(t % F);
        const int c = (int)(t / F);
        double result = qnan();
        bool reflects = false;
        if (!info.bad) {
            const double f_hz = a.freq[f] * 1e6;               // :491
            const double f2 = f_hz * f_hz;                     // f**2
            double h;
            reflects = reflection_height<MODE>(nodes, info.K, f_hz, f2, lane, &h);
            if (reflects) {
                const int i0 = c * sg.chunk_len;
                const int i1 = min(sg.n_points, i0 + sg.chunk_len);
                if (info.unmag)
                    result = integrate_chunk<MODE, TIER, true>(nodes, hint, info, mult, sg.n_points, i0, i1,
                                                               f_hz, f2, h, lane);
                else
                    result = integrate_chunk<MODE, TIER, false>(nodes, hint, info, mult, sg.n_points, i0, i1,
                                                                f_hz, f2, h, lane);
            }
        }
        if (lane == 0) {
            if (C == 1) {
                // :290-292: exact zero means every term was NaN -> NaN; then add min(alt)
                const double vh = (reflects && result != 0.0) ? result + info.alt_min : qnan();
                a.out[sg.out_off + pair_base + f] = vh;
            } else {
                a.partial[sg.partial_off + (pair_base + f) * C + c] = reflects ? result : qnan();
            }
        }
    }
}

}  // namespace

template <int TIER, int THREADS>
__global__ __launch_bounds__(THREADS) void vfo_kernel(const KArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n_alt = (int)a.n_alt;
    Node* nodes = reinterpret_cast<Node*>(smem);
    unsigned short* hint = reinterpret_cast<unsigned short*>(smem + (size_t)n_alt * sizeof(Node));
    double* red = reinterpret_cast<double*>(smem + (size_t)n_alt * sizeof(Node) +
                                            kHintBuckets * sizeof(unsigned short));

    const long long bid = blockIdx.x;
    int s = 0;
    while (s + 1 < a.n_segs && bid >= a.seg[s + 1].block_begin) ++s;
    const SegDev& sg = a.seg[s];
    const long long lb = bid - sg.block_begin;
    const long long prof_local = lb / sg.blocks_per_prof;
    const int block_in_prof = (int)(lb % sg.blocks_per_prof);
    const long long p = sg.prof_begin + prof_local;

    const BlockInfo info = stage_profile<THREADS>(a.den + p * a.prof_stride, a.bmag + p * a.prof_stride,
                                                  a.bpsi + p * a.prof_stride, a.alt + p * a.alt_stride,
                                                  a.freq, (int)a.n_freq, n_alt, nodes, hint, red);
    if (threadIdx.x == 0 && block_in_prof == 0) {
        if (info.bad) atomicOr(a.status, (unsigned)info.bad);
        if (sg.chunks > 1) a.altmin[sg.altmin_off + prof_local] = info.alt_min;
    }
    if (sg.mode == PRHF_KMODE_O)
        run_items<PRHF_KMODE_O, TIER, THREADS>(a, sg, nodes, hint, info, prof_local, block_in_prof);
    else
        run_items<PRHF_KMODE_X, TIER, THREADS>(a, sg, nodes, hint, info, prof_local, block_in_prof);
}

// Chunked pairs: add the chunk sums in a fixed order, then the reference's 0 -> NaN and + min(alt).
__global__ void vfo_finalize_kernel(const KArgs a, int s) {
    const SegDev& sg = a.seg[s];
    const long long n_pairs = (sg.prof_end - sg.prof_begin) * a.n_freq;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_pairs) return;
    const double* part = a.partial + sg.partial_off + t * sg.chunks;
    double sum = 0.0;
    for (int c = 0; c < sg.chunks; ++c) sum = sum + part[c];
    const double amin = a.altmin[sg.altmin_off + t / a.n_freq];
    a.out[sg.out_off + t] = (sum == sum && sum != 0.0) ? sum + amin : qnan();
}

hipError_t launch_vfo(const KArgs& a, long long n_blocks, int tier, size_t lds_bytes, hipStream_t stream) {
    constexpr int THREADS = PRHF_BLOCK_THREADS;
    if (n_blocks <= 0) return hipSuccess;
    if (tier == 0)
        hipLaunchKernelGGL((vfo_kernel<0, THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    else
        hipLaunchKernelGGL((vfo_kernel<1, THREADS>), dim3((unsigned)n_blocks), dim3(THREADS), lds_bytes, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    for (int s = 0; s < a.n_segs; ++s) {
        if (a.seg[s].chunks <= 1) continue;
        const long long n_pairs = (a.seg[s].prof_end - a.seg[s].prof_begin) * a.n_freq;
        const unsigned blocks = (unsigned)((n_pairs + 255) / 256);
        hipLaunchKernelGGL(vfo_finalize_kernel, dim3(blocks), dim3(256), 0, stream, a, s);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t configure_kernels(size_t max_lds_bytes) {
    constexpr int THREADS = PRHF_BLOCK_THREADS;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&vfo_kernel<0, THREADS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&vfo_kernel<1, THREADS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds_bytes);
}

}  // namespace prhf
