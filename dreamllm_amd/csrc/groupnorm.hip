// GroupNorm (+ optional SiLU) on NHWC bf16 activations, forward and input-gradient (frozen affine), for the SD UNet:
// ResnetBlock2D norm1/norm2 (+SiLU), Transformer2DModel.norm, conv_norm_out (+SiLU)  [diffusers 0.24, SURVEY.md A.1].
//
// NHWC keeps a pixel's C channels contiguous (the layout the implicit-GEMM conv and the attention token view want), so
// a group's elements are strided: C/32 channels per pixel.  HBM-bound; three launches, all deterministic:
//   1. partial per-channel sums over pixel chunks: thread owns a fixed 8-channel vector, 16-byte coalesced loads
//   2. finalize: per (image, group) reduce chunks x channels -> per-channel affine  y = x*a + b   (a = rstd*gamma,
//      b = beta - mean*rstd*gamma), stored as [N, C] fp32 (and mean/rstd for the backward)
//   3. apply: y = silu?(x*a + b), 16-byte loads/stores.
// Algorithmic traffic: 6 B/element forward (x read twice, y written once).
// Backward (dx only): dz = dy*silu'(z); dxhat = dz*gamma; dx = rstd*(dxhat - mean_g(dxhat) - xhat*mean_g(dxhat*xhat)).
#include "common.h"

namespace {

// grid (chunks, N); block = CV * R threads (CV = C/8 channel vectors, R rows in flight); x viewed [N][HW][C]
// MODE 0: accumulate (x, x^2); MODE 1 (backward): accumulate (dxhat, dxhat*xhat) with xhat from ab_mean/rstd.
// Pivot of the shifted sums (x - pivot) of one (image, group) slice: the MEDIAN of three of its elements (first pixel / first channel,
// middle pixel / middle channel, last pixel / last channel).  Any value near the slice's mean removes the E[x^2] - mean^2 cancellation
// of a slice with |mean| >> std; a single element as the pivot re-introduces it when that element is an outlier (|pivot - mean| >> std,
// ADVICE r05) -- the median of three ignores one outlier.  Every block / thread of a slice computes the same value.
__device__ __forceinline__ float gn_pivot(const bf16* __restrict__ x, int64_t img_off, int HW, int C, int c0, int cpg) {
    const float a = (float)x[img_off + c0];
    const float b = (float)x[img_off + (int64_t)(HW >> 1) * C + c0 + (cpg >> 1)];
    const float c = (float)x[img_off + (int64_t)(HW - 1) * C + c0 + cpg - 1];
    return __builtin_amdgcn_fmed3f(a, b, c);
}

template <int MODE>
__global__ void gn_partial_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                  float* __restrict__ part, int HW, int C, int G, int pix_per_chunk, int act) {
    extern __shared__ float red[];  // [R][CV][16]
    const int CV = C >> 3;
    const int R = blockDim.x / CV;
    const int cv = threadIdx.x % CV, r = threadIdx.x / CV;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int p0 = chunk * pix_per_chunk, p1 = min(HW, p0 + pix_per_chunk);
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    // MODE 0: sums of (x - pivot), pivot = gn_pivot of the (image, group): with E[x^2] - mean^2 on raw values a group whose mean is
    // 100 x its standard deviation (real SD activations get there) loses four digits of the variance in fp32
    float pv[8];
    if (MODE == 0) {
        const int cpg = C / G;
#pragma unroll
        for (int e = 0; e < 8; ++e) pv[e] = gn_pivot(x, (int64_t)n * HW * C, HW, C, ((cv * 8 + e) / cpg) * cpg, cpg);
    }
    float mu[8], rs[8], ga[8], be[8];
    if (MODE == 1) {
        const int cpg = C / G;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cv * 8 + e;
            mu[e] = mean[n * G + c / cpg];
            rs[e] = rstd[n * G + c / cpg];
            ga[e] = (float)gamma[c];
            be[e] = (float)beta[c];
        }
    }
    if (r < R) {
        for (int p = p0 + r; p < p1; p += R) {
            const int64_t off = ((int64_t)n * HW + p) * C + cv * 8;
            const bf16x8 v = ld_bf16x8(x + off);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e] - pv[e];
                    s1[e] += f;
                    s2[e] += f * f;
                }
            } else {
                const bf16x8 d = ld_bf16x8(dy + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)v[e] - mu[e]) * rs[e];
                    float dz = (float)d[e];
                    if (act) {
                        const float z = xh * ga[e] + be[e];
                        const float sg = sigmoid_f(z);
                        dz *= sg * (1.f + z * (1.f - sg));
                    }
                    const float dxh = dz * ga[e];
                    s1[e] += dxh;
                    s2[e] += dxh * xh;
                }
            }
        }
    }
    float* my = red + ((size_t)r * CV + cv) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        my[e] = s1[e];
        my[8 + e] = s2[e];
    }
    __syncthreads();
    if (r == 0) {
        for (int rr = 1; rr < R; ++rr) {
            const float* o = red + ((size_t)rr * CV + cv) * 16;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                s1[e] += o[e];
                s2[e] += o[8 + e];
            }
        }
        float* dst = part + (((int64_t)n * gridDim.x + chunk) * C + cv * 8) * 2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dst[e * 2] = s1[e];
            dst[e * 2 + 1] = s2[e];
        }
    }
}

// one block (64 threads) per (n, g): reduce part[n][chunk][c][2] over chunks and the group's channels.
// MODE 0: -> mean, rstd, and per-channel a/b.  MODE 1: -> c1 = mean_g(dxhat), c2 = mean_g(dxhat*xhat) into out1/out2.
template <int MODE>
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float* __restrict__ part, const bf16* __restrict__ gamma,
                                                         const bf16* __restrict__ beta, float* __restrict__ out1,
                                                         float* __restrict__ out2, float* __restrict__ ab, int nchunks, int HW,
                                                         int C, int G, float eps, const bf16* __restrict__ x) {
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int cpg = C / G;
    const float pivot = (MODE == 0) ? gn_pivot(x, (int64_t)n * HW * C, HW, C, g * cpg, cpg) : 0.f;  // as gn_partial_kernel<0>
    float s1 = 0.f, s2 = 0.f;
    const int items = nchunks * cpg;
    for (int i = threadIdx.x; i < items; i += 64) {
        const int ch = i / cpg, c = g * cpg + i % cpg;
        const float* p = part + (((int64_t)n * nchunks + ch) * C + c) * 2;
        s1 += p[0];
        s2 += p[1];
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    const float cnt = (float)HW * (float)cpg;
    if (MODE == 0) {
        const float dm = s1 / cnt;                                   // mean - pivot
        const float mean = pivot + dm;
        const float var = fmaxf(s2 / cnt - dm * dm, 0.f);
        const float rstd = rsqrtf(var + eps);
        if (threadIdx.x == 0) {
            out1[n * G + g] = mean;
            out2[n * G + g] = rstd;
        }
        for (int i = threadIdx.x; i < cpg; i += 64) {
            const int c = g * cpg + i;
            const float a = rstd * (float)gamma[c];
            ab[((int64_t)n * C + c) * 2] = a;
            ab[((int64_t)n * C + c) * 2 + 1] = (float)beta[c] - mean * a;
        }
    } else if (threadIdx.x == 0) {
        out1[n * G + g] = s1 / cnt;
        out2[n * G + g] = s2 / cnt;
    }
}

// y = act(x * a[n,c] + b[n,c])
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16* __restrict__ x, const float* __restrict__ ab,
                                                       bf16* __restrict__ y, int64_t total_vec, int HW, int C, int act) {
    const int CV = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const int64_t n = i / ((int64_t)CV * HW);
        const bf16x8 v = ld_bf16x8(x + i * 8);
        const float* p = ab + (n * C + cv * 8) * 2;
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = (float)v[e] * p[e * 2] + p[e * 2 + 1];
            if (act) z = silu_f(z);
            o[e] = (bf16)z;
        }
        st_bf16x8(y + i * 8, o);
    }
}

// Round 6: the same pass with the channel vector FIXED per thread: thread (p, cv) of a 512-thread block owns 16-byte vector cv of pixels
// p, p + PPB, ... of one image (blockIdx.y), so its 8 (a, b) pairs live in registers and the loop body is load - 8 fma (+ SiLU) - store: no
// 64-bit division / modulo per vector, no coefficient loads per element (the grid-stride form above spends ~60 integer and address instructions
// per vector and streams at 3.5 TB/s; the training step's VAE tensors are 2 GB each).  C <= 4096.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void gn_apply_cv_kernel(const bf16* __restrict__ x, const float* __restrict__ ab, bf16* __restrict__ y,
                                                            int HW, int C, int act) {
    const int CV = C >> 3, PPB = BLOCK / CV;
    const int tid = threadIdx.x;
    if (tid >= PPB * CV) return;
    const int p0 = tid / CV, cv = tid - p0 * CV;
    const int64_t n = blockIdx.y;
    float a[8], b[8];
    const float* pc = ab + (n * C + cv * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = pc[e * 2];
        b[e] = pc[e * 2 + 1];
    }
    const bf16* xi = x + n * HW * (int64_t)C + cv * 8;
    bf16* yi = y + n * HW * (int64_t)C + cv * 8;
    for (int pix = blockIdx.x * PPB + p0; pix < HW; pix += gridDim.x * PPB) {
        const bf16x8 v = ld_bf16x8(xi + (int64_t)pix * C);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float z = (float)v[e] * a[e] + b[e];
            if (act) z = silu_f(z);
            o[e] = (bf16)z;
        }
        st_bf16x8(yi + (int64_t)pix * C, o);
    }
}

// Small problems (the UNet at batch 2 inside the denoise loop: 0.3 - 16 MB per tensor, L2 / MALL resident): ONE launch, one block
// per (image, group).  Thread (r, j) owns channel pair j of the group and pixels r, r + R, ...: 4-byte loads (a group is C/G
// channels = 20 - 160 contiguous bytes per pixel), no division in the loops; pass 1 sum / sum of squares, block reduce, pass 2
// re-reads the slice (it is in the XCD's L2) and writes y.  Replaces partial + finalize + apply (3 launches, ~20 us) by ~6 us
// where launch count, not bandwidth, is the cost.
__global__ __launch_bounds__(1024) void gn_small_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gamma,
                                                        const bf16* __restrict__ beta, bf16* __restrict__ y, float* __restrict__ mean_o,
                                                        float* __restrict__ rstd_o, int HW, int C, int G, float eps, int act) {
    __shared__ float red[32];
    const int n = blockIdx.x / G, g = blockIdx.x % G;
    const int cpg = C / G, pp = cpg >> 1;
    const int R = 1024 / pp;
    const int r = threadIdx.x / pp, j = threadIdx.x - r * pp;
    const bool live = r < R;
    const int64_t base = (int64_t)n * HW * C + g * cpg + 2 * j;
    const int64_t stride = (int64_t)R * C;
    // sums of (x - pivot), pivot = gn_pivot of the slice (see gn_partial_kernel): no cancellation when |mean| >> std
    const float pivot = gn_pivot(x, (int64_t)n * HW * C, HW, C, g * cpg, cpg);
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const bf16* px = x + base + (int64_t)r * C;
        int p = r;
        for (; p + 3 * R < HW; p += 4 * R, px += 4 * stride) {  // four independent loads in flight per thread
            const bf16x2 v0 = *reinterpret_cast<const bf16x2*>(px), v1 = *reinterpret_cast<const bf16x2*>(px + stride);
            const bf16x2 v2 = *reinterpret_cast<const bf16x2*>(px + 2 * stride), v3 = *reinterpret_cast<const bf16x2*>(px + 3 * stride);
            const float a0 = (float)v0[0] - pivot, b0 = (float)v0[1] - pivot, a1 = (float)v1[0] - pivot, b1 = (float)v1[1] - pivot;
            const float a2 = (float)v2[0] - pivot, b2 = (float)v2[1] - pivot, a3 = (float)v3[0] - pivot, b3 = (float)v3[1] - pivot;
            s1 += (a0 + b0) + (a1 + b1) + (a2 + b2) + (a3 + b3);
            s2 += (a0 * a0 + b0 * b0) + (a1 * a1 + b1 * b1) + (a2 * a2 + b2 * b2) + (a3 * a3 + b3 * b3);
        }
        for (; p < HW; p += R, px += stride) {
            const bf16x2 v = *reinterpret_cast<const bf16x2*>(px);
            const float a = (float)v[0] - pivot, b = (float)v[1] - pivot;
            s1 += a + b;
            s2 += a * a + b * b;
        }
    }
    s1 = block_sum<16>(s1, red);
    s2 = block_sum<16>(s2, red + 16);
    const float cnt = (float)HW * (float)cpg;
    const float dm = s1 / cnt;
    const float mean = pivot + dm;
    const float rstd = rsqrtf(fmaxf(s2 / cnt - dm * dm, 0.f) + eps);
    if (threadIdx.x == 0) {
        mean_o[n * G + g] = mean;
        rstd_o[n * G + g] = rstd;
    }
    if (live) {
        const int c = g * cpg + 2 * j;
        const float a0 = rstd * (float)gamma[c], a1 = rstd * (float)gamma[c + 1];
        const float b0 = (float)beta[c] - mean * a0, b1 = (float)beta[c + 1] - mean * a1;
        const bf16* px = x + base + (int64_t)r * C;
        bf16* py = y + base + (int64_t)r * C;
#pragma unroll 4
        for (int p = r; p < HW; p += R, px += stride, py += stride) {
            const bf16x2 v = *reinterpret_cast<const bf16x2*>(px);
            float z0 = (float)v[0] * a0 + b0, z1 = (float)v[1] * a1 + b1;
            if (act) {
                z0 = silu_f(z0);
                z1 = silu_f(z1);
            }
            bf16x2 o;
            o[0] = (bf16)z0;
            o[1] = (bf16)z1;
            *reinterpret_cast<bf16x2*>(py) = o;
        }
    }
}

// gn_small_kernel spread over S blocks per (image, group) (round 3).  At batch 2 the UNet's GroupNorms are 64 blocks on a 256-CU
// chip, each a chain of ~10 dependent 4-byte load batches (15 us per launch, 61 launches per denoising step = 11 % of the step).  Here
// S blocks take a quarter of the pixels each; their partial sums meet in a 128-byte slot of a caller-owned, zero-initialised `sync`
// buffer: {arrival count, epoch flag, S x (sum, sum of squares)}, all accessed with agent-scope atomics (the S blocks may sit on
// different XCDs, whose L2s are not coherent for plain accesses).  Every block publishes its sums, waits until they have completed,
// takes a ticket; the last arriver zeroes the count and bumps the epoch, the others spin on the epoch they read before arriving.  The
// totals are then summed in slot order by every block: deterministic, identical in all S blocks.  All NB * G * S blocks are
// co-resident by construction (host: at most 512 blocks of 1024 threads); should they not be (a GPU shared between processes), a
// block that waited ~2 ms computes the whole slice's statistics itself: the wait cannot deadlock and cannot produce a wrong result.
template <int S>
__global__ __launch_bounds__(1024) void gn_small_split_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gamma,
                                                              const bf16* __restrict__ beta, bf16* __restrict__ y,
                                                              float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                              unsigned* __restrict__ sync, int HW, int C, int G, float eps, int act) {
    __shared__ float red[32];
    __shared__ float tot[2];
    __shared__ int alone;
    const int grp = blockIdx.x / S, part = blockIdx.x - grp * S;
    const int n = grp / G, g = grp - n * G;
    const int cpg = C / G, pp = cpg >> 1;
    const int R = 1024 / pp;
    const int r = threadIdx.x / pp, j = threadIdx.x - r * pp;
    const bool live = r < R;
    const int per = HW / S, p0 = part * per, p1 = p0 + per;
    const int64_t base = (int64_t)n * HW * C + g * cpg + 2 * j;
    const int64_t stride = (int64_t)R * C;
    unsigned* slot = sync + (int64_t)grp * 32;
    const float pivot = gn_pivot(x, (int64_t)n * HW * C, HW, C, g * cpg, cpg);  // the same pivot in all S blocks of the slice (gn_small_kernel)
    unsigned f0 = 0;
    if (threadIdx.x == 0) {
        f0 = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epoch is read BEFORE this block arrives
    }
    float s1 = 0.f, s2 = 0.f;
    if (live) {
        const bf16* px = x + base + (int64_t)(p0 + r) * C;
        int p = p0 + r;
        for (; p + 3 * R < p1; p += 4 * R, px += 4 * stride) {
            const bf16x2 v0 = *reinterpret_cast<const bf16x2*>(px), v1 = *reinterpret_cast<const bf16x2*>(px + stride);
            const bf16x2 v2 = *reinterpret_cast<const bf16x2*>(px + 2 * stride), v3 = *reinterpret_cast<const bf16x2*>(px + 3 * stride);
            const float a0 = (float)v0[0] - pivot, b0 = (float)v0[1] - pivot, a1 = (float)v1[0] - pivot, b1 = (float)v1[1] - pivot;
            const float a2 = (float)v2[0] - pivot, b2 = (float)v2[1] - pivot, a3 = (float)v3[0] - pivot, b3 = (float)v3[1] - pivot;
            s1 += (a0 + b0) + (a1 + b1) + (a2 + b2) + (a3 + b3);
            s2 += (a0 * a0 + b0 * b0) + (a1 * a1 + b1 * b1) + (a2 * a2 + b2 * b2) + (a3 * a3 + b3 * b3);
        }
        for (; p < p1; p += R, px += stride) {
            const bf16x2 v = *reinterpret_cast<const bf16x2*>(px);
            const float a = (float)v[0] - pivot, b = (float)v[1] - pivot;
            s1 += a + b;
            s2 += a * a + b * b;
        }
    }
    s1 = block_sum<16>(s1, red);
    s2 = block_sum<16>(s2, red + 16);
    if (threadIdx.x == 0) {
        float* fs = reinterpret_cast<float*>(slot + 2);
        __hip_atomic_store(fs + 2 * part, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(fs + 2 * part + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // published before the ticket is taken
        const unsigned old = __hip_atomic_fetch_add(slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        if (old == (unsigned)(S - 1)) {
            __hip_atomic_store(slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(slot + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // ~2 ms of polling, then give up on the partners (a GPU shared with other processes can keep them from being
            // co-resident): the block then computes the statistics of the whole (image, group) slice itself -- slower, never wrong,
            // never stuck.  The ticket protocol stays consistent: the late partners still arrive and the last one resets the count.
            int spin = 0;
            while (__hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == f0 && spin < (1 << 15)) {
                __builtin_amdgcn_s_sleep(8);
                ++spin;
            }
            ok = spin < (1 << 15);
        }
        // Ordering: the partners' sums are read with agent-scope atomic loads (served by memory, not by this XCD's
        // non-coherent L2 lines) issued after the epoch load that ended the spin; VMEM loads of one wave return in order and the
        // publishers drained their stores (vmcnt(0)) before taking the ticket.  The compiler barrier pins the program order of the
        // relaxed loads.  A formal acquire here would emit `buffer_inv sc1` and drop the slice of x this block is about to
        // re-read for the apply pass from L2 -- the reason the protocol is built from relaxed agent-scope atomics.
        asm volatile("" ::: "memory");
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < S; ++q) {
            t1 += __hip_atomic_load(fs + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t2 += __hip_atomic_load(fs + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        tot[0] = t1;
        tot[1] = t2;
        alone = ok ? 0 : 1;
        // the fallback changes the fp32 summation order of this (image, group): it is COUNTED in word 31 of the slot, so a caller (and
        // the tests) can see that it happened instead of trusting that it does not
        if (!ok) __hip_atomic_fetch_add(slot + 31, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (alone) {   // (uniform) statistics over ALL pixels of the slice, by this block alone
        float a1 = 0.f, a2 = 0.f;
        if (live) {
            const bf16* px = x + base + (int64_t)r * C;
            for (int p = r; p < HW; p += R, px += stride) {
                const bf16x2 v = *reinterpret_cast<const bf16x2*>(px);
                const float a = (float)v[0] - pivot, b = (float)v[1] - pivot;
                a1 += a + b;
                a2 += a * a + b * b;
            }
        }
        __syncthreads();
        a1 = block_sum<16>(a1, red);
        a2 = block_sum<16>(a2, red + 16);
        if (threadIdx.x == 0) {
            tot[0] = a1;
            tot[1] = a2;
        }
        __syncthreads();
    }
    const float cnt = (float)HW * (float)cpg;
    const float dm = tot[0] / cnt;
    const float mean = pivot + dm;
    const float rstd = rsqrtf(fmaxf(tot[1] / cnt - dm * dm, 0.f) + eps);
    if (threadIdx.x == 0 && part == 0) {
        mean_o[n * G + g] = mean;
        rstd_o[n * G + g] = rstd;
    }
    if (live) {
        const int c = g * cpg + 2 * j;
        const float a0 = rstd * (float)gamma[c], a1 = rstd * (float)gamma[c + 1];
        const float b0 = (float)beta[c] - mean * a0, b1 = (float)beta[c + 1] - mean * a1;
        const bf16* px = x + base + (int64_t)(p0 + r) * C;
        bf16* py = y + base + (int64_t)(p0 + r) * C;
#pragma unroll 4
        for (int p = p0 + r; p < p1; p += R, px += stride, py += stride) {
            const bf16x2 v = *reinterpret_cast<const bf16x2*>(px);
            float z0 = (float)v[0] * a0 + b0, z1 = (float)v[1] * a1 + b1;
            if (act) {
                z0 = silu_f(z0);
                z1 = silu_f(z1);
            }
            bf16x2 o;
            o[0] = (bf16)z0;
            o[1] = (bf16)z1;
            *reinterpret_cast<bf16x2*>(py) = o;
        }
    }
}

// dx = rstd * (dxhat - c1 - xhat * c2)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ c1, const float* __restrict__ c2,
                                                           const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                                           bf16* __restrict__ dx, int64_t total_vec, int HW, int C, int G,
                                                           int act) {
    const int CV = C >> 3, cpg = C / G;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const int n = (int)(i / ((int64_t)CV * HW));
        const bf16x8 v = ld_bf16x8(x + i * 8), d = ld_bf16x8(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cv * 8 + e;
            const int ng = n * G + c / cpg;
            const float rs = rstd[ng];
            const float xh = ((float)v[e] - mean[ng]) * rs;
            const float ga = (float)gamma[c];
            float dz = (float)d[e];
            if (act) {
                const float z = xh * ga + (float)beta[c];
                const float sg = sigmoid_f(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            o[e] = (bf16)(rs * (dz * ga - c1[ng] - xh * c2[ng]));
        }
        st_bf16x8(dx + i * 8, o);
    }
}

// the same with the channel vector fixed per thread (see gn_apply_cv_kernel): the per-channel statistics, c1 / c2 and the affine parameters of
// the thread's 8 channels are read ONCE; the grid-stride form above pays 8 integer divisions and 6 scalar-table loads per vector
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void gn_bwd_apply_cv_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ c1, const float* __restrict__ c2,
                                                                const bf16* __restrict__ gamma, const bf16* __restrict__ beta,
                                                                bf16* __restrict__ dx, int HW, int C, int G, int act) {
    const int CV = C >> 3, PPB = BLOCK / CV, cpg = C / G;
    const int tid = threadIdx.x;
    if (tid >= PPB * CV) return;
    const int p0 = tid / CV, cv = tid - p0 * CV;
    const int n = blockIdx.y;
    float rs[8], mu[8], k1[8], k2[8], ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cv * 8 + e;
        const int ng = n * G + c / cpg;
        rs[e] = rstd[ng];
        mu[e] = mean[ng];
        k1[e] = c1[ng];
        k2[e] = c2[ng];
        ga[e] = (float)gamma[c];
        be[e] = (float)beta[c];
    }
    const int64_t base = (int64_t)n * HW * C + cv * 8;
    for (int pix = blockIdx.x * PPB + p0; pix < HW; pix += gridDim.x * PPB) {
        const int64_t off = base + (int64_t)pix * C;
        const bf16x8 v = ld_bf16x8(x + off), d = ld_bf16x8(dy + off);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xh = ((float)v[e] - mu[e]) * rs[e];
            float dz = (float)d[e];
            if (act) {
                const float z = xh * ga[e] + be[e];
                const float sg = sigmoid_f(z);
                dz *= sg * (1.f + z * (1.f - sg));
            }
            o[e] = (bf16)(rs[e] * (dz * ga[e] - k1[e] - xh * k2[e]));
        }
        st_bf16x8(dx + off, o);
    }
}

// 2x2 sum pooling on NHWC: backward of nearest-2x upsampling.  in [N,2H,2W,C] -> out [N,H,W,C]
__global__ __launch_bounds__(256) void sumpool2_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int64_t total_vec,
                                                       int H, int W, int C) {
    const int CV = C >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        int64_t t = i / CV;
        const int w = (int)(t % W);
        t /= W;
        const int h = (int)(t % H);
        const int64_t n = t / H;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dh = 0; dh < 2; ++dh)
#pragma unroll
            for (int dw = 0; dw < 2; ++dw) {
                const bf16x8 v = ld_bf16x8(in + (((n * 2 * H + 2 * h + dh) * 2 * W) + 2 * w + dw) * C + cv * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
        st_bf16x8(out + i * 8, o);
    }
}

inline int gn_geometry(int HW, int C, int NB, int* block, int* nchunks, int* ppc) {
    if (C % 8 != 0) return -1;
    const int CV = C / 8;
    if (CV > 1024) return -1;
    int R = CV <= 256 ? 256 / CV : 1;
    *block = CV * R;
    // aim for >= 1024 blocks overall, chunks of >= 16 pixels
    int chunks = (1024 + NB - 1) / NB;
    int p = (HW + chunks - 1) / chunks;
    if (p < 16) p = 16;
    *ppc = p;
    *nchunks = (HW + p - 1) / p;
    return R;
}

}  // namespace

// grid of the *_cv_kernel<512> apply passes: x = pixel chunks (about 4096 blocks in all: ~16 per CU), y = image
static inline dim3 gn_cv_grid(int NB, int HW, int C) {
    const int ppb = 512 / (C >> 3);
    const int need = (HW + ppb - 1) / ppb;
    int gx = 4096 / (NB > 0 ? NB : 1);
    if (gx < 1) gx = 1;
    if (gx > need) gx = need;
    return dim3((unsigned)gx, (unsigned)NB);
}

extern "C" {

// workspace sizes (floats): partials = N * nchunks * C * 2
int64_t dllm_groupnorm_ws_floats(int NB, int HW, int C) {
    int block, nchunks, ppc;
    if (gn_geometry(HW, C, NB, &block, &nchunks, &ppc) < 0) return -1;
    return (int64_t)NB * nchunks * C * 2;
}

// x,y: [NB, HW, C] bf16 (NHWC); mean,rstd: fp32 [NB, G] outputs (kept for backward); ab: fp32 [NB, C, 2] scratch;
// part: fp32 scratch of dllm_groupnorm_ws_floats().  act: 0 none, 1 SiLU.
int dllm_groupnorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, float* ab,
                       float* part, int NB, int HW, int C, int G, float eps, int act, void* stream) {
    if (NB < 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) != 0) return DLLM_ERR_SHAPE;
    if (NB == 0) return DLLM_OK;
    int block, nchunks, ppc;
    const int R = gn_geometry(HW, C, NB, &block, &nchunks, &ppc);
    if (R < 0) return DLLM_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int cpg = C / G;
    if ((int64_t)NB * HW * C <= ((int64_t)1 << 23) && (cpg & 1) == 0 && cpg <= 512) {  // small: one launch (see gn_small_kernel)
        hipLaunchKernelGGL(gn_small_kernel, dim3(NB * G), dim3(1024), 0, s, (const bf16*)x, (const bf16*)gamma, (const bf16*)beta,
                           (bf16*)y, mean, rstd, HW, C, G, eps, act);
        return dllm_check_launch();
    }
    const size_t lds = (size_t)block * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel<0>, dim3(nchunks, NB), dim3(block), lds, s, (const bf16*)x, nullptr, nullptr, nullptr,
                       nullptr, nullptr, part, HW, C, G, ppc, 0);
    hipLaunchKernelGGL(gn_finalize_kernel<0>, dim3(NB * G), dim3(64), 0, s, part, (const bf16*)gamma, (const bf16*)beta, mean,
                       rstd, ab, nchunks, HW, C, G, eps, (const bf16*)x);
    if ((C & 7) == 0 && C <= 4096 && NB <= 65535) {   // channel vector fixed per thread (round 6)
        const dim3 g = gn_cv_grid(NB, HW, C);
        hipLaunchKernelGGL(gn_apply_cv_kernel<512>, g, dim3(512), 0, s, (const bf16*)x, ab, (bf16*)y, HW, C, act);
        return dllm_check_launch();
    }
    const int64_t tv = (int64_t)NB * HW * (C / 8);
    int grid = (int)((tv + 255) / 256 > 8192 ? 8192 : (tv + 255) / 256);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, ab, (bf16*)y, tv, HW, C, act);
    return dllm_check_launch();
}

// Split form of the single-launch path for tiny batches: `sync` = caller-owned int32 buffer of at least NB * G * 32 words, all zero
// before the first use and left consistent (count zero) by every launch; one buffer per stream in flight.  Returns DLLM_ERR_SHAPE
// when the problem is not eligible (the caller then takes dllm_groupnorm_fwd).
int dllm_groupnorm_fwd_split(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int* sync,
                             int NB, int HW, int C, int G, float eps, int act, void* stream) {
    if (NB <= 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) != 0 || sync == nullptr) return DLLM_ERR_SHAPE;
    const int cpg = C / G;
    constexpr int S = 4;
    if ((int64_t)NB * HW * C > ((int64_t)1 << 23) || (cpg & 1) || cpg > 512 || HW < 1024 || (HW % S) != 0 || NB * G * S > 512)
        return DLLM_ERR_SHAPE;
    hipLaunchKernelGGL(gn_small_split_kernel<S>, dim3(NB * G * S), dim3(1024), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)gamma, (const bf16*)beta, (bf16*)y, mean, rstd, reinterpret_cast<unsigned*>(sync), HW, C, G, eps,
                       act);
    return dllm_check_launch();
}

// dx only (affine parameters frozen).  c1,c2: fp32 [NB, G] scratch.
int dllm_groupnorm_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean, const float* rstd,
                       void* dx, float* c1, float* c2, float* part, int NB, int HW, int C, int G, int act, void* stream) {
    if (NB < 0 || HW <= 0 || C <= 0 || G <= 0 || (C % G) != 0) return DLLM_ERR_SHAPE;
    if (NB == 0) return DLLM_OK;
    int block, nchunks, ppc;
    const int R = gn_geometry(HW, C, NB, &block, &nchunks, &ppc);
    if (R < 0) return DLLM_ERR_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)block * 16 * sizeof(float);
    hipLaunchKernelGGL(gn_partial_kernel<1>, dim3(nchunks, NB), dim3(block), lds, s, (const bf16*)x, (const bf16*)dy, mean, rstd,
                       (const bf16*)gamma, (const bf16*)beta, part, HW, C, G, ppc, act);
    hipLaunchKernelGGL(gn_finalize_kernel<1>, dim3(NB * G), dim3(64), 0, s, part, nullptr, nullptr, c1, c2, nullptr, nchunks, HW,
                       C, G, 0.f, (const bf16*)nullptr);
    if ((C & 7) == 0 && C <= 4096 && NB <= 65535) {
        const dim3 g = gn_cv_grid(NB, HW, C);
        hipLaunchKernelGGL(gn_bwd_apply_cv_kernel<512>, g, dim3(512), 0, s, (const bf16*)x, (const bf16*)dy, mean, rstd, c1, c2, (const bf16*)gamma,
                           (const bf16*)beta, (bf16*)dx, HW, C, G, act);
        return dllm_check_launch();
    }
    const int64_t tv = (int64_t)NB * HW * (C / 8);
    int grid = (int)((tv + 255) / 256 > 8192 ? 8192 : (tv + 255) / 256);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, mean, rstd, c1, c2,
                       (const bf16*)gamma, (const bf16*)beta, (bf16*)dx, tv, HW, C, G, act);
    return dllm_check_launch();
}

// in [NB, 2H, 2W, C] -> out [NB, H, W, C] (sum of each 2x2 window)
int dllm_sumpool2_nhwc(const void* in, void* out, int NB, int H, int W, int C, void* stream) {
    if (NB < 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return DLLM_ERR_SHAPE;
    if (NB == 0) return DLLM_OK;
    const int64_t tv = (int64_t)NB * H * W * (C / 8);
    int grid = (int)((tv + 255) / 256 > 8192 ? 8192 : (tv + 255) / 256);
    hipLaunchKernelGGL(sumpool2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)in, (bf16*)out, tv, H, W, C);
    return dllm_check_launch();
}

}  // extern "C"
