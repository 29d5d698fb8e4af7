// Probe (tools only, not part of the library): what does a VALU "filler" cost beside v_mfma_f32_32x32x16_bf16 on gfx950, for one and
// for two waves per SIMD?  Every work-group runs REP clusters of 16 MFMAs (4 accumulators in rotation, operands in registers) with a
// pattern of fillers after each MFMA; wave 0 (and wave 4 of 8-wave groups) of block 0 reports s_memtime cycles per cluster.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/mfma_filler_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

// PAT: 0 bare, 1 two asm v_add per MFMA, 2 one asm v_add, 3 four asm v_add, 4 two v_exp (asm), 5 two v_fma + two v_exp (asm),
// 6 two compiler-visible fmas (sched_barrier per group), 7 two asm v_add placed BEFORE the MFMA, 8 five asm v_add, 9 two asm
// v_cvt_pk_bf16_f32, 10: two asm v_max3
// ROLE (8-wave groups only): waves 4-7 run pattern PATB instead (e.g. a VALU-only stream: PATB = 100, LDS-less)
template <int PAT>
__device__ __forceinline__ void fillers(float (&x)[8], float y) {
    if constexpr (PAT == 1 || PAT == 7) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[1]) : "v"(y));
    } else if constexpr (PAT == 2) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
    } else if constexpr (PAT == 3) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[1]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[3]) : "v"(y));
    } else if constexpr (PAT == 4) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[1]));
    } else if constexpr (PAT == 5) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(y));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[1]) : "v"(y));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[2]));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[3]));
    } else if constexpr (PAT == 6) {
        x[0] = fmaf(x[0], y, y);
        x[1] = fmaf(x[1], y, y);
    } else if constexpr (PAT == 8) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[1]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[3]) : "v"(y));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[4]) : "v"(y));
    } else if constexpr (PAT == 9) {
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[0]) : "v"(x[2]), "v"(y));
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x[1]) : "v"(x[3]), "v"(y));
    } else if constexpr (PAT == 10) {
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(x[2]), "v"(y));
        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(x[3]), "v"(y));
    }
}

template <int PAT>
__device__ __forceinline__ void cluster(f32x16 (&acc)[4], const bf16x8 (&a)[4], const bf16x8 (&b)[4], float (&x)[8], float y) {
    sfor<0, 16>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (PAT == 7) fillers<PAT>(x, y);
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i & 3], 0, 0, 0);
        if constexpr (PAT != 7) fillers<PAT>(x, y);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// VALU-only partner stream of about the online softmax's size: 32 x (fma, exp) + 32 adds + 16 cvt
__device__ __forceinline__ void valu_stream(float (&x)[8], float y) {
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 7]) : "v"(y));
        asm volatile("v_exp_f32 %0, %0" : "+v"(x[(i + 3) & 7]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[(i + 5) & 7]) : "v"(y));
    }
}

// VALU-only streams of ONE kind (96 instructions per iteration): KIND 0 plain (v_add / v_fma), 1 transcendental (v_exp), 2 v_cvt_pk
template <int KIND>
__device__ __forceinline__ void valu_kind(float (&x)[8], float y) {
#pragma unroll
    for (int i = 0; i < 96; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i & 7]) : "v"(y));
        else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i & 7]));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[i & 7]) : "v"(y));
    }
}
template <int KIND, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void probe_valu(const float* in, float* out, uint64_t* stamps, int rep) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[8];
    const float y = in[lane];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[(lane + 7 * i) & 255] * 0.001f;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rep; ++it) {
        valu_kind<KIND>(x, y);
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) stamps[wave] = t1 - t0;
}
template <int KIND, int NW>
void run_valu(const char* tag, const float* in, float* out, uint64_t* stamps) {
    const int rep = 2000;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_valu<KIND, NW>), dim3(256), dim3(NW * 64), 0, 0, in, out, stamps, rep);
    hipDeviceSynchronize();
    uint64_t h[8];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD %d : %6.2f clk per instruction per wave (SIMD: one per %5.2f clk)\n", tag, NW / 4, (double)h[0] / rep / 96,
           (double)h[0] / rep / 96 / (NW / 4));
}

// LDS-read fillers beside MFMAs: LK 0 one ds_read_b128 per MFMA, 1 two ds_read_b64, 2 two ds_read_b64_tr_b16, 3 one ds_read_b64, 4 none
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_;
template <int LK, int NW>
__global__ __launch_bounds__(NW * 64, NW / 4) void probe_lds(const float* in, float* out, uint64_t* stamps, int rep) {
    __shared__ __attribute__((aligned(16))) char lds[NW * 2048];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = in[(lane + r + i) & 255];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)in[(lane * 3 + e + i) & 255];
            b[i][e] = (__bf16)in[(lane * 5 + e + i) & 255];
        }
    }
    for (int i = threadIdx.x; i < NW * 512; i += NW * 64) reinterpret_cast<float*>(lds)[i] = in[i & 255];
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds + wave * 2048;
    const uint32_t a128 = base + lane * 16, a64 = base + lane * 8;
    u32x4_ r4[4];
    u32x2_ r2[8];
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rep; ++it) {
        sfor<0, 16>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[i >> 2], acc[i & 3], 0, 0, 0);
            if constexpr (LK == 0) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[i & 3]) : "v"(a128), "n"((i & 1) * 1024));
            } else if constexpr (LK == 1) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[(2 * i) & 7]) : "v"(a64), "n"((i & 3) * 512));
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[(2 * i + 1) & 7]) : "v"(a64), "n"(((i + 1) & 3) * 512));
            } else if constexpr (LK == 2) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r2[(2 * i) & 7]) : "v"(a64), "n"((i & 3) * 512));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r2[(2 * i + 1) & 7]) : "v"(a64), "n"(((i + 1) & 3) * 512));
            } else if constexpr (LK == 3) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[i & 7]) : "v"(a64), "n"((i & 3) * 512));
            }
            if constexpr ((i & 3) == 3 && LK != 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (LK == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(r4[i])); s += __uint_as_float(r4[i][0]); }
    } else if (LK != 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { asm volatile("" : "+v"(r2[i])); s += __uint_as_float(r2[i][0]); }
    }
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) stamps[wave] = t1 - t0;
}
template <int LK, int NW>
void run_lds(const char* tag, const float* in, float* out, uint64_t* stamps) {
    const int rep = 2000;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe_lds<LK, NW>), dim3(256), dim3(NW * 64), 0, 0, in, out, stamps, rep);
    hipDeviceSynchronize();
    uint64_t h[8];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD %d : wave0 %7.1f clk per cluster (%5.1f per MFMA)", tag, NW / 4, (double)h[0] / rep, (double)h[0] / rep / 16);
    if (NW == 8) printf("   wave4 %7.1f", (double)h[4] / rep);
    printf("\n");
}

template <int PAT, int NW, int MODE>  // MODE 0: every wave runs the cluster; 1: waves >= 4 run the VALU stream instead; 2: waves >= 4 idle
__global__ __launch_bounds__(NW * 64, NW / 4) void probe(const float* in, float* out, uint64_t* stamps, int rep) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc[4];
    bf16x8 a[4], b[4];
    float x[8];
    const float y = in[lane];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = in[(lane + r + i) & 255];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)in[(lane * 3 + e + i) & 255];
            b[i][e] = (__bf16)in[(lane * 5 + e + i) & 255];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = in[(lane + 7 * i) & 255] * 0.001f;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rep; ++it) {
        if (MODE == 0 || wave < 4) {
            cluster<PAT>(acc, a, b, x, y);
        } else if (MODE == 1) {
            valu_stream(x, y);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && lane == 0) stamps[wave] = t1 - t0;
}

template <int PAT, int NW, int MODE>
void run(const char* tag, const float* in, float* out, uint64_t* stamps) {
    const int rep = 2000;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<PAT, NW, MODE>), dim3(256), dim3(NW * 64), 0, 0, in, out, stamps, rep);
    hipDeviceSynchronize();
    uint64_t h[8];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-58s waves/SIMD %d : wave0 %7.1f clk per cluster (%5.1f per MFMA)", tag, NW / 4, (double)h[0] / rep, (double)h[0] / rep / 16);
    if (NW == 8) printf("   wave4 %7.1f clk per iteration", (double)h[4] / rep);
    printf("\n");
}

int main() {
    float *in, *out;
    uint64_t* stamps;
    hipMalloc(&in, 256 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&stamps, 64);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0.01f * (i % 17) - 0.05f;
    hipMemcpy(in, h.data(), 1024, hipMemcpyHostToDevice);
    run<0, 4, 0>("bare MFMAs", in, out, stamps);
    run<1, 4, 0>("MFMA + 2 asm v_add", in, out, stamps);
    run<2, 4, 0>("MFMA + 1 asm v_add", in, out, stamps);
    run<3, 4, 0>("MFMA + 4 asm v_add", in, out, stamps);
    run<8, 4, 0>("MFMA + 5 asm v_add", in, out, stamps);
    run<4, 4, 0>("MFMA + 2 asm v_exp", in, out, stamps);
    run<5, 4, 0>("MFMA + 2 v_fma + 2 v_exp (asm)", in, out, stamps);
    run<6, 4, 0>("MFMA + 2 compiler fmaf", in, out, stamps);
    run<7, 4, 0>("2 asm v_add BEFORE each MFMA", in, out, stamps);
    run<9, 4, 0>("MFMA + 2 asm v_cvt_pk_bf16_f32", in, out, stamps);
    run<10, 4, 0>("MFMA + 2 asm v_max3", in, out, stamps);
    run<0, 8, 0>("bare MFMAs, both waves of a SIMD", in, out, stamps);
    run<1, 8, 0>("MFMA + 2 asm v_add, both waves", in, out, stamps);
    run<0, 8, 2>("bare MFMAs, partner idle", in, out, stamps);
    run<1, 8, 2>("MFMA + 2 asm v_add, partner idle", in, out, stamps);
    run<0, 8, 1>("bare MFMAs beside a VALU-only partner (96 VALU / iter)", in, out, stamps);
    run<1, 8, 1>("MFMA + 2 asm v_add beside a VALU-only partner", in, out, stamps);
    run<5, 8, 1>("MFMA + 2 fma + 2 exp beside a VALU-only partner", in, out, stamps);
    run_lds<4, 4>("MFMA, no LDS read", in, out, stamps);
    run_lds<0, 4>("MFMA + 1 ds_read_b128 (1 KiB per wave)", in, out, stamps);
    run_lds<1, 4>("MFMA + 2 ds_read_b64 (1 KiB per wave)", in, out, stamps);
    run_lds<2, 4>("MFMA + 2 ds_read_b64_tr_b16 (1 KiB per wave)", in, out, stamps);
    run_lds<3, 4>("MFMA + 1 ds_read_b64 (512 B per wave)", in, out, stamps);
    run_lds<0, 8>("MFMA + 1 ds_read_b128, both waves", in, out, stamps);
    run_lds<1, 8>("MFMA + 2 ds_read_b64, both waves", in, out, stamps);
    run_lds<2, 8>("MFMA + 2 ds_read_b64_tr_b16, both waves", in, out, stamps);
    run_valu<0, 4>("VALU only: v_fma_f32", in, out, stamps);
    run_valu<0, 8>("VALU only: v_fma_f32", in, out, stamps);
    run_valu<0, 16>("VALU only: v_fma_f32", in, out, stamps);
    run_valu<1, 4>("VALU only: v_exp_f32", in, out, stamps);
    run_valu<1, 8>("VALU only: v_exp_f32", in, out, stamps);
    run_valu<1, 16>("VALU only: v_exp_f32", in, out, stamps);
    run_valu<2, 4>("VALU only: v_cvt_pk_bf16_f32", in, out, stamps);
    run_valu<2, 8>("VALU only: v_cvt_pk_bf16_f32", in, out, stamps);
    return 0;
}
