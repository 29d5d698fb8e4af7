// Probe (tools only, not part of the library): what does ONE 1-KiB LDS-DMA request cost the issuing wave beside MFMAs, by
// addressing form?  Round 2 measured ~125 cycles for global_load_lds_dwordx4 with a 64-bit per-lane address
// (profiles/r02_gemm_experiments.md) and every later K-loop experiment ran into that number.  Forms:
//   0  no request (the MFMA stream alone)
//   1  global_load_lds_dwordx4 v[a:a+1], off               64-bit per-lane address (what hipcc emits for the builtin)
//   2  global_load_lds_dwordx4 v_off, s[b:b+1]             64-bit SGPR base + 32-bit per-lane offset
//   3  buffer_load_dwordx4 v_off, s[r:r+3], s_soff offen lds   buffer descriptor + 32-bit per-lane offset + SGPR offset
// Each wave runs REP iterations of { NM MFMAs 16x16x32 (NM accumulators, operands in registers), one request }, at most 8
// requests outstanding (counted vmcnt); source = a 4-MiB window (L2 / Infinity-Cache resident), every CU busy (256 blocks).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dma_issue_probe tools/dma_issue_probe.hip && tools/bin/dma_issue_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

template <int I, int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

template <int FORM, int NW, int NM, int DEPTH = 8, int NRD = 0>
__global__ __launch_bounds__(NW * 64) void probe(const char* src, float* out, uint64_t* stamps, int rep, uint32_t window) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bf16x8 a[4], b;
    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (float)((tid + i + e) & 15));
#pragma unroll
    for (int e = 0; e < 8; ++e) b[e] = (__bf16)(0.002f * (float)((tid + e) & 7));
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // per-lane source: row (lane >> 3) of an 8-row x 128-byte group, 16-byte chunk (lane & 7), row pitch 8 KiB (a K = 4096 bf16 operand)
    const uint32_t voff = (uint32_t)((lane >> 3) * 8192 + (lane & 7) * 16 + wave * 65536);
    // small windows: four blocks share a 1-MiB region (L1 / L2 hits); large windows: every block streams its own slice of `src`
    // `shared` != 0: the 32 blocks of an XCD (blockIdx & 7) walk ONE region together, as the tiles of a GEMM cluster share their panels
    // through the XCD's L2 (shared = 1: all 32 read the same bytes; shared = 2: four groups of 8 blocks, the A-panel / B-panel mix)
    const bool shared = (window >> 31) != 0;
    window &= 0x7fffffffu;
    const char* blk = src + (shared ? (size_t)(blockIdx.x & 7) * (size_t)window + (size_t)((blockIdx.x >> 3) & 3) * 131072u
                                    : (window <= 4096u ? (size_t)(blockIdx.x & 3) * (1u << 20) : (size_t)blockIdx.x * (size_t)window * 1u));
    const char* vaddr = blk + voff;
    const uint32_t lds_dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * 1024u;
    const uint64_t base64 = (uint64_t)(uintptr_t)blk;
    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base64), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base64 >> 32));
    u32x4 rsrc = u32x4{blo, bhi & 0xffffu, 0xffffffffu, 0x00020000u};
    rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]);
    rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
    rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]);
    rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
    const uint64_t sbase = ((uint64_t)bhi << 32) | blo;

    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    uint32_t soff = 0;
    for (int r = 0; r < rep; ++r) {
        sfor<0, NM>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b, acc[i], 0, 0, 0);
        });
        soff = (soff + 128u) & (window - 1u);  // k advances 64 bf16 per request
        if constexpr (NRD > 0) {   // fragment-read traffic beside the request: NRD x ds_read_b128 (one counted wait at the end of the iteration)
            u32x4 fr[NRD];
            const uint32_t ra = lds_dst + (uint32_t)lane * 16u;
            sfor<0, NRD>([&fr, ra](auto jc) {
                constexpr int j = decltype(jc)::value;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fr[j]) : "v"(ra), "n"(j * 2048));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            sfor<0, NRD>([&fr](auto jc) { asm volatile("" ::"v"(fr[decltype(jc)::value])); });
        }
        if constexpr (FORM == 1) {
            const char* p = vaddr + soff;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(lds_dst) : "memory");
        } else if constexpr (FORM == 2) {
            const uint64_t sb = sbase + soff;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sb), "s"(lds_dst) : "memory");
        } else if constexpr (FORM == 3) {
            asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                         : "memory");
        }
        if constexpr (FORM != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + tid] = s + (float)smem[tid & 1023];
    if (blockIdx.x == 17 && lane == 0) stamps[wave] = t1 - t0;
}

template <int FORM, int NW, int NM, int DEPTH = 8, int NRD = 0>
static void run(const char* tag, const char* src, float* out, uint64_t* stamps, int rep, uint32_t window = 4096u) {
    hipFuncSetAttribute((const void*)&probe<FORM, NW, NM, DEPTH, NRD>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int it = 0; it < 2; ++it) {
        hipLaunchKernelGGL((probe<FORM, NW, NM, DEPTH, NRD>), dim3(256), dim3(NW * 64), 65536, 0, src, out, stamps, rep, window);
        hipDeviceSynchronize();
    }
    uint64_t h[8];
    hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-40s w/SIMD %d MFMA/req %2d depth %2d reads %d window %8u%s : wave0 %7.1f clk/iter", tag, NW / 4, NM, DEPTH, NRD, window & 0x7fffffffu,
           (window >> 31) ? " shared/XCD" : "", (double)h[0] / rep);
    if (NW == 8) printf("   wave4 %7.1f", (double)h[4] / rep);
    printf("\n");
}

int main() {
    char* src;
    float* out;
    uint64_t* stamps;
    const size_t SRC = (size_t)256 * (4u << 20) + (16u << 20);   // 256 blocks x 4 MiB (+ slack for the row / wave offsets)
    hipMalloc(&src, SRC);
    hipMemset(src, 0, SRC);
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&stamps, 64);
    const int rep = 4096;
#define ROW(NW, NM)                                                                         \
    run<0, NW, NM>("no request", src, out, stamps, rep);                                    \
    run<1, NW, NM>("global_load_lds 64-bit vaddr", src, out, stamps, rep);                  \
    run<2, NW, NM>("global_load_lds saddr + 32-bit voffset", src, out, stamps, rep);        \
    run<3, NW, NM>("buffer_load ... offen lds (soffset)", src, out, stamps, rep);
    ROW(4, 4)
    ROW(8, 4)
    ROW(4, 8)
    ROW(8, 8)
    ROW(8, 16)
    // the GEMM's regime: 8 waves, 8 MFMAs per request (64 MFMAs + 8 requests per K tile), with fragment reads, by memory level and depth
    const uint32_t W1 = 4096u, W2 = 1u << 18, W3 = 4u << 20;   // L1-resident, L2-resident (256 KiB x 256 blocks = 64 MiB: mostly MALL), streaming
#define ROW2(DEPTH, NRD, W)                                                                      \
    run<0, 8, 8, DEPTH, NRD>("no request", src, out, stamps, rep, W);                            \
    run<1, 8, 8, DEPTH, NRD>("global_load_lds 64-bit vaddr", src, out, stamps, rep, W);          \
    run<3, 8, 8, DEPTH, NRD>("buffer_load ... offen lds", src, out, stamps, rep, W);
    ROW2(8, 3, W1)
    ROW2(8, 3, W2)
    ROW2(8, 3, W3)
    ROW2(4, 3, W3)
    ROW2(16, 3, W3)
    ROW2(8, 0, W3)
    // the GEMM's sharing pattern: the blocks of an XCD read the same region (L2 hits after the first toucher), window = what an XCD walks
    const uint32_t SH = 0x80000000u;
    ROW2(8, 3, SH | (1u << 20))
    ROW2(8, 3, SH | (4u << 20))
    ROW2(8, 3, SH | (16u << 20))
    ROW2(16, 3, SH | (4u << 20))
    ROW2(4, 3, SH | (4u << 20))
    hipError_t e = hipGetLastError();
    printf("last error: %s\n", hipGetErrorString(e));
    return 0;
}
