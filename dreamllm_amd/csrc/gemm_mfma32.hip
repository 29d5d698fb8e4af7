// Experiment (round 4, VERDICT r03 "next" #2a): the 256 x 256 x 64 pipelined LDS-DMA GEMM of gemm.hip with
// v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16 -- same block tile, same 8 waves (2 x 4, wave tile 128 x 64), same
// LDS images and DMA, same software pipeline (16 groups per K tile, one DMA instruction before each of the first 8, fragment reads
// requested one group ahead with counted lgkmcnt waits, the tile barrier before the last group).  What changes is the instruction
// mix: a K tile is 32 matrix instructions per wave instead of 64 (each twice as long), the A / B operand registers are read half as
// often per MAC, the accumulator registers twice as often (K = 16 per instruction instead of 32); LDS fragment traffic is identical
// (16 A + 8 B ds_read_b128 per K tile).  Forward layout only (A and B k-contiguous), full 256-tiles, plain bf16 epilogue: enough
// to measure the K loop against gemm_pipe_kernel on the LLM's forward shapes (tools/gemm_bench.py 259,266); the result decides
// whether the other layouts follow.
#include "gemm_shared.h"

namespace {

constexpr int P32_TILE = 256 * BK * 2;       // one operand tile: 32 KiB
constexpr int P32_STAGE = 2 * P32_TILE;      // A | B
constexpr int P32_LDS = 2 * P32_STAGE;       // two stages: 128 KiB

// fragment (IDX-th 32-row block, KK-th 16-deep k step) of a k-contiguous image: lane (row = lane & 31, half = lane >> 5) reads
// chunk 2 KK + half of its row; base holds chunk `half` (XOR swizzle included), so the k step is an XOR of the address with KK << 5
template <int IDX, int KK>
__device__ __forceinline__ void frag32_issue(FragR<false>& f, uint32_t base) {
    const uint32_t a = base ^ (uint32_t)(KK << 5);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(a), "n"(IDX * 4096));
}

__global__ __launch_bounds__(512, 2) void gemm_pipe32_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MT = 4, NT = 2, NK = 4, NG = NK * MT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / 4) * 128, wn = (wave % 4) * 64;

    const int num_pid_m = (int)(P.M / 256), num_pid_n = (int)(P.N / 256);
    const int nwg = num_pid_m * num_pid_n;
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pid_m, pid_n;
    {
        const int GROUP_M = P.group_m > 0 ? P.group_m : 4;
        const int in_group = GROUP_M * num_pid_n;
        const int group_id = wgid / in_group;
        const int first_m = group_id * GROUP_M;
        const int gsz = min(num_pid_m - first_m, GROUP_M);
        pid_m = first_m + (wgid % in_group) % gsz;
        pid_n = (wgid % in_group) / gsz;
    }
    const int64_t m0 = (int64_t)pid_m * 256, n0 = (int64_t)pid_n * 256;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // DMA: 32 groups of 8 rows (1 KiB) per operand tile, wave w issues groups 4w .. 4w+3 of A and of B
    auto issue_one = [&](int64_t k0, int buf, int q) {
        char* ta = smem + buf * P32_STAGE;
        const int grp = wave * 4 + (q & 3);
        const int r = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        if (q < 4)
            GLDS16(P.A + (m0 + r) * P.lda + k0 + c * 8, ta + grp * 1024);
        else
            GLDS16(P.B + (n0 + r) * P.ldb + k0 + c * 8, ta + P32_TILE + grp * 1024);
    };

    const uint32_t s0 = lds_addr(smem);
    const uint32_t offA = (uint32_t)kc_off(wm + (lane & 31), lane >> 5);
    const uint32_t offB = (uint32_t)P32_TILE + (uint32_t)kc_off(wn + (lane & 31), lane >> 5);
    FragR<false> fa[2];
    FragR<false> fb[2][NT];
    uint32_t ab = s0 + offA, bb = s0 + offB;
    auto first_reads = [&]() {
        static_for<0, NT>([&](auto j) { frag32_issue<decltype(j)::value, 0>(fb[0][decltype(j)::value], bb); });
        frag32_issue<0, 0>(fa[0], ab);
    };

    const int nt = (int)(P.K / BK);
#pragma unroll
    for (int q = 0; q < 8; ++q) issue_one(0, 0, q);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    first_reads();

    for (int t = 0; t < nt; ++t) {
        const bool pf = t + 1 < nt;
        const int64_t kpf = (int64_t)(t + 1) * BK;
        const int nbuf = (t + 1) & 1;
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value, kk = g / MT, i = g % MT;
            if constexpr (g < 8) {
                if (pf) issue_one(kpf, nbuf, g);
            }
            if constexpr (g < NG - 1) {
                constexpr int kn = (g + 1) / MT, in = (g + 1) % MT;
                if constexpr (in == 0)
                    static_for<0, NT>([&](auto j) { frag32_issue<decltype(j)::value, kn>(fb[kn & 1][decltype(j)::value], bb); });
                frag32_issue<in, kn>(fa[(g + 1) & 1], ab);
                fragr_wait<1 + (in == 0 ? NT : 0)>(fa[g & 1]);
            } else {
                fragr_wait<0>(fa[g & 1]);
                if (t + 1 < nt) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    ab = s0 + offA + (uint32_t)(nbuf * P32_STAGE);
                    bb = s0 + offB + (uint32_t)(nbuf * P32_STAGE);
                    first_reads();
                }
            }
            if constexpr (i == 0) static_for<0, NT>([&](auto j) { fragr_touch(fb[kk & 1][decltype(j)::value]); });
            const bf16x8 va = fragr_value(fa[g & 1]);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragr_value(fb[kk & 1][j]), va, acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: operands were swapped (D^T = B^T A^T), so lane (m = lane & 31, h = lane >> 5) holds C[m][n0' + 8 g + 4 h + e],
    // g = 0..3, e = 0..3 of each 32 x 32 block.  Staged through a wave-private 8-KiB LDS region in two 64-row halves (the images of
    // gemm_epilogue_lds: 8-byte slot c8 of row r at c8 ^ 2 ((r >> 1) & 7)), then 16-byte row-contiguous global stores.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* wl = smem + wave * 8192;
    bf16* C = reinterpret_cast<bf16*>(P.C);
    const int64_t mw = m0 + wm, nw = n0 + wn;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = half * 2 + ii;
            const int r = ii * 32 + (lane & 31);
            const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][j][g * 4 + e] * P.alpha);
                    const int c8 = j * 8 + g * 2 + (lane >> 5);
                    *reinterpret_cast<bf16x4*>(wl + r * 128 + ((c8 ^ sw) << 3)) = o;
                }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
            st_bf16x8(C + (mw + half * 64 + row) * P.ldc + nw + p * 8, v);
        }
    }
}

}  // namespace

// eligibility (checked by the caller): forward layout, M % 256 == 0, N % 256 == 0, K % 64 == 0, bf16 output, no bias / activation /
// residual / accumulate / split-K, ldc % 8 == 0, C 16-byte aligned
int dllm_launch_gemm_pipe32(const GemmParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_pipe32_kernel, P32_LDS, lds_ok);
    const int64_t tiles = (P.M / 256) * (P.N / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_pipe32_kernel, dim3((unsigned)tiles), dim3(512), P32_LDS, stream, P);
    return dllm_check_launch();
}
