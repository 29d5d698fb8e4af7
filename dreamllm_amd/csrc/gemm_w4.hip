// Round 6 (VERDICT r05 "next" #1: the hand-scheduled K loop): a FOUR-wave form of the 256 x 256 x 64 bf16 GEMM -- one wave per SIMD,
// wave tile 128 x 128 on v_mfma_f32_32x32x16_bf16 (sixteen 32 x 32 accumulators = 256 registers of the 512 a lone wave owns), operands by
// LDS-DMA in the buffer form.  Rounds 1-2 measured this geometry 10-25 % SLOWER than the 8-wave kernel and explained it by the ~125
// cycles an LDS-DMA request cost the issuing wave, with nobody to cover for a lone wave.  tools/dma_issue_probe.hip (this round) shows
// that figure belonged to the addressing form: at one wave per SIMD a global_load_lds_dwordx4 with 64-bit per-lane addresses costs ~100
// clk beside 8 MFMAs, buffer_load_dwordx4 ... offen lds ~16.  So the experiment is repeated with what changed:
//   * every instruction that is not an MFMA is a FILLER behind an MFMA: per 64-deep K tile a wave issues 64 MFMAs (32 clk each = the whole
//     K tile's 2048 matrix cycles of its SIMD), 32 ds_read_b128 (the 8 fragments of the next 16-deep k step behind the first 8 MFMAs of every
//     step) and 16 DMA requests (behind the last 8 MFMAs of k steps 3 and 0) -- at most one filler per MFMA gap;
//   * a k step's fragments are requested a whole half step (>= 256 clk) before their wait, the tile barrier sits at the start of the LAST k
//     step of a tile (its fragments are in registers by then), so the first reads of the next tile run under 16 MFMAs;
//   * LDS traffic per K tile: 4 waves x 32 reads = 128 KiB (8-wave kernel: 192 KiB), half the waves at the barrier.
// Forward layout (both operands k-contiguous), full 256-tiles, plain bf16 epilogue: enough to measure the K loop against
// gemm_pipe_kernel (tile code 261 beside 259 in tools/gemm_bench.py); the result decides whether the other layouts follow.
//
// RESULT (profiles/r06_gemm_w4_bench.log, r06_pmc_gemm_w4.txt, r06_gemm_w4_clock.log).  hipcc emits exactly the intended
// stream (ISA checked: 64 MFMAs on 256 AGPR accumulators per K tile, one ds_read_b128 or one `s_add m0` + buffer_load ... lds behind each,
// no v_accvgpr moves, 94 VGPRs, no spills), and the wave parks for 14 % of its cycles instead of 32 % (SQ_WAIT_ANY), matrix pipe busy 76 %
// of the GPU cycles against 74 %.  It is nevertheless 2-4 % SLOWER (1257-1300 TF against 1290-1330 on the forward shapes): both kernels sit on
// the 1.4 kW socket power limit, and the denser stream is answered with a lower clock -- 1.76 GHz sustained against 1.91 GHz for the 8-wave
// kernel (tools/clock_probe.py gemm261 / gemm259).  The GEMM is power-bound, not schedule-bound: at this limit a tighter K loop buys clock
// back only through fewer joules per FLOP, and MFMA 32x32x16 moves twice the accumulator registers per FLOP of 16x16x32.  Not the default;
// kept reachable (tile code 261) with its test as the documented end point of the "hand-scheduled K loop" line of work.
#include "gemm_shared.h"

namespace {

constexpr int W4_TILE = 256 * BK * 2;       // one operand tile: 32 KiB
constexpr int W4_STAGE = 2 * W4_TILE;       // A | B
constexpr int W4_LDS = 2 * W4_STAGE;        // two stages: 128 KiB

// fragment (IDX-th 32-row block, KS-th 16-deep k step) of a k-contiguous image: lane (row = lane & 31, half = lane >> 5) reads chunk
// 2 KS + half of its row; `base` holds chunk `half` (XOR swizzle included), so the k step is an XOR of the address with KS << 5
template <int IDX, int KS>
__device__ __forceinline__ void w4_frag(u32x4& f, uint32_t base) {
    const uint32_t a = base ^ (uint32_t)(KS << 5);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(a), "n"(IDX * 4096));
}

// counted wait for the 8 fragment reads of a k step: at most N younger LDS reads may still be outstanding (a function, not a statement
// inside the generic lambdas below: inline-asm operands there do not count as captures)
template <int N>
__device__ __forceinline__ void w4_wait(u32x4 (&a)[4], u32x4 (&b)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3])
                 : "n"(N)
                 : "memory");
}

__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 128;

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

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- LDS-DMA: 32 groups of 8 rows (1 KiB) per operand tile; wave w issues groups 8w .. 8w + 7 of A and of B (16 requests per K tile).
    // Buffer form: wave-uniform descriptor at (first row of the tile, k0), constant per-lane byte offset.
    uint32_t voA[8], voB[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = (wave * 8 + q) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        voA[q] = (uint32_t)(((int64_t)r * P.lda + c * 8) * 2);
        voB[q] = (uint32_t)(((int64_t)r * P.ldb + c * 8) * 2);
    }
    uint64_t curA = (uint64_t)(uintptr_t)P.A + (uint64_t)(m0 * P.lda * 2);
    uint64_t curB = (uint64_t)(uintptr_t)P.B + (uint64_t)(n0 * P.ldb * 2);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curA, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curB, 0, 0x7fffffff, 0x00020000);
    auto advance = [&]() {   // descriptors of the next K tile
#if defined(W4_DIAG) && W4_DIAG == 2
        return;              // diagnostic: every K tile re-reads tile 0 (all cache hits; wrong results)
#endif
        curA += BK * 2;
        curB += BK * 2;
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curA, 0, 0x7fffffff, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curB, 0, 0x7fffffff, 0x00020000);
    };
    // request q (0-7: A groups, 8-15: B groups) of the tile rsA / rsB point at, into stage `buf`
    auto request = [&](int buf, int q) {
#if defined(W4_DIAG) && W4_DIAG == 1
        if (buf >= 0) return;   // diagnostic: no requests at all (the MFMA + fragment-read + barrier stream alone; wrong results)
#endif
        char* st = smem + buf * W4_STAGE;
        if (q < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * 8 + q) * 1024), 16, (int)voA[q], 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + W4_TILE + (wave * 8 + (q - 8)) * 1024), 16,
                                                     (int)voB[q - 8], 0, 0, 0);
    };

    const uint32_t s0 = lds_addr(smem);
    const uint32_t offA = (uint32_t)kc_off(wm + (lane & 31), lane >> 5);
    const uint32_t offB = (uint32_t)W4_TILE + (uint32_t)kc_off(wn + (lane & 31), lane >> 5);
    u32x4 fa[4][4], fb[4][4];   // the fragments of a whole K tile, one set per 16-deep k step

    const int nt = (int)(P.K / BK);
    // ---- prologue: tiles 0 and 1 requested, tile 0 lands, the fragments of its k steps 0 and 1 are read
#pragma unroll
    for (int q = 0; q < 16; ++q) request(0, q);
    if (nt > 1) {
        advance();
#pragma unroll
        for (int q = 0; q < 16; ++q) request(1, q);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    {
        const uint32_t ab = s0 + offA, bb = s0 + offB;
        static_for<0, 4>([&fa, ab](auto i) { w4_frag<decltype(i)::value, 0>(fa[0][decltype(i)::value], ab); });
        static_for<0, 4>([&fb, bb](auto j) { w4_frag<decltype(j)::value, 0>(fb[0][decltype(j)::value], bb); });
        static_for<0, 4>([&fa, ab](auto i) { w4_frag<decltype(i)::value, 1>(fa[1][decltype(i)::value], ab); });
        static_for<0, 4>([&fb, bb](auto j) { w4_frag<decltype(j)::value, 1>(fb[1][decltype(j)::value], bb); });
    }

    // One K tile = 64 MFMAs, g = 16 ks + n; a wave enters with the fragments of k steps 0 and 1 requested.
    //   g  0-15   read the A fragments of k steps 2, 3 (g 0-7), then the B fragments (g 8-15): by g 32 the whole tile is in registers
    //   g 16      barrier: the A half of stage cb is free; g 24: barrier: the B half is free -- HALF A TILE before its MFMAs end
    //   g 16 + 3r request r of tile t + 2 into stage cb (r 0-7 the A groups, 8-15 the B groups): ONE request per three MFMAs per wave =
    //             64 requests of 1 KiB per 2048 clk for the CU -- the vector L1 moves 64 B/clk, i.e. at most one request per 16 clk, so
    //             eight requests per wave inside 256 clk (the first form of this kernel) queue the four waves behind one another
    //   g 48      tile t + 1 (requested at g 16-61 of the tile before: 0.8-1.5 K tiles of MFMA time ago) has landed: counted vmcnt (the
    //             11 requests of tile t + 2 issued so far may be out), barrier, read the fragments of its k steps 0 (g 48-55), 1 (g 56-63)
    // HAS1: a tile t + 1 follows; HAS2: a tile t + 2 follows.
    auto body = [&](int t, auto has1_c, auto has2_c) {
        constexpr bool HAS1 = decltype(has1_c)::value, HAS2 = decltype(has2_c)::value;
        const int cb = t & 1, nb = cb ^ 1;
        const uint32_t ab = s0 + offA + (uint32_t)(cb * W4_STAGE), bb = s0 + offB + (uint32_t)(cb * W4_STAGE);
        const uint32_t abn = s0 + offA + (uint32_t)(nb * W4_STAGE), bbn = s0 + offB + (uint32_t)(nb * W4_STAGE);
        static_for<0, 4>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            static_for<0, 16>([&](auto nc) {
                constexpr int n = decltype(nc)::value, i = n >> 2, j = n & 3, g = ks * 16 + n;
                if constexpr (g == 0) w4_wait<8>(fa[0], fb[0]);     // k step 1's eight reads may still be out
                if constexpr (g == 16) {
                    w4_wait<8>(fa[1], fb[1]);                        // ... and the A reads of k steps 2, 3 are in; the B reads may be out
                    if constexpr (HAS2) {
                        __builtin_amdgcn_s_barrier();                // every wave has read the A half of stage cb
                        advance();
                    }
                }
                if constexpr (g == 24) {
                    w4_wait<0>(fa[2], fb[2]);
                    w4_wait<0>(fa[3], fb[3]);
                    if constexpr (HAS2) __builtin_amdgcn_s_barrier();   // ... and the B half
                }
                if constexpr (g == 48 && HAS1) {
                    if constexpr (HAS2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();                    // every wave's share of tile t + 1 is in stage nb
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[ks][j]), __builtin_bit_cast(bf16x8, fa[ks][i]), acc[i][j], 0, 0, 0);
                // ---- the fillers behind MFMA g
                if constexpr (g < 4) w4_frag<g, 2>(fa[2][g], ab);
                else if constexpr (g < 8) w4_frag<g - 4, 3>(fa[3][g - 4], ab);
                else if constexpr (g < 12) w4_frag<g - 8, 2>(fb[2][g - 8], bb);
                else if constexpr (g < 16) w4_frag<g - 12, 3>(fb[3][g - 12], bb);
                if constexpr (HAS2 && g >= 16 && (g - 16) % 3 == 0 && (g - 16) / 3 < 16) request(cb, (g - 16) / 3);
                if constexpr (HAS1 && g >= 48) {
                    if constexpr (g < 52) w4_frag<g - 48, 0>(fa[0][g - 48], abn);
                    else if constexpr (g < 56) w4_frag<g - 52, 0>(fb[0][g - 52], bbn);
                    else if constexpr (g < 60) w4_frag<g - 56, 1>(fa[1][g - 56], abn);
                    else w4_frag<g - 60, 1>(fb[1][g - 60], bbn);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    int t = 0;
    for (; t + 2 < nt; ++t) body(t, std::true_type{}, std::true_type{});
    if (t + 1 < nt) {
        body(t, std::true_type{}, std::false_type{});
        ++t;
    }
    body(t, std::false_type{}, std::false_type{});

    // ---- epilogue: operands were swapped (D^T = B^T A^T), so lane (m = lane & 31, h = lane >> 5) holds C[32 i + m][32 j + 8 g + 4 h + e], g = 0..3,
    // e = 0..3.  Staged through a wave-private 8-KiB LDS region as four [64 rows][64 columns] quarters (the image of gemm_epilogue_lds: 8-byte
    // slot c8 of row r at c8 ^ 2 ((r >> 1) & 7)), then 16-byte row-contiguous global stores.
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    char* wl = smem + wave * 8192;
    bf16* C = reinterpret_cast<bf16*>(P.C);
    const int64_t mw = m0 + wm, nw = n0 + wn;
#pragma unroll
    for (int rh = 0; rh < 2; ++rh)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = rh * 2 + ii;
                const int r = ii * 32 + (lane & 31);
                const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][ch * 2 + jj][g * 4 + e] * P.alpha);
                        const int c8 = jj * 8 + g * 2 + (lane >> 5);
                        *reinterpret_cast<bf16x4*>(wl + r * 128 + ((c8 ^ sw) << 3)) = o;
                    }
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), p = lane & 7;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
                st_bf16x8(C + (mw + rh * 64 + row) * P.ldc + nw + ch * 64 + p * 8, v);
            }
        }
}

}  // namespace

// eligibility (checked by the caller): forward layout, M % 256 == 0, N % 256 == 0, K % 64 == 0, bf16 output, no bias / activation /
// residual / accumulate / split-K, ldc % 8 == 0, C 16-byte aligned
int dllm_launch_gemm_w4(const GemmParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_w4_kernel, W4_LDS, lds_ok);
    const int64_t tiles = (P.M / 256) * (P.N / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    hipLaunchKernelGGL(gemm_w4_kernel, dim3((unsigned)tiles), dim3(256), W4_LDS, stream, P);
    return dllm_check_launch();
}
