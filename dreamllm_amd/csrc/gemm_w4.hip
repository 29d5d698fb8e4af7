// Round 6 (VERDICT r05 "next" #1: the hand-scheduled K loop): FOUR-wave forms of the 256 x 256 x 64 bf16 GEMM -- one wave per SIMD, wave tile
// 128 x 128, 256 accumulator registers in AGPRs, operands by LDS-DMA in the buffer form, two LDS stages.
//
//   gemm_w4m_kernel<AL, BL, EK>  (MFMA 16x16x32; tile code 280; the launcher's choice for the decoder's linears)   -- the product kernel
//   gemm_w4_kernel               (MFMA 32x32x16; tile code 261; forward layout, plain epilogue)                  -- the experiment it grew from
//
// What the round measured, in order (profiles/r06_power_probe*.log, r06_w4*_diag.log, r06_gemm_w4m_sustained.log):
//  1. The first four-wave kernel (one filler per MFMA, tile barrier in the last k step) ran its matrix pipe 76 % busy and was still 2-4 % slower
//     than the 8-wave kernel; both sat on the 1.4 kW socket limit, and the round first read that as "power-bound, not schedule-bound".
//  2. tools/power_probe.py ended that reading: on ZERO operands (no power limit, 2.4 GHz) the 8-wave kernel does 1.45 PF, that four-wave kernel
//     1.51 PF, hipBLASLt's hand-written kernel for the shape 2.15 PF.  Both of ours were bound by something that does not scale with the clock:
//     the turn-around of an LDS stage.  With two stages a tile's requests can only go out once EVERY wave has read the stage they land in, and
//     have to be back one tile later: less than a K tile (0.85 us) for L2 misses that take longer under load.
//  3. hipBLASLt's kernel (disassembled from its code object: same tile, same MFMA count, same two stages) reads a tile's fragments into
//     registers EARLY -- the second k step's while the first one's MFMAs run -- and releases the stage half a tile before its MFMAs end; its
//     requests then have 1.0-1.45 K tiles to land.  That is the pipeline below: by g = 32 of 128 a wave holds the whole tile in 128 fragment
//     registers, the A half of the stage is released at g = 32, the B half at g = 48, the data of tile t + 1 is awaited at g = 96.
//  4. With all 16 requests of a wave inside 512 clk the four waves queue behind one another on the vector L1 (64 B/clk = one 1-KiB request per
//     16 clk for the CU; an all-L2-hit diagnostic still lost 20 %): one request per six MFMAs (96 clk) per wave.  Zero operands: 1.51 -> 1.69
//     (early release) -> 1.92 PF (spread requests) on MFMA 32x32x16.
//  5. Under the power limit the 32x32x16 form gave most of it back (1.36 PF at 1.69 GHz); the same pipeline on MFMA 16x16x32 draws 12 % less
//     per FLOP: 1.47 PF at 1.81-1.88 GHz on the packed gate|up forward against 1.33 (8-wave) and 1.54-1.59 (hipBLASLt) on the same box.
//  6. hipcc details that cost a day's worth of confusion: (a) the MFMAs are inline asm with "+a" accumulators (the builtin spread the 64
//     accumulators over both register files: ~500 v_accvgpr moves per K tile); hipcc then does not know they are MFMAs and read an accumulator
//     one s_nop behind its last MFMA -- empty "+a" statements behind the closing barrier order the epilogue; (b) with every epilogue inlined the
//     K loop ran out of SGPRs, the request descriptors were spilled to VGPRs and every request became a readfirstlane waterfall loop (-11 %):
//     the epilogue family is a template parameter and the generic gemm_epilogue is not instantiated here (the launcher sends full bf16 tiles only).
// Result on the twelve big linears of the step, sustained, against the 8-wave kernel: +2 ... +11 % each, 29.4 -> 27.2 ms in sum; the training
// step 12.06-12.25 -> 12.81 samples/s, GEMM family 0.496-0.506 -> 0.532 of the bf16 peak.  Same bits as the 8-wave kernel (tests).
#include "gemm_shared.h"
#include "gemm_epilogues.h"

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
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");   // the 4-bit counter: at most 15 LDS operations in flight
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


// ---------------------------------------------------------------------------------------------------------------------------------
// gemm_w4m_kernel<AL, BL>: the same pipeline on v_mfma_f32_16x16x32_bf16, all three dense layouts of the training step (forward: both
// operands k-contiguous; input gradient: B n-contiguous; weight gradient: both m / n-contiguous, read with ds_read_b64_tr_b16), every
// epilogue of gemm_pipe_kernel.  Wave tile 128 x 128 = 8 x 8 accumulators of 16 x 16, two 32-deep k steps per K tile, 128 MFMAs of 16 clk,
// 32 fragment reads (one per two MFMAs), 16 requests (one per six MFMAs).  Accumulators, LDS images, DMA source swizzles and the lane ->
// output mapping are those of gemm_pipe_kernel -- a wave holds two of its 128 x 64 tiles side by side and adds the products in the same
// order -- so its epilogues apply per half and the results are bit-identical to the 8-wave kernel's.
constexpr int w4m_i(int g) { return g < 32 ? ((g & 15) >> 2) : 4 + ((g - 32) >> 3); }   // MFMA order inside a k step: A 0-3 x B 0-3, A 0-3 x B 4-7
constexpr int w4m_j(int g) { return g < 16 ? (g & 3) : (g < 32 ? 4 + (g & 3) : ((g - 32) & 7)); }   // (the order the fragments are read in), rows 4-7
constexpr int w4m_cap(int n) { return n > 15 ? 15 : n; }
// schedule points of a K tile (MFMA index g of 128): release of the A / B half of the stage, first request, request spacing
#ifndef W4M_RA
#define W4M_RA 32
#endif
#ifndef W4M_RB
#define W4M_RB 48
#endif
#ifndef W4M_Q0
#define W4M_Q0 32
#endif
#ifndef W4M_QS
#define W4M_QS 6
#endif
constexpr int w4m_req_at(int g) { return (g >= W4M_Q0 && (g - W4M_Q0) % W4M_QS == 0 && (g - W4M_Q0) / W4M_QS < 16) ? (g - W4M_Q0) / W4M_QS : -1; }
constexpr int w4m_reqs_before(int g) { int n = 0; for (int r = 0; r < 16; ++r) n += (W4M_Q0 + W4M_QS * r < g) ? 1 : 0; return n; }
static_assert(W4M_RA >= 24 && (W4M_RA % 2) == 0 && W4M_RB >= 40 && W4M_RB >= W4M_RA + 8 && W4M_Q0 >= W4M_RA && W4M_Q0 + 8 * W4M_QS >= W4M_RB && W4M_Q0 + 15 * W4M_QS < 128, "requests behind the releases");   // lgkmcnt is a 4-bit counter; waiting for one operation more is always safe

// (a function, not a statement inside a generic lambda: inline-asm operands there do not count as captures)
__device__ __forceinline__ void w4m_tie(f32x4 (&a)[8][4]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(a[i][0]), "+a"(a[i][1]), "+a"(a[i][2]), "+a"(a[i][3]));
}

// EK: the epilogue family, a template parameter so that a launch keeps only ITS epilogue's arguments in SGPRs across the K loop (with all of
// them live the descriptors of the LDS-DMA requests were spilled to VGPRs and every request became a waterfall loop: -11 % on the input
// gradients).  0: gemm_epilogue_lds / gemm_epilogue; 1: RoPE on the q and k column tiles (EPI_ROPE_QKV; v tiles plain); 2: SwiGLU forward
// (A_K, B_K) / backward (A_K, B_N).
template <int AL, int BL, int EK>
__global__ __launch_bounds__(256, 1) void gemm_w4m_kernel(GemmParams P) {
    static_assert((AL == A_K || AL == A_M) && (BL == B_K || BL == B_N), "dense operands");
    static_assert(EK == 0 || (EK == 1 && AL == A_K && BL == B_K) || (EK == 2 && AL == A_K), "fused epilogues: forward layout (SwiGLU backward: A_K, B_N)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 256;
    constexpr bool AMC = (AL == A_M), BMC = (BL == B_N);
    constexpr int OA = AMC ? 2 : 1, OB = BMC ? 2 : 1;   // LDS operations per fragment
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 128;

    const int num_pid_m = (int)((P.M + 255) / 256), num_pid_n = (int)((P.N + 255) / 256);
    const int nwg = P.sk_full > 0 ? P.sk_full : num_pid_m * num_pid_n;   // stream-K: the whole rounds (the tail runs on gemm_pipe_tail_kernel)
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pid_m, pid_n;
    pipe_decode_tile(P, wgid, num_pid_m, num_pid_n, pid_m, pid_n);
    const int64_t m0 = (int64_t)pid_m * 256, n0 = (int64_t)pid_n * 256;

    f32x4 acc[2][8][4];   // [column half][16-row block][16-column block of the half]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- LDS-DMA (buffer form): 32 groups of 1 KiB per operand tile, wave w issues groups 8w .. 8w + 7 of A and of B; offsets as in pipe_tile
    constexpr bool glu_map = (EK == 2 && BL == B_K);
    bool rope_map = false;
    if constexpr (EK == 1) rope_map = n0 < P.rope_cols;
    uint32_t voA[8], voB[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int grp = wave * 8 + q;
        if constexpr (AL == A_K) {
            const int r = grp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            const int64_t rr = min((int64_t)r, P.M - 1 - m0);
            voA[q] = (uint32_t)((rr * P.lda + c * 8) * 2);
        } else {
            const int krow = grp * 2 + (lane >> 5);
            const int pos = (lane & 31) * 16;
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);
            const int64_t col = min(m0 + (lbyte >> 1), P.M - 8);
            voA[q] = (uint32_t)((krow * P.lda + col) * 2);
        }
        if constexpr (BL == B_K) {
            const int r = grp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int64_t rr = min((int64_t)r, P.N - 1 - n0);
            if (glu_map) rr = ((r & 32) ? P.glu_F : 0) + (r >> 6) * 32 + (r & 31);
            if (rope_map) rr = (r >> 7) * 128 + ((r >> 6) & 1) * 32 + (r & 31) + ((r & 32) ? 64 : 0);
            voB[q] = (uint32_t)((rr * P.ldb + c * 8) * 2);
        } else {
            const int krow = grp * 2 + (lane >> 5);
            const int pos = (lane & 31) * 16;
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);
            const int64_t col = min(n0 + (lbyte >> 1), P.N - 8);
            voB[q] = (uint32_t)((krow * P.ldb + col) * 2);
        }
    }
    uint64_t curA = (uint64_t)(uintptr_t)P.A + (uint64_t)((AL == A_K ? m0 * P.lda : (int64_t)0) * 2);
    uint64_t curB = (uint64_t)(uintptr_t)P.B + (uint64_t)((BL == B_K ? (glu_map ? (n0 >> 1) : n0) * P.ldb : (int64_t)0) * 2);
    const uint64_t stepA = (uint64_t)((AL == A_K ? (int64_t)BK : (int64_t)BK * P.lda) * 2);
    const uint64_t stepB = (uint64_t)((BL == B_K ? (int64_t)BK : (int64_t)BK * P.ldb) * 2);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curA, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curB, 0, 0x7fffffff, 0x00020000);
    auto advance = [&]() {   // descriptors of the next K tile
        curA += stepA;
        curB += stepB;
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curA, 0, 0x7fffffff, 0x00020000);
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curB, 0, 0x7fffffff, 0x00020000);
    };
    auto request = [&](int buf, int q) {   // request q (0-7: A groups, 8-15: B groups) of the tile rsA / rsB point at, into stage `buf`
        char* st = smem + buf * W4_STAGE;
        if (q < 8)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * 8 + q) * 1024), 16, (int)voA[q], 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + W4_TILE + (wave * 8 + (q - 8)) * 1024), 16,
                                                     (int)voB[q - 8], 0, 0, 0);
    };

    const int lg = lane >> 4, lt = lane & 15;
    const uint32_t s0 = lds_addr(smem);
    const uint32_t offA = AMC ? (uint32_t)mc_off<T>(lg * 8 + (lt >> 2), wm * 2 + (lt & 3) * 8) : (uint32_t)kc_off(wm + lt, lg);
    const uint32_t offB = (uint32_t)W4_TILE + (BMC ? (uint32_t)mc_off<T>(lg * 8 + (lt >> 2), wn * 2 + (lt & 3) * 8) : (uint32_t)kc_off(wn + lt, lg));
    FragR<AMC> fa[2][8];   // the fragments of a whole K tile, one set per 32-deep k step
    FragR<BMC> fb[2][8];

    // the r-th fragment read of k step 0 / 1 from the stage at (ab, bb): k step 0 as A 0-3, B 0-3, B 4-7, A 4-7 (the order its MFMAs start in),
    // k step 1 as A 0-7, B 0-7 (the A half of the stage is released first)
    auto read0 = [&](auto rc, uint32_t ab, uint32_t bb) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < 4) fragr_issue<AMC, r, 0>(fa[0][r], ab);
        else if constexpr (r < 12) fragr_issue<BMC, r - 4, 0>(fb[0][r - 4], bb);
        else fragr_issue<AMC, r - 8, 0>(fa[0][r - 8], ab);
    };
    auto read1 = [&](auto rc, uint32_t ab, uint32_t bb) {
        constexpr int r = decltype(rc)::value;
        if constexpr (r < 8) fragr_issue<AMC, r, 1>(fa[1][r], ab);
        else fragr_issue<BMC, r - 8, 1>(fb[1][r - 8], bb);
    };
    // counted wait (at most N younger LDS operations outstanding), then every fragment of k step KS is tied behind it
    auto wait_set = [&](auto nc, auto ksc) {
        constexpr int N = decltype(nc)::value, KS = decltype(ksc)::value;
        fragr_wait<w4m_cap(N)>(fa[KS][0]);
        static_for<1, 8>([&](auto i) { fragr_touch(fa[KS][decltype(i)::value]); });
        static_for<0, 8>([&](auto j) { fragr_touch(fb[KS][decltype(j)::value]); });
    };

    const int nt = (int)(P.K / BK);
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
    static_for<0, 4>([&](auto rc) { read0(rc, s0 + offA, s0 + offB); });
    static_for<4, 8>([&](auto rc) { read0(rc, s0 + offA, s0 + offB); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (once per output tile; keeps the 4-bit counter far from its limit in front of the loop)
    static_for<8, 16>([&](auto rc) { read0(rc, s0 + offA, s0 + offB); });

    // One K tile = 128 MFMAs, g = 64 ks + n; a wave enters with the last eight fragment reads of k step 0 outstanding at most.
    //   g 0-30 (even)  the 16 fragment reads of k step 1 (A first)
    //   g 32           A reads in: barrier, the A half of stage cb is free; g 48: the B half -- half a tile before its MFMAs end
    //   g 32 + 6 r     request r of tile t + 2 into stage cb: 64 requests of 1 KiB per 2048 clk for the CU (the vector L1 moves 64 B/clk, i.e.
    //                  at most one request per 16 clk: bursts queue the four waves behind one another)
    //   g 96           tile t + 1 (requested 0.8-1.5 K tiles of MFMA time ago) has landed (vmcnt(11): this tile's first 11 requests may be out),
    //                  barrier; g 96-126 (even): the fragment reads of its k step 0
    auto body = [&](int t, auto has1_c, auto has2_c) {
        constexpr bool HAS1 = decltype(has1_c)::value, HAS2 = decltype(has2_c)::value;
        const int cb = t & 1, nb = cb ^ 1;
        const uint32_t ab = s0 + offA + (uint32_t)(cb * W4_STAGE), bb = s0 + offB + (uint32_t)(cb * W4_STAGE);
        const uint32_t abn = s0 + offA + (uint32_t)(nb * W4_STAGE), bbn = s0 + offB + (uint32_t)(nb * W4_STAGE);
        static_for<0, 128>([&](auto gc) {
            constexpr int g = decltype(gc)::value, ks = g >> 6, n = g & 63, i = w4m_i(n), j = w4m_j(n);
            if constexpr (g == 0) wait_set(std::integral_constant<int, 4 * OA + 4 * OB>{}, std::integral_constant<int, 0>{});   // A 0-3, B 0-3 of k step 0 are in
            if constexpr (g == 16) wait_set(std::integral_constant<int, 8 * OA>{}, std::integral_constant<int, 0>{});           // all of k step 0
            if constexpr (g == W4M_RA) {
                wait_set(std::integral_constant<int, (W4M_RA >= 32 ? 8 : (W4M_RA - 16) / 2) * OB>{}, std::integral_constant<int, 1>{});   // the A reads of k step 1 are in; its B reads may be out
                if constexpr (HAS2) {
                    __builtin_amdgcn_s_barrier();
                    advance();
                }
            }
            if constexpr (g == W4M_RB) {
                wait_set(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
                if constexpr (HAS2) __builtin_amdgcn_s_barrier();
            }
            if constexpr (g == 96 && HAS1) {
                if constexpr (HAS2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(w4m_reqs_before(96)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            // (inline asm with the accumulator tied to an AGPR tuple: left to the builtin, hipcc spreads the 64 accumulators over both register
            // files and shuffles ~500 v_accvgpr moves per K tile between them)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j >> 2][i][j & 3]) : "v"(fragr_value(fb[ks][j])), "v"(fragr_value(fa[ks][i])));
            if constexpr (g < 32 && (g & 1) == 0) read1(std::integral_constant<int, g / 2>{}, ab, bb);
            if constexpr (HAS2 && w4m_req_at(g) >= 0) request(cb, w4m_req_at(g));
            if constexpr (HAS1 && g >= 96 && (g & 1) == 0) read0(std::integral_constant<int, (g - 96) / 2>{}, abn, bbn);
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    int t = 0;
    for (; t + 2 < nt; ++t) body(t, std::true_type{}, std::true_type{});
    if (t + 1 < nt) {
        body(t, std::true_type{}, std::false_type{});
        ++t;
    }
    body(t, std::false_type{}, std::false_type{});

    // every wave is done with its fragment reads: the stages become per-wave staging regions
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // hipcc does not know that the asm statements above are MFMAs: left alone it reads an accumulator one s_nop behind its last MFMA (found as
    // a wrong first 16 x 16 block).  These empty statements tie a half's accumulators behind the barrier (and the second half's behind the
    // first half's epilogue: all 256 copied to VGPRs at once spilled 150 of them to scratch).
    static_for<0, 2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        w4m_tie(acc[h]);
        if constexpr (EK == 2) {   // full tiles by construction of the entry points
            if constexpr (BL == B_K) gemm_epilogue_swiglu_fwd<8>(P, acc[h], smem + wave * 16384, m0 + wm, n0, wn + 64 * h, lane);
            else gemm_epilogue_swiglu_bwd<8>(P, acc[h], smem + wave * 16384, m0 + wm, n0 + wn + 64 * h, lane);
        } else if constexpr (EK == 1) {   // full, aligned tiles by construction of the entry point: the v tiles take the staged plain epilogue
            if (rope_map) gemm_epilogue_rope<8>(P, acc[h], smem + wave * 8192, m0 + wm, n0, wn + 64 * h, lane);
            else gemm_epilogue_lds<8>(P, acc[h], smem + wave * 8192, m0 + wm, n0 + wn + 64 * h, lane);
        } else {   // (the launcher sends only problems whose every tile takes the staged epilogue: dllm_w4m_eligible)
            gemm_epilogue_lds<8>(P, acc[h], smem + wave * 8192, m0 + wm, n0 + wn + 64 * h, lane);
        }
    });
}

}  // namespace

// eligibility (checked by the caller): forward layout, M % 256 == 0, N % 256 == 0, K % 64 == 0, bf16 output, no bias / activation /
// residual / accumulate / split-K, ldc % 8 == 0, C 16-byte aligned
int dllm_launch_gemm_w4(const GemmParams& P, hipStream_t stream) {
    static std::atomic<uint64_t> lds_ok{0};
    const int64_t tiles = (P.M / 256) * (P.N / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    dllm_ensure_dyn_lds(&gemm_w4_kernel, W4_LDS, lds_ok);
    hipLaunchKernelGGL(gemm_w4_kernel, dim3((unsigned)tiles), dim3(256), W4_LDS, stream, P);
    return dllm_check_launch();
}

// gemm_w4m_kernel: `ntiles` blocks (all tiles of the grouped order, or the whole rounds P.sk_full of a stream-K plan).  Eligibility (checked by
// launch_gemm in gemm.hip): dense operands in one of the three layouts, K % 64 == 0, 32-bit DMA offsets, no split-K.
// every tile full and staged through LDS (epilogue_lds_ok for all of them), operands dense in one of the three layouts
bool dllm_w4m_eligible(const GemmParams& P, int layout_a, int layout_b) {
    if (!((layout_a == A_K && (layout_b == B_K || layout_b == B_N)) || (layout_a == A_M && layout_b == B_N))) return false;
    return (P.M % 256) == 0 && (P.N % 256) == 0 && (P.K % BK) == 0 && P.K >= BK && !P.out_f32 && P.splitk <= 1 && P.dbg_noload == 0 && (P.ldc & 7) == 0 &&
           (reinterpret_cast<uintptr_t>(P.C) & 15) == 0 && (P.residual == nullptr || (P.ldr & 3) == 0);
}

int dllm_launch_gemm_w4m(const GemmParams& P, int layout_a, int layout_b, int64_t ntiles, hipStream_t stream) {
    static std::atomic<uint64_t> ok[6] = {};
    if (ntiles <= 0 || ntiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    const int ek = P.epi == EPI_ROPE_QKV ? 1 : ((P.epi == EPI_SWIGLU_FWD || P.epi == EPI_SWIGLU_BWD) ? 2 : 0);
#define W4M_LAUNCH(AL_, BL_, EK_, SLOT)                                                                                        \
    do {                                                                                                                       \
        dllm_ensure_dyn_lds(&gemm_w4m_kernel<AL_, BL_, EK_>, W4_LDS, ok[SLOT]);                                               \
        hipLaunchKernelGGL((gemm_w4m_kernel<AL_, BL_, EK_>), dim3((unsigned)ntiles), dim3(256), W4_LDS, stream, P);           \
        return dllm_check_launch();                                                                                            \
    } while (0)
    if (layout_a == A_K && layout_b == B_K) {
        if (ek == 1) W4M_LAUNCH(A_K, B_K, 1, 0);
        if (ek == 2 && P.epi == EPI_SWIGLU_FWD) W4M_LAUNCH(A_K, B_K, 2, 1);
        if (ek == 0) W4M_LAUNCH(A_K, B_K, 0, 2);
    } else if (layout_a == A_K && layout_b == B_N) {
        if (ek == 2 && P.epi == EPI_SWIGLU_BWD) W4M_LAUNCH(A_K, B_N, 2, 3);
        if (ek == 0) W4M_LAUNCH(A_K, B_N, 0, 4);
    } else if (layout_a == A_M && layout_b == B_N) {
        if (ek == 0) W4M_LAUNCH(A_M, B_N, 0, 5);
    }
#undef W4M_LAUNCH
    return DLLM_ERR_SHAPE;
}
