// Ring-buffered bf16 MFMA GEMM / implicit-GEMM convolution for SMALL GRIDS (gfx950): the UNet's layers inside the denoising loop.
//
// At UNet batch 2 (B_img = 1: M = 128 ... 8192 rows) a launch has 10 ... 400 output tiles and every block's K loop is exposed:
// the register-staged 128-tile kernel (gemm.hip: gemm_bf16_kernel<.., 128>) keeps ONE K tile of global loads in flight per
// block, so each 64-deep K step costs a full memory round trip (~1.5-2 us measured per step against 0.25 us of MFMA work:
// profiles/r03_denoise_kernel_stats.csv, 56 % of the loop).  Little's law per CU: a 128 x 128 x 64 step consumes 32 KiB; at the
// ~1 us latency of a loaded memory system the CU needs ~96 KiB in flight to keep its matrix pipe busy.  Hence:
//   * 128 x 128 x 64 block tile, 8 waves as 4 (M) x 2 (N), wave tile 32 x 64 = 2 x 4 MFMA 16x16x32 tiles (two waves per SIMD:
//     one computes while the other waits on LDS);
//   * a RING of 4 LDS stages of 32 KiB (A 16 KiB | B 16 KiB, the k-contiguous XOR-swizzled image of the family) filled by
//     global_load_lds_dwordx4 (LDS-DMA: no staging registers), THREE stages in flight ahead of the one being consumed; a wave
//     issues 4 DMA instructions per stage (2 A groups + 2 B groups of 8 rows x 128 B), one before each of the K step's four
//     MFMA groups, and waits with a COUNTED s_waitcnt vmcnt(8 / 4 / 0) -- only for the stage the next K step reads;
//   * one raw s_barrier per K step, placed before the step's last MFMA group (whose MFMAs cover the first fragment reads of
//     the next stage), fragment reads as inline asm with counted lgkmcnt (shared with gemm_pipe_kernel);
//   * A operand: k-contiguous rows (nn.Linear forward) or the NHWC conv gather on the DMA's per-lane source address (3x3 / 1x1,
//     C % 64 == 0, plain / fused nearest-2x upsample / transposed stride-2), B operand: k-contiguous weight rows;
//   * split-K over blockIdx.y through fp32 slabs + splitk_reduce_kernel (deterministic), or a direct epilogue (bias, per-image
//     bias, activation, residual);
//   * EPI_GEGLU (diffusers GEGLU, ff.net.0 of BasicTransformerBlock: hidden, gate = proj(x).chunk(2); hidden * gelu(gate)):
//     the B tile's 128 rows are re-mapped on the DMA source address so that every wave holds 32 `hidden` columns and the 32
//     `gate` columns of the SAME outputs; the epilogue multiplies in registers and writes [M, F] -- the [M, 2F] projection never
//     exists in memory and the element-wise launch is gone.
// Reference call sites: the convolutions / linears of diffusers' UNet2DConditionModel as driven by
// omni/models/dreamllm/modeling_plugins.py:806-839.
#include "gemm_shared.h"

namespace {

constexpr int RING_TILE = 128 * BK * 2;          // one operand tile: 16 KiB
constexpr int RING_STAGE = 2 * RING_TILE;        // A | B: 32 KiB; the ring has NS = 4 stages (128 KiB, one block per CU) or NS = 2
                                                 // (64 KiB, TWO blocks per CU: short reductions on multi-round grids, where a block's
                                                 // prologue and epilogue would otherwise sit idle next to nothing -- [65536, 320, 320]
                                                 // is 6 rounds of 5 K steps each)

struct RingConv {
    int64_t pix_off[2];   // element offset of (img, oh*stride - pad, ow*stride - pad, 0) of the lane's row in each of the wave's 2 A groups
    unsigned tapmask[2];  // bit (kh*KW + kw): the tap reads inside the image (shift modes: and, for `even_only`, an even position)
    unsigned org[2];      // shift modes: (ih0 + 2) << 16 | (iw0 + 2)
};

template <bool SHIFT>
__device__ __forceinline__ void ring_conv_init(RingConv& d, const ConvGeom& g, int64_t m0, int64_t M, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int64_t m = m0 + (wave * 2 + q) * 8 + (lane >> 3);
        d.pix_off[q] = 0;
        d.tapmask[q] = 0;
        d.org[q] = 0;
        if (m < M) {
            const int64_t hw = (int64_t)g.OH * g.OW;
            const int64_t img = m / hw;
            const int rem = (int)(m - img * hw);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            d.pix_off[q] = img * (int64_t)g.H * g.W * g.C + (SHIFT ? (int64_t)0 : ((int64_t)ih0 * g.W + iw0) * g.C);
            if constexpr (SHIFT) d.org[q] = ((unsigned)(ih0 + 2) << 16) | (unsigned)(iw0 + 2);
            unsigned mk = 0;
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw) {
                    int ih = ih0 + kh, iw = iw0 + kw;
                    bool ok = ih >= 0 && iw >= 0;
                    if (SHIFT) {  // logical grid = 2x the physical one (nearest-2x upsample, or the zero-stuffed grid of a stride-2 dgrad)
                        if (g.even_only) ok = ok && ((ih & 1) == 0) && ((iw & 1) == 0);
                        ih >>= 1;
                        iw >>= 1;
                    }
                    if (ok && ih < g.H && iw < g.W) mk |= 1u << (kh * g.KW + kw);
                }
            d.tapmask[q] = mk;
        }
    }
}

// AL: A_K (k-contiguous rows), A_CONV (plain gather), A_CONVS (shift gather).  GLU: EPI_GEGLU column pairing (P.N = F outputs).
template <int AL, bool GLU, int RING_NS = 4>
__global__ __launch_bounds__(512, 2) void gemm_ring_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128, BN_OUT = GLU ? 64 : 128, MI = 2, NG = 2 * MI;
    constexpr bool CONV = (AL == A_CONV || AL == A_CONVS), SHIFT = (AL == A_CONVS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 64;

    // XCD-aware grouped tile order (the hardware deals consecutive blocks to the 8 XCDs round-robin)
    const int num_pid_m = (int)((P.M + BM - 1) / BM), num_pid_n = (int)((P.N + BN_OUT - 1) / BN_OUT);
    const int nwg = num_pid_m * num_pid_n;
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pid_m, pid_n;
    {
        const int GROUP_M = P.group_m > 0 ? P.group_m : 8;
        const int in_group = GROUP_M * num_pid_n;
        const int group_id = wgid / in_group;
        const int first_m = group_id * GROUP_M;
        const int gsz = min(num_pid_m - first_m, GROUP_M);
        pid_m = first_m + (wgid % in_group) % gsz;
        pid_n = (wgid % in_group) / gsz;
    }
    const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN_OUT;

#ifdef DLLM_BENCH_MODES
    // timing diagnostic (tile codes 269 / 270, bench library only): wave 0 of block 0 stamps s_memtime into 8 KiB of LDS behind the ring
    // -- at points where its LDS queue is empty anyway -- and copies the stamps to the workspace (uint64) at the end
    const bool ts_on = P.dbg_noload == 4 && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && P.ws != nullptr;
    uint64_t* ts_lds = reinterpret_cast<uint64_t*>(smem + RING_NS * RING_STAGE);
#define RING_TS(slot)                                                 \
    if (ts_on) {                                                      \
        const uint64_t tt_ = __builtin_amdgcn_s_memtime();            \
        if (lane == 0 && (slot) < 1000) ts_lds[(slot)] = tt_;         \
    }
#else
#define RING_TS(slot)
#endif
    RING_TS(0)
    // Placement of a K tile's four LDS-DMA requests.  0 (shipped): one in front of each MFMA group.  The bench library can override it
    // per call (tile codes 271-279): 1 all four at the tile start, 2 all four right behind the tile barrier (for the stage RING_NS tiles
    // ahead), 3 two and two; 4 / 5 are the no-DMA / no-MFMA ablations.  Measured (profiles/r04_ring_timeline.log, r04_denoise_ring_early_
    // requests_ab.log): +-3 % on the four-stage ring; on the two-stage ring placement 2 cuts the stamped block's vmcnt wait from 540-680
    // to 140-150 clocks per tile -- and LOSES 6 % on the denoising loop at B_img = 8 (41.5 -> 39.1 steps/s): the burst of requests of
    // one block lands on its co-resident partner.  Shipped: 0 everywhere.
#ifdef DLLM_BENCH_MODES
    const int dma_mode = (P.dbg_noload == 4 && P.sk_w > 0) ? P.sk_w - 1 : 0;
#else
    constexpr int dma_mode = 0;
#endif
    // this block's K tiles [kt0, kt0 + nt)
    int kt0 = 0, nt = (int)(P.K / BK);
    if (P.splitk > 1) {
        const int per = (nt + P.splitk - 1) / P.splitk;
        kt0 = blockIdx.y * per;
        nt = max(0, min(nt - kt0, per));
    }

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-lane DMA sources: the wave's two 8-row groups of each operand tile -------------------------------------------
    // Round 6: requests are buffer_load_dwordx4 ... offen lds -- a wave-uniform descriptor per operand, a 32-bit per-lane byte offset
    // that is computed ONCE (dense operands, plain conv gather) and the K position in the SGPR offset: no 64-bit per-lane address
    // arithmetic per request (tools/dma_issue_probe.hip: a request beside 8 MFMAs costs the issuing wave ~25 clk less in this form).
    // Conv halo: a lane whose tap falls outside the image sends an offset beyond the descriptor's 2^31-byte extent and the hardware
    // returns zeros (the launcher only sends tensors below 2 GiB here) -- the 16-byte zero page and its 64-bit select are gone.
    uint32_t a_vo[2], b_vo[2];
    RingConv cdma;
    if constexpr (CONV) ring_conv_init<SHIFT>(cdma, P.cv, m0, P.M, wave, lane);
    // plain conv gather: offsets are biased by one image row + one pixel so that the (-1, -1) tap of the first pixel stays non-negative
    const int64_t conv_bias = CONV ? ((int64_t)P.cv.W + 1) * P.cv.C : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int r = (wave * 2 + q) * 8 + (lane >> 3);   // row of the 128-row tile image
        const int c = (lane & 7) ^ ((r >> 1) & 7);        // logical 16-byte chunk held at this lane's LDS position
        if constexpr (!CONV) {
            const int64_t rr = min((int64_t)r, P.M - 1 - m0);
            a_vo[q] = (uint32_t)((rr * P.lda + c * 8) * 2);
        } else {
            a_vo[q] = (uint32_t)((cdma.pix_off[q] + (SHIFT ? 0 : conv_bias) + c * 8) * 2);
        }
        int64_t n;
        if constexpr (GLU) {  // tile rows [64 w, 64 w + 32): `hidden` rows of outputs n0 + 32 w + ..; the next 32: their `gate` rows
            const int wv = r >> 6, rr = r & 63;
            n = n0 + wv * 32 + (rr & 31) + (rr >= 32 ? P.N : 0);
        } else {
            n = n0 + r;
            n = n < P.N ? n : P.N - 1;
        }
        b_vo[q] = (uint32_t)(((n - n0) * P.ldb + c * 8) * 2);
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(uintptr_t)((uint64_t)(uintptr_t)P.A + (uint64_t)((CONV ? (SHIFT ? (int64_t)0 : -conv_bias) : m0 * P.lda) * 2)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)((uint64_t)(uintptr_t)P.B + (uint64_t)(n0 * P.ldb * 2)), 0, 0x7fffffff, 0x00020000);
    constexpr uint32_t kOob = 0x80000000u;   // beyond the descriptor's extent: the request returns zeros
    // conv: tap / first channel of the NEXT stage to be issued (advanced incrementally)
    int ctap = 0, cci = 0;
    if constexpr (CONV) {
        ctap = (int)(((int64_t)kt0 * BK) / P.cv.C);
        cci = (int)(((int64_t)kt0 * BK) % P.cv.C);
    }

    // q-th DMA instruction (0, 1: A groups; 2, 3: B groups) of the stage whose K tile starts at element k0, into ring slot `slot`
    auto issue_one = [&](int64_t k0, int slot, int q) {
        char* st = smem + slot * RING_STAGE;   // (slot < RING_NS)
        if (q < 2) {
            uint32_t vo;
            int so;
            if constexpr (!CONV) {
                vo = a_vo[q];
                so = (int)(k0 * 2);
            } else {
                const int kh = ctap / P.cv.KW, kw = ctap - kh * P.cv.KW;
                const bool ok = (cdma.tapmask[q] >> ctap) & 1u;
                if constexpr (SHIFT) {
                    const int ih = ((int)(cdma.org[q] >> 16) - 2 + kh) >> 1, iw = ((int)(cdma.org[q] & 0xffffu) - 2 + kw) >> 1;
                    vo = ok ? a_vo[q] + (uint32_t)((ih * P.cv.W + iw) * P.cv.C * 2) : kOob;
                    so = cci * 2;
                } else {
                    vo = ok ? a_vo[q] : kOob;
                    so = ((kh * P.cv.W + kw) * P.cv.C + cci) * 2;   // wave-uniform: the tap and the first channel of the K tile
                }
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(st + (wave * 2 + q) * 1024), 16, (int)vo, so, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(st + RING_TILE + (wave * 2 + (q - 2)) * 1024), 16,
                                                     (int)b_vo[q - 2], (int)(k0 * 2), 0, 0);
        }
    };
    auto conv_advance = [&]() {
        if constexpr (CONV) {
            cci += BK;
            if (cci >= P.cv.C) {
                cci -= P.cv.C;
                ++ctap;
            }
        }
    };

    if (nt > 0) {
        // ---- prologue: up to three stages in flight ------------------------------------------------------------------------
        const int pre = nt < RING_NS - 1 ? nt : RING_NS - 1;
        for (int s = 0; s < pre; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_one((int64_t)(kt0 + s) * BK, s, q);
            conv_advance();
        }
        if (RING_NS > 3 && pre >= 3)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (RING_NS > 2 && pre == 2)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        RING_TS(1)

        const int lg = lane >> 4, lt = lane & 15;
        const uint32_t s0 = lds_addr(smem);
        const uint32_t offA = (uint32_t)kc_off(wm + lt, lg);
        const uint32_t offB = (uint32_t)RING_TILE + (uint32_t)kc_off(wn + lt, lg);
        FragR<false> fa[2];
        FragR<false> fb[2][4];
        uint32_t ab = s0 + offA, bb = s0 + offB;
        auto first_reads = [&]() {
            static_for<0, 4>([&](auto j) { fragr_issue<false, decltype(j)::value, 0>(fb[0][decltype(j)::value], bb); });
            fragr_issue<false, 0, 0>(fa[0], ab);
        };
        first_reads();

        for (int t = 0; t < nt; ++t) {
            // stage t+3 goes into the slot stage t-1 was read from: every wave finished those reads before the last barrier
            const bool pf = t + (RING_NS - 1) < nt;
            const int64_t kpf = (int64_t)(kt0 + t + (RING_NS - 1)) * BK;
            const int pslot = (t + (RING_NS - 1)) & (RING_NS - 1);
            static_for<0, NG>([&](auto gc) {
                constexpr int g = decltype(gc)::value, kk = g / MI, i = g % MI;
                if (dma_mode == 0 || dma_mode == 5) {
                    if (pf) issue_one(kpf, pslot, g);   // one DMA instruction per MFMA group
                } else if (dma_mode == 1 || (dma_mode == 2 && t == 0)) {   // all four at the start of the tile
                    if constexpr (g == 0) {
                        if (pf) {
                            issue_one(kpf, pslot, 0); issue_one(kpf, pslot, 1); issue_one(kpf, pslot, 2); issue_one(kpf, pslot, 3);
                        }
                    }
                } else if (dma_mode == 3) {   // two and two
                    if constexpr (g == 0 || g == 2) {
                        if (pf) {
                            issue_one(kpf, pslot, g); issue_one(kpf, pslot, g + 1);
                        }
                    }
                }
                if constexpr (g < NG - 1) {
                    constexpr int kn = (g + 1) / MI, in = (g + 1) % MI;
                    if constexpr (in == 0)
                        static_for<0, 4>([&](auto j) { fragr_issue<false, decltype(j)::value, kn>(fb[kn][decltype(j)::value], bb); });
                    fragr_issue<false, in, kn>(fa[(g + 1) & 1], ab);
                    fragr_wait<1 + (in == 0 ? 4 : 0)>(fa[g & 1]);
                } else {
                    fragr_wait<0>(fa[g & 1]);
                    if (t + 1 < nt) {
                        // stage t+1 must have landed (this wave's share; the barrier extends that to every wave's): the stages
                        // issued after it -- at most two -- stay in flight across the barrier
                        const int ahead = nt - 2 - t;
                        RING_TS(2 + 3 * t)
                        if (RING_NS > 3 && ahead >= 2)
                            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        else if (RING_NS > 2 && ahead >= 1)
                            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        RING_TS(3 + 3 * t)
                        __builtin_amdgcn_s_barrier();
                        RING_TS(4 + 3 * t)
                        const uint32_t so = (uint32_t)(((t + 1) & (RING_NS - 1)) * RING_STAGE);
                        ab = s0 + offA + so;
                        bb = s0 + offB + so;
                        if (dma_mode == 2 && t + RING_NS < nt) {   // stage t + NS into the slot tile t was just read from (every wave is
                            if (pf) conv_advance();                // past the barrier); the stage issued earlier is accounted for first
                            const int64_t k2 = (int64_t)(kt0 + t + RING_NS) * BK;
                            const int s2 = (t + RING_NS) & (RING_NS - 1);
                            issue_one(k2, s2, 0); issue_one(k2, s2, 1); issue_one(k2, s2, 2); issue_one(k2, s2, 3);
                        }
                        first_reads();
                    }
                }
                if constexpr (i == 0) static_for<0, 4>([&](auto j) { fragr_touch(fb[kk][decltype(j)::value]); });
                const bf16x8 va = fragr_value(fa[g & 1]);
#ifdef DLLM_BENCH_MODES
                if (dma_mode == 5) {   // ablation: no matrix instructions (wrong results)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bf16x8 vb = fragr_value(fb[kk][j]);
                        asm volatile("" ::"v"(vb), "v"(va));
                    }
                } else
#endif
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fragr_value(fb[kk][j]), va, acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (pf && !(dma_mode == 2 && t + RING_NS < nt)) conv_advance();
        }
    }

    RING_TS(2 + 3 * (nt > 0 ? nt - 1 : 0))
    if constexpr (GLU) {
        // lane holds, for rows m = m0 + wm + 16 i + (lane & 15): hidden acc[i][0..1] and gate acc[i][2..3] of the output columns
        // n = n0 + 32 (wave & 1) + 16 j + 4 (lane >> 4) + 0..3, j = 0, 1
        bf16* C = reinterpret_cast<bf16*>(P.C);
        const int64_t nb = n0 + (wave & 1) * 32 + (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = m0 + wm + i * 16 + (lane & 15);
            if (m >= P.M) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t n = nb + j * 16;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float h = acc[i][j][e] * P.alpha, gt = acc[i][j + 2][e] * P.alpha;
                    if (P.bias != nullptr) {
                        h += (float)P.bias[n + e];
                        gt += (float)P.bias[P.N + n + e];
                    }
                    o[e] = (bf16)(h * gelu_erf_f(gt));
                }
                st_bf16x4(C + m * P.ldc + n, o);
            }
        }
    } else {
        const bool staged = !P.out_f32 && P.splitk <= 1 && !P.accumulate && (P.N & 7) == 0 && (P.ldc & 7) == 0 &&
                            (reinterpret_cast<uintptr_t>(P.C) & 15) == 0 &&
                            (P.residual == nullptr || ((P.ldr & 7) == 0 && (reinterpret_cast<uintptr_t>(P.residual) & 15) == 0)) &&
                            (P.bias == nullptr || (reinterpret_cast<uintptr_t>(P.bias) & 7) == 0) &&
                            (P.rg_bias == nullptr || (reinterpret_cast<uintptr_t>(P.rg_bias) & 7) == 0);
        if (!staged) {
            // split-K with `counters`: the last K slice of a tile to arrive reduces it inside the launch (gemm_shared.h)
            gemm_epilogue<MI>(P, acc, m0 + wm, n0 + wn, lane, blockIdx.y, pid_m * num_pid_n + pid_n, reinterpret_cast<int*>(smem));
            return;
        }
        // LDS-staged epilogue: the direct form stores 8 bytes per lane (one instruction = 16 rows x 32-byte pieces).  Here the wave
        // passes its 32 x 64 tile through a private 8-KiB fp32 region of the (now idle) ring -- ds_write_b128 in the accumulator
        // layout, 16-byte chunk c of row r at c ^ (r & 15): conflict-free for the 8-lane write groups and the 16-lane read groups
        // -- and leaves with 16-byte stores, 8 rows x 128 contiguous bytes per instruction.  The residual is added in fp32 on the
        // way out (row-contiguous 16-byte loads): one rounding, as in the direct epilogue.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave is done with its fragment reads: the stages are free
        char* wl = smem + wave * 8192;
        const int64_t mw = m0 + wm, nw = n0 + wn;
        bf16* C = reinterpret_cast<bf16*>(P.C);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = nw + j * 16 + (lane >> 4) * 4;
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (P.bias != nullptr && n < P.N) {
                const bf16x4 b = ld_bf16x4(P.bias + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = (float)b[e];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = i * 16 + (lane & 15);
                const int64_t m = mw + r;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * P.alpha + bv[e];
                if (P.rg_bias != nullptr && n < P.N && m < P.M) {
                    const bf16x4 rb = ld_bf16x4(P.rg_bias + (m / P.rg_rows) * P.N + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
                }
                if (P.epi == EPI_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                } else if (P.epi == EPI_QUICK_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
                } else if (P.epi == EPI_SILU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                }
                *reinterpret_cast<f32x4*>(wl + r * 256 + (((j * 4 + (lane >> 4)) ^ (r & 15)) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const int64_t m = mw + row, n = nw + p * 8;
            if (m >= P.M || n >= P.N) continue;
            const f32x4 lo = *reinterpret_cast<const f32x4*>(wl + row * 256 + (((2 * p) ^ (row & 15)) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(wl + row * 256 + (((2 * p + 1) ^ (row & 15)) << 4));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (P.residual != nullptr) {
                const bf16x8 rv = ld_bf16x8(P.residual + m * P.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)rv[e];
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (bf16)v[e];
            st_bf16x8(C + m * P.ldc + n, o);
        }
#ifdef DLLM_BENCH_MODES
        RING_TS(3 + 3 * (nt > 0 ? nt - 1 : 0))
        if (ts_on && lane == 0) {   // stamps: [0] start, [1] prologue done, per K tile t {2+3t before the wait, 3+3t after it, 4+3t after the barrier}
            uint64_t* out = reinterpret_cast<uint64_t*>(P.ws);
            const int n = 4 + 3 * (nt > 0 ? nt - 1 : 0);
            for (int i = 0; i < n && i < 1000; ++i) out[i] = ts_lds[i];
            out[1000] = (uint64_t)nt;
        }
#endif
    }
}

template <int AL, bool GLU>
int launch_ring_t(const GemmParams& P, int two_stage, hipStream_t stream) {
    const int64_t tiles = cdiv64(P.M, 128) * cdiv64(P.N, GLU ? 64 : 128);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_ring_kernel<AL, GLU, 4>, 4 * RING_STAGE, lds_ok);
    const int sk = P.splitk > 1 ? P.splitk : 1;
    // Two stages, two blocks per CU, for grids with more blocks than CUs and K <= 2048: a block's prologue (first DMA round trip) and
    // epilogue (residual loads, LDS staging, stores) then run next to the other block's K loop instead of next to nothing.  Measured
    // at UNet batch 16 against the four-stage ring (profiles/r04_unet_gemm_b16_ring_stages.log): [65536, 320, 320] 35.7 -> 28.4 us,
    // [65536, 2560, 320] 241 -> 194, [16384, 1920, 640] 70.8 -> 57.0, [4096, 10240, 1280] 172 -> 139 (the 256 x 256 pipelined tile:
    // 152), K = 2560 / 5120 tie; on grids of at most one block per CU the deep ring wins ([1024, 1280, 1280] 13.2 vs 17.9 us).
    // Conv gathers keep winning beyond K = 2048 (profiles/r04_unet_conv_b16_ring_stages.log, batch 16: [65536, 320, 2880] 238 -> 175 us
    // against 193 for the 256 x 256 pipelined tile, [4096, 1280, 11520] 252 -> 221, [16384, 320, 2880] stride 2 72 -> 58): no K limit there.
    const bool ns2 = two_stage > 0 || (two_stage == 0 && (P.K <= 32 * BK || AL != A_K) && tiles * sk > (int64_t)dllm_num_cus());
#ifdef DLLM_BENCH_MODES
    if (P.dbg_noload == 4) {   // timing diagnostic: 8 KiB of stamp space behind the ring
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<AL, GLU, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * RING_STAGE + 8192);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring_kernel<AL, GLU, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * RING_STAGE + 8192);
        if (ns2)
            hipLaunchKernelGGL((gemm_ring_kernel<AL, GLU, 2>), dim3((unsigned)tiles, sk), dim3(512), 2 * RING_STAGE + 8192, stream, P);
        else
            hipLaunchKernelGGL((gemm_ring_kernel<AL, GLU, 4>), dim3((unsigned)tiles, sk), dim3(512), 4 * RING_STAGE + 8192, stream, P);
        return dllm_check_launch();
    }
#endif
    if (ns2)
        hipLaunchKernelGGL((gemm_ring_kernel<AL, GLU, 2>), dim3((unsigned)tiles, sk), dim3(512), 2 * RING_STAGE, stream, P);
    else
        hipLaunchKernelGGL((gemm_ring_kernel<AL, GLU, 4>), dim3((unsigned)tiles, sk), dim3(512), 4 * RING_STAGE, stream, P);
    if (sk > 1 && P.counters == nullptr) {
        const int64_t work = P.M * (P.N >> 2);
        const int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, P);
    }
    return dllm_check_launch();
}

}  // namespace

// Preconditions (checked by the caller, gemm.hip: ring_ok): K % 64 == 0, K >= 64, B k-contiguous, conv: C % 64 == 0;
// EPI_GEGLU: P.N = F with F % 64 == 0, no split-K, bf16 output, no residual / per-image bias.
// two_stage: 0 automatic, 1 force the two-stage ring (two blocks per CU), -1 force the four-stage ring
int dllm_launch_gemm_ring(const GemmParams& P, int layout_a, hipStream_t stream, int two_stage) {
    if (P.epi == EPI_GEGLU) {
        if (layout_a != A_K || P.splitk > 1 || (P.N & 63) || P.out_f32 || P.residual != nullptr || P.rg_bias != nullptr || P.accumulate)
            return DLLM_ERR_SHAPE;
        return launch_ring_t<A_K, true>(P, two_stage, stream);
    }
    if (layout_a == A_K) return launch_ring_t<A_K, false>(P, two_stage, stream);
    if (layout_a == A_CONV) {
        if (P.cv.up_shift | P.cv.even_only) return launch_ring_t<A_CONVS, false>(P, two_stage, stream);
        return launch_ring_t<A_CONV, false>(P, two_stage, stream);
    }
    return DLLM_ERR_SHAPE;
}
