// Greedy-decode kernels for gfx950: the batch-1..8 token step of DreamLLMForCausalMLM with a KV cache
// (omni/eval/language_eval/modeling_dreamllm.py:76-97 driving modeling_dreamllm.py:254-400,212-239,1452).
//
// At decode time every Linear is y[M<=8][N] = x[M][K] W[N][K]^T: 2 FLOP per weight byte, i.e. HBM-bound on streaming W once
// (13.2 GB of bf16 weights per token for the 7B model => <= ~600 tokens/s at 8 TB/s).  The MFMA GEMM tiles are the wrong tool
// (a 128-row tile with one useful row, and 32..86 workgroups on 256 CUs), so this file has
//   * dllm_gemv_bf16   : one WAVE per output row n; lanes stride over K with 16-byte loads, UNROLL row chunks in flight before
//                        the first FMA (8 KiB per wave, 16 waves per CU ~ 128 KiB in flight per CU), x re-read from L1/L2,
//                        fp32 accumulate, wave reduction, fused residual add; bf16 or fp32 output (lm_head logits).
//   * dllm_attn_decode : one query token per (batch, head) against the cache [B][S_max][H_kv][D] with a DEVICE-side valid
//                        length (so the launch is identical every step and can live in a hipGraph): split-KV partial
//                        softmax (grid B*H x nsplit) + a combine kernel.  D/8 lanes own one key (16-byte chunks of the row),
//                        64/(D/8) keys per wave per iteration, online softmax in base 2.
// Algorithmic bytes: gemv N*K*2 (+ M*(K+N)*2); attention 2*len*H_kv*D*2 per batch element.
#include "common.h"

namespace {

constexpr float kLog2e = 1.4426950408889634f;

template <int MB, int UNROLL>
__global__ __launch_bounds__(256) void gemv_kernel(const bf16* __restrict__ x, const bf16* __restrict__ W, void* __restrict__ y,
                                                   const bf16* __restrict__ residual, int64_t N, int K, int64_t ldx, int64_t ldw,
                                                   int64_t ldy, int64_t ldr, int out_f32) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const bf16* wrow = W + n * ldw;
    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 512 * UNROLL) {
        bf16x8 w[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {  // all row chunks of this step are requested before any is consumed
            const int k = k0 + u * 512 + lane * 8;
            w[u] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wrow + k)) : zero_bf16x8();
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            if (k < K) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const bf16x8 xv = ld_bf16x8(x + m * ldx + k);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[m] = fmaf((float)w[u][e], (float)xv[e], acc[m]);
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float v = wave_sum(acc[m]);
        if (lane == 0) {
            if (residual != nullptr) v += (float)residual[m * ldr + n];
            if (out_f32)
                reinterpret_cast<float*>(y)[m * ldy + n] = v;
            else
                reinterpret_cast<bf16*>(y)[m * ldy + n] = (bf16)v;
        }
    }
}

template <int MB>
int launch_gemv(const bf16* x, const bf16* W, void* y, const bf16* r, int64_t N, int K, int64_t ldx, int64_t ldw, int64_t ldy,
                int64_t ldr, int out_f32, hipStream_t s) {
    const unsigned grid = (unsigned)cdiv64(N, 4);
    if (K >= 4096)
        hipLaunchKernelGGL((gemv_kernel<MB, 8>), dim3(grid), dim3(256), 0, s, x, W, y, r, N, K, ldx, ldw, ldy, ldr, out_f32);
    else
        hipLaunchKernelGGL((gemv_kernel<MB, 2>), dim3(grid), dim3(256), 0, s, x, W, y, r, N, K, ldx, ldw, ldy, ldr, out_f32);
    return dllm_check_launch();
}

// ---- fused decode GEMVs ---------------------------------------------------------------------------------------------------
// Token step launches drop from 17 to 7 per layer with these (each tiny kernel costs 3-8 us inside the graph):
//   * gemv_fused_kernel<MB, SWIGLU = false>: optional RMSNorm of x folded in (each wave recomputes rstd from the row it reads
//     anyway, with the same lane/vector order as rmsnorm_fwd_kernel so the two roundings t = bf16(x rstd), h = bf16(w t) are
//     reproduced), up to three weight matrices in one launch (q, k, v: rows are concatenated), bf16 or fp32 output.
//   * SWIGLU = true: row n of W0 (gate) and of W1 (up) in the same wave, output act[n] = bf16(silu(bf16 g) * bf16 u) -- the
//     values DreamLLMMLP.forward (modeling_dreamllm.py:237) rounds to.
struct GemvFusedParams {
    const bf16* x;        // [M][ldx]
    const bf16* norm_w;   // [K] or null
    float eps;
    const bf16* W[3];     // [N_i][ldw]
    void* y[3];           // [M][ldy_i]
    int64_t N[3];
    int64_t ldy[3];
    const bf16* residual;  // added to y[0] (single-matrix use) or null
    int64_t ldr;
    int K;
    int64_t ldx, ldw;
    int out_f32;
    // round 6 (gemv_lds_kernel only): x = the merge of split-KV attention partials (ws of dllm_attn_decode[_rope], [M * H][NS][D + 2] floats),
    // computed while the block stages x -- the combine launch between attention and the o projection disappears
    const float* comb_ws;
    int comb_ns, comb_H, comb_D;
};

template <int MB, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_fused_kernel(GemvFusedParams P) {
    constexpr int UNROLL = 4;
    const int lane = threadIdx.x & 63;
    int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    int mat = 0;
    if constexpr (!SWIGLU) {
        if (n >= P.N[0]) {
            n -= P.N[0];
            mat = 1;
            if (n >= P.N[1]) {
                n -= P.N[1];
                mat = 2;
            }
        }
    }
    if (n >= P.N[mat]) return;
    const int K = P.K;
    float rstd[MB];
    if (P.norm_w != nullptr) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float ss = 0.f;
            for (int k = lane * 8; k < K; k += 512) {
                const bf16x8 xv = ld_bf16x8(P.x + m * P.ldx + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) ss += (float)xv[e] * (float)xv[e];
            }
            ss = wave_sum(ss);
            rstd[m] = rsqrtf(ss / (float)K + P.eps);
        }
    }
    const bf16* w0 = P.W[mat] + n * P.ldw;
    const bf16* w1 = SWIGLU ? P.W[1] + n * P.ldw : nullptr;
    float acc[MB], acc1[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = acc1[m] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 512 * UNROLL) {
        bf16x8 wa[UNROLL], wb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            wa[u] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w0 + k)) : zero_bf16x8();
            if constexpr (SWIGLU) wb[u] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(w1 + k)) : zero_bf16x8();
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            if (k < K) {
                bf16x8 nw = zero_bf16x8();
                if (P.norm_w != nullptr) nw = ld_bf16x8(P.norm_w + k);
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const bf16x8 xv = ld_bf16x8(P.x + m * P.ldx + k);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float xe = (float)xv[e];
                        if (P.norm_w != nullptr) {
                            const bf16 t = (bf16)(xe * rstd[m]);          // .to(input_dtype)
                            xe = (float)(bf16)((float)nw[e] * (float)t);  // weight * (.)
                        }
                        acc[m] = fmaf((float)wa[u][e], xe, acc[m]);
                        if constexpr (SWIGLU) acc1[m] = fmaf((float)wb[u][e], xe, acc1[m]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float v = wave_sum(acc[m]);
        float v1 = 0.f;
        if constexpr (SWIGLU) v1 = wave_sum(acc1[m]);
        if (lane == 0) {
            if constexpr (SWIGLU) {
                const float g = (float)(bf16)v, u = (float)(bf16)v1;
                reinterpret_cast<bf16*>(P.y[0])[m * P.ldy[0] + n] = (bf16)(silu_f(g) * u);
            } else {
                if (P.residual != nullptr && mat == 0) v += (float)P.residual[m * P.ldr + n];
                if (P.out_f32)
                    reinterpret_cast<float*>(P.y[mat])[m * P.ldy[mat] + n] = v;
                else
                    reinterpret_cast<bf16*>(P.y[mat])[m * P.ldy[mat] + n] = (bf16)v;
            }
        }
    }
}

template <int MB>
int launch_gemv_fused(const GemvFusedParams& P, int swiglu, hipStream_t s) {
    const int64_t rows = swiglu ? P.N[0] : P.N[0] + P.N[1] + P.N[2];
    const unsigned grid = (unsigned)cdiv64(rows, 4);
    if (swiglu)
        hipLaunchKernelGGL((gemv_fused_kernel<MB, true>), dim3(grid), dim3(256), 0, s, P);
    else
        hipLaunchKernelGGL((gemv_fused_kernel<MB, false>), dim3(grid), dim3(256), 0, s, P);
    return dllm_check_launch();
}

// ---- round 4: block-staged x, R rows per wave, dot2 -------------------------------------------------------------------
// The kernels above re-read x (and, with the norm folded in, the norm weight) from L1/L2 once per WAVE and spend ~10 VALU
// operations per weight element on the two RMSNorm roundings: the q|k|v launch of the 7B decoder streamed its 100 MB at 3.3 TB/s
// and the gate|up launch its 180 MB at 4.4 TB/s (profiles/r04_decode_kernel_stats.csv), 41 % of a token step.  Here a block
//   * requests the first batch of its weight rows BEFORE it touches x (the weights do not depend on the previous kernel's output),
//   * stages x ONCE in LDS -- already normalised when the RMSNorm is folded in: wave 0 computes rstd in the lane / vector order
//     of gemv_fused_kernel / rmsnorm_fwd_kernel, every thread then writes h = bf16(w * bf16(x * rstd)) for its chunks: the same
//     two roundings, once per block instead of once per wave and element,
//   * gives every wave R = 2 (4 for a single matrix) consecutive output rows (more accumulator chains per x fragment read from LDS), the next batch of
//     row chunks requested before the current one is consumed (two named register sets),
//   * multiplies with v_dot2c_f32_bf16 (two bf16 products + fp32 accumulate per instruction): 4 instructions per 16 weight bytes.
// The plain GEMV (o / down projection) runs the same kernel with norm_w = NULL, so "fused == unfused chain" stays bit-exact.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
struct Bf16x8Pairs {
    bf16x2v p[4];
};
__device__ __forceinline__ float dot8(const bf16x8& w, const bf16x8& x, float acc) {
    const Bf16x8Pairs wp = __builtin_bit_cast(Bf16x8Pairs, w), xp = __builtin_bit_cast(Bf16x8Pairs, x);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(wp.p[e], xp.p[e], acc, false);
    return acc;
}

template <int MB, bool SWIGLU>
__global__ __launch_bounds__(256) void gemv_lds_kernel(GemvFusedParams P, int iters) {
    constexpr int R = 2, U = 4;               // rows per wave and row set, 512-element chunks per batch
    constexpr int NW = SWIGLU ? 2 : 1;        // weight streams per row (gate, up)
    constexpr int STEP = 512 * U;
    extern __shared__ __attribute__((aligned(16))) char smem_g[];
    bf16* xs = reinterpret_cast<bf16*>(smem_g);   // [MB][K]
    __shared__ float rstd_s[MB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = P.K;
    const int64_t total_rows = SWIGLU ? P.N[0] : P.N[0] + P.N[1] + P.N[2];
    // a wave owns `iters` consecutive row sets of R rows; its work is the flat sequence of (row set, K batch) pairs
    const int64_t set0 = ((int64_t)blockIdx.x * 4 + wave) * iters;
    const int nbk = (K + STEP - 1) / STEP;
    const int total = iters * nbk;
    // (matrix, row inside it, live) of row r of row set `it`
    auto row_desc = [&](int it, int r, int& mt, int64_t& n, bool& lv) {
        n = (set0 + it) * R + r;
        lv = n < total_rows;
        if (!lv) n = total_rows - 1;          // a valid row: loaded, never stored
        mt = 0;
        if constexpr (!SWIGLU) {
            if (n >= P.N[0]) {
                n -= P.N[0];
                mt = 1;
                if (n >= P.N[1]) {
                    n -= P.N[1];
                    mt = 2;
                }
            }
        }
    };
    auto load_batch = [&](bf16x8 (&dst)[R][NW][U], int j) {
        const int it = j / nbk, k0 = (j - it * nbk) * STEP;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int mt;
            int64_t n;
            bool lv;
            row_desc(it, r, mt, n, lv);
#pragma unroll
            for (int q = 0; q < NW; ++q) {
                const bf16* wr = (SWIGLU ? P.W[q] : P.W[mt]) + n * P.ldw;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + u * 512 + lane * 8;
                    dst[r][q][u] = k < K ? __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wr + k)) : zero_bf16x8();
                }
            }
        }
    };
    bf16x8 wa[R][NW][U], wb[R][NW][U];
    load_batch(wa, 0);   // both register sets are requested before x is touched: the weights do not depend on it
    if (total > 1) load_batch(wb, 1);

    // ---- x -> LDS (normalised when norm_w is given) ------------------------------------------------------------------------
    if (P.norm_w != nullptr) {
        if (wave == 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                float ss = 0.f;
                for (int k = lane * 8; k < K; k += 512) {
                    const bf16x8 xv = ld_bf16x8(P.x + m * P.ldx + k);
#pragma unroll
                    for (int e = 0; e < 8; ++e) ss += (float)xv[e] * (float)xv[e];
                }
                ss = wave_sum(ss);
                if (lane == 0) rstd_s[m] = rsqrtf(ss / (float)K + P.eps);
            }
        }
        __syncthreads();
    }
    if (P.comb_ws != nullptr) {
        // x[m][h * D + d] = attn_decode_combine_kernel's result for (batch m, head h), bit for bit: max over the splits, then l and o summed
        // in split order with the same expressions, one rounding to bf16
        const int D = P.comb_D, NS = P.comb_ns;
        for (int k = threadIdx.x * 8; k < K; k += 2048) {
            const int h = k / D, d0 = k - h * D;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float* src = P.comb_ws + ((int64_t)(m * P.comb_H + h) * NS) * (D + 2);
                float mx = -INFINITY;
                for (int sp = 0; sp < NS; ++sp) mx = fmaxf(mx, src[sp * (D + 2) + D]);
                float l = 0.f, o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = 0.f;
                for (int sp = 0; sp < NS; ++sp) {
                    const float ms = src[sp * (D + 2) + D];
                    if (ms == -INFINITY) continue;
                    const float c = exp2f(ms - mx);
                    l += src[sp * (D + 2) + D + 1] * c;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += src[sp * (D + 2) + d0 + e] * c;
                }
                bf16x8 xv;
#pragma unroll
                for (int e = 0; e < 8; ++e) xv[e] = (bf16)(l > 0.f ? o[e] / l : 0.f);
                *reinterpret_cast<bf16x8*>(xs + (int64_t)m * K + k) = xv;
            }
        }
    } else
    for (int k = threadIdx.x * 8; k < K; k += 2048) {
        bf16x8 nw = zero_bf16x8();
        if (P.norm_w != nullptr) nw = ld_bf16x8(P.norm_w + k);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            bf16x8 xv = ld_bf16x8(P.x + m * P.ldx + k);
            if (P.norm_w != nullptr) {
                const float rs = rstd_s[m];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const bf16 t = (bf16)((float)xv[e] * rs);      // .to(input_dtype)
                    xv[e] = (bf16)((float)nw[e] * (float)t);       // weight * (.)
                }
            }
            *reinterpret_cast<bf16x8*>(xs + (int64_t)m * K + k) = xv;
        }
    }
    __syncthreads();

    float acc[R][NW][MB];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int q = 0; q < NW; ++q)
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[r][q][m] = 0.f;
    // batch j: accumulate; after the last batch of a row set reduce over the wave, store its R rows, clear the accumulators
    auto consume = [&](const bf16x8 (&src)[R][NW][U], int j) {
        const int it = j / nbk, kb = j - it * nbk, k0 = kb * STEP;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * 512 + lane * 8;
            if (k < K) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const bf16x8 xv = *reinterpret_cast<const bf16x8*>(xs + (int64_t)m * K + k);
#pragma unroll
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int q = 0; q < NW; ++q) acc[r][q][m] = dot8(src[r][q][u], xv, acc[r][q][m]);
                }
            }
        }
        if (kb != nbk - 1) return;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int mt;
            int64_t n;
            bool lv;
            row_desc(it, r, mt, n, lv);
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const float v = wave_sum(acc[r][0][m]);
                float v1 = 0.f;
                if constexpr (SWIGLU) v1 = wave_sum(acc[r][1][m]);
                acc[r][0][m] = 0.f;
                if constexpr (SWIGLU) acc[r][1][m] = 0.f;
                if (lane == 0 && lv) {
                    if constexpr (SWIGLU) {
                        const float g = (float)(bf16)v, u_ = (float)(bf16)v1;
                        reinterpret_cast<bf16*>(P.y[0])[m * P.ldy[0] + n] = (bf16)(silu_f(g) * u_);
                    } else {
                        float o = v;
                        if (P.residual != nullptr && mt == 0) o += (float)P.residual[m * P.ldr + n];
                        if (P.out_f32)
                            reinterpret_cast<float*>(P.y[mt])[m * P.ldy[mt] + n] = o;
                        else
                            reinterpret_cast<bf16*>(P.y[mt])[m * P.ldy[mt] + n] = (bf16)o;
                    }
                }
            }
        }
    };
    for (int j = 0; j < total; j += 2) {
        consume(wa, j);
        if (j + 2 < total) load_batch(wa, j + 2);
        if (j + 1 < total) {
            consume(wb, j + 1);
            if (j + 3 < total) load_batch(wb, j + 3);
        }
    }
}

// LDS-staged form: x (MB rows of K bf16) must fit the default 64 KiB of dynamic LDS
static inline bool gemv_lds_ok(int M, int64_t K) { return (int64_t)M * K * 2 <= 60 * 1024 && (K & 7) == 0; }

template <int MB>
int launch_gemv_lds(const GemvFusedParams& P, int swiglu, hipStream_t s) {
    const int64_t rows = swiglu ? P.N[0] : P.N[0] + P.N[1] + P.N[2];
    // row sets (of 2 rows) per wave: as few as keep the grid inside ONE resident round (the gate|up launch at one set per wave was
    // 1376 blocks = 1.8 rounds of its 768 slots: 33.2 -> 30.8 us with two sets; fewer, longer blocks than that lose again: the
    // q|k|v launch on 512 blocks of three sets ran slower than on 1536 of one).
    // (R, U) = (4, 2), (2, 8), (3, 4), (1, 8) for the single-matrix launches all measured within 1 % of or below (2, 4)
    // (profiles/r04_decode_gemv_variants.log).
    const int64_t slots = (int64_t)dllm_num_cus() * (swiglu ? 3 : 5);   // resident blocks: 154 / 88 registers per lane
    int iters = (int)cdiv64(cdiv64(rows, 8), slots);
    iters = iters < 1 ? 1 : (iters > 4 ? 4 : iters);
    const unsigned grid = (unsigned)cdiv64(rows, 8 * (int64_t)iters);
    const size_t lds = (size_t)MB * P.K * 2;
    if (swiglu)
        hipLaunchKernelGGL((gemv_lds_kernel<MB, true>), dim3(grid), dim3(256), lds, s, P, iters);
    else
        hipLaunchKernelGGL((gemv_lds_kernel<MB, false>), dim3(grid), dim3(256), lds, s, P, iters);
    return dllm_check_launch();
}

// RoPE on the new token's q and k (modeling_dreamllm.py:184-209) + append of k, v to the KV cache, one launch.
// q [B][H][D] in place; k [B][Hkv][D] rotated into kcache[b][slot]; v copied into vcache[b][slot]; slot = kv_len[b] - 1 when
// kv_len is given (left-padded prompts: the rotary position is the row's own token count, the cache slot is not), else
// pos[b].  pos / kv_len on device.
__global__ __launch_bounds__(64) void rope_append_kernel(bf16* __restrict__ q, const bf16* __restrict__ k, const bf16* __restrict__ v,
                                                         bf16* __restrict__ kc, bf16* __restrict__ vc, const float* __restrict__ cs,
                                                         const float* __restrict__ sn, const int64_t* __restrict__ pos,
                                                         const int* __restrict__ kv_len, int H, int Hkv, int D, int64_t q_sb, int64_t kv_sb, int64_t c_sb, int64_t c_ss,
                                                         int64_t c_sh) {
    const int b = blockIdx.y, hh = blockIdx.x, half = D >> 1;
    const int64_t p = pos[b];
    const int64_t slot = kv_len ? (int64_t)kv_len[b] - 1 : p;
    const int i = threadIdx.x;  // pair index
    if (i >= half) return;
    const float c = cs[p * half + i], s = sn[p * half + i];
    if (hh < H) {
        bf16* x = q + (int64_t)b * q_sb + (int64_t)hh * D;
        const float x1 = (float)x[i], x2 = (float)x[i + half];
        x[i] = (bf16)(x1 * c - x2 * s);
        x[i + half] = (bf16)(x2 * c + x1 * s);
    } else if (hh < H + Hkv) {
        const int hk = hh - H;
        const bf16* x = k + (int64_t)b * kv_sb + (int64_t)hk * D;
        bf16* dst = kc + (int64_t)b * c_sb + slot * c_ss + (int64_t)hk * c_sh;
        const float x1 = (float)x[i], x2 = (float)x[i + half];
        dst[i] = (bf16)(x1 * c - x2 * s);
        dst[i + half] = (bf16)(x2 * c + x1 * s);
    } else {
        const int hk = hh - H - Hkv;
        const bf16* x = v + (int64_t)b * kv_sb + (int64_t)hk * D;
        bf16* dst = vc + (int64_t)b * c_sb + slot * c_ss + (int64_t)hk * c_sh;
        dst[i] = x[i];
        dst[i + half] = x[i + half];
    }
}

// ---- decode attention -------------------------------------------------------------------------------------------------
struct Partial {  // running softmax state of one lane: 8 of the D output dims of its key group
    float m, l, o[8];
};
__device__ __forceinline__ void merge(Partial& a, float bm, float bl, const float (&bo)[8]) {
    const float mn = fmaxf(a.m, bm);
    if (mn == -INFINITY) return;  // both empty
    const float ca = exp2f(a.m - mn), cb = exp2f(bm - mn);
    a.l = a.l * ca + bl * cb;
#pragma unroll
    for (int e = 0; e < 8; ++e) a.o[e] = a.o[e] * ca + bo[e] * cb;
    a.m = mn;
}

// ws layout per (b, h, split): [D floats of o][m][l]
// ROPE (round 4): the step's RoPE + cache append folded in (one launch less per layer).  q arrives UN-rotated and is rotated
// in registers (rounded to bf16 as rope_append_kernel stores it); the new token's k / v come from `k_new` / `v_new` ([B][Hkv][D],
// batch pitch kv_sb, un-rotated): the lane group that meets the key at the last valid position (cache slot kv_len - 1) uses the
// rotated k / the new v instead of the cache row and writes them there for the following steps (with grouped-query heads every
// query head of a group writes the same bytes).  Exactly one split of a (batch, head) contains that key.
struct RopeNew {
    const bf16* k_new;
    const bf16* v_new;
    int64_t kv_sb;
    const float* cs;
    const float* sn;
    const int64_t* pos;
    // fused combine (optional): one arrival counter per (batch, head), zero at rest; the split that arrives last merges the NS
    // partial states in split order (the order of attn_decode_combine_kernel: same bits) and writes the output row
    int* counters;
    bf16* out;
    int64_t o_sb, o_sh;
};

template <int D, bool ROPE>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const bf16* __restrict__ q, bf16* __restrict__ kc,
                                                                  bf16* __restrict__ vc, const int* __restrict__ kv_len,
                                                                  const int* __restrict__ kv_start, float* __restrict__ ws, int H, int Hkv, int64_t q_sb, int64_t q_sh,
                                                                  int64_t c_sb, int64_t c_ss, int64_t c_sh, float scale, int NS, RopeNew rn) {
    constexpr int LPK = D / 8;    // lanes per key
    constexpr int KPW = 64 / LPK;  // keys per wave per iteration
    __shared__ float lds[4][D + 2];
    const int bh = blockIdx.x, b = bh / H, h = bh % H, split = blockIdx.y;
    const int hk = h / (H / Hkv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPK, sub = lane % LPK;
    const int start = kv_start ? kv_start[b] : 0;  // left-padded prompt: cache slots [0, start) hold pad tokens
    const int len = max(0, kv_len[b] - start);
    const int per = (len + NS - 1) / NS;
    const int k_begin = split * per, k_end = min(len, k_begin + per);

    float qv[8];
    bf16x8 knew = zero_bf16x8(), vnew = zero_bf16x8();
    {
        bf16x8 t = ld_bf16x8(q + (int64_t)b * q_sb + (int64_t)h * q_sh + sub * 8);
        if constexpr (ROPE) {
            // rotate-half: dims i and i + D/2 form a pair; this lane holds dims sub*8 .. +7, its partner lane the other half
            constexpr int half = D / 2, HS = LPK / 2;                 // lanes per half
            const int psub = sub < HS ? sub + HS : sub - HS;
            const int64_t p = rn.pos[b];
            const int ci = (sub % HS) * 8;                            // pair index of this lane's first dim
            float c[8], sn_[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                c[e] = rn.cs[p * half + ci + e];
                sn_[e] = rn.sn[p * half + ci + e];
            }
            const float sgn = sub < HS ? -1.f : 1.f;                  // x1' = x1 c - x2 s ; x2' = x2 c + x1 s
            const bf16x8 tp = ld_bf16x8(q + (int64_t)b * q_sb + (int64_t)h * q_sh + psub * 8);
            const bf16* kn = rn.k_new + (int64_t)b * rn.kv_sb + (int64_t)hk * D;
            const bf16x8 k0 = ld_bf16x8(kn + sub * 8), k1 = ld_bf16x8(kn + psub * 8);
            vnew = ld_bf16x8(rn.v_new + (int64_t)b * rn.kv_sb + (int64_t)hk * D + sub * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t[e] = (bf16)((float)t[e] * c[e] + sgn * (float)tp[e] * sn_[e]);
                knew[e] = (bf16)((float)k0[e] * c[e] + sgn * (float)k1[e] * sn_[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[e] = (float)t[e] * scale * kLog2e;
    }
    Partial st;
    st.m = -INFINITY;
    st.l = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) st.o[e] = 0.f;
    const bf16* kbase = kc + (int64_t)b * c_sb + (int64_t)hk * c_sh + (int64_t)start * c_ss + sub * 8;
    const bf16* vbase = vc + (int64_t)b * c_sb + (int64_t)hk * c_sh + (int64_t)start * c_ss + sub * 8;
    // PF key rows per lane group are requested before the first is consumed (the loop was a chain of dependent 16-byte loads: 4
    // round trips for the 64 keys a split holds at a 512-token context)
    constexpr int PF = 4;
    for (int key0 = k_begin + wave * KPW + grp; key0 < k_end; key0 += 4 * KPW * PF) {
        bf16x8 kq[PF], vq[PF];
#pragma unroll
        for (int p_ = 0; p_ < PF; ++p_) {
            const int key = key0 + p_ * 4 * KPW;
            const int kk = key < k_end ? key : k_end - 1;
            kq[p_] = ld_bf16x8(kbase + (int64_t)kk * c_ss);
            vq[p_] = ld_bf16x8(vbase + (int64_t)kk * c_ss);
        }
#pragma unroll
        for (int p_ = 0; p_ < PF; ++p_) {
            const int key = key0 + p_ * 4 * KPW;
            if (key >= k_end) break;
            bf16x8 kv = kq[p_], vv = vq[p_];
            if constexpr (ROPE) {
                if (key == len - 1) {   // the step's own token: not in the cache yet
                    kv = knew;
                    vv = vnew;
                    st_bf16x8(kc + (int64_t)b * c_sb + (int64_t)hk * c_sh + (int64_t)(start + key) * c_ss + sub * 8, knew);
                    st_bf16x8(vc + (int64_t)b * c_sb + (int64_t)hk * c_sh + (int64_t)(start + key) * c_ss + sub * 8, vnew);
                }
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s = fmaf(qv[e], (float)kv[e], s);
#pragma unroll
            for (int o = LPK / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            const float mn = fmaxf(st.m, s);
            const float c = exp2f(st.m - mn), p = exp2f(s - mn);
            st.l = st.l * c + p;
#pragma unroll
            for (int e = 0; e < 8; ++e) st.o[e] = st.o[e] * c + p * (float)vv[e];
            st.m = mn;
        }
    }
    // key groups of the wave (same sub, different grp): butterfly over the group bits
#pragma unroll
    for (int off = LPK; off < 64; off <<= 1) {
        const float bm = __shfl_xor(st.m, off, 64), bl = __shfl_xor(st.l, off, 64);
        float bo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bo[e] = __shfl_xor(st.o[e], off, 64);
        merge(st, bm, bl, bo);
    }
    if (grp == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[wave][sub * 8 + e] = st.o[e];
        if (sub == 0) {
            lds[wave][D] = st.m;
            lds[wave][D + 1] = st.l;
        }
    }
    __syncthreads();
    const bool fuse = ROPE && rn.counters != nullptr;
    if (wave == 0 && grp == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            float bo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bo[e] = lds[w][sub * 8 + e];
            merge(st, lds[w][D], lds[w][D + 1], bo);
        }
        float* dst = ws + ((int64_t)bh * NS + split) * (D + 2);
        if (fuse) {   // agent-scope 8-byte stores: the other splits of this head run on other XCDs (non-coherent L2s)
#pragma unroll
            for (int e = 0; e < 8; e += 2)
                __hip_atomic_store(reinterpret_cast<f32x2*>(dst + sub * 8 + e), (f32x2{st.o[e], st.o[e + 1]}), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            if (sub == 0)
                __hip_atomic_store(reinterpret_cast<f32x2*>(dst + D), (f32x2{st.m, st.l}), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) dst[sub * 8 + e] = st.o[e];
            if (sub == 0) {
                dst[D] = st.m;
                dst[D + 1] = st.l;
            }
        }
    }
    if constexpr (ROPE) {
        if (!fuse) return;
        // published (stores complete) -> ticket; the last arriver of this (batch, head) combines
        __shared__ int last_s;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int t = __hip_atomic_fetch_add(rn.counters + bh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last_s = (t == NS - 1) ? 1 : 0;
            if (t == NS - 1) __hip_atomic_store(rn.counters + bh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!last_s) return;
        const int d = threadIdx.x;
        if (d < D) {
            const float* src = ws + (int64_t)bh * NS * (D + 2);
            float mx = -INFINITY;
            for (int sp = 0; sp < NS; ++sp) {
                const f32x2 ml = __hip_atomic_load(reinterpret_cast<const f32x2*>(src + sp * (D + 2) + D), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mx = fmaxf(mx, ml[0]);
            }
            float l = 0.f, o = 0.f;
            for (int sp = 0; sp < NS; ++sp) {
                const f32x2 ml = __hip_atomic_load(reinterpret_cast<const f32x2*>(src + sp * (D + 2) + D), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (ml[0] == -INFINITY) continue;
                const float c = exp2f(ml[0] - mx);
                l += ml[1] * c;
                o += __hip_atomic_load(src + sp * (D + 2) + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * c;
            }
            rn.out[(int64_t)b * rn.o_sb + (int64_t)h * rn.o_sh + d] = (bf16)(l > 0.f ? o / l : 0.f);
        }
    }
}

template <int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const float* __restrict__ ws, bf16* __restrict__ out, int H,
                                                                int64_t o_sb, int64_t o_sh, int NS) {
    const int bh = blockIdx.x, b = bh / H, h = bh % H, d = threadIdx.x;
    const float* src = ws + (int64_t)bh * NS * (D + 2);
    float m = -INFINITY;
    for (int s = 0; s < NS; ++s) m = fmaxf(m, src[s * (D + 2) + D]);
    float l = 0.f, o = 0.f;
    for (int s = 0; s < NS; ++s) {
        const float ms = src[s * (D + 2) + D];
        if (ms == -INFINITY) continue;
        const float c = exp2f(ms - m);
        l += src[s * (D + 2) + D + 1] * c;
        o += src[s * (D + 2) + d] * c;
    }
    out[(int64_t)b * o_sb + (int64_t)h * o_sh + d] = (bf16)(l > 0.f ? o / l : 0.f);
}

static int launch_attn_decode(const void* q, void* kcache, void* vcache, const int* kv_len, const int* kv_start, void* out, float* ws,
                              int B, int H, int Hkv, int D, int64_t q_sb, int64_t q_sh, int64_t c_sb, int64_t c_ss, int64_t c_sh,
                              int64_t o_sb, int64_t o_sh, float scale, int nsplit, RopeNew rn, bool rope, hipStream_t s) {
    const dim3 grid((unsigned)(B * H), (unsigned)nsplit);
#define DLLM_ATTN_DEC(DD, RR)                                                                                                     \
    hipLaunchKernelGGL((attn_decode_partial_kernel<DD, RR>), grid, dim3(256), 0, s, (const bf16*)q, (bf16*)kcache, (bf16*)vcache,  \
                       kv_len, kv_start, ws, H, Hkv, q_sb, q_sh, c_sb, c_ss, c_sh, scale, nsplit, rn)
    const bool fused_combine = (rope && rn.counters != nullptr) || out == nullptr;   // out == NULL: the consumer merges the partials (dllm_gemv_attn_combine)
    if (D == 128) {
        if (rope) DLLM_ATTN_DEC(128, true); else DLLM_ATTN_DEC(128, false);
        if (!fused_combine)
            hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3((unsigned)(B * H)), dim3(128), 0, s, ws, (bf16*)out, H, o_sb,
                               o_sh, nsplit);
    } else {
        if (rope) DLLM_ATTN_DEC(64, true); else DLLM_ATTN_DEC(64, false);
        if (!fused_combine)
            hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3((unsigned)(B * H)), dim3(64), 0, s, ws, (bf16*)out, H, o_sb,
                               o_sh, nsplit);
    }
#undef DLLM_ATTN_DEC
    return dllm_check_launch();
}

}  // namespace

extern "C" {

// y[M][N] = x[M][K] W[N][K]^T (+ residual[M][N]); M <= 8; bf16 inputs, fp32 accumulation; out_dtype DLLM_BF16 / DLLM_F32.
// Replaces nn.Linear.forward at decode time (q/k/v/o_proj, gate/up/down_proj, lm_head).
int dllm_gemv_bf16(const void* x, const void* W, void* y, const void* residual, int M, int64_t N, int64_t K, int64_t ldx,
                   int64_t ldw, int64_t ldy, int64_t ldr, int out_dtype, void* stream) {
    if (M < 0 || M > 8 || N < 0 || K <= 0 || K > 0x7fffffff) return DLLM_ERR_SHAPE;
    if ((K & 7) || (ldx & 7) || (ldw & 7)) return DLLM_ERR_ALIGN;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(W)) & 15) return DLLM_ERR_ALIGN;
    if (out_dtype != DLLM_BF16 && out_dtype != DLLM_F32) return DLLM_ERR_DTYPE;
    if (M == 0 || N == 0) return DLLM_OK;
    hipStream_t s = (hipStream_t)stream;
    const bf16* xp = (const bf16*)x;
    const bf16* wp = (const bf16*)W;
    const bf16* rp = (const bf16*)residual;
    const int f32 = out_dtype == DLLM_F32;
    if (M <= 4 && gemv_lds_ok(M, K)) {   // block-staged x, two rows per wave (round 4); the wave-per-row kernels keep the other cases
        GemvFusedParams P{};
        P.x = xp; P.W[0] = wp; P.y[0] = y; P.N[0] = N; P.ldy[0] = ldy; P.residual = rp; P.ldr = ldr; P.K = (int)K; P.ldx = ldx; P.ldw = ldw;
        P.out_f32 = f32;
        switch (M) {
            case 1: return launch_gemv_lds<1>(P, 0, s);
            case 2: return launch_gemv_lds<2>(P, 0, s);
            case 3: return launch_gemv_lds<3>(P, 0, s);
            default: return launch_gemv_lds<4>(P, 0, s);
        }
    }
    switch (M) {
        case 1: return launch_gemv<1>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 2: return launch_gemv<2>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 3: return launch_gemv<3>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 4: return launch_gemv<4>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 5: return launch_gemv<5>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 6: return launch_gemv<6>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        case 7: return launch_gemv<7>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
        default: return launch_gemv<8>(xp, wp, y, rp, N, (int)K, ldx, ldw, ldy, ldr, f32, s);
    }
}

// Fused decode GEMV: y_i[M][N_i] = h W_i^T, h = RMSNorm(x; norm_w, eps) if norm_w else x; i < nmat <= 3 matrices sharing K and ldw
// (q/k/v projections in one launch), optional residual added to y_0.  swiglu != 0: nmat must be 2 (gate, up) and the single
// output y_0[M][N_0] = silu(gate) * up.  M <= 8.
int dllm_gemv_fused(const void* x, const void* norm_w, float eps, const void* W0, const void* W1, const void* W2, void* y0, void* y1,
                    void* y2, const void* residual, int M, int64_t N0, int64_t N1, int64_t N2, int64_t K, int64_t ldx, int64_t ldw,
                    int64_t ldy0, int64_t ldy1, int64_t ldy2, int64_t ldr, int swiglu, int out_dtype, void* stream) {
    if (M < 0 || M > 8 || N0 < 0 || N1 < 0 || N2 < 0 || K <= 0 || K > 0x7fffffff) return DLLM_ERR_SHAPE;
    if ((K & 7) || (ldx & 7) || (ldw & 7)) return DLLM_ERR_ALIGN;
    if (swiglu && (W1 == nullptr || N1 != N0 || N2 != 0 || out_dtype != DLLM_BF16 || residual != nullptr)) return DLLM_ERR_SHAPE;
    if ((N1 > 0 && (W1 == nullptr || (!swiglu && y1 == nullptr))) || (N2 > 0 && (W2 == nullptr || y2 == nullptr))) return DLLM_ERR_SHAPE;
    if (out_dtype != DLLM_BF16 && out_dtype != DLLM_F32) return DLLM_ERR_DTYPE;
    if (M == 0 || N0 == 0) return DLLM_OK;
    GemvFusedParams P{};
    P.x = (const bf16*)x; P.norm_w = (const bf16*)norm_w; P.eps = eps;
    P.W[0] = (const bf16*)W0; P.W[1] = (const bf16*)W1; P.W[2] = (const bf16*)W2;
    P.y[0] = y0; P.y[1] = y1; P.y[2] = y2;
    P.N[0] = N0; P.N[1] = swiglu ? N0 : N1; P.N[2] = N2;
    P.ldy[0] = ldy0; P.ldy[1] = ldy1; P.ldy[2] = ldy2;
    P.residual = (const bf16*)residual; P.ldr = ldr; P.K = (int)K; P.ldx = ldx; P.ldw = ldw; P.out_f32 = out_dtype == DLLM_F32;
    hipStream_t s = (hipStream_t)stream;
    if (M <= 4 && gemv_lds_ok(M, K)) {
        switch (M) {
            case 1: return launch_gemv_lds<1>(P, swiglu, s);
            case 2: return launch_gemv_lds<2>(P, swiglu, s);
            case 3: return launch_gemv_lds<3>(P, swiglu, s);
            default: return launch_gemv_lds<4>(P, swiglu, s);
        }
    }
    switch (M) {
        case 1: return launch_gemv_fused<1>(P, swiglu, s);
        case 2: return launch_gemv_fused<2>(P, swiglu, s);
        case 3: return launch_gemv_fused<3>(P, swiglu, s);
        case 4: return launch_gemv_fused<4>(P, swiglu, s);
        case 5: return launch_gemv_fused<5>(P, swiglu, s);
        case 6: return launch_gemv_fused<6>(P, swiglu, s);
        case 7: return launch_gemv_fused<7>(P, swiglu, s);
        default: return launch_gemv_fused<8>(P, swiglu, s);
    }
}

// RoPE of the step's q (in place) and k with rotary position pos[b] (device int64 [B]); append of rotated k and v to cache slot
// kv_len[b] - 1 (device int32 [B]) or, when kv_len is NULL, slot pos[b].
int dllm_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                     const int64_t* pos, const int* kv_len, int B, int H, int Hkv, int D, int64_t q_sb, int64_t kv_sb, int64_t c_sb, int64_t c_ss,
                     int64_t c_sh, void* stream) {
    if (B < 0 || H <= 0 || Hkv <= 0 || (D != 64 && D != 128) || pos == nullptr) return DLLM_ERR_SHAPE;
    if (B == 0) return DLLM_OK;
    hipLaunchKernelGGL(rope_append_kernel, dim3((unsigned)(H + 2 * Hkv), (unsigned)B), dim3(64), 0, (hipStream_t)stream, (bf16*)q,
                       (const bf16*)k, (const bf16*)v, (bf16*)kcache, (bf16*)vcache, cos_tab, sin_tab, pos, kv_len, H, Hkv, D, q_sb, kv_sb,
                       c_sb, c_ss, c_sh);
    return dllm_check_launch();
}

// floats of workspace dllm_attn_decode needs
int64_t dllm_attn_decode_ws_floats(int B, int H, int D, int nsplit) { return (int64_t)B * H * nsplit * (D + 2); }

// The o projection of a decode step fed DIRECTLY with the split-KV attention partials (round 6): y[M, N] = combine(ws) W^T (+ residual),
// where combine is attn_decode_combine_kernel's merge (same bits), computed while every block stages its x -- call dllm_attn_decode[_rope]
// with out = NULL (partials only) in front of it.  ws: fp32 [M * H][nsplit][D + 2]; W [N][ldw] with K = H * D columns; M <= 4.
// Replaces the combine launch + dllm_gemv_bf16 of the o projection (modeling_dreamllm.py:395 in the token loop of
// omni/eval/language_eval/modeling_dreamllm.py:76-97).
int dllm_gemv_attn_combine(const float* ws, const void* W, void* y, const void* residual, int M, int H, int D, int nsplit, int64_t N,
                           int64_t ldw, int64_t ldy, int64_t ldr, int out_dtype, void* stream) {
    const int64_t K = (int64_t)H * D;
    if (M < 1 || M > 4 || H <= 0 || (D != 64 && D != 128) || nsplit < 1 || nsplit > 64 || N < 0 || !gemv_lds_ok(M, K)) return DLLM_ERR_SHAPE;
    if (ws == nullptr || (ldw & 7) || (reinterpret_cast<uintptr_t>(W) & 15)) return DLLM_ERR_ALIGN;
    if (out_dtype != DLLM_BF16 && out_dtype != DLLM_F32) return DLLM_ERR_DTYPE;
    if (N == 0) return DLLM_OK;
    GemvFusedParams P{};
    P.W[0] = (const bf16*)W; P.y[0] = y; P.N[0] = N; P.ldy[0] = ldy; P.residual = (const bf16*)residual; P.ldr = ldr; P.K = (int)K;
    P.ldw = ldw; P.out_f32 = out_dtype == DLLM_F32;
    P.comb_ws = ws; P.comb_ns = nsplit; P.comb_H = H; P.comb_D = D;
    hipStream_t s = (hipStream_t)stream;
    switch (M) {
        case 1: return launch_gemv_lds<1>(P, 0, s);
        case 2: return launch_gemv_lds<2>(P, 0, s);
        case 3: return launch_gemv_lds<3>(P, 0, s);
        default: return launch_gemv_lds<4>(P, 0, s);
    }
}

// One query token per (b, h) against a KV cache [B][S_max][Hkv][D] (element strides c_sb, c_ss, c_sh; d contiguous) whose valid
// length per batch element is read from DEVICE memory (kv_len[b], positions 0 .. kv_len[b]-1 attended).  q / out: [B][H][D]
// views with strides (q_sb, q_sh) / (o_sb, o_sh).  Same math as the 1-token case of DreamLLMAttention (softmax(q K^T * scale) V).
int dllm_attn_decode(const void* q, const void* kcache, const void* vcache, const int* kv_len, const int* kv_start, void* out, float* ws, int B, int H,
                     int Hkv, int D, int64_t q_sb, int64_t q_sh, int64_t c_sb, int64_t c_ss, int64_t c_sh, int64_t o_sb,
                     int64_t o_sh, float scale, int nsplit, void* stream) {
    if (B < 0 || H <= 0 || Hkv <= 0 || (H % Hkv) != 0 || nsplit < 1 || nsplit > 64) return DLLM_ERR_SHAPE;
    if (D != 64 && D != 128) return DLLM_ERR_SHAPE;
    if ((q_sb | q_sh | c_sb | c_ss | c_sh) & 7) return DLLM_ERR_ALIGN;
    if (kv_len == nullptr || ws == nullptr) return DLLM_ERR_SHAPE;
    if (B == 0) return DLLM_OK;
    return launch_attn_decode(q, const_cast<void*>(kcache), const_cast<void*>(vcache), kv_len, kv_start, out, ws, B, H, Hkv, D, q_sb, q_sh,
                              c_sb, c_ss, c_sh, o_sb, o_sh, scale, nsplit, RopeNew{}, false, (hipStream_t)stream);
}

// dllm_attn_decode with the step's RoPE and KV-cache append folded in (replaces dllm_rope_append + dllm_attn_decode: one launch
// less per layer and token).  q [B][H][D] UN-rotated (left untouched); k_new / v_new [B][Hkv][D] (batch pitch kv_sb, un-rotated k):
// the new token's key / value, written -- k rotated with rotary position pos[b] -- to cache slot kv_len[b] - 1 and attended to in
// the same launch.  cos_tab / sin_tab: fp32 [max_pos][D/2]; pos: int64 [B] on device.  counters: int32 [B * H], zero on entry and
// left zero, or NULL: with counters the split that finishes last merges the partial softmax states inside the launch (no combine
// launch: 5 launches per layer and token); one counter array per stream in flight.
int dllm_attn_decode_rope(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, const float* cos_tab,
                          const float* sin_tab, const int64_t* pos, const int* kv_len, const int* kv_start, void* out, float* ws,
                          int* counters, int B, int H, int Hkv, int D, int64_t q_sb, int64_t q_sh, int64_t kv_sb, int64_t c_sb,
                          int64_t c_ss, int64_t c_sh, int64_t o_sb, int64_t o_sh, float scale, int nsplit, void* stream) {
    if (B < 0 || H <= 0 || Hkv <= 0 || (H % Hkv) != 0 || nsplit < 1 || nsplit > 64) return DLLM_ERR_SHAPE;
    if (D != 64 && D != 128) return DLLM_ERR_SHAPE;
    if ((q_sb | q_sh | c_sb | c_ss | c_sh | kv_sb) & 7) return DLLM_ERR_ALIGN;
    if (kv_len == nullptr || ws == nullptr || pos == nullptr || k_new == nullptr || v_new == nullptr || cos_tab == nullptr ||
        sin_tab == nullptr)
        return DLLM_ERR_SHAPE;
    if (B == 0) return DLLM_OK;
    RopeNew rn{(const bf16*)k_new, (const bf16*)v_new, kv_sb, cos_tab, sin_tab, pos, counters, (bf16*)out, o_sb, o_sh};
    return launch_attn_decode(q, kcache, vcache, kv_len, kv_start, out, ws, B, H, Hkv, D, q_sb, q_sh, c_sb, c_ss, c_sh, o_sb, o_sh,
                              scale, nsplit, rn, true, (hipStream_t)stream);
}

}  // extern "C"
