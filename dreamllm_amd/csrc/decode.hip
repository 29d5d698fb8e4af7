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
template <int D>
__global__ __launch_bounds__(256) void attn_decode_partial_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kc,
                                                                  const bf16* __restrict__ vc, const int* __restrict__ kv_len,
                                                                  const int* __restrict__ kv_start, float* __restrict__ ws, int H, int Hkv, int64_t q_sb, int64_t q_sh,
                                                                  int64_t c_sb, int64_t c_ss, int64_t c_sh, float scale, int NS) {
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
    {
        const bf16x8 t = ld_bf16x8(q + (int64_t)b * q_sb + (int64_t)h * q_sh + sub * 8);
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
    for (int key = k_begin + wave * KPW + grp; key < k_end; key += 4 * KPW) {
        const bf16x8 kv = ld_bf16x8(kbase + (int64_t)key * c_ss);
        const bf16x8 vv = ld_bf16x8(vbase + (int64_t)key * c_ss);
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
    if (wave == 0 && grp == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            float bo[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) bo[e] = lds[w][sub * 8 + e];
            merge(st, lds[w][D], lds[w][D + 1], bo);
        }
        float* dst = ws + ((int64_t)bh * NS + split) * (D + 2);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[sub * 8 + e] = st.o[e];
        if (sub == 0) {
            dst[D] = st.m;
            dst[D + 1] = st.l;
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
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(B * H), (unsigned)nsplit);
    if (D == 128) {
        hipLaunchKernelGGL((attn_decode_partial_kernel<128>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)kcache,
                           (const bf16*)vcache, kv_len, kv_start, ws, H, Hkv, q_sb, q_sh, c_sb, c_ss, c_sh, scale, nsplit);
        hipLaunchKernelGGL((attn_decode_combine_kernel<128>), dim3((unsigned)(B * H)), dim3(128), 0, s, ws, (bf16*)out, H, o_sb,
                           o_sh, nsplit);
    } else {
        hipLaunchKernelGGL((attn_decode_partial_kernel<64>), grid, dim3(256), 0, s, (const bf16*)q, (const bf16*)kcache,
                           (const bf16*)vcache, kv_len, kv_start, ws, H, Hkv, q_sb, q_sh, c_sb, c_ss, c_sh, scale, nsplit);
        hipLaunchKernelGGL((attn_decode_combine_kernel<64>), dim3((unsigned)(B * H)), dim3(64), 0, s, ws, (bf16*)out, H, o_sb,
                           o_sh, nsplit);
    }
    return dllm_check_launch();
}

}  // extern "C"
