// Flash-attention forward for gfx950: causal (LLM, head_dim 128, optional right-padding lengths) and non-causal
// (CLIP-ViT 257 tokens, UNet self-attention 64..4096 tokens and cross-attention over the 64 dream tokens, head_dim 64).
//
// Replaces: eager attention modeling_dreamllm.py:357-379 / flash_attn_func + flash_attn_varlen_func
// modeling_dreamllm.py:532-549 (causal, dropout 0, scale 1/sqrt(Dh), fp32 softmax); CLIP/UNet attention [ext].
//
// Layout: q/k/v/o are [B, S, H, D] views with arbitrary batch/seq/head strides (d contiguous), i.e. exactly what the
// QKV GEMM writes -- no head transposes in HBM.  LSE [B, H, Sq] fp32 (natural log) is kept for the backward.
// Block = 4 waves x 32 queries; KV blocks of 64 keys, K and V staged global -> regs -> LDS (double buffered, one
// barrier per KV block, next block's loads in flight during the MFMAs).  S^T = K Q^T and O^T = V^T P^T with MFMA
// 16x16x32 bf16: a lane owns one query column, so the running max / sum / rescale are lane-local and the
// probabilities feed the PV MFMA straight from registers (no LDS round trip, no permutes).
#include "attn_common.h"

namespace {

template <int D, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnParams P) {
    constexpr int QT = 2;
    constexpr int BQ = 4 * QT * 16;  // 128
    constexpr int BKV = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BKV * D * 2;
    using Img = TileImg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk = CAUSAL ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;  // heavy causal blocks first
    const int hk = h / (P.H / P.Hkv);
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SqE = sp.SqE;
    const int q0 = qblk * BQ, wq0 = q0 + wave * (QT * 16);
    const int coff = sk_len - sq_len;

    bf16* obase = P.o + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh;
    float* lsebase = P.lse ? P.lse + ((int64_t)b * P.H + h) * P.Sq : nullptr;
    if (sp.qst > 0) {  // left padding: rows in front of the sequence are zeros too; then work relative to the first valid row
        if (qblk == 0) {
            zero_head_rows<D, 256>(obase, P.o_ss, sp.qst, tid);
            if (lsebase)
                for (int i = tid; i < sp.qst; i += 256) lsebase[i] = 0.f;
        }
        obase += (int64_t)sp.qst * P.o_ss;
        if (lsebase) lsebase += sp.qst;
    }

    if (q0 >= sq_len) {
        // padded tail of this sequence: zeros (pad_input semantics, modeling_dreamllm.py:545)
        for (int i = tid; i < BQ * (D / 8); i += 256) {
            const int r = q0 + i / (D / 8), c = i % (D / 8);
            if (r < SqE) st_bf16x8(obase + (int64_t)r * P.o_ss + c * 8, zero_bf16x8());
        }
        if (lsebase)
            for (int i = tid; i < BQ; i += 256)
                if (q0 + i < SqE) lsebase[q0 + i] = 0.f;
        return;
    }

    // Q fragments (B operand of S^T = K Q^T): lane = query t of tile qt, d = ds*32 + g*8 ..
    bf16x8 qf[QT][DS];
    {
        const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int qrow = wq0 + qt * 16 + t;
#pragma unroll
            for (int ds = 0; ds < DS; ++ds)
                qf[qt][ds] = ld_bf16x8(qbase + (int64_t)min(qrow, sq_len - 1) * P.q_ss + ds * 32 + g * 8);
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) pin_loaded(qf[qt][ds]);
    }

    int kv_end = sk_len;
    if (CAUSAL) kv_end = min(sk_len, q0 + BQ + coff);
    const int nblk = kv_end > 0 ? (kv_end + BKV - 1) / BKV : 0;

    const bf16* kbase = P.k + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;
    const bf16* vbase = P.v + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;

    f32x4 oacc[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) oacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY;
        l_run[qt] = 0.f;
    }
    const float sl2 = P.scale * kLog2e;

    TileStage<D, BKV, 256> sk, sv;
    if (nblk > 0) {
        sk.gload(kbase, P.k_ss, 0, sk_len, tid);
        sv.gload(vbase, P.k_ss, 0, sk_len, tid);
        sk.lstore_row(smem, tid);
        sv.lstore_col(smem + TILE, tid);
    }
    __syncthreads();

    for (int j = 0; j < nblk; ++j) {
        const int kv0 = j * BKV;
        if (j + 1 < nblk) {
            sk.gload(kbase, P.k_ss, kv0 + BKV, sk_len, tid);
            sv.gload(vbase, P.k_ss, kv0 + BKV, sk_len, tid);
        }
        const char* kt_ = smem + (j & 1) * (2 * TILE);
        const char* vt_ = kt_ + TILE;
        const bool wave_active = (wq0 < sq_len) && !(CAUSAL && kv0 > wq0 + QT * 16 - 1 + coff);
        if (wave_active) {
            f32x4 s[4][QT];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) s[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    const bf16x8 kf = Img::frag_row(kt_, kt * 16, ds, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], s[kt][qt], 0, 0, 0);
                }
            }
            const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
            bf16x8 pb[QT][2];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int qidx = wq0 + qt * 16 + t;
                // running max in the RAW score domain (scale > 0 commutes with max); exp2 argument by one FMA per element
                float mx = -INFINITY;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = s[kt][qt][r];
                        if (need_mask) {
                            const int kidx = kv0 + kt * 16 + g * 4 + r;
                            if (kidx >= sk_len || (CAUSAL && kidx > qidx + coff)) x = -INFINITY;
                            s[kt][qt][r] = x;
                        }
                        mx = fmaxf(mx, x);
                    }
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                const float m_new = fmaxf(m_run[qt], mx);
                const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
                const float alpha = fast_exp2((m_run[qt] - m_use) * sl2);
                const float nm = -m_use * sl2;
                float rs = 0.f;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float p = fast_exp2(fmaf(s[kt][qt][r], sl2, nm));
                        s[kt][qt][r] = p;
                        rs += p;
                    }
                l_run[qt] = l_run[qt] * alpha + rs;
                m_run[qt] = m_new;
                if (!__all(alpha == 1.0f)) {  // wave-uniform: skip the O rescale when no lane's running max moved
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pb[qt][ks][r] = (bf16)s[2 * ks][qt][r];
                        pb[qt][ks][4 + r] = (bf16)s[2 * ks + 1][qt][r];
                    }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const bf16x8 va = Img::frag_col(vt_, dt * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[qt][ks], oacc[dt][qt], 0, 0, 0);
                }
            }
        }
        if (j + 1 < nblk) {
            char* nk = smem + ((j + 1) & 1) * (2 * TILE);
            sk.lstore_row(nk, tid);
            sv.lstore_col(nk + TILE, tid);
        }
        __syncthreads();
    }

    // finalize: lane holds O^T[d = dt*16 + g*4 + r][q = t] of tile qt
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = l_run[qt];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qrow = wq0 + qt * 16 + t;
        const bool valid = qrow < sq_len;
        const float inv = (l > 0.f && valid) ? 1.0f / l : 0.f;
        if (qrow < SqE) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)(oacc[dt][qt][r] * inv);
                st_bf16x4(obase + (int64_t)qrow * P.o_ss + dt * 16 + g * 4, o);
            }
            if (lsebase && g == 0) lsebase[qrow] = (valid && l > 0.f) ? (m_run[qt] * P.scale + logf(l)) : 0.f;
        }
    }
}

template <int D, bool CAUSAL>
int launch_fwd(const AttnParams& P, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 64 * D * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_fwd_kernel<D, CAUSAL>, LDS, lds_ok);
    dim3 grid((P.Sq + 127) / 128, P.H, P.B);
    hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL>), grid, dim3(256), LDS, stream, P);
    return dllm_check_launch();
}

}  // namespace

extern "C" {

// q,o: [B,Sq,H,D] views (element strides sb, ss, sh; d contiguous); k,v: [B,Sk,Hkv,D] views sharing one stride set.
// seqlens: optional int32[B] valid lengths of a padded self-attention batch (Sq == Sk); seqstart: optional int32[B] index of
// the first valid token (left padding; 0 when null).  Keys outside [start, start + len) are masked and query rows outside it
// are written as zeros (pad_input semantics).  With Sq != Sk (KV cache) seqstart masks the first keys only and every query
// is valid.  lse: optional fp32 [B,H,Sq].
int dllm_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* seqlens, const int* seqstart, int B,
                  int H, int Hkv,
                  int Sq, int Sk, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                  int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale, int causal, void* stream) {
    if (B < 0 || H <= 0 || Hkv <= 0 || Sq < 0 || Sk < 0 || (H % Hkv) != 0) return DLLM_ERR_SHAPE;
    if (D != 64 && D != 128) return DLLM_ERR_SHAPE;
    if (B == 0 || Sq == 0) return DLLM_OK;
    if ((q_ss | q_sh | q_sb | k_ss | k_sh | k_sb | o_ss | o_sh | o_sb) & 7) return DLLM_ERR_ALIGN;
    if (seqlens != nullptr && Sq != Sk) return DLLM_ERR_SHAPE;
    AttnParams P{};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.lse = lse; P.seqlens = seqlens;
    P.seqstart = seqstart;
    P.B = B; P.H = H; P.Hkv = Hkv; P.Sq = Sq; P.Sk = Sk;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh; P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh; P.scale = scale; P.causal = causal;
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) return causal ? launch_fwd<128, true>(P, s) : launch_fwd<128, false>(P, s);
    return causal ? launch_fwd<64, true>(P, s) : launch_fwd<64, false>(P, s);
}

}  // extern "C"
