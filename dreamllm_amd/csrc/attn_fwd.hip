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

// ABL (benchmark builds only, -DDLLM_BENCH_MODES): ablation bits that REMOVE one cost each while keeping every value live,
// to find what the loop is bound by (wrong results by design): 1 no softmax VALU, 2 no PV MFMAs, 4 no QK^T MFMAs,
// 8 no global loads / LDS stores in the loop (tile 0 is re-used), 16 no O rescale.
template <int D, bool CAUSAL, int ABL = 0>
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
    constexpr int BQ_ = (4 * 2) * 16;
    const int nqb = (P.Sq + BQ_ - 1) / BQ_;
    const AttnBlock bm = attn_block_map<false>(nqb, P.H, P.B);
    if (!bm.valid) return;
    const int b = bm.b, h = bm.h;
    const int qblk = CAUSAL ? (nqb - 1 - bm.r) : bm.r;  // heavy causal blocks first
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
        if (j + 1 < nblk && !(ABL & 8)) {
            sk.gload(kbase, P.k_ss, kv0 + BKV, sk_len, tid);
            sv.gload(vbase, P.k_ss, kv0 + BKV, sk_len, tid);
        }
        const char* kt_ = smem + ((ABL & 8) ? 0 : (j & 1)) * (2 * TILE);
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
                    if (ABL & 4) {
                        asm volatile("" ::"v"(kf));
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) s[kt][qt][0] += (float)qf[qt][ds][0];
                        continue;
                    }
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], s[kt][qt], 0, 0, 0);
                }
            }
            const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
            bf16x8 pb[QT][2];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (ABL & 1) {  // no softmax: scores go straight to bf16
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            pb[qt][ks][r] = (bf16)s[2 * ks][qt][r];
                            pb[qt][ks][4 + r] = (bf16)s[2 * ks + 1][qt][r];
                        }
                    continue;
                }
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
                if (!(ABL & 16) && !__all(alpha == 1.0f)) {  // wave-uniform: skip the O rescale when no lane's running max moved
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
                    if (ABL & 2) {
                        asm volatile("" ::"v"(va));
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) oacc[dt][qt][0] += (float)pb[qt][ks][0];
                        continue;
                    }
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[qt][ks], oacc[dt][qt], 0, 0, 0);
                }
            }
        }
        if (j + 1 < nblk && !(ABL & 8)) {
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


// ------------------------------------------------------------------------------------------------ 8-wave pipelined forward
// Same math and LDS images as attn_fwd_kernel, restructured for the long-sequence shapes (LLM prefill / training, UNet 1024+
// tokens) after an ablation of the 4-wave kernel at B16 x S2048 x H32 x D128 (tools/attn_ablate.py, profiles/r02_attn_ablate.log):
// softmax VALU 35 % and tile staging 28 % of the time, the MFMAs 10 %, everything serialised inside a wave.
//   * block = 8 waves x 32 queries = 256 queries: a K/V tile is fetched and written to LDS once per 256 queries (staging
//     instructions per FLOP halve); full tiles use uniform-base + one per-thread offset addressing.
//   * software pipeline inside the wave: S(j+1) = K(j+1) Q^T is issued BEFORE the softmax of S(j), so the MFMA pipe works
//     under the softmax's VALU instructions (K runs one tile ahead of V in LDS; scores are double-buffered in registers).
//   * softmax diet: row max through v_permlane16/32_swap (no LDS round trip of ds_bpermute), row sums on the matrix pipe
//     (one extra MFMA with an all-ones A operand per 32 keys instead of 32 VALU adds per lane), v_max3 / cvt_pk by the compiler.
constexpr float kNegBig = -1.0e30f;

// single-instruction maxima: with NaNs possible (-fno-finite-math-only: the masks use infinities) fmaxf() costs an extra
// canonicalising v_max per MFMA output; the scores here are never NaN
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// max(seed, x over the 4 lanes {t, t+16, t+32, t+48})
__device__ __forceinline__ float group_max4(float x, float seed) {
    const uint32_t u = __float_as_uint(x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float m = vmax2(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const uint32_t v = __float_as_uint(m);
    auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return vmax3(seed, __uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_fwd8_kernel(AttnParams P) {
    constexpr int QT = 2, NW = 8;
    constexpr int BQ = NW * QT * 16;  // 256
    constexpr int BKV = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BKV * D * 2;
    using Img = TileImg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K[2] then V[2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    constexpr int BQ_ = (8 * 2) * 16;
    const int nqb = (P.Sq + BQ_ - 1) / BQ_;
    // Causal: one work-group takes the PAIR of query blocks (nqb-1-r, r) -- heaviest with lightest -- so every group does the same
    // (nqb + 1) key tiles: the grid is uniform (no tail of stragglers), launch / prologue / epilogue cost is paid per pair, and the
    // head-major order keeps a head's groups together on one XCD's L2.
    const int nitems = CAUSAL ? (nqb + 1) / 2 : nqb;
    const AttnBlock bm = attn_block_map<false>(nitems, P.H, P.B);
    if (!bm.valid) return;
    const int b = bm.b, h = bm.h;
    const int npass = (CAUSAL && nqb - 1 - bm.r != bm.r) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int qblk = CAUSAL ? (pass == 0 ? nqb - 1 - bm.r : bm.r) : bm.r;
        const int hk = h / (P.H / P.Hkv);
        const AttnSpan sp = attn_span(P, b);
        const int sq_len = sp.sq_len, sk_len = sp.sk_len, SqE = sp.SqE;
        const int q0 = qblk * BQ, wq0 = q0 + wave * (QT * 16);
        const int coff = sk_len - sq_len;

        bf16* obase = P.o + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh;
        float* lsebase = P.lse ? P.lse + ((int64_t)b * P.H + h) * P.Sq : nullptr;
        if (sp.qst > 0) {
            if (qblk == 0) {
                zero_head_rows<D, 512>(obase, P.o_ss, sp.qst, tid);
                if (lsebase)
                    for (int i = tid; i < sp.qst; i += 512) lsebase[i] = 0.f;
            }
            obase += (int64_t)sp.qst * P.o_ss;
            if (lsebase) lsebase += sp.qst;
        }
        if (q0 >= sq_len) {  // padded tail: zeros (pad_input semantics, modeling_dreamllm.py:545)
            for (int i = tid; i < BQ * (D / 8); i += 512) {
                const int r = q0 + i / (D / 8), c = i % (D / 8);
                if (r < SqE) st_bf16x8(obase + (int64_t)r * P.o_ss + c * 8, zero_bf16x8());
            }
            if (lsebase)
                for (int i = tid; i < BQ; i += 512)
                    if (q0 + i < SqE) lsebase[q0 + i] = 0.f;
            continue;
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
        f32x4 lacc[QT];  // row sums on the matrix pipe: every row of this accumulator holds sum_k P[k][q]
        float m_run[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            lacc[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
            m_run[qt] = kNegBig;  // finite "minus infinity": no special cases in the bookkeeping (exp2 of -huge is exactly 0)
        }
        const float sl2 = P.scale * kLog2e;
        bf16x8 ones;
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

        char* const Kb0 = smem;
        char* const Vb0 = smem + 2 * TILE;
        // K / V tiles arrive by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass, no address VALU per
        // store).  The DMA writes wave-uniform base + lane * 16, so the XOR swizzles of the two LDS images are applied to the
        // per-lane SOURCE chunk (a permutation inside one row: coalescing is unaffected).  A wave moves NDMA 1-KiB groups of RPG
        // rows per tile; every LDS read of this kernel is inline asm (below), so hipcc has no visible LDS read in front of which it
        // would drain the DMA queue.
        constexpr int CPR = D / 8;             // 16-byte chunks per row
        constexpr int RPG = 64 / CPR;          // rows per 1-KiB group (4 at D = 128, 8 at D = 64)
        constexpr int NDMA = (BKV / RPG) / NW; // groups per wave per tile (2 / 1)
        const int drow = lane / CPR, dpos = lane % CPR;
        int k_src_chunk[NDMA], v_src_chunk[NDMA];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) {
            const int r = (wave * NDMA + i) * RPG + drow;
            if constexpr (D == 128) {
                k_src_chunk[i] = dpos ^ (r & 15);
                v_src_chunk[i] = (((dpos >> 1) ^ (r & 7)) << 1) | (dpos & 1);
            } else {
                k_src_chunk[i] = dpos ^ ((r >> 1) & 7);
                v_src_chunk[i] = (((dpos >> 1) ^ ((r >> 1) & 3)) << 1) | (dpos & 1);
            }
        }
        auto dma_tile = [&](int row0, int buf) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int grp = wave * NDMA + i;
                const int row = min(row0 + grp * RPG + drow, sk_len - 1);  // rows past the end: any finite data (masked later)
                GLDS16_(kbase + (int64_t)row * P.k_ss + k_src_chunk[i] * 8, Kb0 + buf * TILE + grp * 1024);
                GLDS16_(vbase + (int64_t)row * P.k_ss + v_src_chunk[i] * 8, Vb0 + buf * TILE + grp * 1024);
            }
        };
        // a wave takes part in tile j iff one of its queries can see one of the tile's keys
        auto active = [&](int j) { return (wq0 < sq_len) && !(CAUSAL && j * BKV > wq0 + QT * 16 - 1 + coff); };
        // Online-softmax bookkeeping of one 32-key half: mask (diagonal / ragged tiles only), row max, and the RARE rescale.
        // The running max is only advanced when some row's max grew by more than 2^kDefer (in the exp2 domain): otherwise the
        // probabilities of this half are taken against the old max (they are then bounded by 2^kDefer instead of 1, harmless in
        // fp32 / bf16 and invisible in O = sum(P V) / sum(P)), and neither O nor the row sums need the multiply.  Returns the
        // exponent offsets.  Everything that follows (exp2 + bf16 pack) is branch-free, so the compiler interleaves it with the
        // MFMAs of the next group.
        constexpr float kDefer = 6.0f;
        auto max_half = [&](f32x4 (&s)[2][QT], int kbase_idx, bool need_mask, float (&nm)[QT], bf16x8 (*pending)[QT]) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                if (need_mask) {
                    const int qidx = wq0 + qt * 16 + t;
#pragma unroll
                    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kidx = kbase_idx + kt * 16 + g * 4 + r;
                            const bool dead = kidx >= sk_len || (CAUSAL && kidx > qidx + coff);
                            s[kt][qt][r] = dead ? -INFINITY : s[kt][qt][r];
                        }
                }
                float mx = vmax3(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
                mx = vmax3(mx, s[0][qt][3], s[1][qt][0]);
                mx = vmax3(mx, s[1][qt][1], s[1][qt][2]);
                mx = vmax2(mx, s[1][qt][3]);
                const float m_new = group_max4(mx, m_run[qt]);
                if (__any((m_new - m_run[qt]) * sl2 > kDefer)) {  // wave-uniform, rare after the first tiles
                    const float alpha = fast_exp2((m_run[qt] - m_new) * sl2);  // 0 while the old max is the finite "minus infinity"
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) oacc[dt][qt] *= alpha;
                    lacc[qt] *= alpha;
                    // probabilities of the previous half were exponentiated against the OLD max and have not entered O / the row
                    // sums yet: they take the same factor (everything still at the old scale is rescaled exactly once)
                    if (pending != nullptr) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) (*pending)[qt][e] = (bf16)((float)(*pending)[qt][e] * alpha);
                    }
                    m_run[qt] = m_new;
                }
                nm[qt] = -m_run[qt] * sl2;  // a row that has seen no key yet: +huge, and exp2(-inf + huge) = 0
            }
        };
        // exp2 + bf16 pack of elements [e0, e0 + n) of a half (element e: qt = e / 8, slot = e % 8; slot < 4 -> key tile 0 of the
        // half).  The packed words are pinned with an empty asm at the point where they are produced: LLVM otherwise SINKS the
        // whole exponential chain into the block of its first use (the P V MFMAs), i.e. behind the MFMAs it is meant to overlap.
        auto exp_slice = [&](const f32x4 (&s)[2][QT], const float (&nm)[QT], bf16x8 (&pb)[QT], int e0, int n) {
#pragma unroll
            for (int e = e0; e < e0 + n; e += 2) {
                const int qt = e >> 3, sl = e & 7;
                bf16x2 w;
                w[0] = (bf16)fast_exp2(fmaf(s[sl >> 2][qt][sl & 3], sl2, nm[qt]));
                w[1] = (bf16)fast_exp2(fmaf(s[(sl + 1) >> 2][qt][(sl + 1) & 3], sl2, nm[qt]));
                uint32_t u = __builtin_bit_cast(uint32_t, w);
                asm volatile("" : "+v"(u));
                w = __builtin_bit_cast(bf16x2, u);
                pb[qt][sl] = w[0];
                pb[qt][sl + 1] = w[1];
            }
        };
        constexpr int NE = QT * 8;  // exp elements per half

        // ---- the fragment stream of one tile: 32 steps = {K frags of keys 0..31} {K frags of keys 32..63} {V frags 0..31} {V frags
        // 32..63}, DS*2 / DT steps each (D = 128: 8 / 8 / 8 / 8).  Every LDS read is inline asm, requested PRE steps ahead of its
        // MFMAs and retired by a COUNTED s_waitcnt lgkmcnt(n) (LDS returns in order): with the compiler's own placement each step
        // waited out a full LDS round trip (~200 cycles x 32 steps per tile -- the reason the 4-wave kernel ran its MFMAs 10 % of
        // the time).  Steps are compile-time indices, so ring slots and immediates are static.
        constexpr int NSK = DS * 2, NSV = DT;                 // steps per K half / per V half
        constexpr int NSTEP = 2 * NSK + 2 * NSV;
        constexpr int PRE = 5;                                // fragments in flight (<= 15 LDS operations outstanding)
        constexpr int RING = PRE + 1;
        u32x4 kring[RING];
        u32x2 vlo[RING], vhi[RING];
        // per-lane LDS byte addresses (tile buffer 0): K row image at (row t, chunk ds*4 + g), V col image at (row g*4 + t/4, d = 4*(t&3))
        uint32_t kaddr0[DS];
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) kaddr0[ds] = lds_addr32(Kb0 + Img::row_off(t, ds * 4 + g));
        const uint32_t vaddr0 = lds_addr32(Vb0 + (g * 4 + (t >> 2)) * Img::PITCH + (t & 3) * 8);
        const uint32_t vswz = (uint32_t)((D == 128 ? ((g * 4 + (t >> 2)) & 7) : (((g * 4 + (t >> 2)) >> 1) & 3)) << 5);
        uint32_t kaddr[DS], vaddr = vaddr0;
        auto frag_ops = [](int f) { return f < 2 * NSK ? 1 : 2; };  // LDS operations of fragment f
        auto issue = [&kring, &vlo, &vhi, &kaddr, &vaddr, vswz](auto fc) {
            constexpr int f = decltype(fc)::value;
            if constexpr (f < 2 * NSK) {
                constexpr int hh = f / NSK, ds = (f % NSK) >> 1, kt = (f % NSK) & 1;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kring[f % RING]) : "v"(kaddr[ds]), "n"((2 * hh + kt) * 16 * Img::PITCH));
            } else {
                constexpr int hh = (f - 2 * NSK) / NSV, dt = (f - 2 * NSK) % NSV;
                const uint32_t a = vaddr + ((uint32_t)(dt << 5) ^ vswz);
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[f % RING]) : "v"(a), "n"((2 * hh) * 16 * Img::PITCH));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[f % RING]) : "v"(a), "n"((2 * hh + 1) * 16 * Img::PITCH));
            }
        };
        auto wait_frag = [&kring, &vlo, &vhi](auto fc) {  // retire fragment f: everything requested after it may stay in flight
            constexpr int f = decltype(fc)::value;
            constexpr int last = f + PRE < NSTEP ? f + PRE : NSTEP - 1;
            constexpr int younger = [](int f0, int l0) { int n = 0; for (int i = f0 + 1; i <= l0; ++i) n += (i < 2 * (D / 32) * 2 ? 1 : 2); return n; }(f, last);
            static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
            if constexpr (f < 2 * NSK)
                asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(kring[f % RING]) : "n"(younger) : "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(vlo[f % RING]), "+v"(vhi[f % RING]) : "n"(younger) : "memory");
        };
        (void)frag_ops;

        // ---- prologue: tile 0 -> LDS
        if (nblk > 0) dma_tile(0, 0);

        // operand loads AFTER the first tile's DMA is in flight: the two latencies overlap instead of adding up
        bf16x8 qf[QT][DS];
        {
            const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int qrow = min(wq0 + qt * 16 + t, sq_len - 1);
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) qf[qt][ds] = ld_bf16x8(qbase + (int64_t)qrow * P.q_ss + ds * 32 + g * 8);
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) pin_loaded(qf[qt][ds]);
        }
        __syncthreads();

        for (int j = 0; j < nblk; ++j) {
            const int kv0 = j * BKV;
            if (j + 1 < nblk) dma_tile(kv0 + BKV, (j + 1) & 1);  // the next tile flies under this tile's compute
            if (active(j)) {
                const uint32_t boff = (uint32_t)((j & 1) * TILE);
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) kaddr[ds] = kaddr0[ds] + boff;
                vaddr = vaddr0 + boff;
                const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
                f32x4 sa[2][QT], sb[2][QT];
                bf16x8 pa[QT], pbb[QT];
                float nma[QT], nmb[QT];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        sa[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        sb[kt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                static_for_<0, PRE>([&](auto fc) { issue(fc); });
                static_for_<0, NSTEP>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    if constexpr (st + PRE < NSTEP) issue(std::integral_constant<int, st + PRE>{});
                    // segment boundaries: the short VALU-only bookkeeping of the online softmax
                    if constexpr (st == NSK) max_half(sa, kv0, need_mask, nma, nullptr);
                    if constexpr (st == 2 * NSK) max_half(sb, kv0 + 32, need_mask, nmb, &pa);
                    if constexpr (st == 2 * NSK || st == 2 * NSK + NSV) {  // row sums of the half entering P V, on the matrix pipe
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            lacc[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, st == 2 * NSK ? pa[qt] : pbb[qt], lacc[qt], 0, 0, 0);
                    }
                    wait_frag(sc);
                    if constexpr (st < 2 * NSK) {
                        constexpr int hh = st / NSK, ds = (st % NSK) >> 1, kt = (st % NSK) & 1;
                        const bf16x8 kf = __builtin_bit_cast(bf16x8, kring[st % RING]);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            if constexpr (hh == 0)
                                sa[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], sa[kt][qt], 0, 0, 0);
                            else
                                sb[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[qt][ds], sb[kt][qt], 0, 0, 0);
                        }
                        // keys 32..63: under the exponentials of keys 0..31
                        if constexpr (hh == 1) exp_slice(sa, nma, pa, (st - NSK) * NE / NSK, NE / NSK);
                    } else {
                        constexpr int hh = (st - 2 * NSK) / NSV, dt = (st - 2 * NSK) % NSV;
                        const bf16x8 va = join2(vlo[st % RING], vhi[st % RING]);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            oacc[dt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, hh == 0 ? pa[qt] : pbb[qt], oacc[dt][qt], 0, 0, 0);
                        // P V of keys 0..31: under the exponentials of keys 32..63
                        if constexpr (hh == 0) exp_slice(sb, nmb, pbb, dt * NE / NSV, NE / NSV);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            __syncthreads();  // tile j + 1 has landed (vmcnt(0) of the DMA) and every wave is done reading tile j
        }

        // finalize: lane holds O^T[d = dt*16 + g*4 + r][q = t] of tile qt; every row of lacc holds the row sum of query t
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float l = lacc[qt][0];
            const int qrow = wq0 + qt * 16 + t;
            const bool valid = qrow < sq_len;
            const float inv = (l > 0.f && valid) ? 1.0f / l : 0.f;
            // Store tail (round 4; MI355X guide T21 with the 16-lane swap): a lane holds 4 consecutive d (8 bytes) of its query row per
            // 16-wide d tile, the lane 16 further the next 4.  One v_permlane16_swap per dword and pair of d tiles (dt, dt + 1) gives the
            // even lane group [own dt | partner's dt] and the odd one [partner's dt + 1 | own dt + 1]: 16 contiguous bytes each, so a
            // row leaves in DT / 2 16-byte stores per lane instead of DT 8-byte ones (the tail is store-ISSUE bound, not bandwidth bound).
            // Both lanes of a pair hold the same query row, so the predicate is uniform across the exchange.
#pragma unroll
            for (int dt = 0; dt < DT; dt += 2) {
                union {
                    bf16x4 v;
                    uint32_t w[2];
                } a, b;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    a.v[r] = (bf16)(oacc[dt][qt][r] * inv);
                    b.v[r] = (bf16)(oacc[dt + 1][qt][r] * inv);
                }
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const auto sw = __builtin_amdgcn_permlane16_swap(a.w[w], b.w[w], false, false);
                    a.w[w] = sw[0];
                    b.w[w] = sw[1];
                }
                if (qrow < SqE) {
                    const u32x4 o = u32x4{a.w[0], a.w[1], b.w[0], b.w[1]};
                    const int d0 = (g & 1) ? (dt + 1) * 16 + (g - 1) * 4 : dt * 16 + g * 4;
                    *reinterpret_cast<u32x4*>(obase + (int64_t)qrow * P.o_ss + d0) = o;
                }
            }
            if (qrow < SqE && lsebase && g == 0) lsebase[qrow] = (valid && l > 0.f) ? (m_run[qt] * P.scale + logf(l)) : 0.f;
        }
    }  // pass
}

template <int D, bool CAUSAL>
int launch_fwd8(const AttnParams& P, hipStream_t stream) {
    constexpr int LDS = 4 * 64 * D * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_fwd8_kernel<D, CAUSAL>, LDS, lds_ok);
    const int nqb = (P.Sq + 255) / 256;
    const dim3 grid(attn_grid(CAUSAL ? (nqb + 1) / 2 : nqb, P.H, P.B));  // causal: one group per pair of query blocks
    hipLaunchKernelGGL((attn_fwd8_kernel<D, CAUSAL>), grid, dim3(512), LDS, stream, P);
    return dllm_check_launch();
}

template <int D, bool CAUSAL, int ABL = 0>
int launch_fwd(const AttnParams& P, hipStream_t stream) {
    constexpr int LDS = 2 * 2 * 64 * D * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_fwd_kernel<D, CAUSAL, ABL>, LDS, lds_ok);
    const dim3 grid(attn_grid((P.Sq + 127) / 128, P.H, P.B));
    hipLaunchKernelGGL((attn_fwd_kernel<D, CAUSAL, ABL>), grid, dim3(256), LDS, stream, P);
    return dllm_check_launch();
}

}  // namespace

__attribute__((visibility("hidden"))) int dllm_launch_attn_fwd_pp(const AttnParams& P, int D, int causal, hipStream_t stream);  // attn_fwd_pp.hip

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
    // 16-byte loads of q / k / v rows and (round 4: the widened store tail) 16-byte stores of o rows
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o)) & 15)
        return DLLM_ERR_ALIGN;
    if (seqlens != nullptr && Sq != Sk) return DLLM_ERR_SHAPE;
    AttnParams P{};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.lse = lse; P.seqlens = seqlens;
    P.seqstart = seqstart;
    P.B = B; P.H = H; P.Hkv = Hkv; P.Sq = Sq; P.Sk = Sk;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh; P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh; P.scale = scale; P.causal = causal;
    hipStream_t s = (hipStream_t)stream;
    // kernel choice: bit 1 of `causal` forces the 4-wave kernel, bit 2 the 8-wave pipelined one (tests cover both on every
    // shape); automatic = 8-wave for long query sequences, 4-wave (128-query blocks fill the chip better) for short ones
    const int force = (causal >> 1) & 3;
    causal &= 1;
    P.causal = causal;
    // force 3 = the ping-pong kernel of attn_fwd_pp.hip (32-bit per-lane source byte offsets: one (batch, head) key axis must span less than 1 GiB)
    const bool pp_ok = (int64_t)Sk * k_ss < (1ll << 29);
    const bool wide = force == 2 || force == 3 || (force == 0 && Sq >= 512);
    // automatic choice on long query axes = the ping-pong kernel (round 5: 830 vs 761 TF at B16 S2048 H32 D128 causal, 880 vs 754 TF at
    // the UNet's S = 4096 d64 shape, profiles/r05_attn_bench.log); force 2 keeps the 8-wave pipelined kernel reachable for tests / tools
    if ((force == 3 || (force == 0 && wide)) && pp_ok) return dllm_launch_attn_fwd_pp(P, D, causal, s);
    if (wide) {
        if (D == 128) return causal ? launch_fwd8<128, true>(P, s) : launch_fwd8<128, false>(P, s);
        return causal ? launch_fwd8<64, true>(P, s) : launch_fwd8<64, false>(P, s);
    }
    if (D == 128) return causal ? launch_fwd<128, true>(P, s) : launch_fwd<128, false>(P, s);
    return causal ? launch_fwd<64, true>(P, s) : launch_fwd<64, false>(P, s);
}

#ifdef DLLM_BENCH_MODES
// benchmark-only: the ping-pong forward (causal, head_dim 128, [B,S,H,D] contiguous) with s_memtime stamps of work-group 0 written
// to `stamps` (2 x 512 uint64: wave 0, wave 4; two stamps around every barrier of the first pass)
int dllm_attn_fwd_pp_timeline(const void* q, const void* k, const void* v, void* o, float* lse, void* stamps, int B, int H, int Sq,
                              int abl, void* stream) {
    AttnParams P{};
    const int D = 128;
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.lse = lse; P.delta = (float*)stamps;
    P.B = B; P.H = H; P.Hkv = H; P.Sq = Sq; P.Sk = Sq;
    P.q_sb = P.k_sb = P.o_sb = (int64_t)Sq * H * D; P.q_ss = P.k_ss = P.o_ss = (int64_t)H * D; P.q_sh = P.k_sh = P.o_sh = D;
    P.scale = 0.08838834764f; P.causal = 1;
    return dllm_launch_attn_fwd_pp(P, D, 1 | (abl << 8), (hipStream_t)stream);
}
// benchmark-only: the causal head_dim-128 forward with one cost removed (see ABL above); results are wrong by design
int dllm_attn_fwd_ablate(const void* q, const void* k, const void* v, void* o, float* lse, int B, int H, int Sq, int D, int abl,
                         void* stream) {
    if (D != 128) return DLLM_ERR_SHAPE;
    AttnParams P{};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.lse = lse;
    P.B = B; P.H = H; P.Hkv = H; P.Sq = Sq; P.Sk = Sq;
    P.q_sb = P.k_sb = P.o_sb = (int64_t)Sq * H * D; P.q_ss = P.k_ss = P.o_ss = (int64_t)H * D; P.q_sh = P.k_sh = P.o_sh = D;
    P.scale = 0.08838834764f; P.causal = 1;
    hipStream_t s = (hipStream_t)stream;
    switch (abl) {
        case 0: return launch_fwd<128, true, 0>(P, s);
        case 1: return launch_fwd<128, true, 1>(P, s);
        case 2: return launch_fwd<128, true, 2>(P, s);
        case 4: return launch_fwd<128, true, 4>(P, s);
        case 8: return launch_fwd<128, true, 8>(P, s);
        case 16: return launch_fwd<128, true, 16>(P, s);
        case 3: return launch_fwd<128, true, 3>(P, s);
        case 6: return launch_fwd<128, true, 6>(P, s);
        case 7: return launch_fwd<128, true, 7>(P, s);
        case 15: return launch_fwd<128, true, 15>(P, s);
        case 9: return launch_fwd<128, true, 9>(P, s);
        default: return DLLM_ERR_SHAPE;
    }
}
#endif

}  // extern "C"
