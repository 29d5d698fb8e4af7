// Flash-attention backward for gfx950 (dQ, dK, dV), same layouts and masking rules as attn_fwd.hip.
//
// Replaces the autograd of modeling_dreamllm.py:357-379 / flash_attn's backward behind modeling_dreamllm.py:532-549
// and the attention backward inside the frozen UNet (dgrad only) [ext].
//
// Three launches, no atomics, deterministic:
//   1. delta[b,h,q] = sum_d dO * O                                   (HBM-bound preprocess)
//   2. dQ  : block = 4 waves x (QT*16) queries, loops over KV blocks; recomputes S^T = K Q^T and dP^T = V dO^T with
//            K and V tiles in LDS; dQ^T += K^T dS^T reads the same K tile through transpose reads.
//   3. dK,dV: block = 4 waves x (KT*16) keys held in registers, loops over Q blocks (and over the query heads of a
//            GQA group); recomputes S = Q K^T and dP = dO V^T from Q / dO tiles in LDS, then dV^T += dO^T P and
//            dK^T += Q^T dS read the same tiles through transpose reads.
// P = exp(scale*S - LSE) uses the forward's saved log-sum-exp; dS = P * (dP - delta).
#include "attn_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ delta
template <int D>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnParams P) {
    constexpr int TPR = D / 8;  // threads per (b,q,h) row
    const int64_t rows = (int64_t)P.B * P.Sq * P.H;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t row = gid / TPR;
    const int c = (int)(gid % TPR) * 8;
    float s = 0.f;
    int b = 0, q = 0, h = 0;
    if (row < rows) {
        h = (int)(row % P.H);
        const int64_t bq = row / P.H;
        q = (int)(bq % P.Sq);
        b = (int)(bq / P.Sq);
        const int64_t off = (int64_t)b * P.o_sb + (int64_t)q * P.o_ss + (int64_t)h * P.o_sh + c;
        const bf16x8 a = ld_bf16x8(P.o + off), d = ld_bf16x8(P.dout + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)a[e] * (float)d[e];
    }
#pragma unroll
    for (int o = TPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (row < rows && (threadIdx.x % TPR) == 0) P.delta[((int64_t)b * P.H + h) * P.Sq + q] = s;
}

// ------------------------------------------------------------------------------------------------ dQ
template <int D, bool CAUSAL, int QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnParams P) {
    constexpr int BQ = 4 * QT * 16;
    constexpr int BKV = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BKV * D * 2;
    using Img = TileImg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y;
    const int qblk = CAUSAL ? (gridDim.x - 1 - blockIdx.x) : blockIdx.x;
    const int hk = h / (P.H / P.Hkv);
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SqE = sp.SqE;
    const int q0 = qblk * BQ, wq0 = q0 + wave * (QT * 16);
    const int coff = sk_len - sq_len;
    bf16* dqbase = P.dq + (int64_t)b * P.dq_sb + (int64_t)h * P.dq_sh;
    const int64_t dq_ss = P.dq_ss;
    if (sp.qst > 0) {
        if (qblk == 0) zero_head_rows<D, 256>(dqbase, dq_ss, sp.qst, tid);
        dqbase += (int64_t)sp.qst * dq_ss;
    }

    if (q0 >= sq_len) {
        for (int i = tid; i < BQ * (D / 8); i += 256) {
            const int r = q0 + i / (D / 8), c = i % (D / 8);
            if (r < SqE) st_bf16x8(dqbase + (int64_t)r * dq_ss + c * 8, zero_bf16x8());
        }
        return;
    }

    bf16x8 qf[QT][DS], dof[QT][DS];
    float lse2[QT], dlt[QT];
    {
        const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
        const bf16* dobase = P.dout + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh + (int64_t)sp.qst * P.o_ss;
        const float* lsep = P.lse + ((int64_t)b * P.H + h) * P.Sq + sp.qst;
        const float* dlp = P.delta + ((int64_t)b * P.H + h) * P.Sq + sp.qst;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int qrow = wq0 + qt * 16 + t;
            const bool ok = qrow < sq_len;
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                qf[qt][ds] = ok ? ld_bf16x8(qbase + (int64_t)qrow * P.q_ss + ds * 32 + g * 8) : zero_bf16x8();
                dof[qt][ds] = ok ? ld_bf16x8(dobase + (int64_t)qrow * P.o_ss + ds * 32 + g * 8) : zero_bf16x8();
            }
            lse2[qt] = ok ? lsep[qrow] * kLog2e : 0.f;
            dlt[qt] = ok ? dlp[qrow] : 0.f;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                pin_loaded(qf[qt][ds]);
                pin_loaded(dof[qt][ds]);
            }
            pin_loaded(lse2[qt]);
            pin_loaded(dlt[qt]);
        }
    }

    int kv_end = sk_len;
    if (CAUSAL) kv_end = min(sk_len, q0 + BQ + coff);
    const int nblk = kv_end > 0 ? (kv_end + BKV - 1) / BKV : 0;
    const bf16* kbase = P.k + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;
    const bf16* vbase = P.v + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;

    f32x4 dqacc[DT][QT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) dqacc[dt][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float sl2 = P.scale * kLog2e;

    TileStage<D, BKV, 256> sk, sv;
    if (nblk > 0) {
        sk.gload(kbase, P.k_ss, 0, sk_len, tid);
        sv.gload(vbase, P.k_ss, 0, sk_len, tid);
        sk.lstore_row(smem, tid);
        sv.lstore_row(smem + TILE, tid);
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
            const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
            bf16x8 dsb[QT][2];
            // Fragment reads run one batch of 4 ahead of the MFMAs that consume them (double-buffered registers, counted
            // lgkmcnt from the compiler), as in the dK/dV kernel: DS steps of K row-fragments (S), 4 steps of V row-fragments
            // (dP, softmax after each key tile), 2*DT/4 steps of K column-fragments (dQ).
            bf16x8 fr[2][4];
            constexpr int NA = DS, NB = 4 * (DS / 4 > 0 ? DS / 4 : 1), NC = 2 * (DT / 4);
            static_assert(DS == 2 || DS == 4, "head dims 64 / 128");
            constexpr int VB = DS == 4 ? 1 : 2;  // key tiles per V batch (4 fragments = VB key tiles x DS d steps)
            constexpr int NBs = 4 / VB;          // V steps
            auto load_step = [&](int buf, int st) {
                if (st < NA) {  // K row-fragments of d step st, key tiles 0..3
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) fr[buf][kt] = Img::frag_row(kt_, kt * 16, st, lane);
                } else if (st < NA + NBs) {  // V row-fragments of VB key tiles, all d steps
                    const int b0 = (st - NA) * VB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) fr[buf][e] = Img::frag_row(vt_, (b0 + e / DS) * 16, e % DS, lane);
                } else {  // K column-fragments: q-row half ks, d tiles 4*dq4 .. +3
                    const int c = st - NA - NBs, ks = c / (DT / 4), dq4 = c % (DT / 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        fr[buf][e] = Img::frag_col_rowimg(kt_, (dq4 * 4 + e) * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
                }
            };
            constexpr int NST = NA + NBs + NC;
            (void)NB;
            load_step(0, 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int buf = st & 1;
                if (st + 1 < NST) load_step(buf ^ 1, st + 1);
                if (st < NA) {
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            s[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[buf][kt], qf[qt][st], s[kt][qt], 0, 0, 0);
                } else if (st < NA + NBs) {
                    const int b0 = (st - NA) * VB;
#pragma unroll
                    for (int vb = 0; vb < VB; ++vb) {
                        const int kt = b0 + vb;
                        f32x4 dp[QT];
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) dp[qt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ds = 0; ds < DS; ++ds)
#pragma unroll
                            for (int qt = 0; qt < QT; ++qt)
                                dp[qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[buf][vb * DS + ds], dof[qt][ds], dp[qt], 0, 0, 0);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            const int qidx = wq0 + qt * 16 + t;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float p = fast_exp2(fmaf(s[kt][qt][r], sl2, -lse2[qt]));
                                if (need_mask) {
                                    const int kidx = kv0 + kt * 16 + g * 4 + r;
                                    if (kidx >= sk_len || (CAUSAL && kidx > qidx + coff)) p = 0.f;
                                }
                                if (qidx >= sq_len) p = 0.f;
                                const float dsv = p * (dp[qt][r] - dlt[qt]);
                                dsb[qt][kt >> 1][(kt & 1) * 4 + r] = (bf16)dsv;
                            }
                        }
                    }
                } else {
                    const int c = st - NA - NBs, ks = c / (DT / 4), dq4 = c % (DT / 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            dqacc[dq4 * 4 + e][qt] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[buf][e], dsb[qt][ks], dqacc[dq4 * 4 + e][qt], 0, 0, 0);
                }
            }
        }
        if (j + 1 < nblk) {
            char* nk = smem + ((j + 1) & 1) * (2 * TILE);
            sk.lstore_row(nk, tid);
            sv.lstore_row(nk + TILE, tid);
        }
        __syncthreads();
    }

#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int qrow = wq0 + qt * 16 + t;
        if (qrow < SqE) {
            const float sc = (qrow < sq_len) ? P.scale : 0.f;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (bf16)(dqacc[dt][qt][r] * sc);
                st_bf16x4(dqbase + (int64_t)qrow * dq_ss + dt * 16 + g * 4, o);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ dK, dV
template <int D, bool CAUSAL, int KT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(AttnParams P) {
    constexpr int BKEYS = 4 * KT * 16;
    constexpr int BQ = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BQ * D * 2;
    using Img = TileImg<D>;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][Q tile | dO tile | lse 64 f32 | delta 64 f32]
    constexpr int BUF = 2 * TILE + 512;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, t = lane & 15;
    const int b = blockIdx.z, hk = blockIdx.y;
    const int group = P.H / P.Hkv;
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SkE = sp.SkE;
    const int k0 = blockIdx.x * BKEYS, wk0 = k0 + wave * (KT * 16);
    const int coff = sk_len - sq_len;
    bf16* dkbase = P.dk + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
    bf16* dvbase = P.dv + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
    const int64_t dk_ss = P.dk_ss;
    if (sp.kst > 0) {
        if (blockIdx.x == 0) {
            zero_head_rows<D, 256>(dkbase, dk_ss, sp.kst, tid);
            zero_head_rows<D, 256>(dvbase, dk_ss, sp.kst, tid);
        }
        dkbase += (int64_t)sp.kst * dk_ss;
        dvbase += (int64_t)sp.kst * dk_ss;
    }

    // this wave's keys as B operands: lane = key t of tile kt, d = ds*32 + g*8 ..
    bf16x8 kfB[KT][DS], vfB[KT][DS];
    {
        const bf16* kbase = P.k + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;
        const bf16* vbase = P.v + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int krow = wk0 + kt * 16 + t;
            const bool ok = krow < sk_len;
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                kfB[kt][ds] = ok ? ld_bf16x8(kbase + (int64_t)krow * P.k_ss + ds * 32 + g * 8) : zero_bf16x8();
                vfB[kt][ds] = ok ? ld_bf16x8(vbase + (int64_t)krow * P.k_ss + ds * 32 + g * 8) : zero_bf16x8();
            }
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) {
                pin_loaded(kfB[kt][ds]);
                pin_loaded(vfB[kt][ds]);
            }
    }
    f32x4 dkacc[DT][KT], dvacc[DT][KT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            dkacc[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dvacc[dt][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const float sl2 = P.scale * kLog2e;

    // q blocks that can see this key block
    int qb_begin = 0;
    if (CAUSAL) qb_begin = max(0, k0 - coff) / BQ;
    const int qb_end = (sq_len + BQ - 1) / BQ;
    const int nq = (k0 < sk_len && qb_end > qb_begin) ? (qb_end - qb_begin) : 0;
    const int niter = nq * group;  // iterate (head in group, q block)

    TileStage<D, BQ, 256> sq, sdo;
    float stat = 0.f;
    const uint32_t goff_q = TileStage<D, BQ, 256>::thread_goff(P.q_ss, tid);
    const uint32_t goff_o = TileStage<D, BQ, 256>::thread_goff(P.o_ss, tid);
    const uint32_t loff = TileStage<D, BQ, 256>::thread_loff_row(tid);
    auto gload = [&](int it) {
        const int hq = hk * group + it / nq;
        const int qb = qb_begin + it % nq;
        const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)hq * P.q_sh + (int64_t)sp.qst * P.q_ss;
        const bf16* dobase = P.dout + (int64_t)b * P.o_sb + (int64_t)hq * P.o_sh + (int64_t)sp.qst * P.o_ss;
        if (qb * BQ + BQ <= sq_len) {  // whole tile inside the sequence: strength-reduced addressing
            sq.gload_full(qbase, P.q_ss, qb * BQ, goff_q);
            sdo.gload_full(dobase, P.o_ss, qb * BQ, goff_o);
        } else {
            sq.gload(qbase, P.q_ss, qb * BQ, sq_len, tid);
            sdo.gload(dobase, P.o_ss, qb * BQ, sq_len, tid);
        }
        if (tid < 128) {
            const int qi = qb * BQ + (tid & 63);
            const float* src = (tid < 64 ? P.lse : P.delta) + ((int64_t)b * P.H + hq) * P.Sq + sp.qst;
            stat = (qi < sq_len) ? src[qi] : 0.f;
        }
    };
    auto lstore = [&](int buf) {
        char* base = smem + buf * BUF;
        sq.lstore_row_full(base, loff);
        sdo.lstore_row_full(base + TILE, loff);
        if (tid < 128) reinterpret_cast<float*>(base + 2 * TILE)[tid] = (tid < 64) ? stat * kLog2e : stat;
    };

    if (niter > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();

    for (int it = 0; it < niter; ++it) {
        if (it + 1 < niter) gload(it + 1);
        const char* qt_ = smem + (it & 1) * BUF;
        const char* dot_ = qt_ + TILE;
        const float* lse_s = reinterpret_cast<const float*>(qt_ + 2 * TILE);
        const float* dlt_s = lse_s + 64;
        const int qb0 = (qb_begin + it % nq) * BQ;
        const bool wave_active = (wk0 < sk_len) && !(CAUSAL && wk0 > qb0 + BQ - 1 + coff);
        if (wave_active) {
            const bool need_mask = (qb0 + BQ > sq_len) || (wk0 + KT * 16 > sk_len) || (CAUSAL && (wk0 + KT * 16 - 1 > qb0 + coff));
            bf16x8 pb[KT][2], dsb[KT][2];
            // Fragment reads are issued one batch (4 fragments = the operands of the next 4*KT MFMAs) AHEAD of their use and
            // double-buffered in registers: the compiler's counted lgkmcnt then lets a batch's MFMAs start while the next
            // batch is still in flight.  Read-then-use in source order made every MFMA pair wait out a full LDS round trip
            // (78 s_waitcnt per 64 MFMAs, MFMA pipe 18 % busy in rocprofv3 PMC).
            bf16x8 fr[2][4];
            auto load_p1 = [&](int buf, int qt, int h) {  // Q and dO row-fragments of q tile qt, d steps 2h and 2h+1
                fr[buf][0] = Img::frag_row(qt_, qt * 16, 2 * h, lane);
                fr[buf][1] = Img::frag_row(dot_, qt * 16, 2 * h, lane);
                fr[buf][2] = Img::frag_row(qt_, qt * 16, 2 * h + 1, lane);
                fr[buf][3] = Img::frag_row(dot_, qt * 16, 2 * h + 1, lane);
            };
            auto load_p2 = [&](int buf, int ks, int dp2) {  // dO and Q column-fragments of d tiles 2*dp2, 2*dp2+1, q rows of ks
                fr[buf][0] = Img::frag_col_rowimg(dot_, (2 * dp2) * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
                fr[buf][1] = Img::frag_col_rowimg(qt_, (2 * dp2) * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
                fr[buf][2] = Img::frag_col_rowimg(dot_, (2 * dp2 + 1) * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
                fr[buf][3] = Img::frag_col_rowimg(qt_, (2 * dp2 + 1) * 16, (2 * ks) * 16, (2 * ks + 1) * 16, lane);
            };
            static_assert(DS == 2 || DS == 4, "two d steps per batch");
            constexpr int NH = DS / 2;          // batches per q tile in phase 1
            constexpr int NP1 = 4 * NH;         // phase-1 steps
            constexpr int NP2 = 2 * (DT / 2);   // phase-2 steps
            f32x4 s[KT], dp[KT];
            load_p1(0, 0, 0);
#pragma unroll
            for (int st = 0; st < NP1; ++st) {
                const int qt = st / NH, h = st % NH;
                if (st + 1 < NP1)
                    load_p1((st + 1) & 1, (st + 1) / NH, (st + 1) % NH);
                else
                    load_p2((st + 1) & 1, 0, 0);
                if (h == 0) {
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                        dp[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int ds = 2 * h + e;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[st & 1][2 * e], kfB[kt][ds], s[kt], 0, 0, 0);
                        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[st & 1][2 * e + 1], vfB[kt][ds], dp[kt], 0, 0, 0);
                    }
                }
                if (h == NH - 1) {
                    // acc: lane col = key t, rows q = qt*16 + g*4 + r
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 16 + g * 4);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(dlt_s + qt * 16 + g * 4);
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        const int kidx = wk0 + kt * 16 + t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float p = fast_exp2(fmaf(s[kt][r], sl2, -l4[r]));
                            if (need_mask) {
                                const int qidx = qb0 + qt * 16 + g * 4 + r;
                                if (qidx >= sq_len || kidx >= sk_len || (CAUSAL && kidx > qidx + coff)) p = 0.f;
                            }
                            pb[kt][qt >> 1][(qt & 1) * 4 + r] = (bf16)p;
                            dsb[kt][qt >> 1][(qt & 1) * 4 + r] = (bf16)(p * (dp[kt][r] - d4[r]));
                        }
                    }
                }
            }
#pragma unroll
            for (int st = 0; st < NP2; ++st) {
                const int ks = st / (DT / 2), dp2 = st % (DT / 2);
                const int buf = (NP1 + st) & 1;
                if (st + 1 < NP2) load_p2(buf ^ 1, (st + 1) / (DT / 2), (st + 1) % (DT / 2));
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int dt = 2 * dp2 + e;
#pragma unroll
                    for (int kt = 0; kt < KT; ++kt) {
                        dvacc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[buf][2 * e], pb[kt][ks], dvacc[dt][kt], 0, 0, 0);
                        dkacc[dt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[buf][2 * e + 1], dsb[kt][ks], dkacc[dt][kt], 0, 0, 0);
                    }
                }
            }
        }
        if (it + 1 < niter) lstore((it + 1) & 1);
        __syncthreads();
    }

    // lane holds d = dt*16 + g*4 + r of key t
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int krow = wk0 + kt * 16 + t;
        if (krow < SkE) {
            const bool ok = krow < sk_len;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                bf16x4 ok_, ov_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ok_[r] = (bf16)(ok ? dkacc[dt][kt][r] * P.scale : 0.f);
                    ov_[r] = (bf16)(ok ? dvacc[dt][kt][r] : 0.f);
                }
                st_bf16x4(dkbase + (int64_t)krow * dk_ss + dt * 16 + g * 4, ok_);
                st_bf16x4(dvbase + (int64_t)krow * dk_ss + dt * 16 + g * 4, ov_);
            }
        }
    }
}

template <int D, bool CAUSAL>
int launch_bwd(const AttnParams& P, hipStream_t stream) {
    constexpr int QT = (D == 128) ? 1 : 2;
    constexpr int KT = (D == 128) ? 1 : 2;
    constexpr int LDS_DQ = 2 * 2 * 64 * D * 2;
    constexpr int LDS_DKV = 2 * (2 * 64 * D * 2 + 512);
    static std::atomic<uint64_t> lds_ok{0}, lds2_ok{0};
    dllm_ensure_dyn_lds(&attn_bwd_dq_kernel<D, CAUSAL, QT>, LDS_DQ, lds_ok);
    dllm_ensure_dyn_lds(&attn_bwd_dkv_kernel<D, CAUSAL, KT>, LDS_DKV, lds2_ok);
    const int64_t rows = (int64_t)P.B * P.Sq * P.H;
    const int64_t nthreads = rows * (D / 8);
    hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, P);
    constexpr int BQ = 4 * QT * 16, BKEYS = 4 * KT * 16;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<D, CAUSAL, QT>), dim3((P.Sq + BQ - 1) / BQ, P.H, P.B), dim3(256), LDS_DQ, stream, P);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, CAUSAL, KT>), dim3((P.Sk + BKEYS - 1) / BKEYS, P.Hkv, P.B), dim3(256), LDS_DKV,
                       stream, P);
    return dllm_check_launch();
}

}  // namespace

extern "C" {

// dout/o share the o strides; q and k/v strides as in dllm_attn_fwd; dq is a [B,Sq,H,D] view with strides dq_*, dk/dv are
// [B,Sk,Hkv,D] views sharing strides dk_* (so gradients can be written straight into a packed dQKV buffer);
// delta: fp32 [B,H,Sq] workspace.
int dllm_attn_bwd(const void* dout, const void* q, const void* k, const void* v, const void* o, const float* lse, float* delta,
                  void* dq, void* dk, void* dv, const int* seqlens, const int* seqstart, int B, int H, int Hkv, int Sq, int Sk, int D,
                  int64_t q_sb,
                  int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                  int64_t dq_sb, int64_t dq_ss, int64_t dq_sh, int64_t dk_sb, int64_t dk_ss, int64_t dk_sh, float scale,
                  int causal, void* stream) {
    if (B < 0 || H <= 0 || Hkv <= 0 || Sq < 0 || Sk < 0 || (H % Hkv) != 0) return DLLM_ERR_SHAPE;
    if (D != 64 && D != 128) return DLLM_ERR_SHAPE;
    if (B == 0 || Sq == 0 || Sk == 0) return DLLM_OK;
    if ((q_ss | q_sh | q_sb | k_ss | k_sh | k_sb | o_ss | o_sh | o_sb) & 7) return DLLM_ERR_ALIGN;
    if ((dq_ss | dq_sh | dq_sb | dk_ss | dk_sh | dk_sb) & 3) return DLLM_ERR_ALIGN;
    if (seqlens != nullptr && Sq != Sk) return DLLM_ERR_SHAPE;
    if (lse == nullptr || delta == nullptr) return DLLM_ERR_SHAPE;
    AttnParams P{};
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.dout = (const bf16*)dout;
    P.dq = (bf16*)dq; P.dk = (bf16*)dk; P.dv = (bf16*)dv; P.lse = (float*)lse; P.delta = delta; P.seqlens = seqlens;
    P.seqstart = seqstart;
    P.B = B; P.H = H; P.Hkv = Hkv; P.Sq = Sq; P.Sk = Sk;
    P.q_sb = q_sb; P.q_ss = q_ss; P.q_sh = q_sh; P.k_sb = k_sb; P.k_ss = k_ss; P.k_sh = k_sh;
    P.o_sb = o_sb; P.o_ss = o_ss; P.o_sh = o_sh; P.scale = scale; P.causal = causal;
    P.dq_sb = dq_sb; P.dq_ss = dq_ss; P.dq_sh = dq_sh; P.dk_sb = dk_sb; P.dk_ss = dk_ss; P.dk_sh = dk_sh;
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) return causal ? launch_bwd<128, true>(P, s) : launch_bwd<128, false>(P, s);
    return causal ? launch_bwd<64, true>(P, s) : launch_bwd<64, false>(P, s);
}

}  // extern "C"
