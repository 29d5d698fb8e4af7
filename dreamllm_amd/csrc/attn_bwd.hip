// Flash-attention backward for gfx950 (dQ, dK, dV), same layouts and masking rules as attn_fwd.hip.
//
// Replaces the autograd of modeling_dreamllm.py:357-379 / flash_attn's backward behind modeling_dreamllm.py:532-549
// and the attention backward inside the frozen UNet (dgrad only) [ext].
//
// No atomics, deterministic.  Two kernel families (chosen per call like the forward's):
//   4-wave kernels (short axes):
//   1. delta[b,h,q] = sum_d dO * O                                   (HBM-bound preprocess)
//   2. dQ  : block = 4 waves x (QT*16) queries, loops over KV blocks; recomputes S^T = K Q^T and dP^T = V dO^T with
//            K and V tiles in LDS; dQ^T += K^T dS^T reads the same K tile through transpose reads.
//   3. dK,dV: block = 4 waves x (KT*16) keys held in registers, loops over Q blocks (and over the query heads of a
//            GQA group); recomputes S = Q K^T and dP = dO V^T from Q / dO tiles in LDS, then dV^T += dO^T P and
//            dK^T += Q^T dS read the same tiles through transpose reads.
//   8-wave pipelined kernels (axes >= 512): attn_bwd_dq8_kernel (computes delta itself and publishes the statistic planes),
//   then attn_bwd_dkv8_kernel (fused at head_dim 64, split into a dK and a dV pass at head_dim 128); see their headers below.
// P = exp(scale*S - LSE) uses the forward's saved log-sum-exp; dS = P * (dP - delta).
#include "attn_common.h"

__attribute__((visibility("hidden"))) int dllm_launch_attn_bwd_dq_pp(const AttnParams& P, int causal, void* tl_out, hipStream_t stream);

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
    if (row < rows && (threadIdx.x % TPR) == 0) {
        // workspace planes: 0 = delta, 1 = -delta, 2 = -lse / scale (the accumulator seeds of the 8-wave dK/dV kernel)
        const int64_t i = ((int64_t)b * P.H + h) * P.Sq + q;
        P.delta[i] = s;
        P.delta[rows + i] = -s;
        P.delta[2 * rows + i] = -P.lse[i] / P.scale;
    }
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
    const int nqb = (P.Sq + BQ - 1) / BQ;
    const AttnBlock bm = attn_block_map<false>(nqb, P.H, P.B);
    if (!bm.valid) return;
    const int b = bm.b, h = bm.h;
    const int qblk = CAUSAL ? (nqb - 1 - bm.r) : bm.r;
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

// ================================================================================================ 8-wave pipelined dQ
// Same skeleton as attn_fwd8_kernel (attn_fwd.hip): 8 waves x 32 queries, K / V tiles of 64 keys by LDS-DMA into the UNIFIED
// image (attn_common.h: K is read as row fragments for S^T = K Q^T and as column fragments for dQ^T += K^T dS^T), every LDS read
// inline asm, requested PRE fragments ahead and retired by counted lgkmcnt.  A 32-query wave tile halves the LDS bytes per MFMA
// of the 16-query 4-wave kernel above, which at head_dim 128 is LDS-bandwidth bound at half the matrix rate.
// One tile = 6 segments of 2*DS fragment steps: {K,V row frags (alternating) of key tile kt -> S^T[kt], dP^T[kt]} for kt = 0..3, then
// {K col frags -> dQ^T} for the two 32-key halves.  The softmax algebra of key tile kt (exp2, * (dP - delta), bf16 pack) is
// spread over the steps of the NEXT segment, i.e. under MFMAs that do not depend on it; delta is folded into the dP
// accumulator's initial value.
template <int D, bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq8_kernel(AttnParams P) {
    constexpr int QT = 2, NW = 8;
    constexpr int BQ = NW * QT * 16;  // 256
    constexpr int BKV = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BKV * D * 2;
    constexpr int PITCH = D * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K[2] then V[2]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    const int nqb = (P.Sq + BQ - 1) / BQ;
    // causal: one work-group per PAIR of query blocks (nqb-1-r, r), as in attn_fwd8_kernel: uniform work, head-major order
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
        bf16* dqbase = P.dq + (int64_t)b * P.dq_sb + (int64_t)h * P.dq_sh;
        const int64_t dq_ss = P.dq_ss;
        if (sp.qst > 0) {
            if (qblk == 0) zero_head_rows<D, 512>(dqbase, dq_ss, sp.qst, tid);
            dqbase += (int64_t)sp.qst * dq_ss;
        }
        if (q0 >= sq_len) {
            for (int i = tid; i < BQ * (D / 8); i += 512) {
                const int r = q0 + i / (D / 8), c = i % (D / 8);
                if (r < SqE) st_bf16x8(dqbase + (int64_t)r * dq_ss + c * 8, zero_bf16x8());
            }
            continue;
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

        char* const Kb0 = smem;
        char* const Vb0 = smem + 2 * TILE;
        // LDS-DMA of one tile: the swizzle of the unified image goes on the per-lane SOURCE chunk (the DMA writes lane-linear)
        constexpr int CPR = D / 8, RPG = 64 / CPR, NDMA = (BKV / RPG) / NW;
        const int drow = lane / CPR, dpos = lane % CPR;
        int src_chunk[NDMA];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) src_chunk[i] = dpos ^ uni_f<D>((wave * NDMA + i) * RPG + drow);
        auto dma_tile = [&](int row0, int buf) {
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int grp = wave * NDMA + i;
                const int row = min(row0 + grp * RPG + drow, sk_len - 1);  // rows past the end: finite data, masked below
                GLDS16_(kbase + (int64_t)row * P.k_ss + src_chunk[i] * 8, Kb0 + buf * TILE + grp * 1024);
                GLDS16_(vbase + (int64_t)row * P.k_ss + src_chunk[i] * 8, Vb0 + buf * TILE + grp * 1024);
            }
        };
        auto active = [&](int j) { return (wq0 < sq_len) && !(CAUSAL && j * BKV > wq0 + QT * 16 - 1 + coff); };

        // ---- fragment stream
        constexpr int NS = 2 * DS;        // steps per segment (= DT)
        constexpr int NSTEP = 6 * NS;
        constexpr int PRE = 3, RING = PRE + 1;  // fragments in flight; 2...5 measure the same (+-1 %), 6 is 2 % slower
        u32x4 ring[RING];  // one ring for both fragment kinds: a column fragment = two transpose reads composed into one register quad
        uint32_t raddr[DS];   // row fragments: (row t, chunk ds*4 + g) of the unified image; start at buffer 0, toggled per tile
#pragma unroll
        for (int ds = 0; ds < DS; ++ds) raddr[ds] = lds_addr32(Kb0 + t * PITCH + (((ds * 4 + g) ^ uni_f<D>(t)) << 4));
        // column fragments: row rr = g*4 + t/4, 8 bytes at d = dt*16 + 4*(t&3): chunk dt*2 + ((t>>1)&1), byte (t&1)*8
        const int rr = g * 4 + (t >> 2);
        uint32_t caddr = lds_addr32(Kb0 + rr * PITCH + (t & 1) * 8 + ((((t >> 1) & 1) ^ (uni_f<D>(rr) & 1)) << 4));
        const uint32_t cswz = (uint32_t)((uni_f<D>(rr) >> 1) << 5);
        auto is_col = [](int f) { return f >= 4 * (2 * (D / 32)); };
        auto issue = [&ring, &raddr, &caddr, cswz](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int seg = f / NS, i = f % NS;
            if constexpr (seg < 4) {  // K (even i) or V row fragment of key tile seg, d step i / 2: four accumulator chains in rotation
                constexpr int off = seg * 16 * PITCH + ((i & 1) == 0 ? 0 : 2 * TILE);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % RING]) : "v"(raddr[i >> 1]), "n"(off));
            } else {                  // K column fragment of d tile i, keys of half seg - 4
                constexpr int ks = seg - 4;
                const uint32_t a = caddr + ((uint32_t)(i << 5) ^ cswz);
                u32x2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"((2 * ks) * 16 * PITCH));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"((2 * ks + 1) * 16 * PITCH));
                // NOTE: the quad is composed right after the two transpose reads are ISSUED, i.e. before their data has arrived: this
            // is correct only because hipcc coalesces lo / hi into the sub-registers of the quad (no v_mov is emitted; the
            // counted s_waitcnt in wait_frag() pins the quad).  Copies here would read registers still in flight -- every
            // attention test would fail, which is the guard.
            ring[f % RING] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        };
        auto wait_frag = [&ring](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int last = f + PRE < NSTEP ? f + PRE : NSTEP - 1;
            constexpr int younger = [](int f0, int l0) { int n = 0; for (int i = f0 + 1; i <= l0; ++i) n += (i < 8 * (D / 32) ? 1 : 2); return n; }(f, last);
            static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[f % RING]) : "n"(younger) : "memory");
        };
        (void)is_col;

        if (nblk > 0) dma_tile(0, 0);

        // operand loads AFTER the first tile's DMA is in flight: the two latencies overlap instead of adding up
        bf16x8 qf[QT][DS], dof[QT][DS];
        float nlse2[QT], ndlt[QT];
        {
            const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
            const bf16* dobase = P.dout + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh + (int64_t)sp.qst * P.o_ss;
            const bf16* obase = P.o + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh + (int64_t)sp.qst * P.o_ss;
            const int64_t stat0 = ((int64_t)b * P.H + h) * P.Sq + sp.qst, plane = (int64_t)P.B * P.H * P.Sq;
            const float* lsep = P.lse + stat0;
            // delta = rowsum(dO * O) is computed HERE (this is the first backward kernel; no separate preprocess launch): a lane
            // holds 32 of its row's d values per operand, the other three quarters sit in lanes t+16, t+32, t+48.  The three
            // statistic planes of the workspace (delta, -delta, -lse/scale) are published for the dK / dV kernels that follow.
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const int qrow = wq0 + qt * 16 + t;
                const bool ok = qrow < sq_len;  // rows past the end: all-zero operands => S = 0, P = 1, dP - delta = 0, dS = 0
                float dsum = 0.f;
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    qf[qt][ds] = ok ? ld_bf16x8(qbase + (int64_t)qrow * P.q_ss + ds * 32 + g * 8) : zero_bf16x8();
                    dof[qt][ds] = ok ? ld_bf16x8(dobase + (int64_t)qrow * P.o_ss + ds * 32 + g * 8) : zero_bf16x8();
                    const bf16x8 of = ok ? ld_bf16x8(obase + (int64_t)qrow * P.o_ss + ds * 32 + g * 8) : zero_bf16x8();
#pragma unroll
                    for (int e = 0; e < 8; ++e) dsum += (float)of[e] * (float)dof[qt][ds][e];
                }
                dsum += __shfl_xor(dsum, 16, 64);
                dsum += __shfl_xor(dsum, 32, 64);
                const float lse = ok ? lsep[qrow] : 0.f;
                nlse2[qt] = -lse * kLog2e;
                ndlt[qt] = -dsum;
                if (ok && g == 0) {
                    P.delta[stat0 + qrow] = dsum;
                    P.delta[plane + stat0 + qrow] = -dsum;
                    P.delta[2 * plane + stat0 + qrow] = -lse / P.scale;
                }
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    pin_loaded(qf[qt][ds]);
                    pin_loaded(dof[qt][ds]);
                }
                pin_loaded(nlse2[qt]);
                pin_loaded(ndlt[qt]);
            }
        }
        __syncthreads();

        for (int j = 0; j < nblk; ++j) {
            const int kv0 = j * BKV;
            if (j + 1 < nblk) dma_tile(kv0 + BKV, (j + 1) & 1);
            if (active(j)) {
                const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
                f32x4 s[2][QT], dp[2][QT];   // scores / dP - delta of the key tile being accumulated and of the previous one
                bf16x8 dsb[QT][2];           // dS^T as B operands: [k = 32 keys of half ks][n = query]
                // softmax algebra of elements [e0, e0 + n) of key tile kt (element e: qt = e / 4, r = e % 4)
                auto sm_slice = [&](auto ktc, int e0, int n) {
                    constexpr int kt = decltype(ktc)::value;
#pragma unroll
                    for (int e = e0; e < e0 + n; e += 2) {
                        const int qt = e >> 2, r = e & 3;
                        const float p0 = fast_exp2(fmaf(s[kt & 1][qt][r], sl2, nlse2[qt]));
                        const float p1 = fast_exp2(fmaf(s[kt & 1][qt][r + 1], sl2, nlse2[qt]));
                        bf16x2 w;
                        w[0] = (bf16)(p0 * dp[kt & 1][qt][r]);
                        w[1] = (bf16)(p1 * dp[kt & 1][qt][r + 1]);
                        uint32_t u = __builtin_bit_cast(uint32_t, w);
                        asm volatile("" : "+v"(u));
                        w = __builtin_bit_cast(bf16x2, u);
                        dsb[qt][kt >> 1][(kt & 1) * 4 + r] = w[0];
                        dsb[qt][kt >> 1][(kt & 1) * 4 + r + 1] = w[1];
                    }
                };
                auto mask_tile = [&](auto ktc) {  // diagonal / ragged tiles only: dead (key, query) pairs get P = exp2(-inf) = 0
                    constexpr int kt = decltype(ktc)::value;
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const int qidx = wq0 + qt * 16 + t;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int kidx = kv0 + kt * 16 + g * 4 + r;
                            const bool dead = kidx >= sk_len || (CAUSAL && kidx > qidx + coff);
                            s[kt & 1][qt][r] = dead ? -INFINITY : s[kt & 1][qt][r];
                        }
                    }
                };
                constexpr int NE = QT * 4;  // softmax elements per key tile and lane
                static_for_<0, PRE>([&](auto fc) { issue(fc); });
                static_for_<0, NSTEP>([&](auto sc) {
                    constexpr int st = decltype(sc)::value;
                    constexpr int seg = st / NS, i = st % NS;
                    if constexpr (st + PRE < NSTEP) issue(std::integral_constant<int, st + PRE>{});
                    if constexpr (seg < 4 && i == 0) {
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            s[seg & 1][qt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            dp[seg & 1][qt] = f32x4{ndlt[qt], ndlt[qt], ndlt[qt], ndlt[qt]};
                        }
                    }
                    if constexpr (seg >= 1 && seg <= 4 && i == 0) {
                        if (need_mask) mask_tile(std::integral_constant<int, seg - 1>{});
                    }
                    wait_frag(sc);
                    if constexpr (seg < 4) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[st % RING]);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            if constexpr ((i & 1) == 0)
                                s[seg & 1][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[qt][i >> 1], s[seg & 1][qt], 0, 0, 0);
                            else
                                dp[seg & 1][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dof[qt][i >> 1], dp[seg & 1][qt], 0, 0, 0);
                        }
                    } else {
                        constexpr int ks = seg - 4;
                        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[st % RING]);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
                            dqacc[i][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dsb[qt][ks], dqacc[i][qt], 0, 0, 0);
                    }
                    // the previous key tile's softmax algebra, under this segment's MFMAs
                    if constexpr (seg >= 1 && seg <= 4) sm_slice(std::integral_constant<int, seg - 1>{}, i * NE / NS, NE / NS);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
#pragma unroll
            for (int ds = 0; ds < DS; ++ds) raddr[ds] ^= (uint32_t)TILE;  // the other K / V buffer (dynamic LDS starts at address 0)
            caddr ^= (uint32_t)TILE;
            __syncthreads();
        }

#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int qrow = wq0 + qt * 16 + t;
            if (qrow < SqE) {
                const bool ok = qrow < sq_len;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (bf16)(ok ? dqacc[dt][qt][r] * P.scale : 0.f);
                    st_bf16x4(dqbase + (int64_t)qrow * dq_ss + dt * 16 + g * 4, o);
                }
            }
        }
    }  // pass
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
    const AttnBlock bm = attn_block_map<false>((P.Sk + BKEYS - 1) / BKEYS, P.Hkv, P.B);  // key block 0 sees the most queries: heavy first
    if (!bm.valid) return;
    const int b = bm.b, hk = bm.h;
    const int group = P.H / P.Hkv;
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SkE = sp.SkE;
    const int k0 = bm.r * BKEYS, wk0 = k0 + wave * (KT * 16);
    const int coff = sk_len - sq_len;
    bf16* dkbase = P.dk + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
    bf16* dvbase = P.dv + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
    const int64_t dk_ss = P.dk_ss;
    if (sp.kst > 0) {
        if (bm.r == 0) {
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

// ================================================================================================ 8-wave pipelined dK, dV
// 8 waves x 32 keys (K and V of the wave's keys live in registers as B operands, dK^T / dV^T accumulate in registers), Q / dO
// tiles of 64 query rows stream through LDS by DMA into the unified image (read as row fragments for S = Q K^T, dP = dO V^T and as
// column fragments for dV^T += dO^T P, dK^T += Q^T dS), plus a 128-float statistics tile {-lse/scale, -delta} which seeds the S
// and dP accumulators (first MFMA of each chain takes it as src C): P = exp2(scale*log2e * S'), dS = P * dP' with no further
// per-element bookkeeping.  32-key wave tiles halve the LDS bytes per MFMA of the 16-key 4-wave kernel above; at head_dim 128
// that leaves the 256-register budget (two waves per SIMD) with no room for double-buffered scores, so the exponentials of one
// wave run beside the MFMAs of its SIMD partner rather than beside its own.
// Per 64-row tile and wave: 2 chunks of {2 x [stat, Q/dO row fragments of a 16-row tile -> S, dP; softmax algebra] ; dO^T / Q^T
// column fragments of the 32 rows -> dV^T, dK^T}.
#define GLDS4_(gptr, lptr)                                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                           \
                                     (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// MODE 0: dK and dV in one pass (head_dim 64).  At head_dim 128 the two 32-key accumulators (128 registers) plus the K / V operands
// (64) do not leave room for the rest in 256 registers (hipcc spills the operands into the loop), so the work is split:
// MODE 1 = dV only (S, P, dV: K resident), MODE 2 = dK only (S, dP, dS, dK: K and V resident) -- 5 instead of 4 MFMA units per
// (query, key) pair, each at the LDS bytes-per-MFMA of the 32-key tile.
struct DkvFrag {
    int kind;   // 0 L (seed of S'), 1 Dl (seed of dP'), 2 Q row, 3 dO row, 4 dO^T col (dV), 5 Q^T col (dK)
    int c, u, x;  // chunk, 16-row tile in the chunk, d step (rows) / d tile (cols)
    int first;  // row fragments: first MFMA of its chain (takes the seed read one step earlier)
};
template <int D, int MODE>
struct DkvStream {
    static constexpr int DS = D / 32, DT = D / 16;
    static constexpr bool kDV = MODE != 2, kDK = MODE != 1;
    static constexpr int NP1 = 1 + DS + (kDK ? 1 + DS : 0);
    static constexpr int NFC = 2 * NP1 + (kDV ? DT : 0) + (kDK ? DT : 0);
    static constexpr int NF = 2 * NFC;
    static constexpr DkvFrag at(int f) {
        const int c = f / NFC, i = f % NFC;
        if (i < 2 * NP1) {
            const int u = i / NP1, k = i % NP1;
            if (!kDK) return k == 0 ? DkvFrag{0, c, u, 0, 0} : DkvFrag{2, c, u, k - 1, k == 1};
            if (k == 0) return DkvFrag{0, c, u, 0, 0};
            if (k == 1) return DkvFrag{2, c, u, 0, 1};
            if (k == 2) return DkvFrag{1, c, u, 0, 0};
            if (k == 3) return DkvFrag{3, c, u, 0, 1};
            return DkvFrag{2 + (k & 1), c, u, (k - 2) >> 1, 0};
        }
        const int j = i - 2 * NP1;
        if (kDV && kDK) return DkvFrag{4 + (j & 1), c, 0, j >> 1, 0};
        return DkvFrag{kDV ? 4 : 5, c, 0, j, 0};
    }
    static constexpr int ops(int f) { return at(f).kind >= 4 ? 2 : 1; }
    static constexpr bool last_of_p1(int f) { return (f % NFC) < 2 * NP1 && ((f % NFC) % NP1) == NP1 - 1; }
};

template <int D, bool CAUSAL, int MODE>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv8_kernel(AttnParams P) {
    using St = DkvStream<D, MODE>;
    constexpr bool kDV = St::kDV, kDK = St::kDK;
    constexpr int KT = 2, NW = 8;
    constexpr int BKEYS = NW * KT * 16;  // 256
    constexpr int BQ = 64;
    constexpr int DS = D / 32, DT = D / 16;
    constexpr int TILE = BQ * D * 2, PITCH = D * 2;
    constexpr int BUF = 2 * TILE;        // Q tile | dO tile; two buffers, then 2 x {64 x -lse/scale, 64 x -delta}
    constexpr int STAT0 = 2 * BUF;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, t = lane & 15;
    // causal: one work-group per PAIR of key blocks (r, nkb-1-r) -- key block 0 sees every query, the last one only its own rows
    const int nkb = (P.Sk + BKEYS - 1) / BKEYS;
    const int nitems = CAUSAL ? (nkb + 1) / 2 : nkb;
    const AttnBlock bm = attn_block_map<false>(nitems, P.Hkv, P.B);
    if (!bm.valid) return;
    const int b = bm.b, hk = bm.h;
    const int group = P.H / P.Hkv;
    const int npass = (CAUSAL && nkb - 1 - bm.r != bm.r) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int kblk = (CAUSAL && pass == 1) ? nkb - 1 - bm.r : bm.r;
        const AttnSpan sp = attn_span(P, b);
        const int sq_len = sp.sq_len, sk_len = sp.sk_len, SkE = sp.SkE;
        const int k0 = kblk * BKEYS, wk0 = k0 + wave * (KT * 16);
        const int coff = sk_len - sq_len;
        bf16* dkbase = P.dk + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
        bf16* dvbase = P.dv + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
        const int64_t dk_ss = P.dk_ss;
        if (sp.kst > 0) {
            if (kblk == 0) {
                if constexpr (kDK) zero_head_rows<D, 512>(dkbase, dk_ss, sp.kst, tid);
                if constexpr (kDV) zero_head_rows<D, 512>(dvbase, dk_ss, sp.kst, tid);
            }
            dkbase += (int64_t)sp.kst * dk_ss;
            dvbase += (int64_t)sp.kst * dk_ss;
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

        int qb_begin = 0;
        if (CAUSAL) qb_begin = max(0, k0 - coff) / BQ;
        const int qb_end = (sq_len + BQ - 1) / BQ;
        const int nq = (k0 < sk_len && qb_end > qb_begin) ? (qb_end - qb_begin) : 0;
        const int niter = nq * group;

        // LDS-DMA of one Q / dO tile pair + statistics (swizzle on the per-lane source chunk; rows past the end are clamped: finite
        // data, masked below)
        constexpr int CPR = D / 8, RPG = 64 / CPR, NDMA = (BQ / RPG) / NW;
        const int drow = lane / CPR, dpos = lane % CPR;
        int src_chunk[NDMA];
#pragma unroll
        for (int i = 0; i < NDMA; ++i) src_chunk[i] = dpos ^ uni_f<D>((wave * NDMA + i) * RPG + drow);
        const int64_t plane = (int64_t)P.B * P.H * P.Sq;
        auto dma_tile = [&](int it, int buf) {
            const int hq = hk * group + it / nq;
            const int row0 = (qb_begin + it % nq) * BQ;
            const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)hq * P.q_sh + (int64_t)sp.qst * P.q_ss;
            const bf16* dobase = P.dout + (int64_t)b * P.o_sb + (int64_t)hq * P.o_sh + (int64_t)sp.qst * P.o_ss;
#pragma unroll
            for (int i = 0; i < NDMA; ++i) {
                const int grp = wave * NDMA + i;
                const int row = min(row0 + grp * RPG + drow, sq_len - 1);
                GLDS16_(qbase + (int64_t)row * P.q_ss + src_chunk[i] * 8, smem + buf * BUF + grp * 1024);
                GLDS16_(dobase + (int64_t)row * P.o_ss + src_chunk[i] * 8, smem + buf * BUF + TILE + grp * 1024);
            }
            if (wave < (kDK ? 2 : 1)) {  // planes 2 (-lse/scale) and 1 (-delta) of the statistics workspace
                const float* src = P.delta + (wave == 0 ? 2 : 1) * plane + ((int64_t)b * P.H + hq) * P.Sq + sp.qst;
                GLDS4_(src + min(row0 + lane, sq_len - 1), smem + STAT0 + buf * 512 + wave * 256);
            }
        };

        // ---- fragment stream of one tile (DkvStream).  ring slot = f % RING with RING = PRE + 2: a slot is overwritten two steps
        // after its own step, so a seed vector is still there when the next step's MFMA takes it as src C.
        constexpr int NF = St::NF;
        constexpr int PRE = 3, RING = PRE + 2;
        u32x4 ring[RING];
        // row fragments: tile row t, chunk (ds*4 + g) ^ uni_f(t): base + ((ds << 6) ^ rswz); column fragments as in the dQ kernel
        uint32_t rbase = lds_addr32(smem + t * PITCH + ((g ^ (uni_f<D>(t) & 3)) << 4));
        const uint32_t rswz = (uint32_t)((uni_f<D>(t) >> 2) << 6);
        const int rr = g * 4 + (t >> 2);
        uint32_t cbase = lds_addr32(smem + rr * PITCH + (t & 1) * 8 + ((((t >> 1) & 1) ^ (uni_f<D>(rr) & 1)) << 4));
        const uint32_t cswz = (uint32_t)((uni_f<D>(rr) >> 1) << 5);
        uint32_t sbase = lds_addr32(smem + STAT0 + g * 16);
        auto issue = [&ring, &rbase, &cbase, &sbase, rswz, cswz](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr DkvFrag fr = St::at(f);
            constexpr int qi = 2 * fr.c + fr.u;
            if constexpr (fr.kind < 2) {  // 4 consecutive rows g*4.. of the 16-row tile
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % RING]) : "v"(sbase), "n"(qi * 64 + fr.kind * 256));
            } else if constexpr (fr.kind < 4) {
                const uint32_t a = rbase + ((uint32_t)(fr.x << 6) ^ rswz);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[f % RING]) : "v"(a), "n"((fr.kind - 2) * TILE + qi * 16 * PITCH));
            } else {
                const uint32_t a = cbase + ((uint32_t)(fr.x << 5) ^ cswz);
                u32x2 lo, hi;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(a), "n"((5 - fr.kind) * TILE + (2 * fr.c) * 16 * PITCH));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(a), "n"((5 - fr.kind) * TILE + (2 * fr.c + 1) * 16 * PITCH));
                ring[f % RING] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
        };
        auto wait_frag = [&ring](auto fc) {
            constexpr int f = decltype(fc)::value;
            constexpr int last = f + PRE < NF ? f + PRE : NF - 1;
            constexpr int younger = [](int f0, int l0) { int n = 0; for (int i = f0 + 1; i <= l0; ++i) n += St::ops(i); return n; }(f, last);
            static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
            asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[f % RING]) : "n"(younger) : "memory");
        };

        if (niter > 0) dma_tile(0, 0);

        // operand loads AFTER the first tile's DMA is in flight: the two latencies overlap instead of adding up
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
                    if constexpr (kDK) vfB[kt][ds] = ok ? ld_bf16x8(vbase + (int64_t)krow * P.k_ss + ds * 32 + g * 8) : zero_bf16x8();
                }
            }
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int ds = 0; ds < DS; ++ds) {
                    pin_loaded(kfB[kt][ds]);
                    if constexpr (kDK) pin_loaded(vfB[kt][ds]);
                }
        }
        __syncthreads();

        for (int it = 0; it < niter; ++it) {
            if (it + 1 < niter) dma_tile(it + 1, (it + 1) & 1);
            const int qb0 = (qb_begin + it % nq) * BQ;
            const bool wave_active = (wk0 < sk_len) && !(CAUSAL && wk0 > qb0 + BQ - 1 + coff);
            if (wave_active) {
                const bool need_mask = (qb0 + BQ > sq_len) || (wk0 + KT * 16 > sk_len) || (CAUSAL && (wk0 + KT * 16 - 1 > qb0 + coff));
                f32x4 sc[KT], dp[KT];
                bf16x8 pb[KT], dsb[KT];
                static_for_<0, PRE>([&](auto fc) { issue(fc); });
                static_for_<0, NF>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    constexpr DkvFrag fr = St::at(f);
                    if constexpr (f + PRE < NF) issue(std::integral_constant<int, f + PRE>{});
                    wait_frag(fc);
                    if constexpr (fr.kind == 2 || fr.kind == 3) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[f % RING]);
                        if constexpr (fr.first) {  // first MFMA of the chain: src C = the statistic vector read one step earlier
                            const f32x4 seed = __builtin_bit_cast(f32x4, ring[(f - 1) % RING]);
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
                                if constexpr (fr.kind == 2)
                                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kfB[kt][fr.x], seed, 0, 0, 0);
                                else
                                    dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, vfB[kt][fr.x], seed, 0, 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
                                if constexpr (fr.kind == 2)
                                    sc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, kfB[kt][fr.x], sc[kt], 0, 0, 0);
                                else
                                    dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, vfB[kt][fr.x], dp[kt], 0, 0, 0);
                            }
                        }
                    } else if constexpr (fr.kind >= 4) {
                        const bf16x8 a = __builtin_bit_cast(bf16x8, ring[f % RING]);
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt) {
                            if constexpr (fr.kind == 4)
                                dvacc[fr.x][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[kt], dvacc[fr.x][kt], 0, 0, 0);
                            else
                                dkacc[fr.x][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, dsb[kt], dkacc[fr.x][kt], 0, 0, 0);
                        }
                    }
                    if constexpr (St::last_of_p1(f)) {  // S' (and dP') of 16 rows x 32 keys complete: P, dS -> B operands of phase 2
                        __builtin_amdgcn_sched_barrier(0);
                        if (need_mask) {
#pragma unroll
                            for (int kt = 0; kt < KT; ++kt) {
                                const int kidx = wk0 + kt * 16 + t;
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const int qidx = qb0 + (2 * fr.c + fr.u) * 16 + g * 4 + r;
                                    const bool dead = qidx >= sq_len || kidx >= sk_len || (CAUSAL && kidx > qidx + coff);
                                    sc[kt][r] = dead ? -INFINITY : sc[kt][r];
                                }
                            }
                        }
#pragma unroll
                        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                            for (int r = 0; r < 4; r += 2) {
                                const float p0 = fast_exp2(sc[kt][r] * sl2), p1 = fast_exp2(sc[kt][r + 1] * sl2);
                                if constexpr (kDV) {
                                    pb[kt][fr.u * 4 + r] = (bf16)p0;
                                    pb[kt][fr.u * 4 + r + 1] = (bf16)p1;
                                }
                                if constexpr (kDK) {
                                    dsb[kt][fr.u * 4 + r] = (bf16)(p0 * dp[kt][r]);
                                    dsb[kt][fr.u * 4 + r + 1] = (bf16)(p1 * dp[kt][r + 1]);
                                }
                            }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            rbase ^= BUF;  // the other tile buffer (the dynamic LDS segment starts at address 0)
            cbase ^= BUF;
            sbase ^= 512;
            __syncthreads();
        }

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
                    if constexpr (kDK) st_bf16x4(dkbase + (int64_t)krow * dk_ss + dt * 16 + g * 4, ok_);
                    if constexpr (kDV) st_bf16x4(dvbase + (int64_t)krow * dk_ss + dt * 16 + g * 4, ov_);
                }
            }
        }
    }  // pass
}

template <int D, bool CAUSAL>
int launch_bwd(const AttnParams& P, hipStream_t stream, bool wide_dq, bool wide_dkv, bool pp_dq) {
    constexpr int QT = (D == 128) ? 1 : 2;
    constexpr int KT = (D == 128) ? 1 : 2;
    constexpr int LDS_DQ = 2 * 2 * 64 * D * 2;
    constexpr int LDS_DKV = 2 * (2 * 64 * D * 2 + 512);
    static std::atomic<uint64_t> lds_ok{0}, lds2_ok{0}, lds3_ok{0}, lds4_ok{0}, lds5_ok{0};
    const int64_t rows = (int64_t)P.B * P.Sq * P.H;
    const int64_t nthreads = rows * (D / 8);
    if (!wide_dq)  // the 8-wave dQ kernel computes delta and publishes the statistic planes itself
        hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, stream, P);
    constexpr int BQ = 4 * QT * 16, BKEYS = 4 * KT * 16;
    if (wide_dq && pp_dq && D == 128) {  // ping-pong dQ kernel (attn_bwd_pp.hip): computes delta and the statistic planes as well
        const int rc = dllm_launch_attn_bwd_dq_pp(P, CAUSAL ? 1 : 0, nullptr, stream);
        if (rc != DLLM_OK) return rc;
    } else if (wide_dq) {
        dllm_ensure_dyn_lds(&attn_bwd_dq8_kernel<D, CAUSAL>, LDS_DQ, lds3_ok);
        const int nqb8 = (P.Sq + 255) / 256;  // causal: one group per pair of row blocks
                hipLaunchKernelGGL((attn_bwd_dq8_kernel<D, CAUSAL>), dim3(attn_grid(CAUSAL ? (nqb8 + 1) / 2 : nqb8, P.H, P.B)), dim3(512), LDS_DQ, stream, P);
    } else {
        dllm_ensure_dyn_lds(&attn_bwd_dq_kernel<D, CAUSAL, QT>, LDS_DQ, lds_ok);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<D, CAUSAL, QT>), dim3(attn_grid((P.Sq + BQ - 1) / BQ, P.H, P.B)), dim3(256), LDS_DQ, stream, P);
    }
    if (wide_dkv) {
        constexpr int LDS_DKV8 = 2 * (2 * 64 * D * 2) + 1024;
        static_assert(((2 * 64 * D * 2) & (2 * 64 * D * 2 - 1)) == 0, "buffer toggling by XOR");
        const int nkb8 = (P.Sk + 255) / 256;
        const dim3 grid8(attn_grid(CAUSAL ? (nkb8 + 1) / 2 : nkb8, P.Hkv, P.B));
        if constexpr (D == 128) {
            dllm_ensure_dyn_lds(&attn_bwd_dkv8_kernel<D, CAUSAL, 1>, LDS_DKV8, lds4_ok);
            dllm_ensure_dyn_lds(&attn_bwd_dkv8_kernel<D, CAUSAL, 2>, LDS_DKV8, lds5_ok);
            hipLaunchKernelGGL((attn_bwd_dkv8_kernel<D, CAUSAL, 2>), grid8, dim3(512), LDS_DKV8, stream, P);
            hipLaunchKernelGGL((attn_bwd_dkv8_kernel<D, CAUSAL, 1>), grid8, dim3(512), LDS_DKV8, stream, P);
        } else {
            dllm_ensure_dyn_lds(&attn_bwd_dkv8_kernel<D, CAUSAL, 0>, LDS_DKV8, lds4_ok);
            hipLaunchKernelGGL((attn_bwd_dkv8_kernel<D, CAUSAL, 0>), grid8, dim3(512), LDS_DKV8, stream, P);
        }
    } else {
        dllm_ensure_dyn_lds(&attn_bwd_dkv_kernel<D, CAUSAL, KT>, LDS_DKV, lds2_ok);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<D, CAUSAL, KT>), dim3(attn_grid((P.Sk + BKEYS - 1) / BKEYS, P.Hkv, P.B)), dim3(256), LDS_DKV,
                           stream, P);
    }
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
    // kernel choice as in dllm_attn_fwd: bits 1-2 of `causal` force the 4-wave (1) or the 8-wave pipelined (2) kernels; higher bits
    // are not part of the ABI and are rejected
    if (causal & ~7) return DLLM_ERR_SHAPE;
    const int force = (causal >> 1) & 3;
    causal &= 1;
    P.causal = causal;
    // automatic: the 256-row kernels where the row axis they tile is long enough to fill their blocks (UNet cross-attention has
    // Sq = 4096 queries over Sk = 64 dream tokens: wide dQ, 4-wave dK/dV)
    const bool wq = force >= 2 || (force == 0 && Sq >= 512), wk = force >= 2 || (force == 0 && Sk >= 512);  // force is 0..3
    // 3 / automatic: the ping-pong dQ kernel (D = 128; 32-bit key-axis offsets and 16-byte dQ stores are its preconditions)
    const bool pp = force != 2 && (int64_t)Sk * k_ss < (1ll << 29) && ((dq_ss | dq_sh | dq_sb) & 7) == 0 && ((uintptr_t)dq & 15) == 0;
    // (The dK / dV passes in the same form measured 0.05 / 0.07 ms SLOWER than the 8-wave kernels, profiles/r05_attn_bwd_pp_history.md;
    // that kernel file is kept as profiles/patches/r05_attn_bwd_dkv_pp.hip, outside the library.)
    if (D == 128) return causal ? launch_bwd<128, true>(P, s, wq, wk, pp) : launch_bwd<128, false>(P, s, wq, wk, pp);
    return causal ? launch_bwd<64, true>(P, s, wq, wk, false) : launch_bwd<64, false>(P, s, wq, wk, false);
}

#ifdef DLLM_BENCH_MODES
// benchmark-only: the ping-pong dQ kernel alone (causal, head_dim 128, [B,S,H,D] contiguous operands) with s_memtime stamps of one
// work-group written to `stamps` (2 x 512 uint64: wave 0, wave 4; two stamps around every barrier of the first pass); stamps == null
// runs the plain kernel (for timing it alone)
int dllm_attn_bwd_dq_pp_timeline(const void* dout, const void* q, const void* k, const void* v, const void* o, const float* lse,
                                 float* delta, void* dq, void* stamps, int B, int H, int Sq, void* stream) {
    AttnParams P{};
    const int D = 128;
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.o = (bf16*)o; P.dout = (const bf16*)dout; P.dq = (bf16*)dq;
    P.lse = (float*)lse; P.delta = delta;
    P.B = B; P.H = H; P.Hkv = H; P.Sq = Sq; P.Sk = Sq;
    P.q_sb = P.k_sb = P.o_sb = P.dq_sb = (int64_t)Sq * H * D; P.q_ss = P.k_ss = P.o_ss = P.dq_ss = (int64_t)H * D;
    P.q_sh = P.k_sh = P.o_sh = P.dq_sh = D;
    P.scale = 0.08838834764f; P.causal = 1;
    return dllm_launch_attn_bwd_dq_pp(P, 1, stamps, (hipStream_t)stream);
}
#endif

}  // extern "C"
