// Flash-attention backward, dQ kernel in "ping-pong" form for gfx950 (round 5): the long-axis dQ kernel of dllm_attn_bwd at D = 128.
//
// Replaces the same reference code as attn_bwd.hip (the backward of flash_attn_func / flash_attn_varlen_func as called from
// modeling_dreamllm.py:532-549) with the same layouts, masking rules, statistic planes and results contract as attn_bwd_dq8_kernel.
//
// Structure (the forward's, csrc/attn_fwd_pp.hip, carried over to the three GEMMs of dQ):
//   * 8 waves x 32 query rows; waves w and w + 4 share a SIMD and group B (waves 4-7) runs ONE barrier interval behind group A;
//   * MFMA 32x32x16: a lane owns ONE query (column lane & 31) and 16 of the 32 keys of a key block, so P = exp2(S - lse) and
//     dS = P (dP - delta) are lane-local, and dS^T feeds the dQ MFMAs as B operand from the registers it was computed in;
//   * a 64-key tile is two half tiles (32 keys), each two intervals:
//         C12 : S^T = K Q^T and dP^T - delta = V dO^T, 16 MFMAs in two alternating chains, out of row fragments in registers
//         E+C3: the softmax algebra of the half tile (VALU only), then dQ^T += K^T dS^T, 8 MFMAs out of transposed fragments
//     so while one wave of a SIMD runs the 16-MFMA cluster its partner runs VALU + the 8-MFMA cluster;
//   * fragments time-share 64 registers in two banks: [rows d 0-63] -> [rows d 64-127] -> [K^T] -> [next rows d 0-63], every LDS
//     read is issued one cluster ahead of its use beside the MFMAs of the cluster before;
//   * K / V tiles arrive by LDS-DMA in a PF-deep ring of (K, V) slots; ONE image per tile serves the row reads (ds_read_b128) and
//     the transposed reads (ds_read_b64_tr_b16) conflict-free: chunk c of row r sits at c ^ (((r & 3) << 2) | ((r >> 2) & 3)).
#include "attn_common.h"


namespace {

__device__ __forceinline__ uint32_t bp_cvt_pk(float lo, float hi) {
    bf16x2 w;
    w[0] = (bf16)lo;
    w[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, w);
}
__device__ __forceinline__ void bp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int bp_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
#define BP_LAUNDER(x) asm volatile("" : "+v"(x))

constexpr unsigned kBpTlBlock = 1024;

// TL (bench library only): lane 0 of waves 0 and 4 of work-group kBpTlBlock stamps s_memtime at both sides of every barrier of the
// first pass into LDS behind the rings; the stamps go to tl_out ([2][512] words: wave 0, wave 4; then the pass phases at word 1024).
// (The round-5 build-time experiments -- no laundering, staggered first round, the three wrong-result ablations; results in
// profiles/r05_attn_bwd_pp_history.md -- live in profiles/patches/r05_attn_bwd_pp_experiments.patch, not in this file.)
template <int D, bool CAUSAL, int PF, bool TL = false>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_pp_kernel(AttnParams P, uint64_t* tl_out) {
    static_assert(D == 128, "the d = 64 shapes stay on attn_bwd_dq8_kernel");
    constexpr int NW = 8, QW = 32, BQ = NW * QW, BKV = 64;
    constexpr int DSN = D / 16;  // d steps of S^T = K Q^T and dP^T = V dO^T (8)
    constexpr int DBN = D / 32;  // 32-wide d blocks of dQ^T (4)
    constexpr int PITCH = D * 2, TILE = BKV * PITCH, SLOT = 2 * TILE;  // a ring slot = K tile, then V tile
    constexpr int CPR = D / 8, RPG = 64 / CPR, NDMA = (BKV / RPG) / NW;  // 1-KiB DMA groups per wave, tile and operand (2)
    static_assert(NDMA == 2, "four requests per wave and tile, one per interval");
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpB = wave >= 4;
    const int lq = lane & 31, hi = lane >> 5;
    [[maybe_unused]] uint32_t tl_addr = 0;
    [[maybe_unused]] bool tl_on = false;
    if constexpr (TL) {
        tl_on = blockIdx.x == kBpTlBlock && (wave == 0 || wave == 4);
        tl_addr = lds_addr32(smem) + PF * SLOT + (wave == 4 ? 4096 : 0);
    }
#define BP_STAMP()                                                                                         \
    do {                                                                                                   \
        if constexpr (TL) {                                                                                \
            if (tl_on) {                                                                                   \
                const uint64_t tt_ = __builtin_amdgcn_s_memtime();                                         \
                if (lane == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(tl_addr), "v"(tt_) : "memory");    \
                tl_addr += 8;                                                                              \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

#define BP_PHASE(idx)                                                                                         \
    do {                                                                                                      \
        if constexpr (TL) {                                                                                   \
            if (blockIdx.x == kBpTlBlock && (wave == 0 || wave == 4) && lane == 0)                            \
                tl_out[1024 + (wave == 4 ? 64 : 0) + (idx)] = __builtin_amdgcn_s_memtime();                   \
        }                                                                                                     \
    } while (0)
    BP_PHASE(0);

    const int nqb = (P.Sq + BQ - 1) / BQ;
    const int nitems = CAUSAL ? (nqb + 1) / 2 : nqb;
    const AttnBlock bm = attn_block_map<false>(nitems, P.H, P.B);
    if (!bm.valid) return;
    const int b = bm.b, h = bm.h;
    const int hk = h / (P.H / P.Hkv);
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SqE = sp.SqE;
    const int coff = sk_len - sq_len;
    const float sl2 = P.scale * kLog2e;
    const bf16* kbase = P.k + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;
    const bf16* vbase = P.v + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)sp.kst * P.k_ss;

    // ---- LDS-DMA: lane -> (row of its 1-KiB group, 16-byte position); the position holds source chunk (position ^ swizzle(row))
    const int drow = lane / CPR, dpos = lane % CPR;
    uint32_t goff[NDMA];  // byte offset of this lane's source chunk relative to the tile's first row (K and V alike)
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int r = (wave * NDMA + i) * RPG + drow;
        goff[i] = (uint32_t)(r * (int)P.k_ss + (dpos ^ bp_swz(r)) * 8) * 2u;
    }
    const int kss2 = (int)P.k_ss * 2;
    // one 1-KiB group of key tile `t` (clamped to the last tile of the pass: the request stream stays uniform) into slot `slot`
    auto dma_src = [&](const bf16* base, int t, int nblk_, int i) -> const char* {
        const int row0 = min(t, nblk_ - 1) * BKV;
        const char* tb = reinterpret_cast<const char*>(base) + (uint32_t)(row0 * kss2);
        uint32_t o = goff[i];
        if (row0 + BKV > sk_len) {  // ragged last tile: rows past the end re-read the last valid row (finite data; masked later)
            asm volatile("" ::: "memory");
            const int r = (wave * NDMA + i) * RPG + drow;
            if (row0 + r > sk_len - 1) o = o - (uint32_t)(r * kss2) + (uint32_t)((sk_len - 1 - row0) * kss2);
        }
        return tb + o;
    };
    auto dma_dst = [&](int opnd, int slot, int i) -> char* { return smem + slot * SLOT + opnd * TILE + (wave * NDMA + i) * 1024; };
    auto dma_one = [&](const bf16* base, int opnd, int t, int nblk_, int slot, int i) { GLDS16_(dma_src(base, t, nblk_, i), dma_dst(opnd, slot, i)); };

    // ---- fragment addresses (slot 0, K tile; the V tile is TILE bytes further).
    // Row fragment (kb, ds): row 32 kb + lq, chunk (2 ds + hi) ^ swizzle(row): one XOR with ds << 5.
    const uint32_t ra0 = lds_addr32(smem) + (uint32_t)(lq * PITCH + ((hi ^ bp_swz(lq)) << 4));
    // Transposed K fragment (ks, db) = two transpose reads of [4 keys][16 d] blocks per 16-lane group (attn_fwd_pp.hip, V fragments):
    // lane (gb = (lane >> 4) & 1, t = lane & 15) addresses row 16 ks + 4 hi + (t >> 2) (+ 8), bytes 64 db + 32 gb + 8 (t & 3); with the
    // unified swizzle the row's 64-byte granule db sits at granule db ^ (t >> 2) and the chunk inside it at (2 gb + (t & 3) / 2) ^
    // ((row >> 2) & 3), where (row >> 2) & 3 = hi for the first read and hi + 2 for the second (XOR 32 on the byte address).
    uint32_t kt0;
    {
        const int t = lane & 15, gb = (lane >> 4) & 1;
        const int chunk = 4 * (t >> 2) + ((2 * gb + ((t >> 1) & 1)) ^ hi);
        kt0 = lds_addr32(smem) + (uint32_t)((4 * hi + (t >> 2)) * PITCH + chunk * 16 + 8 * (t & 1));
    }

    const bf16* qhead = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
    const bf16* dohead = P.dout + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh + (int64_t)sp.qst * P.o_ss;
    const bf16* ohead = P.o + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh + (int64_t)sp.qst * P.o_ss;
    const int64_t stat0 = ((int64_t)b * P.H + h) * P.Sq + sp.qst, plane = (int64_t)P.B * P.H * P.Sq;

    // Requests of a pass: its first two key tiles, then this wave's operand rows -- Q and dO as B operands (lane = query lq, d = 16 ds +
    // 8 hi ..), O for delta = rowsum(dO * O), the row's lse.  (Measured and not kept, profiles/r05_attn_bwd_pp_history.md: requesting the
    // second pass of a causal pair in front of the first pass's store tail -- 96 more live registers there, +5 % kernel time; fencing
    // the loads into one batch -- no difference.)
    bf16x8 qf[DSN], dof[DSN], of[DSN];
    float lse_v = 0.f;
    auto request_pass = [&](int qblk_, int nblk_) {
        const int qr = min(qblk_ * BQ + wave * QW + lq, sq_len - 1);
        int64_t qo = (int64_t)qr * P.q_ss + hi * 8, oo = (int64_t)qr * P.o_ss + hi * 8, lo = stat0 + qr;
        const char* src[2][2][NDMA];  // [tile][operand][piece]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NDMA; ++u) {
                src[t][0][u] = dma_src(kbase, t, max(nblk_, 1), u);
                src[t][1][u] = dma_src(vbase, t, max(nblk_, 1), u);
            }
        const bf16 *qp = qhead + qo, *dop = dohead + oo, *op = ohead + oo;
        if (nblk_ > 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int o2 = 0; o2 < 2; ++o2)
#pragma unroll
                    for (int u = 0; u < NDMA; ++u) GLDS16_(src[t][o2][u], dma_dst(o2, t, u));
        }
        lse_v = P.lse[lo];
#pragma unroll
        for (int ds = 0; ds < DSN; ++ds) {
            of[ds] = ld_bf16x8(op + ds * 16);
            dof[ds] = ld_bf16x8(dop + ds * 16);
        }
#pragma unroll
        for (int ds = 0; ds < DSN; ++ds) qf[ds] = ld_bf16x8(qp + ds * 16);
    };
    const int npass = (CAUSAL && nqb - 1 - bm.r != bm.r) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int qblk = CAUSAL ? (pass == 0 ? nqb - 1 - bm.r : bm.r) : bm.r;
        const int q0 = qblk * BQ, wq0 = q0 + wave * QW;
        bf16* dqbase = P.dq + (int64_t)b * P.dq_sb + (int64_t)h * P.dq_sh;
        if (sp.qst > 0) {
            if (qblk == 0) zero_head_rows<D, 512>(dqbase, P.dq_ss, sp.qst, tid);
            dqbase += (int64_t)sp.qst * P.dq_ss;
        }
        if (q0 >= sq_len) {
            for (int i = tid; i < BQ * (D / 8); i += 512) {
                const int r = q0 + i / (D / 8), c = i % (D / 8);
                if (r < SqE) st_bf16x8(dqbase + (int64_t)r * P.dq_ss + c * 8, zero_bf16x8());
            }
            continue;
        }
        int kv_end = sk_len;
        if (CAUSAL) kv_end = min(sk_len, q0 + BQ + coff);
        const int nblk = kv_end > 0 ? (kv_end + BKV - 1) / BKV : 0;

        BP_PHASE(1 + 8 * pass);
        // ---- operands and statistics: this is the first backward kernel, it computes delta and (with the store tail) publishes the
        // statistic planes delta, -delta and -lse / scale for the dK / dV kernels, exactly as attn_bwd_dq8_kernel does
        float nlse2, ndlt;
        {
            request_pass(qblk, nblk);
            BP_PHASE(2 + 8 * pass);
            const float lse = lse_v;
            float dsum = 0.f;
#pragma unroll
            for (int ds = 0; ds < DSN; ++ds)
#pragma unroll
                for (int e = 0; e < 8; ++e) dsum += (float)of[ds][e] * (float)dof[ds][e];
            dsum += __shfl_xor(dsum, 32, 64);
            nlse2 = -lse * kLog2e;
            ndlt = -dsum;
#pragma unroll
            for (int ds = 0; ds < DSN; ++ds) {
                pin_loaded(qf[ds]);
                pin_loaded(dof[ds]);
            }
            pin_loaded(nlse2);
            pin_loaded(ndlt);
        }
        BP_PHASE(3 + 8 * pass);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first tiles landed (this wave's shares); the barrier publishes them

        f32x16 dqacc[DBN];
#pragma unroll
        for (int db = 0; db < DBN; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) dqacc[db][r] = 0.f;
        int nact = 0;
        if (wq0 < sq_len) nact = CAUSAL ? max(0, min(nblk, (wq0 + QW - 1 + coff) / BKV + 1)) : nblk;
        if (CAUSAL && wq0 + QW - 1 + coff < 0) nact = 0;

        bp_barrier();
        BP_PHASE(4 + 8 * pass);
        // row fragments d 0-63 of the next half tile (loop carried): rf[2 i] = K fragment of d step i, rf[2 i + 1] = V fragment
        u32x4 rf[8];
        auto load_rows = [ra0](u32x4 (&dst)[8], auto kbc, auto d0c, uint32_t slotoff) {
            constexpr int kb = decltype(kbc)::value, d0 = decltype(d0c)::value;
            const uint32_t ra = ra0 + slotoff;
            static_for_<0, 4>([&dst, ra](auto ic) {
                constexpr int i = decltype(ic)::value;
                const uint32_t a = ra ^ (uint32_t)((d0 + i) << 5);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2 * i]) : "v"(a), "n"(kb * 32 * PITCH));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[2 * i + 1]) : "v"(a), "n"(TILE + kb * 32 * PITCH));
            });
        };
        if (nact > 0) {
            load_rows(rf, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for_<0, 8>([&rf](auto fc) { asm volatile("" : "+v"(rf[decltype(fc)::value])); });
        }
        if (grpB) bp_barrier();  // group B starts one interval late ...
        BP_PHASE(5 + 8 * pass);

        int slot = 0;
        int j = 0;
        for (; j < nact; ++j) {
            const int kv0 = j * BKV;
            const int nslot = slot + 1 == PF ? 0 : slot + 1;
            const int pslot = slot == 0 ? PF - 1 : slot - 1;  // slot of tile j + 2 (PF = 3) = slot of tile j - 1
            const uint32_t so = (uint32_t)(slot * SLOT), son = (uint32_t)(nslot * SLOT);
            const bool more = j + 2 < nblk;  // the last two tiles of a pass request nothing (uniform branch)
            const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
            const int lim = (CAUSAL ? min(wq0 + lq + coff, sk_len - 1) : sk_len - 1) - kv0 - 4 * hi;

            // one half tile: r1 = its row fragments of d 0-63 (in registers), rn = those of the next half tile (requested here)
            auto half = [&](auto kbc, u32x4 (&r1)[8], u32x4 (&rn)[8]) {
                constexpr int kb = decltype(kbc)::value;
                f32x16 s, dp;
                u32x4 r2[8];
                u32x2 tlo[8], thi[8];
                // ---------------------------------------------------------------- C12: 16 MFMAs, two chains
                static_for_<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const bf16x8 ka = __builtin_bit_cast(bf16x8, r1[2 * i]), va = __builtin_bit_cast(bf16x8, r1[2 * i + 1]);
                    if constexpr (i == 0) {
                        f32x16 z, nd;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            z[r] = 0.f;
                            nd[r] = ndlt;
                        }
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[i], z, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[i], nd, 0, 0, 0);
                    } else {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[i], s, 0, 0, 0);
                        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[i], dp, 0, 0, 0);
                    }
                    if constexpr (i < 2) {  // row fragments of d 64-127, four per step
                        uint32_t ra = ra0 + so;
                        BP_LAUNDER(ra);  // per-use address arithmetic: hipcc otherwise keeps all twenty XOR variants of a tile live
                        static_for_<2 * i, 2 * i + 2>([&r2, ra](auto uc) {
                            constexpr int u = decltype(uc)::value;
                            const uint32_t a = ra ^ (uint32_t)((4 + u) << 5);
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2[2 * u]) : "v"(a), "n"(kb * 32 * PITCH));
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2[2 * u + 1]) : "v"(a), "n"(TILE + kb * 32 * PITCH));
                        });
                    }
                    // the four requests of tile j + 2 are spread over the four intervals of tile j (K piece 0 | K piece 1 | V piece 0 |
                    // V piece 1): the CU's address path takes them one at a time, four at once per wave stall the issuing waves
                    if constexpr (i == 0) {
                        if (more) dma_one(kb == 0 ? kbase : vbase, kb, j + 2, nblk, pslot, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&r2](auto fc) { asm volatile("" : "+v"(r2[decltype(fc)::value])); });
                static_for_<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const bf16x8 ka = __builtin_bit_cast(bf16x8, r2[2 * i]), va = __builtin_bit_cast(bf16x8, r2[2 * i + 1]);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[4 + i], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[4 + i], dp, 0, 0, 0);
                    {  // transposed K fragments f = 2 i, 2 i + 1 (f = 4 jj + db: 16-key step 2 kb + jj, d block db)
                        uint32_t kt = kt0 + so;
                        BP_LAUNDER(kt);
                        static_for_<2 * i, 2 * i + 2>([&tlo, &thi, kt](auto fc) {
                            constexpr int f = decltype(fc)::value;
                            constexpr int ks = 2 * kb + (f >> 2), db = f & 3;
                            const uint32_t a = kt ^ (uint32_t)(db << 6);
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tlo[f]) : "v"(a), "n"(ks * 16 * PITCH));
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(thi[f]) : "v"(a ^ 32u), "n"(ks * 16 * PITCH + 8 * PITCH));
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                BP_STAMP();
                bp_barrier();
                BP_STAMP();
                if (more) dma_one(kb == 0 ? kbase : vbase, kb, j + 2, nblk, pslot, 1);
                // ---------------------------------------------------------------- E: P = exp2(S scale log2e - lse log2e), dS = P (dP - delta)
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 32 * kb;
                        s[r] = (c > lim) ? -INFINITY : s[r];
                    }
                }
                // Exposed part: the exponents of all 16 scores (packed FMAs) and dS of the first 16-key step; the second step's exp2 / multiply /
                // pack run in the shadow of the first step's four MFMAs, each one MFMA group after the results it depends on were issued
                // (an in-order wave stalls on a dependent VALU: attn_fwd_pp.hip).  Results are pinned where they are produced.
                uint32_t dsb[2][4];
                {
                    const f32x2 sl22 = f32x2{sl2, sl2}, nl2 = f32x2{nlse2, nlse2};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        f32x2 a = __builtin_elementwise_fma(f32x2{s[r], s[r + 1]}, sl22, nl2);
                        s[r] = a[0];
                        s[r + 1] = a[1];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(s[r]));
                auto exp4 = [&](int r0) {
#pragma unroll
                    for (int r = r0; r < r0 + 4; ++r) {
                        float x = fast_exp2(s[r]);
                        asm volatile("" : "+v"(x));
                        s[r] = x;
                    }
                };
                auto mulpack4 = [&](int r0) {  // dS of scores r0 .. r0 + 3 -> two packed words of dsb
#pragma unroll
                    for (int r = r0; r < r0 + 4; r += 2) {
                        const f32x2 d = f32x2{s[r], s[r + 1]} * f32x2{dp[r], dp[r + 1]};
                        uint32_t u = bp_cvt_pk(d[0], d[1]);
                        asm volatile("" : "+v"(u));
                        dsb[r >> 3][(r & 7) >> 1] = u;
                    }
                };
                exp4(0);
                exp4(4);
                mulpack4(0);
                mulpack4(4);
                exp4(8);
                exp4(12);
                mulpack4(8);
                mulpack4(12);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&tlo, &thi](auto fc) { asm volatile("" : "+v"(tlo[decltype(fc)::value]), "+v"(thi[decltype(fc)::value])); });
                // ---------------------------------------------------------------- C3: dQ^T += K^T dS^T, 8 MFMAs
                static_for_<0, 8>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    constexpr int jj = f >> 2, db = f & 3;
                    const bf16x8 a = join2(tlo[f], thi[f]);
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, u32x4{dsb[jj][0], dsb[jj][1], dsb[jj][2], dsb[jj][3]});
                    dqacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, dqacc[db], 0, 0, 0);
                    if constexpr (f < 4) {  // row fragments d 0-63 of the next half tile, two per step
                        uint32_t ra = ra0 + (kb == 0 ? so : son);
                        BP_LAUNDER(ra);
                        const uint32_t a2 = ra ^ (uint32_t)(f << 5);
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rn[2 * f]) : "v"(a2), "n"((1 - kb) * 32 * PITCH));
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rn[2 * f + 1]) : "v"(a2), "n"(TILE + (1 - kb) * 32 * PITCH));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (kb == 0) {  // this wave's share of tile j + 1 landed (younger: two requests of tile j + 2, if there is one)
                    if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&rn](auto fc) { asm volatile("" : "+v"(rn[decltype(fc)::value])); });
                BP_STAMP();
                bp_barrier();
                BP_STAMP();
            };
            u32x4 rm[8];
            half(std::integral_constant<int, 0>{}, rf, rm);
            half(std::integral_constant<int, 1>{}, rm, rf);
            slot = nslot;
        }
        for (; j < nblk; ++j) {  // idle part (causal: tiles above this wave's rows): keep the DMA shares and the barriers going
            const int pslot = slot == 0 ? PF - 1 : slot - 1;
            const bool more = j + 2 < nblk;
            if (more) dma_one(kbase, 0, j + 2, nblk, pslot, 0);
            bp_barrier();
            if (more) {
                dma_one(kbase, 0, j + 2, nblk, pslot, 1);
                asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            bp_barrier();
            if (more) dma_one(vbase, 1, j + 2, nblk, pslot, 0);
            bp_barrier();
            if (more) dma_one(vbase, 1, j + 2, nblk, pslot, 1);
            bp_barrier();
            slot = slot + 1 == PF ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BP_PHASE(6 + 8 * pass);
        if (!grpB) bp_barrier();  // ... and group A waits for it at the end: every wave has passed its last LDS read
        BP_PHASE(7 + 8 * pass);
        if constexpr (TL) {
            BP_STAMP();
            if (tl_on) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t base = lds_addr32(smem) + PF * SLOT + (wave == 4 ? 4096 : 0);
                uint64_t* dst = tl_out + (wave == 4 ? 512 : 0);
                for (int i = lane; i < 512; i += 64) {
                    u32x2 w;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(base + 8 * i) : "memory");
                    dst[i] = (i < (int)((tl_addr - base) >> 3)) ? (((uint64_t)w[1] << 32) | w[0]) : 0ull;
                }
                tl_on = false;
            }
        }

        {  // statistic planes of this wave's rows (delta, -delta, -lse / scale) for the dK / dV kernels
            const int qrow = wq0 + lq;
            if (qrow < sq_len && hi == 0) {
                P.delta[stat0 + qrow] = -ndlt;
                P.delta[plane + stat0 + qrow] = ndlt;
                P.delta[2 * plane + stat0 + qrow] = -P.lse[stat0 + qrow] / P.scale;  // (from the register copy instead: same kernel time)
            }
        }
        // ---- store: lane (q = lq, hi) holds dQ^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][q]; as the forward's store tail
        {
            const int qrow = wq0 + lq;
            const float f = qrow < sq_len ? P.scale : 0.f;
            bf16* orow = dqbase + (int64_t)qrow * P.dq_ss + hi * 8;
#pragma unroll
            for (int db = 0; db < DBN; ++db)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    uint32_t a0 = bp_cvt_pk(dqacc[db][8 * m + 0] * f, dqacc[db][8 * m + 1] * f);
                    uint32_t a1 = bp_cvt_pk(dqacc[db][8 * m + 2] * f, dqacc[db][8 * m + 3] * f);
                    uint32_t b0 = bp_cvt_pk(dqacc[db][8 * m + 4] * f, dqacc[db][8 * m + 5] * f);
                    uint32_t b1 = bp_cvt_pk(dqacc[db][8 * m + 6] * f, dqacc[db][8 * m + 7] * f);
                    const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    if (qrow < SqE) *reinterpret_cast<u32x4*>(orow + db * 32 + m * 16) = u32x4{x0[0], x1[0], x0[1], x1[1]};
                }
        }
        BP_PHASE(8 + 8 * pass);
    }  // pass
#undef BP_STAMP
#undef BP_PHASE
}

template <bool CAUSAL, int PF, bool TL = false>
int launch_dq_pp(const AttnParams& P, uint64_t* tl_out, hipStream_t stream) {
    constexpr int D = 128;
    constexpr int LDS = PF * 2 * 64 * D * 2 + (TL ? 8192 : 0);
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_bwd_dq_pp_kernel<D, CAUSAL, PF, TL>, LDS, lds_ok);
    const int nqb = (P.Sq + 255) / 256;
    const dim3 grid(attn_grid(CAUSAL ? (nqb + 1) / 2 : nqb, P.H, P.B));
    hipLaunchKernelGGL((attn_bwd_dq_pp_kernel<D, CAUSAL, PF, TL>), grid, dim3(512), LDS, stream, P, tl_out);
    return dllm_check_launch();
}

}  // namespace

// Called by dllm_attn_bwd (attn_bwd.hip) for D = 128 on long query axes; the caller has checked shapes and alignment.
__attribute__((visibility("hidden"))) int dllm_launch_attn_bwd_dq_pp(const AttnParams& P, int causal, void* tl_out, hipStream_t stream) {
#ifdef DLLM_BENCH_MODES
    if (tl_out != nullptr && causal) return launch_dq_pp<true, 3, true>(P, (uint64_t*)tl_out, stream);
#endif
    (void)tl_out;
    return causal ? launch_dq_pp<true, 3>(P, nullptr, stream) : launch_dq_pp<false, 3>(P, nullptr, stream);
}
