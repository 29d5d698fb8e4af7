// Flash-attention backward, dK and dV passes in "ping-pong" form for gfx950 (round 5): replace attn_bwd_dkv8_kernel<128, *, 2 / 1> on
// long key axes without GQA.  Same reference code, layouts, masking rules and results contract as attn_bwd.hip (the backward of
// flash_attn_func / flash_attn_varlen_func as called from modeling_dreamllm.py:532-549); they read the statistic planes (-delta,
// -lse/scale) the dQ kernel of the same call publishes.
//
// The skeleton is the dQ kernel's (csrc/attn_bwd_pp.hip) with the roles swapped:
//   * 8 waves x 32 KEYS; a wave's K rows (and, in the dK pass, V rows) are its resident B operands (lane = key lane & 31,
//     d = 16 ds + 8 hi ..); Q and dO tiles of 64 query rows stream through a 3-deep LDS-DMA ring of (Q, dO) slots, both in the unified
//     image that serves ds_read_b128 rows and ds_read_b64_tr_b16 transposes conflict-free, the tile's statistics beside them;
//   * MFMA 32x32x16: S[q x key] and dP[q x key] come out with a lane owning ONE key and 16 of a block's 32 queries; P (dV pass) or
//     dS = P dP' (dK pass) is packed in place as the B operand of dV^T[d x key] += dO^T P / dK^T[d x key] += Q^T dS;
//   * the per-query statistics enter through the matrix pipe: -lse/scale (and -delta) of query q, split into three bf16 pieces (exact
//     to 24 bits), are one extra k step of the S (dP) chain against a B operand of ones -- one ds_read_b32 and one register per
//     32 queries and plane, where an accumulator seed would be 16 registers per chain;
//   * a 64-query tile is two half tiles (32 queries), each two barrier intervals: [S (and dP): 9 (18) MFMAs] [exp2 / multiply / pack,
//     then dV (dK): 8 MFMAs]; waves 4-7 run one interval behind waves 0-3; the five (six) DMA requests of tile j + 2 are spread
//     over the four intervals of tile j.
// MODE 1 = dV pass (S, P, dV: K resident), MODE 2 = dK pass (S, dP, dS, dK: K and V resident), as in attn_bwd_dkv8_kernel.
//
// MEASURED AND NOT SHIPPED (profiles/r05_attn_bwd_pp_history.md): both passes agree with the 8-wave kernels to 4e-5 rel-L2 on the first
// run and compile without a spill in the loop, but at B16 S2048 H32 D128 causal the dK pass takes 1.16 ms against 1.10 and the dV pass
// 0.84 against 0.76: the statistics' extra k step (+8 % MFMAs), four barriers per tile around 9-MFMA intervals in the dV pass, and an
// 8-wave kernel that streams its fragments with less per-tile overhead than the dQ one did.  The file is compiled into
// DLLM_BENCH_MODES libraries only (tools/attn_bwd_ab.py switches the passes per call); the shipped library does not contain it.
#include "attn_common.h"

#ifdef DLLM_BENCH_MODES

namespace {

__device__ __forceinline__ uint32_t kp_cvt_pk(float lo, float hi) {
    bf16x2 w;
    w[0] = (bf16)lo;
    w[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, w);
}
__device__ __forceinline__ void kp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int kp_swz(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }
#define KP_LAUNDER(x) asm volatile("" : "+v"(x))  // per-use address arithmetic (see attn_bwd_pp.hip: keeps spills out of the loop)
#define KP_GLDS4(gptr, lptr)                                                                                           \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                           \
                                     (__attribute__((address_space(3))) void*)(lptr), 4, 0, 0)

// the A operand of the statistic's k step: lanes hi = 0 carry v = h1 + h2 + h3 (bf16 pieces by truncation, exact) in k slots 0 .. 2
__device__ __forceinline__ bf16x8 kp_stat_operand(uint32_t vbits, int hi) {
    const uint32_t h1 = vbits & 0xffff0000u;
    const float r1 = __uint_as_float(vbits) - __uint_as_float(h1);
    const uint32_t h2 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(h2);
    const uint32_t h3 = __float_as_uint(r2) & 0xffff0000u;
    const uint32_t w0 = hi ? 0u : ((h1 >> 16) | h2), w1 = hi ? 0u : (h3 >> 16);
    return __builtin_bit_cast(bf16x8, u32x4{w0, w1, 0u, 0u});
}

template <bool CAUSAL, int PF, int MODE>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_pp_kernel(AttnParams P) {
    constexpr int D = 128;
    constexpr bool kDK = MODE == 2;
    constexpr int NW = 8, KW = 32, BKEYS = NW * KW, BQ = 64;
    constexpr int DBN = D / 32;
    constexpr int PITCH = D * 2, TILE = BQ * PITCH, SLOT = 2 * TILE;  // a ring slot = Q tile, then dO tile
    constexpr int CPR = D / 8, RPG = 64 / CPR, NDMA = (BQ / RPG) / NW;
    static_assert(NDMA == 2 && PF >= 3, "request schedule and wait counts below");
    // Tile j + PF - 1 is requested during tile j.  At the end of a tile's second interval the wave's share of tile j + 1 must have landed:
    // younger than it are PF - 3 whole tiles (4 requests, 5 for a wave that carries a statistics plane) and the first two intervals'
    // requests of tile j + PF - 1 (2, or 3).
    constexpr int WAIT_PLAIN = (PF - 3) * 4 + 2, WAIT_STAT = (PF - 3) * 5 + 3;
    constexpr int AHEAD = PF - 1;
    constexpr int STAT0 = PF * SLOT;  // statistics ring: PF x {64 x -lse/scale, 64 x -delta}
    constexpr int NSTATW = kDK ? 2 : 1;  // waves 0 (and 1) carry one statistics request per tile on top of the two Q pieces
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpB = wave >= 4;
    const int lq = lane & 31, hi = lane >> 5;

    const int nkb = (P.Sk + BKEYS - 1) / BKEYS;
    const int nitems = CAUSAL ? (nkb + 1) / 2 : nkb;  // causal: key block 0 sees every query, the last one only its own rows: pairs
    const AttnBlock bm = attn_block_map<false>(nitems, P.Hkv, P.B);
    if (!bm.valid) return;
    const int b = bm.b, hk = bm.h;  // no GQA here: query head = key head
    const AttnSpan sp = attn_span(P, b);
    const int sq_len = sp.sq_len, sk_len = sp.sk_len, SkE = sp.SkE;
    const int coff = sk_len - sq_len;
    const float sl2 = P.scale * kLog2e;
    const bf16* qbase = P.q + (int64_t)b * P.q_sb + (int64_t)hk * P.q_sh + (int64_t)sp.qst * P.q_ss;
    const bf16* dobase = P.dout + (int64_t)b * P.o_sb + (int64_t)hk * P.o_sh + (int64_t)sp.qst * P.o_ss;
    const int64_t plane = (int64_t)P.B * P.H * P.Sq;
    const float* statbase = P.delta + ((int64_t)b * P.H + hk) * P.Sq + sp.qst;  // + 2 plane: -lse / scale, + plane: -delta

    // ---- LDS-DMA: lane -> (row of its 1-KiB group, 16-byte position); the position holds source chunk (position ^ swizzle(row))
    const int drow = lane / CPR, dpos = lane % CPR;
    uint32_t qoff[NDMA], ooff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int r = (wave * NDMA + i) * RPG + drow;
        qoff[i] = (uint32_t)(r * (int)P.q_ss + (dpos ^ kp_swz(r)) * 8) * 2u;
        ooff[i] = (uint32_t)(r * (int)P.o_ss + (dpos ^ kp_swz(r)) * 8) * 2u;
    }
    const int qss2 = (int)P.q_ss * 2, oss2 = (int)P.o_ss * 2;
    int qb_begin = 0, nq = 0;  // first query tile and number of query tiles of the pass
    // one 1-KiB group of query tile `t` of the pass into ring slot `slot`: opnd 0 = Q, 1 = dO
    auto dma_one = [&](int opnd, int t, int slot, int i) {
        char* dst = smem + slot * SLOT + opnd * TILE + (wave * NDMA + i) * 1024;
        const int row0 = (qb_begin + min(t, nq - 1)) * BQ;
        const int ss2 = opnd ? oss2 : qss2;
        const char* tb = reinterpret_cast<const char*>(opnd ? dobase : qbase) + (uint32_t)(row0 * ss2);
        uint32_t o = opnd ? ooff[i] : qoff[i];
        if (row0 + BQ > sq_len) {  // ragged last tile: rows past the end re-read the last valid row (finite data; masked later)
            asm volatile("" ::: "memory");
            const int r = (wave * NDMA + i) * RPG + drow;
            if (row0 + r > sq_len - 1) o = o - (uint32_t)(r * ss2) + (uint32_t)((sq_len - 1 - row0) * ss2);
        }
        GLDS16_(tb + o, dst);
    };
    // the tile's 64 statistics of one plane: requested by wave `pl` only (its counted waits allow for one request more)
    auto dma_stat = [&](int pl, int t, int slot) {
        if (wave != pl) return;
        const int row0 = (qb_begin + min(t, nq - 1)) * BQ;
        KP_GLDS4(statbase + (pl ? plane : 2 * plane) + min(row0 + lane, sq_len - 1), smem + STAT0 + slot * 512 + pl * 256);
    };
    // the requests of tile t by interval (0 .. 3): Q piece 0 + lse plane | Q piece 1 (+ delta plane) | dO piece 0 | dO piece 1
    auto dma_part = [&](int part, int t, int slot) {
        if (part == 0) {
            dma_one(0, t, slot, 0);
            dma_stat(0, t, slot);
        } else if (part == 1) {
            dma_one(0, t, slot, 1);
            if constexpr (kDK) dma_stat(1, t, slot);
        } else {
            dma_one(1, t, slot, part - 2);
        }
    };

    // ---- fragment addresses (slot 0, Q tile; the dO tile is TILE bytes further): as in attn_bwd_pp.hip
    const uint32_t ra0 = lds_addr32(smem) + (uint32_t)(lq * PITCH + ((hi ^ kp_swz(lq)) << 4));
    uint32_t kt0;
    {
        const int t = lane & 15, gb = (lane >> 4) & 1;
        const int chunk = 4 * (t >> 2) + ((2 * gb + ((t >> 1) & 1)) ^ hi);
        kt0 = lds_addr32(smem) + (uint32_t)((4 * hi + (t >> 2)) * PITCH + chunk * 16 + 8 * (t & 1)) + (kDK ? 0u : (uint32_t)TILE);
    }
    const uint32_t sa0 = lds_addr32(smem) + STAT0 + (uint32_t)(lq * 4);

    const int npass = (CAUSAL && nkb - 1 - bm.r != bm.r) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int kblk = (CAUSAL && pass == 1) ? nkb - 1 - bm.r : bm.r;
        const int k0 = kblk * BKEYS, wk0 = k0 + wave * KW;
        bf16* obase = (kDK ? P.dk : P.dv) + (int64_t)b * P.dk_sb + (int64_t)hk * P.dk_sh;
        if (sp.kst > 0) {
            if (kblk == 0) zero_head_rows<D, 512>(obase, P.dk_ss, sp.kst, tid);
            obase += (int64_t)sp.kst * P.dk_ss;
        }
        qb_begin = CAUSAL ? max(0, k0 - coff) / BQ : 0;
        const int qb_end = (sq_len + BQ - 1) / BQ;
        nq = (k0 < sk_len && qb_end > qb_begin) ? (qb_end - qb_begin) : 0;

        // ---- prologue: tiles 0 .. PF - 2 of the pass, then this wave's K (and V) rows
        if (nq > 0) {
#pragma unroll
            for (int t = 0; t < PF - 1; ++t)
#pragma unroll
                for (int part = 0; part < 4; ++part) dma_part(part, t, t);
        }
        bf16x8 kf[8], vf[8];
        {
            const int krow = min(wk0 + lq, max(sk_len - 1, 0));  // keys past the end: finite data, masked below
            const bf16* kp = P.k + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)(sp.kst + krow) * P.k_ss + hi * 8;
            const bf16* vp = P.v + (int64_t)b * P.k_sb + (int64_t)hk * P.k_sh + (int64_t)(sp.kst + krow) * P.k_ss + hi * 8;
#pragma unroll
            for (int ds = 0; ds < 8; ++ds) {
                kf[ds] = ld_bf16x8(kp + ds * 16);
                if constexpr (kDK) vf[ds] = ld_bf16x8(vp + ds * 16);
            }
#pragma unroll
            for (int ds = 0; ds < 8; ++ds) {
                pin_loaded(kf[ds]);
                if constexpr (kDK) pin_loaded(vf[ds]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        f32x16 acc[DBN];
#pragma unroll
        for (int db = 0; db < DBN; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[db][r] = 0.f;

        // A wave takes part in tile j iff one of its keys is visible to one of the tile's queries: the tiles of a pass go UP the query
        // axis, so the idle ones come first -- tiles [0, nidle) only keep the DMA / barrier protocol going.
        int nidle = 0;
        if (wk0 >= sk_len) nidle = nq;
        else if (CAUSAL && wk0 - coff > 0) nidle = min(nq, max(0, (wk0 - coff) / BQ - qb_begin));

        kp_barrier();
        if (grpB) kp_barrier();  // group B starts one interval late ...

        int slot = 0;
        int j = 0;
        for (; j < nidle; ++j) {
            const int pslot = slot == 0 ? PF - 1 : slot - 1;
            const bool more = j + AHEAD < nq;
            if (more) dma_part(0, j + AHEAD, pslot);
            kp_barrier();
            if (more) {
                dma_part(1, j + AHEAD, pslot);
                if (wave < NSTATW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_STAT) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PLAIN) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            kp_barrier();
            if (more) dma_part(2, j + AHEAD, pslot);
            kp_barrier();
            if (more) dma_part(3, j + AHEAD, pslot);
            kp_barrier();
            slot = slot + 1 == PF ? 0 : slot + 1;
        }
        // row fragments d 0-63 and statistics of the next half tile (loop carried): rf[2 i] = Q fragment of d step i, rf[2 i + 1] = dO
        u32x4 rf[8];
        uint32_t stl = 0, std_ = 0;
        if (j < nq) {  // first active tile (landed and published at least two barriers ago): one exposed LDS latency per pass
            const uint32_t ra = ra0 + (uint32_t)(slot * SLOT);
            static_for_<0, 4>([&rf, ra](auto ic) {
                constexpr int i = decltype(ic)::value;
                const uint32_t a = ra ^ (uint32_t)(i << 5);
                asm volatile("ds_read_b128 %0, %1" : "=v"(rf[2 * i]) : "v"(a));
                if constexpr (kDK) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rf[2 * i + 1]) : "v"(a), "n"(TILE));
            });
            const uint32_t sa = sa0 + (uint32_t)(slot * 512);
            asm volatile("ds_read_b32 %0, %1" : "=v"(stl) : "v"(sa));
            if constexpr (kDK) asm volatile("ds_read_b32 %0, %1 offset:256" : "=v"(std_) : "v"(sa));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for_<0, 8>([&rf](auto fc) {
                if constexpr (kDK || (decltype(fc)::value & 1) == 0) asm volatile("" : "+v"(rf[decltype(fc)::value]));
            });
            asm volatile("" : "+v"(stl), "+v"(std_));
        }

        for (; j < nq; ++j) {
            const int qb0 = (qb_begin + j) * BQ;
            const int nslot = slot + 1 == PF ? 0 : slot + 1;
            const int pslot = slot == 0 ? PF - 1 : slot - 1;
            const uint32_t so = (uint32_t)(slot * SLOT), son = (uint32_t)(nslot * SLOT);
            const bool more = j + AHEAD < nq;
            const bool need_mask = (qb0 + BQ > sq_len) || (wk0 + KW > sk_len) || (CAUSAL && (wk0 + KW - 1 > qb0 + coff));
            // query qb0 + c + 4 hi (c = the register's compile-time offset) is dead for key kidx iff it lies below kidx - coff (causal) or
            // past the last query; a key past the end kills its whole column
            const int kidx = wk0 + lq;
            const int lo = (kidx >= sk_len) ? (1 << 20) : (CAUSAL ? kidx - coff - qb0 - 4 * hi : -(1 << 20));
            const int up = sq_len - qb0 - 4 * hi;

            // one half tile (32 queries): r1 / st* = its row fragments of d 0-63 and statistics, rn / sn* = those of the next half tile
            auto half = [&](auto kbc, u32x4 (&r1)[8], uint32_t sl_in, uint32_t sd_in, u32x4 (&rn)[8], uint32_t& sl_out, uint32_t& sd_out) {
                constexpr int kb = decltype(kbc)::value;
                f32x16 s, dp;
                u32x4 r2[8];
                u32x2 tlo[8], thi[8];
                // ---------------------------------------------------------------- C12: the statistics' k step, then 8 (16) MFMAs
                {
                    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kp_stat_operand(sl_in, hi), ones, z, 0, 0, 0);
                    if constexpr (kDK) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kp_stat_operand(sd_in, hi), ones, z, 0, 0, 0);
                }
                static_for_<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r1[2 * i]), kf[i], s, 0, 0, 0);
                    if constexpr (kDK) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r1[2 * i + 1]), vf[i], dp, 0, 0, 0);
                    if constexpr (i < 2) {  // row fragments of d 64-127, two (four) per step
                        uint32_t ra = ra0 + so;
                        KP_LAUNDER(ra);
                        static_for_<2 * i, 2 * i + 2>([&r2, ra](auto uc) {
                            constexpr int u = decltype(uc)::value, kb = decltype(kbc)::value;
                            const uint32_t a = ra ^ (uint32_t)((4 + u) << 5);
                            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2[2 * u]) : "v"(a), "n"(kb * 32 * PITCH));
                            if constexpr (kDK) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r2[2 * u + 1]) : "v"(a), "n"(TILE + kb * 32 * PITCH));
                        });
                    }
                    if constexpr (i == 0) {
                        if (more) dma_part(2 * kb, j + AHEAD, pslot);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&r2](auto fc) {
                    if constexpr (kDK || (decltype(fc)::value & 1) == 0) asm volatile("" : "+v"(r2[decltype(fc)::value]));
                });
                static_for_<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r2[2 * i]), kf[4 + i], s, 0, 0, 0);
                    if constexpr (kDK) dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r2[2 * i + 1]), vf[4 + i], dp, 0, 0, 0);
                    {  // transposed Q (dK) / dO (dV) fragments f = 2 i, 2 i + 1 (f = 4 jj + db: 16-query step 2 kb + jj, d block db)
                        uint32_t kt = kt0 + so;
                        KP_LAUNDER(kt);
                        static_for_<2 * i, 2 * i + 2>([&tlo, &thi, kt](auto fc) {
                            constexpr int f = decltype(fc)::value, kb = decltype(kbc)::value;
                            constexpr int ks = 2 * kb + (f >> 2), db = f & 3;
                            const uint32_t a = kt ^ (uint32_t)(db << 6);
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(tlo[f]) : "v"(a), "n"(ks * 16 * PITCH));
                            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(thi[f]) : "v"(a ^ 32u), "n"(ks * 16 * PITCH + 8 * PITCH));
                        });
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                kp_barrier();
                if (more) dma_part(2 * kb + 1, j + AHEAD, pslot);
                // ---------------------------------------------------------------- E: P = exp2(scale log2e S'), dS = P dP'
                if (need_mask) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2) + 32 * kb;
                        s[r] = (c < lo || c >= up) ? -INFINITY : s[r];
                    }
                }
                uint32_t dsb[2][4];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = fast_exp2(s[r] * sl2);
                    asm volatile("" : "+v"(x));
                    s[r] = x;
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    f32x2 d = f32x2{s[r], s[r + 1]};
                    if constexpr (kDK) d = d * f32x2{dp[r], dp[r + 1]};
                    uint32_t u = kp_cvt_pk(d[0], d[1]);
                    asm volatile("" : "+v"(u));
                    dsb[r >> 3][(r & 7) >> 1] = u;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&tlo, &thi](auto fc) { asm volatile("" : "+v"(tlo[decltype(fc)::value]), "+v"(thi[decltype(fc)::value])); });
                // ---------------------------------------------------------------- C3: dK^T += Q^T dS / dV^T += dO^T P, 8 MFMAs
                static_for_<0, 8>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    constexpr int jj = f >> 2, db = f & 3;
                    const bf16x8 a = join2(tlo[f], thi[f]);
                    const bf16x8 bb = __builtin_bit_cast(bf16x8, u32x4{dsb[jj][0], dsb[jj][1], dsb[jj][2], dsb[jj][3]});
                    acc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc[db], 0, 0, 0);
                    if constexpr (f < 4) {  // row fragments d 0-63 of the next half tile
                        uint32_t ra = ra0 + (kb == 0 ? so : son);
                        KP_LAUNDER(ra);
                        const uint32_t a2 = ra ^ (uint32_t)(f << 5);
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rn[2 * f]) : "v"(a2), "n"((1 - kb) * 32 * PITCH));
                        if constexpr (kDK) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rn[2 * f + 1]) : "v"(a2), "n"(TILE + (1 - kb) * 32 * PITCH));
                    }
                    if constexpr (f == 4) {  // ... and its statistics
                        const uint32_t sa = sa0 + (uint32_t)((kb == 0 ? slot : nslot) * 512);
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sl_out) : "v"(sa), "n"((1 - kb) * 128));
                        if constexpr (kDK) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sd_out) : "v"(sa), "n"(256 + (1 - kb) * 128));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (kb == 0) {  // this wave's share of tile j + 1 landed (younger: the first two intervals' requests of tile j + 2)
                    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (wave < NSTATW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_STAT) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_PLAIN) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                static_for_<0, 8>([&rn](auto fc) {
                    if constexpr (kDK || (decltype(fc)::value & 1) == 0) asm volatile("" : "+v"(rn[decltype(fc)::value]));
                });
                asm volatile("" : "+v"(sl_out), "+v"(sd_out));
                kp_barrier();
            };
            u32x4 rm[8];
            uint32_t sml = 0, smd = 0;
            half(std::integral_constant<int, 0>{}, rf, stl, std_, rm, sml, smd);
            half(std::integral_constant<int, 1>{}, rm, sml, smd, rf, stl, std_);
            slot = nslot;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!grpB) kp_barrier();  // ... and group A waits for it at the end: every wave has passed its last LDS read

        // ---- store: lane (key = lq, hi) holds dK^T / dV^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][key]; as the forward's store tail
        {
            const int krow = wk0 + lq;
            const float f = krow < sk_len ? (kDK ? P.scale : 1.f) : 0.f;
            bf16* orow = obase + (int64_t)krow * P.dk_ss + hi * 8;
#pragma unroll
            for (int db = 0; db < DBN; ++db)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    uint32_t a0 = kp_cvt_pk(acc[db][8 * m + 0] * f, acc[db][8 * m + 1] * f);
                    uint32_t a1 = kp_cvt_pk(acc[db][8 * m + 2] * f, acc[db][8 * m + 3] * f);
                    uint32_t b0 = kp_cvt_pk(acc[db][8 * m + 4] * f, acc[db][8 * m + 5] * f);
                    uint32_t b1 = kp_cvt_pk(acc[db][8 * m + 6] * f, acc[db][8 * m + 7] * f);
                    const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    if (krow < SkE) *reinterpret_cast<u32x4*>(orow + db * 32 + m * 16) = u32x4{x0[0], x1[0], x0[1], x1[1]};
                }
        }
    }  // pass
}

template <bool CAUSAL, int PF, int MODE>
int launch_dkv_pp(const AttnParams& P, hipStream_t stream) {
    constexpr int LDS = PF * 2 * 64 * 128 * 2 + PF * 512;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_bwd_dkv_pp_kernel<CAUSAL, PF, MODE>, LDS, lds_ok);
    const int nkb = (P.Sk + 255) / 256;
    const dim3 grid(attn_grid(CAUSAL ? (nkb + 1) / 2 : nkb, P.Hkv, P.B));
    hipLaunchKernelGGL((attn_bwd_dkv_pp_kernel<CAUSAL, PF, MODE>), grid, dim3(512), LDS, stream, P);
    return dllm_check_launch();
}

}  // namespace

// Called by dllm_attn_bwd (attn_bwd.hip) for D = 128, H == Hkv, long key axes; the caller has checked shapes and alignment.  mode 1 =
// dV pass, 2 = dK pass.  Must run after the dQ kernel of the same call (reads the statistic planes of the workspace).
__attribute__((visibility("hidden"))) int dllm_launch_attn_bwd_dkv_pp(const AttnParams& P, int causal, int mode, hipStream_t stream) {
#ifndef KP_PF
#define KP_PF 3  // 4 (tile j + 3 requested during tile j: 130 KiB of LDS) measured the same: the request -> landed path is not what holds these passes
#endif
    if (mode == 1) return causal ? launch_dkv_pp<true, KP_PF, 1>(P, stream) : launch_dkv_pp<false, KP_PF, 1>(P, stream);
    return causal ? launch_dkv_pp<true, KP_PF, 2>(P, stream) : launch_dkv_pp<false, KP_PF, 2>(P, stream);
}
#endif  // DLLM_BENCH_MODES
