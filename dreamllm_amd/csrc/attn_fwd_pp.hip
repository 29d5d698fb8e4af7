// Flash-attention forward, "ping-pong" form for gfx950 (round 5): the long-axis kernel of dllm_attn_fwd.
//
// Replaces the same reference code as attn_fwd.hip (flash_attn_func / flash_attn_varlen_func, modeling_dreamllm.py:532-549; eager
// attention :357-379) with the same layouts, masking rules and results contract.
//
// Why a third kernel: the 8-wave pipelined kernel (attn_fwd8_kernel) streams LDS fragments BETWEEN its MFMAs and keeps both waves of
// a SIMD in the same phase (one barrier per key tile), so the two waves want the matrix pipe at the same time and the softmax VALU
// at the same time: 5800 cycles per 64-key tile against 2176 matrix cycles (profiles/r04_attn_bench.log, r03_pmc_attention.txt:
// 38 % of the wave cycles parked at waits, 31 % issue-stalled).  Here (MI355X_MICROARCH.md "Two waves per SIMD"):
//   * a work-group is 8 waves x 32 query rows; waves 0-3 (group A) and 4-7 (group B) share the four SIMDs pairwise, and group B runs
//     ONE barrier interval behind group A.  A key tile is four intervals per wave:
//         L_K  : K fragments LDS -> registers (16 ds_read_b128), DMA requests for a later tile
//         C_QK : S^T = K Q^T, 16 MFMA 32x32x16 straight out of registers (no LDS operation inside the cluster)
//         L_V  : V fragments LDS -> registers (32 ds_read_b64_tr_b16) with the WHOLE online softmax under their latency
//         C_PV : O^T += V^T P^T, 16 MFMAs out of registers, the row sums (32 adds) in their shadow
//     so while one wave of a SIMD is in a matrix cluster its partner is in a load / softmax segment: matrix beside memory + VALU.
//   * MFMA 32x32x16: a K or V fragment read (1 KiB per wave) feeds 32 matrix cycles instead of 16 -- half the LDS traffic per FLOP of
//     the 16x16x32 kernels; a lane owns ONE query (column lane & 31) and 16 of the 32 keys of a block, so the softmax state is
//     lane-local up to one v_permlane32_swap, and P feeds the PV MFMAs from the registers it was computed in (the k-slot order of
//     the V^T fragments is permuted to match: no cross-lane move of P at all).
//   * K and V fragments time-share one 64-register block; K / V tiles arrive by LDS-DMA into PF-deep rings (K row image with the
//     b128 swizzle, V image with a 64-byte-granule swizzle for the 32-lane transpose reads), swizzles applied on the per-lane SOURCE
//     address; every LDS read is inline asm with explicit waits (hipcc would drain the DMA queue in front of a visible LDS read).
#include "attn_common.h"

namespace {

constexpr float kNegBigPP = -1.0e30f;
constexpr unsigned kTlBlock = 1024;  // the work-group the timeline diagnostic stamps: mid-grid (warm caches, steady state), XCD 0

__device__ __forceinline__ float pp_max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float pp_max2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max of a lane's 16 scores in ONE asm statement: hipcc pads every asm whose result feeds the next VALU with an s_nop, and fmaxf()
// costs a canonicalising v_max per MFMA output under -fno-finite-math-only (the scores are never NaN)
__device__ __forceinline__ float pp_max16(const f32x16& s) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3\n\t"
        "v_max3_f32 %0, %0, %4, %5\n\t"
        "v_max3_f32 %0, %0, %6, %7\n\t"
        "v_max3_f32 %0, %0, %8, %9\n\t"
        "v_max3_f32 %0, %0, %10, %11\n\t"
        "v_max3_f32 %0, %0, %12, %13\n\t"
        "v_max3_f32 %0, %0, %14, %15\n\t"
        "v_max_f32 %0, %0, %16"
        : "=&v"(r)
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(s[6]), "v"(s[7]), "v"(s[8]), "v"(s[9]), "v"(s[10]),
          "v"(s[11]), "v"(s[12]), "v"(s[13]), "v"(s[14]), "v"(s[15]));
    return r;
}
// max over elements [8 h, 8 h + 8) of BOTH score blocks, two interleaved chains (an in-order wave stalls on a dependent VALU)
template <int H>
__device__ __forceinline__ void pp_max8x2(const f32x16& a, const f32x16& b, float& ma, float& mb) {
    if constexpr (H == 0) {
        asm("v_max3_f32 %0, %2, %3, %4\n\t"
            "v_max3_f32 %1, %10, %11, %12\n\t"
            "v_max3_f32 %0, %0, %5, %6\n\t"
            "v_max3_f32 %1, %1, %13, %14\n\t"
            "v_max3_f32 %0, %0, %7, %8\n\t"
            "v_max3_f32 %1, %1, %15, %16\n\t"
            "v_max_f32 %0, %0, %9\n\t"
            "v_max_f32 %1, %1, %17"
            : "=&v"(ma), "=&v"(mb)
            : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(b[0]), "v"(b[1]), "v"(b[2]),
              "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    } else {
        asm("v_max3_f32 %0, %0, %2, %3\n\t"
            "v_max3_f32 %1, %1, %10, %11\n\t"
            "v_max3_f32 %0, %0, %4, %5\n\t"
            "v_max3_f32 %1, %1, %12, %13\n\t"
            "v_max3_f32 %0, %0, %6, %7\n\t"
            "v_max3_f32 %1, %1, %14, %15\n\t"
            "v_max3_f32 %0, %0, %8, %9\n\t"
            "v_max3_f32 %1, %1, %16, %17"
            : "+v"(ma), "+v"(mb)
            : "v"(a[8]), "v"(a[9]), "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[8]), "v"(b[9]), "v"(b[10]),
              "v"(b[11]), "v"(b[12]), "v"(b[13]), "v"(b[14]), "v"(b[15]));
    }
}
__device__ __forceinline__ uint32_t pp_cvt_pk(float lo, float hi) {
    bf16x2 w;
    w[0] = (bf16)lo;
    w[1] = (bf16)hi;
    return __builtin_bit_cast(uint32_t, w);
}
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}

// TL (bench library only): lane 0 of waves 0 and 4 of work-group 0 stamps s_memtime at both ends of every segment of the first pass
// into LDS behind the rings (asm ds_write: invisible to hipcc's wait insertion); the stamps go to P.delta after the pass.
// ABL (bench library only): ablations with wrong results by design -- 1 no in-loop DMA requests, 2 no VALU fillers beside the P V MFMAs,
// 4 no softmax head -- and priority experiments with correct results -- 16 group B at s_setprio 1 throughout, 32 every wave at priority
// 1 inside its MFMA clusters, 64 group B at priority 1 in X only.
template <int D, bool CAUSAL, int PF, bool TL = false, int ABL = 0>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(AttnParams P) {
    constexpr int NW = 8, QW = 32, BQ = NW * QW, BKV = 64;
    constexpr int DSN = D / 16;          // d steps of S^T = K Q^T          (8 / 4)
    constexpr int DBN = D / 32;          // 32-wide d blocks of O^T         (4 / 2)
    constexpr int KS = BKV / 16;         // 16-key steps of O^T += V^T P^T  (4)
    constexpr int NF = 2 * DSN;          // fragments per tile, K and V alike (16 / 8)
    static_assert(NF == DBN * KS, "K and V fragments share one register block");
    constexpr int PITCH = D * 2, TILE = BKV * PITCH;
    constexpr int CPR = D / 8, RPG = 64 / CPR, NDMA = (BKV / RPG) / NW;  // 1-KiB DMA groups per wave and tile (2 / 1)
    extern __shared__ __attribute__((aligned(1024))) char smem[];        // K ring [PF][TILE], V ring [PF][TILE]
    char* const Ksm = smem;
    char* const Vsm = smem + PF * TILE;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool grpB = wave >= 4;  // waves w and w + 4 share a SIMD; the second group runs one interval behind the first
    const int lq = lane & 31, hi = lane >> 5;
    [[maybe_unused]] uint32_t tl_addr = 0;
    [[maybe_unused]] bool tl_on = false;
    if constexpr (TL) {
        tl_on = blockIdx.x == kTlBlock && (wave == 0 || wave == 4);
        tl_addr = lds_addr32(smem) + 2 * PF * TILE + (wave == 4 ? 4096 : 0);
    }
#define PP_STAMP()                                                                                         \
    do {                                                                                                   \
        if constexpr (TL) {                                                                                \
            if (tl_on) {                                                                                   \
                const uint64_t tt_ = __builtin_amdgcn_s_memtime();                                         \
                if (lane == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(tl_addr), "v"(tt_) : "memory");    \
                tl_addr += 8;                                                                              \
            }                                                                                              \
        }                                                                                                  \
    } while (0)

#define PP_PHASE(idx)                                                                                                   \
    do {                                                                                                                \
        if constexpr (TL) {                                                                                             \
            if (blockIdx.x == kTlBlock && (wave == 0 || wave == 4) && lane == 0)                                            \
                reinterpret_cast<uint64_t*>(P.delta)[1024 + (wave == 4 ? 64 : 0) + (idx)] = __builtin_amdgcn_s_memtime(); \
        }                                                                                                               \
    } while (0)
    PP_PHASE(0);

    const int nqb = (P.Sq + BQ - 1) / BQ;
    const int nitems = CAUSAL ? (nqb + 1) / 2 : nqb;  // causal: the pair (heaviest, lightest) remaining row block per work-group
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
    uint32_t koff[NDMA], voff[NDMA];  // byte offsets of this lane's source chunk relative to the tile's first row
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int r = (wave * NDMA + i) * RPG + drow;
        int kc, vc;
        if constexpr (D == 128) {
            kc = dpos ^ (r & 15);
            vc = dpos ^ ((r & 3) << 2);
        } else {
            kc = dpos ^ ((r >> 1) & 7);
            vc = dpos ^ (((r >> 1) & 1) << 2);
        }
        koff[i] = (uint32_t)(r * (int)P.k_ss + kc * 8) * 2u;
        voff[i] = (uint32_t)(r * (int)P.k_ss + vc * 8) * 2u;
    }
    // One 1-KiB DMA group (NDMA per wave and tile) of key tile `t` into ring slot `slot`: uniform 64-bit tile base + the lane's 32-bit
    // byte offset (dllm_attn_fwd only sends shapes here whose key axis spans less than 1 GiB).  Tiles past the end of the pass re-fetch
    // the last one (into a slot nobody reads any more): the request stream -- and with it every counted wait -- stays uniform.
    const int kss2 = (int)P.k_ss * 2;
    auto dma_one = [&](const bf16* base, const uint32_t (&offb)[NDMA], char* ring, int t, int nblk_, int slot, int i) {
        char* dst = ring + slot * TILE + (wave * NDMA + i) * 1024;
        const int row0 = min(t, nblk_ - 1) * BKV;
        const char* tb = reinterpret_cast<const char*>(base) + (uint32_t)(row0 * kss2);
        uint32_t o = offb[i];
        if (row0 + BKV > sk_len) {  // ragged last tile (rare): rows past the end re-read the last valid row (finite data; masked later)
            asm volatile("" ::: "memory");  // keeps this path a branch (not if-converted into the common one)
            const int r = (wave * NDMA + i) * RPG + drow;
            if (row0 + r > sk_len - 1) o = o - (uint32_t)(r * kss2) + (uint32_t)((sk_len - 1 - row0) * kss2);
        }
        GLDS16_(tb + o, dst);
    };

    // ---- fragment addresses (slot 0).  K fragment (kb, ds): row 32 kb + lq, chunk (2 ds + hi) ^ swizzle(row) = one XOR with ds << 5.
    uint32_t ka0, va0;
    {
        const int swk = (D == 128) ? (lq & 15) : ((lq >> 1) & 7);
        ka0 = lds_addr32(Ksm) + (uint32_t)(lq * PITCH + ((hi ^ swk) << 4));
        // V fragment (db, ks) = two transpose reads of [4 keys][16 d] blocks per 16-lane group: lane (g = lane >> 4, t = lane & 15)
        // addresses row 16 ks + 4 hi + (t >> 2) (+ 8), bytes 64 db + 32 (g & 1) + 8 (t & 3), and receives column 16 (g & 1) + t of the
        // block's four keys: exactly the A operand of a 32x32x16 MFMA whose k slots are {16 ks + 4 hi + 0..3, 16 ks + 8 + 4 hi + 0..3}
        // -- the keys whose scores the SAME half-wave holds after S^T = K Q^T.  The 64-byte granule db sits at granule db ^ (row & 3)
        // ((row >> 1) & 1 at 128-byte rows): the four rows a half-wave reads land on four different 64-byte bank windows.
        const int t = lane & 15, gb = (lane >> 4) & 1;
        const int swg = (D == 128) ? (t >> 2) : ((t >> 3) & 1);
        va0 = lds_addr32(Vsm) + (uint32_t)((4 * hi + (t >> 2)) * PITCH + 32 * gb + 8 * (t & 3) + (swg << 6));
    }

    const bf16* qhead = P.q + (int64_t)b * P.q_sb + (int64_t)h * P.q_sh + (int64_t)sp.qst * P.q_ss;
    // first tiles of a pass (K / V tiles 0 .. PF - 2 of the head) and this wave's query rows (B operand of S^T: lane = query lq, d = 16 ds + 8 hi ..)
    bf16x8 qf[DSN];
    auto request_pass = [&](int qblk_, int nblk_) {
        // a pass without a visible key (causal Sq > Sk: the first query blocks; an empty key axis) requests nothing -- dma_one's
        // clamp min(t, nblk_ - 1) would otherwise address the tile in front of the key base
        for (int i = 0; nblk_ > 0 && i + 1 < PF; ++i) {
#pragma unroll
            for (int u = 0; u < NDMA; ++u) dma_one(kbase, koff, Ksm, i, nblk_, i, u);
#pragma unroll
            for (int u = 0; u < NDMA; ++u) dma_one(vbase, voff, Vsm, i, nblk_, i, u);
        }
        const bf16* qrow = qhead + (int64_t)min(qblk_ * BQ + wave * QW + lq, sq_len - 1) * P.q_ss;
#pragma unroll
        for (int ds = 0; ds < DSN; ++ds) qf[ds] = ld_bf16x8(qrow + ds * 16 + hi * 8);
    };
    bool requested = false;  // the pass's first tiles / query rows were requested in front of the previous pass's store tail

    const int npass = (CAUSAL && nqb - 1 - bm.r != bm.r) ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
        const int qblk = CAUSAL ? (pass == 0 ? nqb - 1 - bm.r : bm.r) : bm.r;
        const int q0 = qblk * BQ, wq0 = q0 + wave * QW;
        bf16* obase = P.o + (int64_t)b * P.o_sb + (int64_t)h * P.o_sh;
        float* lsebase = P.lse ? P.lse + ((int64_t)b * P.H + h) * P.Sq : nullptr;
        if (sp.qst > 0) {  // left padding: the rows in front of the sequence are zeros, then work relative to the first valid row
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
        PP_PHASE(1 + 4 * pass);

        // ---- Structure of a pass (round-5 "v5").  Per key tile j a wave runs two segments, separated by work-group barriers:
        //   X(j): C_QK(j) = the 2 DSN MFMAs of S^T = K Q^T out of the K fragments that are already in registers; beside them (fillers)
        //         the transpose reads of V(j) -- into the registers the K fragments vacate -- and the DMA requests of tile j + PF - 1.
        //   Y(j): the serial head of the online softmax (row max, rescale decision, exponents, P of the first 16 keys), then C_PV(j) =
        //         the NF MFMAs of O^T += V^T P^T; beside them the exponentials / packs / row sums of the other 48 keys, the reads of
        //         K(j + 1) into the registers the V fragments vacate.
        //   The older wave of a SIMD wins every arbitration (matrix pipe and VALU: tools/mfma_filler_probe.hip -- two waves in MFMA
        //   clusters at once run one after the other), so the cut is placed where the partner's work is complementary: a wave's
        //   softmax head (no MFMA) opens the segment whose partner segment is the pure cluster X.
        // Group B (waves 4-7) runs ONE segment behind group A, so one wave of a SIMD is in X while its partner is in Y.  There is no
        // load-only segment: every LDS read has a whole cluster between its request and its wait.
        // DMA stream per wave: K0 V0 .. K(PF-2) V(PF-2) (prologue), then V(j + PF - 1) and K(j + PF - 1) beside the MFMAs of X(j) (the
        // shorter segment); requests retire in order, so the waits at the segment ends are counted (below).  Slots: both reuse slot
        // (j - 1) % PF -- V(j - 1) was read in X(j - 1) and K(j - 1) in Y(j - 2) by both groups before either reaches X(j).
        constexpr int YOUNGER_X = (1 + 2 * (PF - 2)) * NDMA;  // end of X(j): V(j + 1) landed <=> at most K(j + 1) .. K(j + PF - 1) outstanding
        constexpr int YOUNGER_Y = 2 * (PF - 3) * NDMA;        // end of Y(j): K(j + 2) landed <=> at most V(j + 3) .. K(j + PF - 1) outstanding
        static_assert(PF >= 3, "K(j + 2) is requested in X(j + 3 - PF)");
        if (!requested) request_pass(qblk, nblk);
#pragma unroll
        for (int ds = 0; ds < DSN; ++ds) pin_loaded(qf[ds]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tiles landed (this wave's shares); the barrier below publishes them
        f32x16 oacc[DBN];
#pragma unroll
        for (int db = 0; db < DBN; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
        f32x16 s0, s1;  // scores / probabilities of key blocks 0 and 1 of the tile
        uint32_t pb[KS][4];
        float m_run = kNegBigPP, l_run = 0.f;
        // A wave takes part in tile j iff one of its queries can see one of the tile's keys: tiles [0, nact), then it only keeps the
        // DMA / barrier protocol going (two separate loops: fragments and scores are not live across the idle part).
        int nact = 0;
        if (wq0 < sq_len) nact = CAUSAL ? max(0, min(nblk, (wq0 + QW - 1 + coff) / BKV + 1)) : nblk;
        if (CAUSAL && wq0 + QW - 1 + coff < 0) nact = 0;

        pp_barrier();
        u32x4 kf[NF];  // K fragments of the NEXT C_QK (loop carried: requested beside the P V MFMAs of the tile before)
        if (nact > 0) {
            static_for_<0, NF>([&kf, ka0](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int ds = f >> 1, kb = f & 1;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[f]) : "v"(ka0 ^ (uint32_t)(ds << 5)), "n"(kb * 32 * PITCH));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for_<0, NF>([&kf](auto fc) { asm volatile("" : "+v"(kf[decltype(fc)::value])); });
        }
        if (grpB) pp_barrier();  // group B starts one segment late ...
        if constexpr ((ABL & 16) != 0) {
            if (grpB) __builtin_amdgcn_s_setprio(1);
        }
        PP_PHASE(2 + 4 * pass);

        int slot = 0;  // j % PF
        int j = 0;
        for (; j < nact; ++j) {
            const int kv0 = j * BKV;
            const int pslot = slot == 0 ? PF - 1 : slot - 1;       // (j - 1) % PF: slot of V(j + PF - 1)
            const int nslot = slot + 1 == PF ? 0 : slot + 1;       // (j + 1) % PF: slot of K(j + 1)
            u32x2 vlo[NF], vhi[NF];   // V fragments (take over the K fragments' registers as those are consumed)
            // ------------------------------------------------------------------ X(j): C_QK  (+ V(j) fragment reads, DMA of V(j + PF - 1))
            if constexpr ((ABL & 32) != 0) __builtin_amdgcn_s_setprio(1);
            if constexpr ((ABL & 64) != 0) {
                if (grpB) __builtin_amdgcn_s_setprio(1);
            }
            {
                const uint32_t va = va0 + (uint32_t)(slot * TILE);
                static_for_<0, DSN>([&](auto dc) {
                    constexpr int ds = decltype(dc)::value;
                    const bf16x8 k0 = __builtin_bit_cast(bf16x8, kf[2 * ds]), k1 = __builtin_bit_cast(bf16x8, kf[2 * ds + 1]);
                    if constexpr (ds == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ds], z, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ds], z, 0, 0, 0);
                    } else {
                        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[ds], s0, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[ds], s1, 0, 0, 0);
                    }
                    static_for_<2 * ds, 2 * ds + 2>([&vlo, &vhi, va](auto fc) {
                        constexpr int f = decltype(fc)::value;
                        constexpr int ks = f / DBN, db = f % DBN;
                        const uint32_t a = va ^ (uint32_t)(db << 6);
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[f]) : "v"(a), "n"(ks * 16 * PITCH));
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[f]) : "v"(a), "n"(ks * 16 * PITCH + 8 * PITCH));
                    });
                    if constexpr (ds >= 1 && ds - 1 < NDMA && !(ABL & 1)) dma_one(vbase, voff, Vsm, j + PF - 1, nblk, pslot, ds - 1);
                    if constexpr (ds >= 1 + DSN / 2 && ds - 1 - DSN / 2 < NDMA && !(ABL & 1))
                        dma_one(kbase, koff, Ksm, j + PF - 1, nblk, pslot, ds - 1 - DSN / 2);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            // ------------------------------------------------------------------ X(j) tail: mask and row max (still beside the partner's Y)
            float m_new = m_run;
            if constexpr (!(ABL & 4)) {
                const bool need_mask = (kv0 + BKV > sk_len) || (CAUSAL && (kv0 + BKV - 1 > wq0 + coff));
                if (need_mask) {
                    // key kv0 + c + 4 hi (c = the register's compile-time offset) is dead iff it lies past the lane's last visible key
                    const int last = CAUSAL ? min(wq0 + lq + coff, sk_len - 1) : sk_len - 1;
                    const int lim = last - kv0 - 4 * hi;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = (r & 3) + 8 * (r >> 2);
                        s0[r] = (c > lim) ? -INFINITY : s0[r];
                        s1[r] = (c + 32 > lim) ? -INFINITY : s1[r];
                    }
                }
                // row max over the lane's 32 keys (two interleaved chains), then across the two half-waves that share the query
                float mxa, mxb;
                pp_max8x2<0>(s0, s1, mxa, mxb);
                pp_max8x2<1>(s0, s1, mxa, mxb);
                const float mb = pp_max2(mxa, mxb);
                const uint32_t mu = __float_as_uint(mb);
                const auto sw = __builtin_amdgcn_permlane32_swap(mu, mu, false, false);
                m_new = pp_max3(m_run, __uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_X) : "memory");  // this wave's share of V(j + 1) has landed
            PP_STAMP();
            pp_barrier();
            PP_STAMP();
            // ------------------------------------------------------------------ Y(j): rescale decision, exponents, P of step 0
            // Y is the long segment: its wave runs at priority 1, so the partner's X (priority 0, ~300 cycles of slack) is the one that waits
            // at the arbitration -- with equal priorities the OLDER wave wins whatever it is doing (-6 % per tile, profiles r05 timeline).
            __builtin_amdgcn_s_setprio(1);
            float nm = 0.f;
            if constexpr (!(ABL & 4)) {
                // The running max only advances when some row's max grew by more than 2^kDefer (exp2 domain): otherwise this tile's
                // probabilities are taken against the old max (bounded by 2^kDefer, invisible in O = sum(P V) / sum(P)).  At this point
                // nothing is pending at the old scale except O and l themselves (P V of the previous tile is complete, its row sums
                // are in l): they take the factor exactly once.
                constexpr float kDefer = 6.0f;
                if (__any((m_new - m_run) * sl2 > kDefer)) {
                    const float alpha = fast_exp2((m_run - m_new) * sl2);  // 0 while the old max is the finite "minus infinity"
#pragma unroll
                    for (int db = 0; db < DBN; ++db) oacc[db] *= alpha;
                    l_run *= alpha;
                    m_run = m_new;
                }
                nm = -m_run * sl2;
                // Exponents of the whole tile (independent FMAs), then the probabilities of the first 16-key step only (they feed the first
                // DBN MFMAs of C_PV); the other steps are exponentiated in the shadow of the MFMAs, one group ahead of their pack, so
                // that no filler depends on a result of its own group (an in-order wave stalls on a dependent VALU / transcendental: the
                // dependent form measured 71 cycles per MFMA group).  P^T as B operands: the k slots of step ks = 2 kb + jj are the
                // lane's scores 8 jj .. 8 jj + 7 of key block kb, in order.  Packed words and the exponent registers are pinned where
                // they are produced (empty volatile asm): LLVM otherwise SINKS the chain behind the barrier.
                {
                    const f32x2 sl22 = f32x2{sl2, sl2}, nm2 = f32x2{nm, nm};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {  // v_pk_fma_f32: the head is issue bound, not shadowed by MFMAs
                        f32x2 a = f32x2{s0[r], s0[r + 1]}, c = f32x2{s1[r], s1[r + 1]};
                        a = __builtin_elementwise_fma(a, sl22, nm2);
                        c = __builtin_elementwise_fma(c, sl22, nm2);
                        s0[r] = a[0]; s0[r + 1] = a[1];
                        s1[r] = c[0]; s1[r + 1] = c[1];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(s0[r]), "+v"(s1[r]));
#pragma unroll
                for (int r = 0; r < 8; ++r) s0[r] = fast_exp2(s0[r]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    uint32_t u = pp_cvt_pk(s0[2 * w], s0[2 * w + 1]);
                    asm volatile("" : "+v"(u));
                    pb[0][w] = u;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // V(j) fragments are in their registers (requested a segment ago)
            static_for_<0, NF>([&vlo, &vhi](auto fc) { asm volatile("" : "+v"(vlo[decltype(fc)::value]), "+v"(vhi[decltype(fc)::value])); });
            // ------------------------------------------------------------------ Y(j): C_PV
            if constexpr ((ABL & 32) != 0) __builtin_amdgcn_s_setprio(1);
            if constexpr ((ABL & 64) != 0) {
                if (grpB) __builtin_amdgcn_s_setprio(0);
            }
            // Fillers in FRONT of MFMA g (~5 VALU issues hide under one 32x32x16 MFMA, tools/mfma_filler_probe.hip), pairs p = 0 .. 11 of
            // scores of steps 1 .. KS - 1 in order:  exp2 of pair g  |  pack + row-sum adds of pair g - 1 (exponentiated one group earlier)
            // |  the row-sum adds of step 0 in the last groups  |  behind the MFMA: the read of K(j + 1) fragment g, DMA of K(j + PF).
            {
                float la = 0.f, lb = 0.f;
                constexpr int PPG = (12 + NF - 1) / NF;  // score pairs per group: the 12 pairs of steps 1..3 over NF groups (1 / 2)
                // score `half` of pair p (pairs of steps 1..3 in order: step 1 + p / 4, packed word p % 4)
                auto sget = [&](int p, int half) -> float {
                    const int ks = 1 + p / 4, e = 2 * (p % 4) + half;
                    return ks < 2 ? s0[8 * ks + e] : s1[8 * (ks - 2) + e];
                };
                auto sset = [&](int p, int half, float x) {
                    const int ks = 1 + p / 4, e = 2 * (p % 4) + half;
                    if (ks < 2) s0[8 * ks + e] = x;
                    else s1[8 * (ks - 2) + e] = x;
                };
                const uint32_t kan = ka0 + (uint32_t)(nslot * TILE);
                static_for_<0, NF>([&](auto fc) {
                    constexpr int f = decltype(fc)::value;
                    constexpr int ks = f / DBN, db = f % DBN;
#pragma unroll
                    for (int p = f * PPG; p < ((ABL & 2) ? 0 : (f + 1) * PPG) && p < 12; ++p) {  // exp2 of this group's pairs
                        float x0 = fast_exp2(sget(p, 0)), x1 = fast_exp2(sget(p, 1));
                        asm volatile("" : "+v"(x0), "+v"(x1));
                        sset(p, 0, x0);
                        sset(p, 1, x1);
                    }
                    if constexpr (f > 0 && !(ABL & 2)) {
#pragma unroll
                        for (int p = (f - 1) * PPG; p < f * PPG && p < 12; ++p) {  // pack + row sums of the previous group's pairs
                            const float x0 = sget(p, 0), x1 = sget(p, 1);
                            uint32_t u = pp_cvt_pk(x0, x1);
                            asm volatile("" : "+v"(u));
                            pb[1 + p / 4][p % 4] = u;
                            asm volatile("v_add_f32 %0, %0, %1" : "+v"(la) : "v"(x0));
                            asm volatile("v_add_f32 %0, %0, %1" : "+v"(lb) : "v"(x1));
                        }
                    }
                    if constexpr (f >= NF - 4 && !(ABL & 2)) {  // step 0's eight probabilities (exponentiated in X): two adds in each of the last 4 groups
                        constexpr int e = 2 * (f - (NF - 4));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(la) : "v"(s0[e]));
                        asm volatile("v_add_f32 %0, %0, %1" : "+v"(lb) : "v"(s0[e + 1]));
                    }
                    const bf16x8 va_ = join2(vlo[f], vhi[f]);
                    const bf16x8 pv = __builtin_bit_cast(bf16x8, u32x4{pb[ks][0], pb[ks][1], pb[ks][2], pb[ks][3]});
                    oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_, pv, oacc[db], 0, 0, 0);
                    {  // K(j + 1) fragment f (order of the next C_QK: d step f / 2, key block f % 2)
                        constexpr int ds = f >> 1, kb = f & 1;
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[f]) : "v"(kan ^ (uint32_t)(ds << 5)), "n"(kb * 32 * PITCH));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                l_run += la + lb;
            }
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_Y) : "memory");  // this wave's share of K(j + 2) has landed
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                // K(j + 1) fragments are in their registers
            static_for_<0, NF>([&kf](auto fc) { asm volatile("" : "+v"(kf[decltype(fc)::value])); });
            PP_STAMP();
            pp_barrier();
            PP_STAMP();
            slot = nslot;
        }
        for (; j < nblk; ++j) {  // idle part (causal: tiles above this wave's rows): keep the DMA shares and the barriers going
            const int pslot = slot == 0 ? PF - 1 : slot - 1;
#pragma unroll
            for (int u = 0; u < NDMA; ++u) dma_one(vbase, voff, Vsm, j + PF - 1, nblk, pslot, u);
#pragma unroll
            for (int u = 0; u < NDMA; ++u) dma_one(kbase, koff, Ksm, j + PF - 1, nblk, pslot, u);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_X) : "memory");
            pp_barrier();
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_Y) : "memory");
            pp_barrier();
            slot = slot + 1 == PF ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the re-fetches past the end have landed before the next pass reuses the rings
        PP_PHASE(3 + 4 * pass);
        if (!grpB) pp_barrier();  // ... and group A waits for it at the end: every wave has passed its last LDS read
        if constexpr (TL) {
            PP_STAMP();
            if (tl_on) {  // stamps of this pass -> P.delta ([2][512] 64-bit words: wave 0, wave 4), then no more stamping
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const uint32_t base = lds_addr32(smem) + 2 * PF * TILE + (wave == 4 ? 4096 : 0);
                uint64_t* dst = reinterpret_cast<uint64_t*>(P.delta) + (wave == 4 ? 512 : 0);
                for (int i = lane; i < 512; i += 64) {
                    u32x2 w;
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(w) : "v"(base + 8 * i) : "memory");
                    dst[i] = (i < (int)((tl_addr - base) >> 3)) ? (((uint64_t)w[1] << 32) | w[0]) : 0ull;
                }
                tl_on = false;
            }
        }

        // ---- second pass of a causal pair: its first tiles and query rows are requested NOW (every wave is past its last LDS read, the
        // query registers are free), so the ~9.5 k cycles of prologue latency run under this pass's store tail
        requested = false;
        if (CAUSAL && pass + 1 < npass) {
            const int qblk_n = bm.r, q0n = qblk_n * BQ;
            const int kv_end_n = min(sk_len, q0n + BQ + coff);
            if (q0n < sq_len && kv_end_n > 0) {
                request_pass(qblk_n, (kv_end_n + BKV - 1) / BKV);
                requested = true;
            }
        }

        // ---- finalize: lane (q = lq, hi) holds O^T[d = 32 db + (r & 3) + 8 (r >> 2) + 4 hi][q]; row sums are split over the two half-waves
        {
            const uint32_t lu = __float_as_uint(l_run);
            const auto sw = __builtin_amdgcn_permlane32_swap(lu, lu, false, false);
            const float l = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            const int qrow = wq0 + lq;
            const bool valid = qrow < sq_len;
            const float inv = (l > 0.f && valid) ? 1.0f / l : 0.f;
            // store tail (guide T21): the half-waves hold alternating 4-column groups of a row; one v_permlane32_swap per dword and pair
            // of groups gives lanes < 32 columns [16 m, 16 m + 8) and lanes >= 32 columns [16 m + 8, 16 m + 16) of block db: 16-byte stores
            bf16* orow = obase + (int64_t)qrow * P.o_ss + hi * 8;
#pragma unroll
            for (int db = 0; db < DBN; ++db)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    uint32_t a0 = pp_cvt_pk(oacc[db][8 * m + 0] * inv, oacc[db][8 * m + 1] * inv);
                    uint32_t a1 = pp_cvt_pk(oacc[db][8 * m + 2] * inv, oacc[db][8 * m + 3] * inv);
                    uint32_t b0 = pp_cvt_pk(oacc[db][8 * m + 4] * inv, oacc[db][8 * m + 5] * inv);
                    uint32_t b1 = pp_cvt_pk(oacc[db][8 * m + 6] * inv, oacc[db][8 * m + 7] * inv);
                    const auto x0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto x1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    if (qrow < SqE) *reinterpret_cast<u32x4*>(orow + db * 32 + m * 16) = u32x4{x0[0], x1[0], x0[1], x1[1]};
                }
            if (qrow < SqE && lsebase && hi == 0) lsebase[qrow] = (valid && l > 0.f) ? (m_run * P.scale + logf(l)) : 0.f;
        }
        PP_PHASE(4 + 4 * pass);
    }  // pass
}

template <int D, bool CAUSAL, int PF, bool TL = false, int ABL = 0>
int launch_pp(const AttnParams& P, hipStream_t stream) {
    constexpr int LDS = 2 * PF * 64 * D * 2 + (TL ? 8192 : 0);
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&attn_fwd_pp_kernel<D, CAUSAL, PF, TL, ABL>, LDS, lds_ok);
    const int nqb = (P.Sq + 255) / 256;
    const dim3 grid(attn_grid(CAUSAL ? (nqb + 1) / 2 : nqb, P.H, P.B));
    hipLaunchKernelGGL((attn_fwd_pp_kernel<D, CAUSAL, PF, TL, ABL>), grid, dim3(512), LDS, stream, P);
    return dllm_check_launch();
}

}  // namespace

// Called by dllm_attn_fwd (attn_fwd.hip) for the long-axis shapes; same argument checks apply there.
__attribute__((visibility("hidden"))) int dllm_launch_attn_fwd_pp(const AttnParams& P, int D, int causal, hipStream_t stream) {
#ifdef DLLM_BENCH_MODES
    if (P.delta != nullptr && D == 128 && causal) {  // timeline diagnostic; bits 8.. of `causal` = ablation
        switch (causal >> 8) {
            case 0: return launch_pp<128, true, 3, true, 0>(P, stream);
            case 1: return launch_pp<128, true, 3, true, 1>(P, stream);
            case 2: return launch_pp<128, true, 3, true, 2>(P, stream);
            case 3: return launch_pp<128, true, 3, true, 3>(P, stream);
            case 4: return launch_pp<128, true, 3, true, 4>(P, stream);
            case 7: return launch_pp<128, true, 3, true, 7>(P, stream);
            case 8: return launch_pp<128, true, 3, true, 8>(P, stream);
            case 15: return launch_pp<128, true, 3, true, 15>(P, stream);
            case 16: return launch_pp<128, true, 3, true, 16>(P, stream);
            case 32: return launch_pp<128, true, 3, true, 32>(P, stream);
            case 64: return launch_pp<128, true, 3, true, 64>(P, stream);
            case 128: return launch_pp<128, true, 3, true, 128>(P, stream);
            default: return DLLM_ERR_SHAPE;
        }
    }
#endif
    if (D == 128) return causal ? launch_pp<128, true, 3>(P, stream) : launch_pp<128, false, 3>(P, stream);
    return causal ? launch_pp<64, true, 3>(P, stream) : launch_pp<64, false, 3>(P, stream);
}
