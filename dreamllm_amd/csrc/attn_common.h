// Shared pieces of the flash-attention kernels (forward, dQ, dK/dV): LDS images of [rows][D] bf16 tiles,
// fragment reads, and global -> register -> LDS staging.
//
// All three kernels work in the "transposed" formulation: S^T = K Q^T, O^T = V^T P^T, so a lane owns one query
// (column lane&15) and softmax statistics never leave the lane's 16-lane column group.  A [rows][D] tile is kept
// in LDS once, in its natural row-major image, and read two ways:
//   * row-fragments  (16 rows x 32 d, d contiguous)      -> ds_read_b128 from an XOR-swizzled image  (QK^T / dO V^T)
//   * col-fragments  (16 d   x 32 rows, rows = MFMA k)   -> ds_read_b64_tr_b16 transpose reads       (PV / dS K ...)
// Two images are therefore defined: ROW image (b128-friendly swizzle) and COL image (tr-read-friendly swizzle).
#pragma once
#include "common.h"
#include <type_traits>

template <int D>
struct TileImg {
    static constexpr int PITCH = D * 2;    // bytes per row
    static constexpr int CPR = D / 8;      // 16-byte chunks per row
    // ROW image: chunk c of row r lives at chunk position c ^ sw(r); 16 rows x same chunk -> 16 distinct 16-B slots.
    __device__ static __forceinline__ int row_off(int r, int c) {
        if constexpr (D == 128)
            return r * PITCH + ((c ^ (r & 15)) << 4);
        else
            return r * PITCH + ((c ^ ((r >> 1) & 7)) << 4);
    }
    // COL image: 32-byte slot s of row r lives at slot s ^ sw(r); the 8 consecutive rows a half-wave touches in one
    // transpose read land on distinct bank groups.
    __device__ static __forceinline__ int col_off(int r, int byte_in_row) {
        const int s = byte_in_row >> 5;
        int sw;
        if constexpr (D == 128)
            sw = r & 7;
        else
            sw = (r >> 1) & 3;
        return r * PITCH + ((s ^ sw) << 5) + (byte_in_row & 31);
    }
    // row-fragment: lane gets row (rbase + lane&15), d = dstep*32 + (lane>>4)*8 .. +7
    __device__ static __forceinline__ bf16x8 frag_row(const char* tile, int rbase, int dstep, int lane) {
        return *reinterpret_cast<const bf16x8*>(tile + row_off(rbase + (lane & 15), dstep * 4 + (lane >> 4)));
    }
    // col-fragment over two groups of 4 rows: lane (g = lane>>4, t = lane&15) gets column d = dbase + t and
    // rows {r_lo + g*4 + 0..3} then {r_hi + g*4 + 0..3} as the 8 MFMA k-slots of its group.
    __device__ static __forceinline__ bf16x8 frag_col(const char* tile, int dbase, int r_lo, int r_hi, int lane) {
        const int g = lane >> 4, t = lane & 15;
        const int bcol = dbase * 2 + (t & 3) * 8;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + col_off(r_lo + g * 4 + (t >> 2), bcol)));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + col_off(r_hi + g * 4 + (t >> 2), bcol)));
        union {
            struct { short4v a, b; } s;
            bf16x8 v;
        } u;
        u.s.a = lo;
        u.s.b = hi;
        return u.v;
    }
    // col-fragment read out of a ROW image (tile that is also read with frag_row): 2-way bank conflict, no second copy.
    __device__ static __forceinline__ bf16x8 frag_col_rowimg(const char* tile, int dbase, int r_lo, int r_hi, int lane) {
        const int g = lane >> 4, t = lane & 15;
        const int bcol = dbase * 2 + (t & 3) * 8;
        const int c = bcol >> 4, w = bcol & 15;
        short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + row_off(r_lo + g * 4 + (t >> 2), c) + w));
        short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + row_off(r_hi + g * 4 + (t >> 2), c) + w));
        union {
            struct { short4v a, b; } s;
            bf16x8 v;
        } u;
        u.s.a = lo;
        u.s.b = hi;
        return u.v;
    }
};

// Staging of a [ROWS][D] tile by NT threads: chunk index i = tid + NT*p -> row i / CPR, chunk i % CPR.
template <int D, int ROWS, int NT>
struct TileStage {
    static constexpr int CPR = D / 8;
    static constexpr int NP = (ROWS * CPR) / NT;
    bf16x8 v[NP];
    __device__ __forceinline__ void gload(const bf16* base, int64_t row_stride, int row0, int nrows_valid, int tid) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + NT * p;
            const int r = i / CPR, c = i % CPR;
            // rows past the end are clamped to the last valid row (finite data; every use of them is masked), so the
            // loads are unconditional: no exec-mask branches around the prefetch
            const int row = min(row0 + r, nrows_valid - 1);
            v[p] = ld_bf16x8(base + (int64_t)row * row_stride + c * 8);
        }
    }
    // ---- strength-reduced forms for tiles that lie fully inside the sequence (the common case) -------------------------
    // chunk i = tid + NT*p is row (tid / CPR) + (NT / CPR) p, column chunk tid % CPR: one per-thread offset plus a uniform
    // multiple of p.  gload_full: address = uniform base (SGPR pair) + 32-bit per-thread offset, no per-load 64-bit VALU
    // math; lstore_row_full: LDS address = per-thread offset + immediate.  (The clamped gload above costs ~12 VALU
    // instructions per load; the attention kernels are VALU-bound, DESIGN.md.)
    static constexpr int RPP = NT / CPR;  // rows advanced per p
    __device__ static __forceinline__ uint32_t thread_goff(int64_t row_stride, int tid) {
        return (uint32_t)((tid / CPR) * row_stride + (tid % CPR) * 8);
    }
    __device__ static __forceinline__ uint32_t thread_loff_row(int tid) {
        static_assert(RPP % 16 == 0, "row swizzle must not depend on p");
        return (uint32_t)TileImg<D>::row_off(tid / CPR, tid % CPR);
    }
    __device__ __forceinline__ void gload_full(const bf16* base, int64_t row_stride, int row0, uint32_t goff) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const bf16* tb = base + (int64_t)(row0 + RPP * p) * row_stride;  // uniform
            v[p] = ld_bf16x8(tb + goff);
        }
    }
    __device__ __forceinline__ void lstore_row_full(char* tile, uint32_t loff) const {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<bf16x8*>(tile + loff + p * (RPP * TileImg<D>::PITCH)) = v[p];
    }
    __device__ __forceinline__ void lstore_row(char* tile, int tid) const {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + NT * p;
            *reinterpret_cast<bf16x8*>(tile + TileImg<D>::row_off(i / CPR, i % CPR)) = v[p];
        }
    }
    __device__ __forceinline__ void lstore_col(char* tile, int tid) const {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int i = tid + NT * p;
            *reinterpret_cast<bf16x8*>(tile + TileImg<D>::col_off(i / CPR, (i % CPR) * 16)) = v[p];
        }
    }
};

struct AttnParams {
    const bf16* q; const bf16* k; const bf16* v; bf16* o;
    const bf16* dout; bf16* dq; bf16* dk; bf16* dv;
    float* lse;          // [B, H, Sq]
    float* delta;        // [B, H, Sq]  rowsum(dO * O)
    const int* seqlens;  // [B] number of valid tokens of a padded self-attention batch (Sq == Sk), or null
    const int* seqstart; // [B] index of the first valid key, or null (0): keys [start, start + len) are attended.  With
                         // Sq == Sk the queries share the span (left / right padding, modeling_dreamllm.py:523-583); with
                         // Sq != Sk (KV cache) every query is valid and sits at the END of the key axis.
    int B, H, Hkv, Sq, Sk;
    int64_t q_sb, q_ss, q_sh;      // element strides of q / o / dout / dq
    int64_t k_sb, k_ss, k_sh;      // element strides of k / v / dk / dv
    int64_t o_sb, o_ss, o_sh;
    int64_t dq_sb, dq_ss, dq_sh;   // element strides of dq
    int64_t dk_sb, dk_ss, dk_sh;   // element strides of dk / dv
    float scale;
    int causal;
};

// Per-batch valid span after shifting every base pointer to the first valid token: all kernels index rows relative to it.
struct AttnSpan {
    int kst, qst;        // first valid key / query row (pointer shift)
    int SqE, SkE;        // rows left on each axis after the shift (bounds of what may be written)
    int sq_len, sk_len;  // valid queries / keys
};
__device__ __forceinline__ AttnSpan attn_span(const AttnParams& P, int b) {
    AttnSpan s;
    s.kst = P.seqstart ? max(0, min(P.seqstart[b], P.Sk)) : 0;
    s.qst = (P.Sq == P.Sk) ? s.kst : 0;
    s.SqE = P.Sq - s.qst;
    s.SkE = P.Sk - s.kst;
    s.sk_len = P.seqlens ? max(0, min(P.seqlens[b], s.SkE)) : s.SkE;
    s.sq_len = (P.Sq == P.Sk) ? s.sk_len : s.SqE;
    return s;
}
// rows [0, n) of a [rows][D] bf16 view := 0 (the pad rows in front of a left-padded sequence)
template <int D, int NT>
__device__ __forceinline__ void zero_head_rows(bf16* base, int64_t row_stride, int n, int tid) {
    for (int i = tid; i < n * (D / 8); i += NT) st_bf16x8(base + (int64_t)(i / (D / 8)) * row_stride + (i % (D / 8)) * 8, zero_bf16x8());
}

// Work-group -> (row block, head, batch) map of every attention kernel.  The grid is 1-D; the hardware deals consecutive work-group
// ids round-robin to the 8 XCDs (id % 8).  With the natural (row block fastest) order and 8 row blocks per head, XCD k received
// row block k of EVERY head: under a causal mask XCD 0 then holds all the heaviest blocks (32 key tiles) and XCD 7 all the lightest
// (4) -- the kernel ran at the pace of XCD 0, 1.8x the balanced time (rocprofv3: 1.08 resident waves per SIMD on average in a
// kernel that fills a CU with one 8-wave group).  Here XCD k owns heads k, k + 8, ...; the row blocks of one head go to
// consecutive slots of the SAME XCD (they run concurrently on neighbouring CUs and share that XCD's L2 for the head's K / V or
// Q / dO), heaviest first.  Groups whose head index falls past the end (head count not a multiple of 8) exit.
struct AttnBlock {
    int r, h, b;   // row block (in dispatch order: 0 = first), head, batch
    bool valid;
};
// Causal 8-wave kernels do not hand out single row blocks at all: a work-group takes the PAIR (heaviest remaining, lightest
// remaining) = (nrb-1-r, r), so every group does the same number of key tiles.  (An LPT order -- an XCD walks the row blocks of
// ALL its heads heaviest first -- balanced the tail as well, +8 %, but gave up the L2 sharing between a head's blocks; pairing
// gives both, another +5...10 %.  With pairing, dealing consecutive groups of an XCD to DIFFERENT heads instead of head-major
// measured 4-9 % slower on every kernel.)
template <bool UNUSED = false>
__device__ __forceinline__ AttnBlock attn_block_map(int nrow_blocks, int heads, int B) {
    const int L = blockIdx.x;
    const int xcd = L & 7, s = L >> 3;
    AttnBlock m;
    const int slot = s / nrow_blocks;
    m.r = s - slot * nrow_blocks;
    const int bh = slot * 8 + xcd;
    m.h = bh % heads;
    m.b = bh / heads;
    m.valid = bh < heads * B;
    return m;
}
inline unsigned attn_grid(int nrow_blocks, int heads, int B) { return (unsigned)(((heads * B + 7) / 8) * 8 * nrow_blocks); }

// Pins a value loaded from global memory BEFORE a loop as "arrived": the empty asm consumes the register, so hipcc places the
// s_waitcnt vmcnt for it here, once.  Without this the loop header merges "still pending" (the path around the guarded
// prologue) into the loop, and the first MFMA that reads the register is preceded by s_waitcnt vmcnt(0..1) IN EVERY ITERATION,
// i.e. right after the next tile's global loads were issued: the wave then sits out a full HBM round trip per iteration.
template <typename T>
__device__ __forceinline__ void pin_loaded(const T& v) {
    asm volatile("" ::"v"(v));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // raw v_exp_f32 (x <= 0 here)

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---- pieces of the 8-wave pipelined kernels (forward, dQ, dK/dV) ------------------------------------------------------------
#define GLDS16_(gptr, lptr)                                                                                            \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                           \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ uint32_t lds_addr32(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ bf16x8 join2(u32x2 lo, u32x2 hi) {
    union {
        struct { u32x2 a, b; } s;
        bf16x8 v;
    } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for_(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_<I + 1, N>(f);
    }
}


// UNIFIED image of a [rows][D] tile that is read BOTH ways: chunk c (16 bytes) of row r sits at chunk position c ^ uni_f(r).
// LDS services a ds_read_b128 in 4 groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same +32
// (MI355X_MICROARCH.md, LDS) -- i.e. a row fragment's group holds rows {0-3,12-15} at chunk c and rows {4-11} at chunk c^1, and
// a ds_read_b64_tr_b16 in 2 groups of 32 lanes = 8 consecutive rows x 32 bytes.  uni_f is a bijection of r & 15 with
//   * bits 1..: the 32-byte slot swizzle, distinct over rows 0-7 and over rows 8-15 (transpose reads conflict-free);
//   * rows {4-7} and {8-11} (and {0-3}, {12-15}) share their slots pairwise and differ in bit 0, so each b128 lane group's 16
//     (row, chunk) pairs land on 16 distinct 16-byte positions (row fragments conflict-free).
// (The ROW image read by transpose reads, as the 4-wave backward kernels do, is 2-way conflicted.)
template <int D>
__device__ __forceinline__ int uni_f(int r) {
    if constexpr (D == 128) {
        return ((((r & 7) ^ ((r & 8) >> 1))) << 1) | ((r >> 3) & 1);
    } else {  // 128-byte rows: two rows per 256-byte bank window, the swizzle acts on k = r / 2
        const int k = (r >> 1) & 7;
        return (((k & 3) ^ ((k & 4) >> 1)) << 1) | (k >> 2);
    }
}
