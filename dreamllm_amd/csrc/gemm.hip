// bf16 MFMA GEMM family for gfx950 (fp32 accumulate) with fused epilogues, plus NHWC implicit-GEMM convolution.
//
//   C[M,N] = epilogue( A[M,K] * B[K,N] )
//
// One kernel template covers every contraction on the DreamLLM hot path:
//   forward  linear  y = x W^T      : A k-contiguous ([M,K]),  B k-contiguous (weight [N,K])      -> (A_K, B_K)
//   dgrad            dx = dy W      : A k-contiguous ([M,N']), B n-contiguous (weight as [K',N])  -> (A_K, B_N)
//   wgrad            dW = dy^T x    : A m-contiguous (dy as [K',M]), B n-contiguous (x as [K',N]) -> (A_M, B_N)
//   conv3x3 / 1x1 on NHWC           : A gathered per (row, tap) with zero padding                 -> (A_CONV, B_K)
// Reference call sites replaced: nn.Linear in DreamLLMAttention/DreamLLMMLP/lm_head
// (omni/models/dreamllm/modeling_dreamllm.py:219-237,273-276,1216), the projectors
// (omni/models/projector/mlp_projector.py:19,39-44) and the linears/convs inside CLIPVisionModel and
// UNet2DConditionModel ([ext]).
//
// Structure (template T): T=128: 128x128x64 block tile, 4 waves (2x2), wave tile 64x64 = 4x4 MFMA 16x16x32 tiles, 64 KiB LDS;
//                         T=256: 256x256x64 block tile, 8 waves (2x4), wave tile 128x64 = 8x4 MFMA tiles, 128 KiB LDS
//                         (one block per CU, 2 waves per SIMD; used when the grid still covers the 256 CUs >= 1.5x).
// Operands are staged global -> registers -> LDS (double buffered, one barrier per K tile, the next tile's global
// loads are in flight during the MFMAs).  k-contiguous tiles are read with ds_read_b128 from an XOR-swizzled
// [rows][64] image; reduction-dim-strided tiles keep their natural [64][128] image and are read with the gfx950
// transpose read ds_read_b64_tr_b16, so no operand is ever transposed in HBM.  The MFMA is issued with the operands
// swapped (D^T = B^T A^T) so every lane ends up with 4 consecutive output columns of one row -> 8/16-byte stores.
// blockIdx is remapped so each XCD (private 4 MiB L2) works on a contiguous group of tiles (GROUP_M swizzle).
#include "gemm_shared.h"
#include "gemm_epilogues.h"

namespace {


// Kernel choice is a per-call argument (`variant`, see include/dreamllm_hip.h): the library keeps no mutable state.
struct Variant {
    int force_tile = 0;  // 0 auto, 128, 256
    int use_glds = 1;    // direct-to-LDS 256-tile kernel when eligible
    int glds_pipe = 1;   // software-pipelined direct-to-LDS kernel (default); 0 = plain direct-to-LDS kernel
    int dbg_noload = 0;  // benchmark-only wrong-result modes; compiled in with -DDLLM_BENCH_MODES only
    int group_m = 0;     // 0: per-layout default
    int force_n128 = 0;  // tile code 262: the 256 x 128 pipelined kernel wherever it is eligible (tests)
    int persist = 0;     // bit 24: XCD-synchronised persistent walk of the pipelined 256-tile kernel (needs the workspace)
    int force_ring = 0;  // tile code 264: the 128 x 128 ring-buffered kernel (gemm_ring.hip) wherever it is eligible (tests, tools)
    int streamk_ws = 0;  // bit 26: with splitk <= 1 the caller's workspace is a stream-K workspace of dllm_gemm_streamk_ws_bytes() bytes (ADVICE r03:
                         // without the bit a workspace passed with splitk <= 1 is ignored, as before round 3 -- no unchecked 128 MiB writes)
    int ring_stages = 0;  // tile codes 267 / 268: force the four- / two-stage ring (0: the launcher decides)
    int no_ring = 0;     // bit 25: never pick the ring-buffered kernel by itself (the round-3 selection: A/B knob of tools / bench)
    int dma_mode = -1;     // bench builds: tile codes 271-276, placement of the ring kernel's LDS-DMA requests (experiment)
    int force_mfma32 = 0;  // tile code 266: the MFMA 32x32x16 experiment (gemm_mfma32.hip) wherever it is eligible (tools)
    int force_w4 = 0;      // tile code 261: the four-wave MFMA 32x32x16 experiment (gemm_w4.hip) wherever it is eligible (tools)
    int w4m = 0;           // the four-wave 16x16x32 kernel (gemm_w4.hip: gemm_w4m_kernel) instead of gemm_pipe_kernel: 1 = where the launcher
                           // finds it faster (tile code 0), 2 = wherever it is eligible (tile code 280: tests, tools); 259 keeps the 8-wave kernel
};
static inline int parse_variant(int variant, Variant& v) {
    const int tile = variant & 0xffff;
    v.group_m = (variant >> 16) & 0xff;
    bool ok = tile == 0 || tile == 128 || tile == 256 || tile == 257 || tile == 259 || tile == 261 || tile == 262 || tile == 280 || tile == 264 || tile == 266 || tile == 267 || tile == 268;
#ifdef DLLM_BENCH_MODES
    ok = ok || tile == 258 || tile == 260 || tile == 263 || tile == 265 || (tile >= 269 && tile <= 279);
    v.dbg_noload = (tile == 258 || tile == 260) ? 1 : (tile == 263 ? 2 : (tile == 265 ? 3 : ((tile >= 269 && tile <= 279) ? 4 : 0)));
    // 271 / 272 / 273: 269 with the K tile's four LDS-DMA requests placed differently (all at the tile start / right behind the barrier /
    // two and two); 274 / 275 / 276: the same for the two-stage ring
    // 269 / 270: the ring kernel (four- / two-stage) with s_memtime stamps of block 0 / wave 0 around every K tile's wait and barrier,
    // written to the workspace as uint64 (tools/ring_timeline.py): correct results, timing diagnostic only
#endif
    if (!ok || (variant >> 29) != 0) return DLLM_ERR_SHAPE;
    if ((variant >> 27) & 1) v.ring_stages = -1;   // bit 27: the ring kernel always with four stages (round-4 A/B knob)
    v.no_ring = (variant >> 25) & 1;
    v.streamk_ws = (variant >> 26) & 1;
    v.persist = (variant >> 24) & 1;
    v.use_glds = (tile == 0 || tile >= 257);
    v.glds_pipe = (tile == 0 || tile == 259 || tile == 260 || tile == 262 || tile == 263 || tile == 265);
    v.force_tile = tile >= 257 ? 256 : tile;
    v.force_n128 = tile == 262;
    v.force_ring = tile == 264 || tile == 267 || tile == 268 || (tile >= 269 && tile <= 279);   // 267: four-stage ring forced, 268: two-stage ring forced (tools)
    if (tile == 267 || tile == 268 || (tile >= 269 && tile <= 279)) v.ring_stages = (tile == 267 || tile == 269 || (tile >= 271 && tile <= 273) || tile >= 277) ? -1 : 1;
    v.dma_mode = (tile >= 271 && tile <= 273) ? tile - 270 : ((tile >= 274 && tile <= 276) ? tile - 273 : ((tile == 277 || tile == 278) ? tile - 273 : (tile == 279 ? 0 : -1)));   // -1: the kernel's default; 279: two-stage with one request per MFMA group   // 277: no LDS-DMA in the loop, 278: no MFMAs (ablations, two-stage flag ignored: four-stage)
    v.force_mfma32 = tile == 266;
    v.force_w4 = tile == 261;
    v.w4m = tile == 280 ? 2 : (tile == 0 ? 1 : 0);
    if (tile == 280) v.glds_pipe = 1;
    if (((variant >> 28) & 1) && v.w4m == 1) v.w4m = 0;   // bit 28: never choose the four-wave kernel automatically (A/B knob)
    if (v.force_ring) v.force_tile = 0;
    return DLLM_OK;
}



// ---- global -> register staging -----------------------------------------------------------------------------------
struct Stage {
    bf16x8 v[4];
};

// k-contiguous operand: thread t loads rows (t>>3) + (T/4)p, chunk t&7   (2T threads, T rows, 8 chunks per row).
template <int T>
__device__ __forceinline__ void gload_kc(Stage& s, const bf16* base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0,
                                         int64_t K, int tid) {
    const int chunk = tid & 7;
    const int64_t k = k0 + chunk * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t row = row0 + (tid >> 3) + (T / 4) * p;
        if (row < nrows && k < K)
            s.v[p] = ld_bf16x8(base + row * ld + k);
        else
            s.v[p] = zero_bf16x8();
    }
}
template <int T>
__device__ __forceinline__ void lstore_kc(const Stage& s, char* tile, int tid) {
    const int chunk = tid & 7;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int row = (tid >> 3) + (T / 4) * p;
        *reinterpret_cast<bf16x8*>(tile + kc_off(row, chunk)) = s.v[p];
    }
}
// m-contiguous operand ([K][ld] with the tile's T m/n columns contiguous): T/8 chunks per k row, 16 k rows per pass.
template <int T>
__device__ __forceinline__ void gload_mc(Stage& s, const bf16* base, int64_t ld, int64_t col0, int64_t ncols, int64_t k0,
                                         int64_t K, int tid) {
    constexpr int CPR = T / 8;
    const int c16 = tid % CPR;
    const int64_t col = col0 + c16 * 8;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t k = k0 + (tid / CPR) + 16 * p;
        if (k < K && col < ncols)
            s.v[p] = ld_bf16x8(base + k * ld + col);
        else
            s.v[p] = zero_bf16x8();
    }
}
template <int T>
__device__ __forceinline__ void lstore_mc(const Stage& s, char* tile, int tid) {
    constexpr int CPR = T / 8;
    const int c16 = tid % CPR;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int krow = (tid / CPR) + 16 * p;
        *reinterpret_cast<bf16x8*>(tile + mc_off<T>(krow, c16 * 16)) = s.v[p];
    }
}

// implicit-GEMM gather for NHWC convolution.  Row m = (img, oh, ow); k = (kh, kw, ci) with ci contiguous.
struct ConvRows {
    int64_t img_base[4];  // element offset of the image (img * H * W * C), or -1 if the row is out of range
    int ih0[4], iw0[4];   // logical top-left input coordinate (oh*stride - pad, ow*stride - pad)
    int64_t pix_off[4];   // plain modes: element offset of (img, ih0, iw0, 0) -- may lie before the image for halo rows
    unsigned tapmask[4];  // plain modes: bit (kh*KW + kw) set when that tap reads inside the image
};
template <int T>
__device__ __forceinline__ void conv_rows_init(ConvRows& r, const ConvGeom& g, int64_t row0, int64_t M, int tid) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int64_t m = row0 + (tid >> 3) + (T / 4) * p;
        if (m < M) {
            const int64_t hw = (int64_t)g.OH * g.OW;
            const int64_t img = m / hw;
            const int rem = (int)(m - img * hw);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            r.img_base[p] = img * (int64_t)g.H * g.W * g.C;
            r.ih0[p] = oh * g.stride - g.pad;
            r.iw0[p] = ow * g.stride - g.pad;
            r.pix_off[p] = r.img_base[p] + ((int64_t)r.ih0[p] * g.W + r.iw0[p]) * g.C;
            unsigned mk = 0;
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw) {
                    const int ih = r.ih0[p] + kh, iw = r.iw0[p] + kw;
                    if (ih >= 0 && ih < g.H && iw >= 0 && iw < g.W) mk |= 1u << (kh * g.KW + kw);
                }
            r.tapmask[p] = mk;
        } else {
            r.img_base[p] = -1;
            r.ih0[p] = 0;
            r.iw0[p] = 0;
            r.pix_off[p] = 0;
            r.tapmask[p] = 0;
        }
    }
}
template <int T>
__device__ __forceinline__ void gload_conv(Stage& s, const bf16* base, const ConvGeom& g, const ConvRows& r, int64_t k0,
                                           int64_t K, int tid) {
    // 32-bit index math only (K = KH*KW*C < 2^31): one unsigned division per thread per K tile, none when the tile start is
    // uniform and C >= 64 is a multiple of 8 (the chunk wraps into the next tap at most once)
    const unsigned k0u = (unsigned)k0, C = (unsigned)g.C;
    const unsigned k = k0u + (unsigned)(tid & 7) * 8u;
    const bool kvalid = k < (unsigned)K;
    unsigned tap = k0u / C;  // uniform across the block: scalarised by the compiler
    unsigned ci = k0u - tap * C + (unsigned)(tid & 7) * 8u;
    if (C >= 64u) {
        if (ci >= C) {
            ci -= C;
            ++tap;
        }
    } else {
        tap = k / C;
        ci = k - tap * C;
    }
    const int kh = (int)(tap / (unsigned)g.KW), kw = (int)tap - kh * g.KW;
    const bool plain = !(g.up_shift | g.even_only);
    if (plain) {
        // unconditional loads from a always-valid address + select: no exec-mask juggling around the 4 gathers
        const int64_t tapoff = (int64_t)((kh * g.W + kw) * g.C + (int)ci);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const bool ok = kvalid && ((r.tapmask[p] >> tap) & 1u);
            const bf16* src = base + (ok ? r.pix_off[p] + tapoff : (int64_t)0);
            const bf16x8 v = ld_bf16x8(src);
            s.v[p] = ok ? v : zero_bf16x8();
        }
        return;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        int ih = r.ih0[p] + kh, iw = r.iw0[p] + kw;
        bool ok = kvalid && r.img_base[p] >= 0 && ih >= 0 && iw >= 0;
        if (!plain) {
            if (g.even_only) ok = ok && ((ih & 1) == 0) && ((iw & 1) == 0);
            ih >>= 1;
            iw >>= 1;
        }
        ok = ok && ih < g.H && iw < g.W;
        if (ok)
            s.v[p] = ld_bf16x8(base + r.img_base[p] + (int64_t)((ih * g.W + iw) * g.C + (int)ci));
        else
            s.v[p] = zero_bf16x8();
    }
}


// ---- kernel -------------------------------------------------------------------------------------------------------
template <int AL, int BL, int T>
__global__ __launch_bounds__(2 * T, 2) void gemm_bf16_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = T, BN = T;
    constexpr int TILE_BYTES = T * BK * 2;   // one operand tile: 16 KiB (T=128) / 32 KiB (T=256)
    constexpr int STAGE = 2 * TILE_BYTES;    // [buf][A | B]
    constexpr int WC = T / 64;               // wave columns (each wave: T/2 rows x 64 cols)
    constexpr int MI = T / 32;               // 16-row MFMA sub-tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / WC) * (T / 2), wn = (wave % WC) * 64;

    // XCD-aware, grouped tile order
    const int num_pid_m = (int)((P.M + BM - 1) / BM), num_pid_n = (int)((P.N + BN - 1) / BN);
    const int nwg = num_pid_m * num_pid_n;
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int in_group = GROUP_M * num_pid_n;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(num_pid_m - first_m, GROUP_M);
    const int pid_m = first_m + (wgid % in_group) % gsz;
    const int pid_n = (wgid % in_group) / gsz;
    const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stage sa, sb;
    ConvRows crow;
    if constexpr (AL == A_CONV) conv_rows_init<T>(crow, P.cv, m0, P.M, tid);

    auto gload = [&](int64_t k0) {
        if constexpr (AL == A_K)
            gload_kc<T>(sa, P.A, P.lda, m0, P.M, k0, P.K, tid);
        else if constexpr (AL == A_M)
            gload_mc<T>(sa, P.A, P.lda, m0, P.M, k0, P.K, tid);
        else
            gload_conv<T>(sa, P.A, P.cv, crow, k0, P.K, tid);
        if constexpr (BL == B_K)
            gload_kc<T>(sb, P.B, P.ldb, n0, P.N, k0, P.K, tid);
        else
            gload_mc<T>(sb, P.B, P.ldb, n0, P.N, k0, P.K, tid);
    };
    auto lstore = [&](int buf) {
        char* ta = smem + buf * STAGE;
        char* tb = ta + TILE_BYTES;
        if constexpr (AL == A_M)
            lstore_mc<T>(sa, ta, tid);
        else
            lstore_kc<T>(sa, ta, tid);
        if constexpr (BL == B_K)
            lstore_kc<T>(sb, tb, tid);
        else
            lstore_mc<T>(sb, tb, tid);
    };

    int t_begin = 0, nt = (int)((P.K + BK - 1) / BK);
    if (P.splitk > 1) {
        const int per = (nt + P.splitk - 1) / P.splitk;
        t_begin = blockIdx.y * per;
        nt = max(0, min(nt - t_begin, per));
    }
    if (nt > 0) {
        gload((int64_t)t_begin * BK);
        lstore(0);
    }
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) gload((int64_t)(t_begin + t + 1) * BK);
        const char* ta = smem + (t & 1) * STAGE;
        const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fb[4];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if constexpr (AL == A_M)
                    fa[i] = frag_mc<T>(ta, wm + i * 16, kk, lane);
                else
                    fa[i] = frag_kc(ta, wm + i * 16 + (lane & 15), kk, lane);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (BL == B_K)
                    fb[j] = frag_kc(tb, wn + j * 16 + (lane & 15), kk, lane);
                else
                    fb[j] = frag_mc<T>(tb, wn + j * 16, kk, lane);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) lstore((t + 1) & 1);
        __syncthreads();
    }

    if (epilogue_lds_ok(P, m0, n0, T))  // the loop ended on a barrier: the stages are free to serve as per-wave staging regions
        gemm_epilogue_lds<MI>(P, acc, smem + wave * 8192, m0 + wm, n0 + wn, lane);
    else
        gemm_epilogue<MI>(P, acc, m0 + wm, n0 + wn, lane, blockIdx.y, pid_m * num_pid_n + pid_n, reinterpret_cast<int*>(smem));
}


// ---- direct-to-LDS variant (T = 256, K % 64 == 0, dense operands) -------------------------------------------------
// Same tile geometry and LDS images as gemm_bf16_kernel<.., 256>, but tiles arrive through global_load_lds_dwordx4
// (LDS-DMA: no staging VGPRs, no ds_write pass).  The DMA writes wave-uniform-base + lane*16, i.e. LDS stays linear, so
// the XOR swizzles of the images are applied to the per-lane SOURCE address (within a 128/256-byte segment: coalescing is
// unaffected).  Tile t+1 is issued before the MFMAs of tile t into the other buffer; one vmcnt(0)+barrier per K tile.
// Rows/columns beyond M/N are clamped (they only feed outputs that are never stored); K must be a multiple of 64.

__device__ __forceinline__ void glds_kc_tile(const bf16* base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile,
                                             int wave, int lane) {
    // 256 rows x 128 B: 32 groups of 8 rows (1 KiB); wave w issues groups 4w .. 4w+3
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int grp = wave * 4 + p;
        const int r = grp * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int64_t row = row0 + r;
        row = row < nrows ? row : nrows - 1;
        GLDS16(base + row * ld + k0 + c * 8, tile + grp * 1024);
    }
}
__device__ __forceinline__ void glds_mc_tile(const bf16* base, int64_t ld, int64_t col0, int64_t ncols, int64_t k0, char* tile,
                                             int wave, int lane) {
    // 64 k rows x 512 B: 32 groups of 2 k rows (1 KiB)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int grp = wave * 4 + p;
        const int krow = grp * 2 + (lane >> 5);
        const int pos = (lane & 31) * 16;                       // byte position inside the 512-B row image
        const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
        const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);  // logical byte held at that position
        int64_t col = col0 + (lbyte >> 1);
        col = col < ncols ? col : ncols - 8;
        GLDS16(base + (k0 + krow) * ld + col, tile + grp * 1024);
    }
}

// group `grp` (8 rows x 128 B) of a k-contiguous tile, for tiles that are not 4 groups per wave
__device__ __forceinline__ void glds_kc_grp(const bf16* base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile,
                                            int grp, int lane) {
    const int r = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t row = row0 + r;
    row = row < nrows ? row : nrows - 1;
    GLDS16(base + row * ld + k0 + c * 8, tile + grp * 1024);
}

// one 1-KiB group (p-th of this wave's four) of the tiles above, for kernels that spread the DMA issue over the k loop
__device__ __forceinline__ void glds_kc_one(const bf16* base, int64_t ld, int64_t row0, int64_t nrows, int64_t k0, char* tile,
                                            int wave, int lane, int p) {
    const int grp = wave * 4 + p;
    const int r = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int64_t row = row0 + r;
    row = row < nrows ? row : nrows - 1;
    GLDS16(base + row * ld + k0 + c * 8, tile + grp * 1024);
}
__device__ __forceinline__ void glds_mc_one(const bf16* base, int64_t ld, int64_t col0, int64_t ncols, int64_t k0, char* tile,
                                            int wave, int lane, int p) {
    const int grp = wave * 4 + p;
    const int krow = grp * 2 + (lane >> 5);
    const int pos = (lane & 31) * 16;
    const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
    const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);
    int64_t col = col0 + (lbyte >> 1);
    col = col < ncols ? col : ncols - 8;
    GLDS16(base + (k0 + krow) * ld + col, tile + grp * 1024);
}

// Transpose reads issued as inline asm: with LDS-DMA in flight hipcc puts a vmcnt(0) in front of every
// __builtin_amdgcn_ds_read_tr16_b64 (it cannot prove the read does not alias the DMA), which drains the prefetch of the
// next tile right after it is issued.  The asm form is invisible to that pass; its completion is waited for explicitly
// (tr_wait: lgkmcnt(0) naming every destination, so no consumer is scheduled above it).
__device__ __forceinline__ void tr_read_asm(u32x2& dst, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr));
}
template <int T>
__device__ __forceinline__ void frag_mc_issue(u32x2& lo, u32x2& hi, const char* tile, int mbase, int kk, int lane) {
    const int g = lane >> 4, t = lane & 15;
    const int k0 = kk * 32 + g * 8 + (t >> 2);
    const int bcol = mbase * 2 + (t & 3) * 8;
    tr_read_asm(lo, lds_addr(tile + mc_off<T>(k0, bcol)));
    tr_read_asm(hi, lds_addr(tile + mc_off<T>(k0 + 4, bcol)));
}
template <int N>
__device__ __forceinline__ void tr_wait(u32x2 (&lo)[N], u32x2 (&hi)[N]) {
    if constexpr (N == 8)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(lo[4]), "+v"(lo[5]), "+v"(lo[6]), "+v"(lo[7]),
                       "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3]), "+v"(hi[4]), "+v"(hi[5]), "+v"(hi[6]), "+v"(hi[7])
                     :: "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(lo[0]), "+v"(lo[1]), "+v"(lo[2]), "+v"(lo[3]), "+v"(hi[0]), "+v"(hi[1]), "+v"(hi[2]), "+v"(hi[3])
                     :: "memory");
}

template <int AL, int BL>
__global__ __launch_bounds__(512, 2) void gemm_glds_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 256, BM = T, BN = T;
    constexpr int TILE_BYTES = T * BK * 2, STAGE = 2 * TILE_BYTES;
    constexpr int WC = 4, MI = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WC) * (T / 2), wn = (wave % WC) * 64;

    const int num_pid_m = (int)((P.M + BM - 1) / BM), num_pid_n = (int)((P.N + BN - 1) / BN);
    const int nwg = num_pid_m * num_pid_n;
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    constexpr int GROUP_M = 8;
    const int in_group = GROUP_M * num_pid_n;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(num_pid_m - first_m, GROUP_M);
    const int pid_m = first_m + (wgid % in_group) % gsz;
    const int pid_n = (wgid % in_group) / gsz;
    const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int64_t k0, int buf) {
        char* ta = smem + buf * STAGE;
        char* tb = ta + TILE_BYTES;
        if constexpr (AL == A_K)
            glds_kc_tile(P.A, P.lda, m0, P.M, k0, ta, wave, lane);
        else
            glds_mc_tile(P.A, P.lda, m0, P.M, k0, ta, wave, lane);
        if constexpr (BL == B_K)
            glds_kc_tile(P.B, P.ldb, n0, P.N, k0, tb, wave, lane);
        else
            glds_mc_tile(P.B, P.ldb, n0, P.N, k0, tb, wave, lane);
    };

    const int nt = (int)(P.K / BK);
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt && !P.dbg_noload) issue((int64_t)(t + 1) * BK, (t + 1) & 1);
        const char* ta = smem + (t & 1) * STAGE;
        const char* tb = ta + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 fa[MI], fb[4];
            u32x2 alo[MI], ahi[MI], blo[4], bhi[4];
            if constexpr (AL == A_M) {
#pragma unroll
                for (int i = 0; i < MI; ++i) frag_mc_issue<T>(alo[i], ahi[i], ta, wm + i * 16, kk, lane);
            } else {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = frag_kc(ta, wm + i * 16 + (lane & 15), kk, lane);
            }
            if constexpr (BL == B_N) {
#pragma unroll
                for (int j = 0; j < 4; ++j) frag_mc_issue<T>(blo[j], bhi[j], tb, wn + j * 16, kk, lane);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = frag_kc(tb, wn + j * 16 + (lane & 15), kk, lane);
            }
            if constexpr (AL == A_M) {
                tr_wait<MI>(alo, ahi);
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[i] = join_frag(alo[i], ahi[i]);
            }
            if constexpr (BL == B_N) {
                tr_wait<4>(blo, bhi);
#pragma unroll
                for (int j = 0; j < 4; ++j) fb[j] = join_frag(blo[j], bhi[j]);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    gemm_epilogue<MI>(P, acc, m0 + wm, n0 + wn, lane);
}

// ---- software-pipelined direct-to-LDS variant ------------------------------------------------------------------------
// Same tiles, LDS images and DMA as gemm_glds_kernel; what changes is the order inside a wave.  The plain kernel reads all
// 12 fragments of a 32-deep k step, waits, then issues 32 MFMAs: after every barrier both waves of a SIMD sit in that wait
// together and the matrix pipe idles (PMC: SQ_WAIT_ANY 46 % of wave time, MFMA busy 49 %).  Here a k step is 8 groups of
// 4 MFMAs (one A fragment x the 4 B fragments); the A fragment of group g+1 (and, at a k-step boundary, the next 4 B
// fragments) is requested before group g's MFMAs, so each wait is for data requested >= 64 matrix cycles earlier.  The
// tile barrier sits before the LAST group of a tile: the wave has consumed the tile from LDS, crosses the barrier, requests
// the first fragments of the next tile, and only then issues the last 4 MFMAs, which cover that LDS latency.
// Every LDS read is inline asm with counted s_waitcnt lgkmcnt(n) (LDS returns in order), for two reasons: hipcc would
// otherwise drain vmcnt before any LDS read it can see while LDS-DMA is in flight, and its scheduler clusters reads.

// ---- implicit-GEMM gather through LDS-DMA (3x3 / 1x1 NHWC convolutions on the pipelined kernel) ----------------------
// global_load_lds takes a per-lane source address, so the im2col gather costs nothing extra: lane (row r of the DMA group,
// 16-byte chunk c) reads channels ci..ci+7 of tap (kh, kw) of output pixel m0 + r.  Taps that fall outside the image (and
// rows past M) read a 16-byte zero page instead.  Requires C % 64 == 0 (a 64-deep K tile never straddles two taps): every
// SD-2.1 / SDXL UNet and VAE conv except conv_in.  Round 3: the fused nearest-2x upsample and the transposed stride-2 gather
// (physical pixel = logical >> 1, parity mask for the zero-stuffed grid) ride on the same per-lane source address.

struct ConvDma {
    int64_t pix_off[4];   // element offset of (img, oh*stride - pad, ow*stride - pad, 0) for the lane's row in each A group;
                          // shift modes (fused nearest-2x upsample / transposed stride-2 gather): offset of the image only
    unsigned tapmask[4];  // bit (kh*KW + kw): that tap reads inside the image (shift modes: and, for `even_only`, an even position)
    unsigned org[4];      // shift modes: (ih0 + 2) << 16 | (iw0 + 2), the row's LOGICAL top-left input coordinate (>= -1), biased
};
template <bool SHIFT>
__device__ __forceinline__ void conv_dma_init(ConvDma& d, const ConvGeom& g, int64_t m0, int64_t M, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t m = m0 + (wave * 4 + q) * 8 + (lane >> 3);
        d.pix_off[q] = 0;
        d.tapmask[q] = 0;
        d.org[q] = 0;
        if (m < M) {
            const int64_t hw = (int64_t)g.OH * g.OW;
            const int64_t img = m / hw;
            const int rem = (int)(m - img * hw);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            constexpr bool shift = SHIFT;   // a separate instantiation: the plain gather keeps its round-2 instruction stream
            d.pix_off[q] = img * (int64_t)g.H * g.W * g.C + (shift ? (int64_t)0 : ((int64_t)ih0 * g.W + iw0) * g.C);
            if constexpr (SHIFT) d.org[q] = ((unsigned)(ih0 + 2) << 16) | (unsigned)(iw0 + 2);
            unsigned mk = 0;
            for (int kh = 0; kh < g.KH; ++kh)
                for (int kw = 0; kw < g.KW; ++kw) {
                    int ih = ih0 + kh, iw = iw0 + kw;
                    bool ok = ih >= 0 && iw >= 0;
                    if (shift) {   // logical grid = 2x the physical one: nearest-2x upsample, or the zero-stuffed grid of a stride-2 dgrad
                        if (g.even_only) ok = ok && ((ih & 1) == 0) && ((iw & 1) == 0);
                        ih >>= 1;
                        iw >>= 1;
                    }
                    if (ok && ih < g.H && iw < g.W) mk |= 1u << (kh * g.KW + kw);
                }
            d.tapmask[q] = mk;
        }
    }
}
// q-th A group of the wave for the K tile that starts at channel ci0 of tap `tap` (both uniform)
template <bool SHIFT>
__device__ __forceinline__ void glds_conv_one(const bf16* x, const ConvGeom& g, const ConvDma& d, int tap, int ci0, char* tile,
                                              int wave, int lane, int q) {
    const int grp = wave * 4 + q;
    const int r = grp * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    const bool ok = (d.tapmask[q] >> tap) & 1u;
    const bf16* src;
    if constexpr (SHIFT) {   // physical pixel = logical >> 1
        const int ih = ((int)(d.org[q] >> 16) - 2 + kh) >> 1, iw = ((int)(d.org[q] & 0xffffu) - 2 + kw) >> 1;
        src = ok ? x + d.pix_off[q] + (int64_t)((ih * g.W + iw) * g.C + ci0 + c * 8) : g_zero_page;
    } else {
        src = ok ? x + d.pix_off[q] + (int64_t)((kh * g.W + kw) * g.C + ci0 + c * 8) : g_zero_page;
    }
    GLDS16(src, tile + grp * 1024);
}

// BN_ = 256: 2 x 4 waves of 128 x 64; BN_ = 128 (k-contiguous B only): 4 x 2 waves of 64 x 64 -- a 256 x 128 block tile for narrow
// outputs (the 320-channel UNet layers are 3 x 128 = 83 % full instead of 2 x 256 = 62 %), same wave-level stream.
//
// pipe_tile: the K loop of ONE output tile over the K tiles [kt0, kt1) (kt1 > kt0), accumulating into `acc`.  The whole-tile
// kernel calls it with [0, K / 64); the stream-K tail kernel with a slice of a tile's K loop.
// LDS operations issued AFTER the read of A fragment g inside a K tile, with a lookahead of `la` groups: A(g + 1 .. g + la) and the four B
// fragments that go out in front of the first A fragment of a k step
constexpr int pipe_younger_ops(int g, int la, int ng, int mi, int oa, int ob) {
    int n = 0;
    for (int j = 1; j <= la; ++j)
        if (g + j < ng) n += oa + ((g + j) % mi == 0 ? 4 * ob : 0);
    return n;
}

template <int BN_>
struct PipeGeom {
    static constexpr int WC = BN_ / 64, MI = (256 / (8 / WC)) / 16;  // wave grid (8 / WC) x WC, MI 16-row MFMA tiles per wave
};

template <int AL, int BL, int BN_>
__device__ __forceinline__ void pipe_tile(const GemmParams& P, char* smem, int64_t m0, int64_t n0, int kt0, int kt1,
                                          f32x4 (&acc)[PipeGeom<BN_>::MI][4]) {
    constexpr int T = 256, BM = T, BN = BN_;
    static_assert(BN == 256 || (BN == 128 && BL == B_K), "the 128-wide tile reads a k-contiguous B image");
    constexpr int TILE_BYTES = BM * BK * 2, TILE_B_BYTES = BN * BK * 2, STAGE = TILE_BYTES + TILE_B_BYTES;
    constexpr int WC = PipeGeom<BN_>::WC, MI = PipeGeom<BN_>::MI;
    constexpr int NBD = BN / 64;                            // DMA instructions per wave for the B tile (A: 4)
    constexpr int NG = 2 * MI;                              // MFMA groups per K tile
    constexpr bool AMC = (AL == A_M), BMC = (BL == B_N);
    constexpr int OA = AMC ? 2 : 1, OB = BMC ? 2 : 1;  // LDS operations per fragment
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WC) * (MI * 16), wn = (wave % WC) * 64;

    constexpr bool CONV = (AL == A_CONV || AL == A_CONVS), SHIFT = (AL == A_CONVS);
    ConvDma cdma;
    if constexpr (CONV) conv_dma_init<SHIFT>(cdma, P.cv, m0, P.M, wave, lane);
    // conv: tap / first channel of the K tile being prefetched (advanced incrementally: no division inside the loop)
    int ctap = 0, cci = 0;
    if constexpr (CONV) {
        if (kt0 != 0) {
            ctap = (int)(((int64_t)kt0 * BK) / P.cv.C);
            cci = (int)(((int64_t)kt0 * BK) % P.cv.C);
        }
    }

    // ---- LDS-DMA requests of the dense operands: buffer_load_dwordx4 ... offen lds (round 6).  The request carries a wave-uniform
    // buffer descriptor (the operand's origin for the K tile, advanced by SALU once per tile) and ONE 32-bit per-lane byte offset that never
    // changes: no 64-bit VALU address arithmetic per request, half the address VGPRs read at issue.  tools/dma_issue_probe.hip (8 waves,
    // 8 MFMAs per request, L2-resident panels shared by an XCD's blocks as in a GEMM cluster): 303 clk per iteration against 337 for
    // global_load_lds_dwordx4 with 64-bit per-lane addresses and 308 for no request at all (profiles/r06_dma_issue_probe.log).
    // Offsets: A_K / B_K image = rows of the tile x 128 bytes of k (descriptor at (row0, k0), offset (r * ld + 8 c) * 2, r < 256);
    // A_M / B_N image = 64 k rows x the tile's 512 bytes (descriptor at (k0, 0), offset (krow * ld + col) * 2).  Rows / columns past
    // the edge are clamped as before (they only feed outputs that are never stored).
    constexpr bool BUFA = (AL == A_K || AL == A_M);
    uint32_t voA[4], voB[NBD];
    uint64_t curA = 0, curB = 0, stepA = 0, stepB = 0;   // byte address of the descriptor origin for the NEXT K tile to request; per-tile advance
    if constexpr (BUFA) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int grp = wave * 4 + q;
            if constexpr (AL == A_K) {
                const int r = grp * 8 + (lane >> 3);
                const int c = (lane & 7) ^ ((r >> 1) & 7);
                const int64_t rr = min((int64_t)r, P.M - 1 - m0);
                voA[q] = (uint32_t)((rr * P.lda + c * 8) * 2);
            } else {
                const int krow = grp * 2 + (lane >> 5);
                const int pos = (lane & 31) * 16;
                const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
                const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);
                const int64_t col = min(m0 + (lbyte >> 1), P.M - 8);
                voA[q] = (uint32_t)((krow * P.lda + col) * 2);
            }
        }
        curA = (uint64_t)(uintptr_t)P.A + (uint64_t)((AL == A_K ? m0 * P.lda + (int64_t)kt0 * BK : (int64_t)kt0 * BK * P.lda) * 2);
        stepA = (uint64_t)((AL == A_K ? (int64_t)BK : (int64_t)BK * P.lda) * 2);
    }
    // EPI_SWIGLU_FWD: the B tile's 256 rows are re-mapped so that every wave holds 32 gate columns and the 32 up columns of the SAME
    // outputs (tile row r of wave column block r >> 6: rows 0-31 = gate rows, 32-63 = the matching up rows F further down the packed
    // weight); block column pid_n then covers the outputs [n0 / 2, n0 / 2 + 128).  Free on the DMA's per-lane offset.
    bool glu_map = false, rope_map = false;
    if constexpr (AL == A_K && BL == B_K && BN == 256) {
        glu_map = P.epi == EPI_SWIGLU_FWD;
        // EPI_ROPE_QKV (head_dim 128: a tile = two heads): wave column block wc holds dims [32 (wc & 1), +32) of head wc >> 1 in its first 32
        // columns and the partner dims 64 further in the other 32 -- rotate_half's pairs meet in one lane
        rope_map = P.epi == EPI_ROPE_QKV && n0 < P.rope_cols;
    }
#pragma unroll
    for (int q = 0; q < NBD; ++q) {
        if constexpr (BL == B_K) {
            const int grp = (BN == 128 ? wave * NBD : wave * 4) + q;
            const int r = grp * 8 + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            int64_t rr = min((int64_t)r, P.N - 1 - n0);
            if (glu_map) rr = ((r & 32) ? P.glu_F : 0) + (r >> 6) * 32 + (r & 31);   // relative to weight row n0 / 2 (the descriptor origin below)
            if (rope_map) rr = (r >> 7) * 128 + ((r >> 6) & 1) * 32 + (r & 31) + ((r & 32) ? 64 : 0);
            voB[q] = (uint32_t)((rr * P.ldb + c * 8) * 2);
        } else {
            const int grp = wave * 4 + q;
            const int krow = grp * 2 + (lane >> 5);
            const int pos = (lane & 31) * 16;
            const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
            const int lbyte = ((((pos >> 5) ^ f)) << 5) + (pos & 31);
            const int64_t col = min(n0 + (lbyte >> 1), P.N - 8);
            voB[q] = (uint32_t)((krow * P.ldb + col) * 2);
        }
    }
    curB = (uint64_t)(uintptr_t)P.B + (uint64_t)((BL == B_K ? (glu_map ? (n0 >> 1) : n0) * P.ldb + (int64_t)kt0 * BK : (int64_t)kt0 * BK * P.ldb) * 2);
    stepB = (uint64_t)((BL == B_K ? (int64_t)BK : (int64_t)BK * P.ldb) * 2);
    auto dma_buf = [&](uint64_t origin, uint32_t vo, char* dst) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)origin, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, 0, 0, 0);
    };

    auto issue = [&](int64_t k0, int buf) {   // the whole first tile (prologue): k0 is the tile curA / curB point at
        char* ta = smem + buf * STAGE;
        char* tb = ta + TILE_BYTES;
        if constexpr (BUFA) {
#pragma unroll
            for (int q = 0; q < 4; ++q) dma_buf(curA, voA[q], ta + (wave * 4 + q) * 1024);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds_conv_one<SHIFT>(P.A, P.cv, cdma, (int)(k0 / P.cv.C), (int)(k0 % P.cv.C), ta, wave, lane, q);
        }
#pragma unroll
        for (int q = 0; q < NBD; ++q) dma_buf(curB, voB[q], tb + ((BN == 128 ? wave * NBD : wave * 4) + q) * 1024);
    };

    // per-lane byte offset of fragment (idx 0, k step 0) inside an operand tile
    const int lg = lane >> 4, lt = lane & 15;
    const uint32_t s0 = lds_addr(smem);
    const uint32_t offA = AMC ? (uint32_t)mc_off<T>(lg * 8 + (lt >> 2), wm * 2 + (lt & 3) * 8) : (uint32_t)kc_off(wm + lt, lg);
    const uint32_t offB =
        (uint32_t)TILE_BYTES + (BMC ? (uint32_t)mc_off<T>(lg * 8 + (lt >> 2), wn * 2 + (lt & 3) * 8) : (uint32_t)kc_off(wn + lt, lg));

    // A fragments are requested LA groups ahead of their MFMAs (round 6: 2; rounds 1-5: 1 = 64 matrix cycles, less than an LDS round trip
    // under load) into a ring of 4 (NG % 4 == 0: the slots of a tile's last groups never collide with the first reads of the next tile,
    // which are issued in front of the last group's MFMAs); the B fragments of a k step go out in front of the first A fragment of that step.
    constexpr int LA = 2;
    static_assert(NG % 4 == 0 && LA <= 2, "ring of 4 A-fragment slots");
    FragR<AMC> fa[4];
    FragR<BMC> fb[2][4];
    uint32_t ab = s0 + offA, bb = s0 + offB;
    auto first_reads = [&]() {
        static_for<0, 4>([&](auto j) { fragr_issue<BMC, decltype(j)::value, 0>(fb[0][decltype(j)::value], bb); });
        static_for<0, LA>([&](auto g0) { fragr_issue<AMC, decltype(g0)::value % MI, decltype(g0)::value / MI>(fa[decltype(g0)::value], ab); });
    };

    // q-th of the wave's 8 DMA instructions for the K tile curA / curB point at (4 per operand)
    auto issue_one = [&](int buf, int q, const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rb) {
        char* ta = smem + buf * STAGE;
        char* tb = ta + TILE_BYTES;
        if (q < 4) {
            if constexpr (BUFA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(ta + (wave * 4 + q) * 1024), 16, (int)voA[q], 0, 0, 0);
            else
                glds_conv_one<SHIFT>(P.A, P.cv, cdma, ctap, cci, ta, wave, lane, q);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(tb + ((BN == 128 ? wave * NBD : wave * 4) + (q - 4)) * 1024), 16,
                                                     (int)voB[q - 4], 0, 0, 0);
        }
    };

    issue((int64_t)kt0 * BK, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    first_reads();

    // The body of K tile t.  NEXT = a tile t + 1 follows: its requests ride on the first 4 + NBD MFMA groups and the tile barrier + the first
    // fragment reads of tile t + 1 sit in front of the last group.  The last tile of the range runs the NEXT = false copy, so the loop itself
    // carries no per-request branch (round 6; the shipped loop tested `t + 1 < kt1` in front of each of the 8 requests).
    __amdgpu_buffer_rsrc_t rsA, rsB;
    auto body = [&](int t, auto next_c) {
        constexpr bool NEXT = decltype(next_c)::value;
        // tile t+1 goes into the buffer tile t-1 was read from: every wave finished those reads before the last barrier
#ifdef DLLM_BENCH_MODES
        const bool pf = NEXT && P.dbg_noload != 1;
        if (NEXT && P.dbg_noload != 2) {   // (bench mode 2: every prefetch re-reads the first K tile)
            curA += stepA;
            curB += stepB;
        }
#else
        constexpr bool pf = NEXT;
        if constexpr (NEXT) {
            curA += stepA;
            curB += stepB;
        }
#endif
        if constexpr (NEXT) {   // the two descriptors of tile t + 1: built once per tile (wave-uniform SALU), shared by its 4 + NBD requests
            if constexpr (BUFA) rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curA, 0, 0x7fffffff, 0x00020000);
            rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)curB, 0, 0x7fffffff, 0x00020000);
        }
        const int nbuf = (t - kt0 + 1) & 1;
        if constexpr (CONV && NEXT) {  // K tile t+1 starts BK channels further; wraps into the next tap at C
            cci += BK;
            if (cci >= P.cv.C) {
                cci -= P.cv.C;
                ++ctap;
            }
        }
        static_for<0, NG>([&](auto gc) {
            constexpr int g = decltype(gc)::value, kk = g / MI, i = g % MI;
            // one DMA instruction per group from the start of the tile: no 64-KiB burst per CU.  (Round 6, with the cheaper buffer form: two per
            // group in the first four groups measured -6 % on the weight gradients, -1...-3 % dgrad, -4...+0.7 % forward:
            // profiles/r06_gemm_rpg2_ab.log.  An L2 "touch" of the lines two K tiles ahead -- one dword per lane and tile behind the
            // last request, counted vmcnt(1) at the barrier -- measured -0.2 ... 0.0 % on all twelve shapes: first-touch L2 misses are not
            // what the tile barrier waits for; profiles/r06_gemm_l2touch_ab.log, profiles/patches/r06_gemm_l2_touch.patch)
            if constexpr (NEXT && g < 4 + NBD) {
                if (pf) issue_one(nbuf, g, rsA, rsB);
            }
            if constexpr (g + LA < NG) {   // request A(g + LA) (and, in front of the first A fragment of a k step, that step's B fragments)
                constexpr int kn = (g + LA) / MI, in = (g + LA) % MI;
                if constexpr (in == 0)
                    static_for<0, 4>([&](auto j) { fragr_issue<BMC, decltype(j)::value, kn>(fb[kn][decltype(j)::value], bb); });
                fragr_issue<AMC, in, kn>(fa[(g + LA) & 3], ab);
            }
            // LDS returns in order: A(g) has landed once at most the operations issued after it are outstanding
            constexpr int younger = pipe_younger_ops(g, LA, NG, MI, OA, OB);
            if constexpr (g < NG - 1) {
                fragr_wait<younger>(fa[g & 3]);
            } else {
                fragr_wait<0>(fa[g & 3]);
                if constexpr (NEXT) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    ab = s0 + offA + (uint32_t)(nbuf * STAGE);
                    bb = s0 + offB + (uint32_t)(nbuf * STAGE);
                    first_reads();
                }
            }
            if constexpr (i == 0) static_for<0, 4>([&](auto j) { fragr_touch(fb[kk][decltype(j)::value]); });
            const bf16x8 va = fragr_value(fa[g & 3]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fragr_value(fb[kk][j]), va, acc[i][j], 0, 0, 0);
            // Round 5 (profiles/r05_gemm_prio_ab.log): the groups that carry the tile's LDS-DMA requests run at priority 1, the rest at 0 --
            // both waves of a SIMD walk the K tile in phase and the older one wins every arbitration, so without this the younger wave's
            // requests for tile t + 1 queue behind the older wave's MFMAs.  Forward layout (both operands k-contiguous): +0.4 ... +3.7 %
            // sustained (packed gate|up 1319 -> 1355 TF); the dgrad / wgrad layouts lose 0.5-1 % with it and keep equal priorities.
            // (the implicit-GEMM conv layouts take it too: B_img 8 denoise loop 43.14 -> 43.32 steps/s in two interleaved rounds,
            // profiles/r05_denoise_prio_ab.log; on the ring kernel both placements tried there measured equal or slower)
            constexpr bool kPrioLayout = (AL == A_K || AL == A_CONV || AL == A_CONVS) && BL == B_K;
            if constexpr (kPrioLayout && NEXT) {   // (the seven other placements measured: profiles/patches/r05_gemm_prio_variants.patch)
                if constexpr (g + 1 < 4 + NBD) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    for (int t = kt0; t + 1 < kt1; ++t) body(t, std::true_type{});
    body(kt1 - 1, std::false_type{});
    __builtin_amdgcn_s_setprio(0);
}

// (the fused SwiGLU / RoPE epilogues and pipe_decode_tile live in gemm_epilogues.h: gemm_w4.hip shares them)

template <int AL, int BL, int BN_ = 256>
__global__ __launch_bounds__(512, 2) void gemm_pipe_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 256, BN = BN_;
    constexpr int WC = PipeGeom<BN_>::WC, MI = PipeGeom<BN_>::MI;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / WC) * (MI * 16), wn = (wave % WC) * 64;

    const int num_pid_m = (int)((P.M + BM - 1) / BM), num_pid_n = (int)((P.N + BN - 1) / BN);
    // stream-K tail (sk_full > 0): this launch covers the first sk_full tiles of the grouped order (whole rounds of 256 CUs); the
    // remaining tiles' K loops are spread evenly over the CUs by gemm_pipe_tail_kernel
    const int nwg = P.sk_full > 0 ? P.sk_full : num_pid_m * num_pid_n;
    int wgid;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int pid_m, pid_n;
    pipe_decode_tile(P, wgid, num_pid_m, num_pid_n, pid_m, pid_n);
    const int64_t m0 = (int64_t)pid_m * BM, n0 = (int64_t)pid_n * BN;

    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    pipe_tile<AL, BL, BN_>(P, smem, m0, n0, 0, (int)(P.K / BK), acc);

    if constexpr (AL == A_K && BL == B_K && BN_ == 256) {   // fused RoPE epilogue of the q and k column tiles (the v tiles take the plain path below)
        if (P.epi == EPI_ROPE_QKV && n0 < P.rope_cols) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            gemm_epilogue_rope<MI>(P, acc, smem + wave * 8192, m0 + wm, n0, wn, lane);
            return;
        }
    }
    if constexpr (AL == A_K && BN_ == 256) {   // fused SwiGLU epilogues (full tiles by construction of their entry points)
        if (P.epi == (BL == B_K ? EPI_SWIGLU_FWD : EPI_SWIGLU_BWD)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // every wave is done with its fragment reads: the stages become per-wave staging regions
            if constexpr (BL == B_K)
                gemm_epilogue_swiglu_fwd<MI>(P, acc, smem + wave * 16384, m0 + wm, n0, wn, lane);
            else
                gemm_epilogue_swiglu_bwd<MI>(P, acc, smem + wave * 16384, m0 + wm, n0 + wn, lane);
            return;
        }
    }
    if (epilogue_lds_ok(P, m0, n0, BM, BN)) {
        // every wave must be done with its fragment reads before the stages are reused as per-wave staging regions
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        gemm_epilogue_lds<MI>(P, acc, smem + wave * 8192, m0 + wm, n0 + wn, lane);
    } else {
        gemm_epilogue<MI>(P, acc, m0 + wm, n0 + wn, lane);
    }
}

// ---- XCD-synchronised persistent walk ----------------------------------------------------------------------------------------
// One block per CU; the hardware deals block b to XCD b & 7.  Block (xcd, slot) walks tiles xcd_start + slot + 32 r of the grouped
// order, r = 0, 1, ... -- the same 32-tile clusters per XCD as the one-tile-per-block launch -- but the 32 blocks of an XCD start
// every tile TOGETHER (a counter in global memory, all 32 on the same L2): a cluster's tiles then move through K in lockstep and
// share their A / B panels through the XCD's 4 MiB L2 (32 tiles need 12 panels x 32 KiB per K step when aligned).  Without it the
// blocks of later rounds start whenever a CU frees up, drift apart by more than the ~10 K steps the L2 can bridge and re-fetch the
// panels from the fabric (PMC: 12.6 GB fetched for the packed gate|up forward against 8.5 GB for aligned clusters).  The barrier
// carries no data dependency: a block that waits "too long" (spin limit) simply goes on, so a missing co-resident block cannot hang it.
__device__ __forceinline__ void xcd_barrier(unsigned* counter, unsigned target) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; spin < (1 << 16); ++spin) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __syncthreads();
}

template <int AL, int BL>
__global__ __launch_bounds__(512, 2) void gemm_pipe_persist_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave / 4) * 128, wn = (wave % 4) * 64;
    const int num_pid_m = (int)((P.M + 255) / 256), num_pid_n = (int)((P.N + 255) / 256);
    const int nwg = P.sk_full > 0 ? P.sk_full : num_pid_m * num_pid_n;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;   // per = co-resident blocks per XCD (32)
    const int q = nwg >> 3, r8 = nwg & 7;
    const int start = xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q;
    const int count = q + (xcd < r8 ? 1 : 0);
    const int rounds = (count + per - 1) / per;
    unsigned* counter = reinterpret_cast<unsigned*>(P.ws) + xcd * 16;                 // one 64-byte line per XCD
    for (int r = 0; r < rounds; ++r) {
        if (r > 0) xcd_barrier(counter, (unsigned)(per * r));   // round 0: the launch itself started the blocks together
        const int it = r * per + slot;
        if (it >= count) continue;
        int pid_m, pid_n;
        pipe_decode_tile(P, start + it, num_pid_m, num_pid_n, pid_m, pid_n);
        const int64_t m0 = (int64_t)pid_m * 256, n0 = (int64_t)pid_n * 256;
        f32x4 acc[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        pipe_tile<AL, BL, 256>(P, smem, m0, n0, 0, (int)(P.K / BK), acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // fragment reads done: the stages become epilogue staging, then the next tile's first DMA target
        if (epilogue_lds_ok(P, m0, n0, 256, 256))
            gemm_epilogue_lds<MI>(P, acc, smem + wave * 8192, m0 + wm, n0 + wn, lane);
        else
            gemm_epilogue<MI>(P, acc, m0 + wm, n0 + wn, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave's staging reads are done before the next tile's DMA overwrites the stages
    }
    // leave the counter at zero for the next launch: every block arrives once more; the last arrival of the launch resets it
    if (rounds > 1 && tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(per * rounds) - 1u) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- stream-K tail of the pipelined kernel ---------------------------------------------------------------------------------
// One 256 x 256 block per CU means a grid runs in rounds of 256 tiles; the weight gradients of the MLP have 1376 (gate|up) and 688
// (down) tiles = 5.4 and 2.7 rounds, so their last round leaves 62 % / 31 % of the chip idle (10 % of those launches).  With a
// workspace the launch is split: the whole rounds run as usual (gemm_pipe_kernel over the first sk_full tiles of the grouped
// order); the K loops of the remaining sk_tail tiles are CONCATENATED into one iteration space of sk_tail * (K / 64) K tiles and
// cut into equal contiguous ranges of sk_w K tiles, one per CU (a range covers the end of one tile and the start of the next: at
// most two segments, because sk_w <= K / 64 when sk_tail <= 256).  Each segment's raw fp32 accumulators go to a workspace slab in
// the thread-linear layout [(i, j)][tid] (8 KiB contiguous per store instruction); gemm_pipe_fixup_kernel sums a tile's slabs in
// ascending K order (fixed order: deterministic, bit-identical run to run) and applies the ordinary epilogue.
constexpr int64_t SK_SLAB_FLOATS = 256 * 256;
constexpr int64_t SK_HEAD_FLOATS = 1024;  // first 4 KiB of the workspace: per-XCD counters of the persistent walk (zero between launches)
constexpr int64_t SK_WS_BYTES = (2 * 256 * SK_SLAB_FLOATS + SK_HEAD_FLOATS) * 4;  // two segments per tail block, 256 tail blocks: 128 MiB

template <int AL, int BL>
__global__ __launch_bounds__(512, 2) void gemm_pipe_tail_kernel(GemmParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MI = 8;
    const int tid = threadIdx.x;
    const int num_pid_m = (int)((P.M + 255) / 256), num_pid_n = (int)((P.N + 255) / 256);
    const int nt = (int)(P.K / BK);
    const int64_t total = (int64_t)P.sk_tail * nt;
    int64_t it0 = (int64_t)blockIdx.x * P.sk_w;
    const int64_t it1 = min(it0 + (int64_t)P.sk_w, total);
    for (int seg = 0; it0 < it1; ++seg) {
        const int j = (int)(it0 / nt);
        const int kt0 = (int)(it0 - (int64_t)j * nt);
        const int kt1 = (int)min((int64_t)nt, (int64_t)kt0 + (it1 - it0));
        int pid_m, pid_n;
        pipe_decode_tile(P, P.sk_full + j, num_pid_m, num_pid_n, pid_m, pid_n);
        f32x4 acc[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        pipe_tile<AL, BL, 256>(P, smem, (int64_t)pid_m * 256, (int64_t)pid_n * 256, kt0, kt1, acc);
        float* slab = P.ws + SK_HEAD_FLOATS + ((int64_t)blockIdx.x * 2 + seg) * SK_SLAB_FLOATS;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(slab + ((int64_t)(i * 4 + q) * 512 + tid) * 4) = acc[i][q];
        it0 += kt1 - kt0;
        // the next segment's first DMA overwrites the LDS stages: every wave must have finished its fragment reads
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
}

// one block per tail tile: sum the slabs of the tail blocks that covered its K loop, in K order, then the ordinary epilogue
__global__ __launch_bounds__(512) void gemm_pipe_fixup_kernel(GemmParams P) {
    constexpr int MI = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave / 4) * 128, wn = (wave % 4) * 64;
    const int num_pid_m = (int)((P.M + 255) / 256), num_pid_n = (int)((P.N + 255) / 256);
    const int nt = (int)(P.K / BK);
    const int j = blockIdx.x;
    int pid_m, pid_n;
    pipe_decode_tile(P, P.sk_full + j, num_pid_m, num_pid_n, pid_m, pid_n);
    const int64_t a = (int64_t)j * nt, b = a + nt - 1;
    const int u0 = (int)(a / P.sk_w), u1 = (int)(b / P.sk_w);
    f32x4 acc[MI][4];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int u = u0; u <= u1; ++u) {
        const int seg = ((int64_t)u * P.sk_w) / nt == j ? 0 : 1;  // the block's range starts in this tile (segment 0) or in the one before
        const float* slab = P.ws + SK_HEAD_FLOATS + ((int64_t)u * 2 + seg) * SK_SLAB_FLOATS;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][q] += *reinterpret_cast<const f32x4*>(slab + ((int64_t)(i * 4 + q) * 512 + tid) * 4);
    }
    gemm_epilogue<MI>(P, acc, (int64_t)pid_m * 256 + wm, (int64_t)pid_n * 256 + wn, lane);
}

// Stream-K tail plan for an [M, N, K] problem on the 256 x 256 pipelined kernel: whole rounds first, the rest evenly over the CUs.
// Returns false when there is nothing to gain.  Measured on MI355X (tools/streamk_ab.py, profiles/r03_streamk_ab.log): the tail
// blocks walk DIFFERENT K ranges of their tiles, so -- unlike the whole-tile rounds, where an XCD's 32 concurrent tiles share A / B
// panels at the same K position through its L2 -- every tail block streams its own 64 KiB per K tile, and a tail that still fills
// most of the chip is bandwidth-bound: the gate|up weight gradient (1376 tiles, remainder 96 = 37 % of a round, K = 32768) gains
// 8.6 %, the down weight gradient (688 tiles, remainder 176 = 69 %) LOSES 4 %, the down input gradient (remainder 128, K = 4096:
// the half round saved is 50 us, less than two extra launches and the slab traffic) loses 1 %.  Hence: remainder at most half a
// round, and the idle share of that round worth at least 160 K tiles (~0.26 ms) of one CU.
static inline bool streamk_plan(int64_t M, int64_t N, int64_t K, int& full, int& tail, int& w, int& blocks) {
    const int64_t tiles = cdiv64(M, 256) * cdiv64(N, 256);
    const int64_t nt = K / BK;
    const int64_t r = tiles % 256;
    if (tiles <= 256 || tiles > 0x3fffffff || r == 0 || r > 128 || (256 - r) * nt < 160 * 256) return false;
    full = (int)(tiles - r);
    tail = (int)r;
    const int64_t total = r * nt;
    w = (int)cdiv64(total, 256);
    blocks = (int)cdiv64(total, w);
    return true;
}

// Small grids with a deep reduction (the UNet's 1280-channel 3x3 convs at 16 x 16: M = 4096 at batch 16 = 80 tiles of 256 x 256
// with K = 11520 = 180 K tiles; a whole-tile launch fills 31 % of the chip, the 256 x 128 tiles 62 %): ALL tiles go through the
// stream-K machinery (sk_full = 0): 256 blocks take tiles * nt / 256 K tiles each.
static inline bool streamk_plan_small(int64_t M, int64_t N, int64_t K, int& tail, int& w, int& blocks) {
    const int64_t tiles = cdiv64(M, 256) * cdiv64(N, 256);
    const int64_t nt = K / BK;
    // N >= 512: at N = 320 (the UNet's first level) the 256-wide tiles waste 37 % of their columns and the fix-up launch alone costs
    // 60 us; the ring kernel runs [8192, 320, 8640] in 97 us against 140 us for this path (profiles/r04_denoise_kernel_stats.csv)
    // (round 6: at most HALF a round of tiles instead of 208 -- at 192 tiles (the UNet's 640-channel convs at 32 x 32, batch 16) the plain whole-tile
    // launch measured 18-19 % faster at K = 8640 / 17280 and equal at 11520; at 160 tiles (the 1280-channel convs at 16 x 16, batch 32) 15 % / 10 %
    // faster at K = 11520 / 17280 and 12 % slower at 23040: profiles/r06_unet_conv_family_ab.log)
    if (tiles < 48 || tiles > 128 || nt < 96 || (K % BK) != 0 || N < 512) return false;
    const int64_t total = tiles * nt;
    w = (int)cdiv64(total, 256);
    if (w < 32) return false;
    tail = (int)tiles;
    blocks = (int)cdiv64(total, w);
    return true;
}

template <int AL, int BL, int T>
int launch_gemm_t(const GemmParams& P, hipStream_t stream) {
    const int64_t tiles = cdiv64(P.M, T) * cdiv64(P.N, T);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    constexpr int LDS = 2 * 2 * T * BK * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_bf16_kernel<AL, BL, T>, LDS, lds_ok);
    const int sk = P.splitk > 1 ? P.splitk : 1;
    hipLaunchKernelGGL((gemm_bf16_kernel<AL, BL, T>), dim3((unsigned)tiles, sk), dim3(2 * T), LDS, stream, P);
    if (sk > 1 && P.counters == nullptr) {
        int64_t work = P.M * (P.N >> 2);
        int grid = (int)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, stream, P);
    }
    return dllm_check_launch();
}

// Tile choice: estimated efficiency = (useful / padded output area) x (occupied / available block slots over the rounds the
// grid needs on 256 CUs) x relative kernel speed (128-tile 0.85, register-staged 256-tile 1.0, direct-to-LDS 256-tile 1.15).
static inline double tile_eff(int64_t M, int64_t N, int T, double speed, int blocks_per_cu = 0) {
    const int64_t tm = cdiv64(M, T), tn = cdiv64(N, T), tiles = tm * tn;
    const int64_t slots = 256 * (blocks_per_cu > 0 ? blocks_per_cu : (T == 128 ? 2 : 1));
    const int64_t rounds = cdiv64(tiles, slots);
    return ((double)M * N) / ((double)tm * tn * T * T) * ((double)tiles / (double)(rounds * slots)) * speed;
}

// Three-way choice when the pipelined kernels are eligible: estimated time = rounds x (fixed + K tiles x per-K-tile cost) with the
// per-round figures measured on the UNet conv shapes (tools/microbench.py --only conv --tile {128,259,262}; microseconds):
// 256 x 256 pipelined 9 + 1.9 k, 256 x 128 pipelined 6 + 1.3 k (0.69 of the big tile's time for half its outputs), 128 x 128
// register-staged 6 + 1.75 k with two blocks per CU.  Only grids of at least 128 of the 256 x 128 tiles are compared (smaller
// ones are split-K or keep the padding-based choice above).
static inline double tile_cost_us(int64_t M, int64_t N, int64_t K, int TM, int TN, int64_t slots, double fixed, double perk) {
    const int64_t tiles = cdiv64(M, TM) * cdiv64(N, TN);
    return (double)cdiv64(tiles, slots) * (fixed + (double)K / 64.0 * perk);
}

// The LDS-DMA kernels address their dense operands with a buffer descriptor per K tile and 32-bit per-lane byte offsets (round 6): a tile's
// rows (k-contiguous image: 256 rows of ld elements; m-contiguous image: 64 k rows of ld elements + the column) must span less than 2 GiB.
// Operands with larger leading dimensions (strided views of huge buffers) take the register-staged kernels.
template <int AL, int BL>
static inline bool pipe_offsets_ok(const GemmParams& P) {
    constexpr int64_t kLim = ((int64_t)1 << 31) - 4096;
    if (AL == A_K && 256 * P.lda * 2 >= kLim) return false;
    if (AL == A_M && (64 * P.lda + P.M) * 2 >= kLim) return false;
    if (BL == B_K && 256 * P.ldb * 2 >= kLim) return false;
    if (BL == B_N && (64 * P.ldb + P.N) * 2 >= kLim) return false;
    return true;
}

// The ring-buffered 128 x 128 kernel (gemm_ring.hip) replaces the register-staged 128-tile kernel wherever it is eligible: forward
// linears and NHWC convs with K % 64 == 0 (conv: C % 64 == 0), with or without split-K (separate reduce launch only).
template <int AL, int BL>
static inline bool ring_ok(const GemmParams& P) {
    if constexpr (BL != B_K || (AL != A_K && AL != A_CONV)) return false;
    if (P.dbg_noload != 0 && P.dbg_noload != 4) return false;
    if ((P.K % BK) != 0 || P.K < BK) return false;
    if (AL == A_CONV && (P.cv.C % BK) != 0) return false;
    // the ring kernel addresses its operands through buffer descriptors with 32-bit byte offsets (round 6); the conv gather additionally relies
    // on offsets >= 2^31 being out of range (halo taps read zeros): tensors of 2 GiB and more stay on the other kernels
    constexpr int64_t kLim = (int64_t)1 << 31;
    if (AL == A_CONV) {
        if (((int64_t)P.M / ((int64_t)P.cv.OH * P.cv.OW)) * P.cv.H * P.cv.W * P.cv.C * 2 + ((int64_t)P.cv.W + 2) * P.cv.C * 2 >= kLim) return false;
    } else if (128 * P.lda * 2 + P.K * 2 >= kLim) {
        return false;
    }
    // B rows are addressed relative to the tile's first weight row (GEGLU: the `gate` rows lie N rows behind their `hidden` rows)
    if (((P.epi == EPI_GEGLU ? P.N : 0) + 128) * P.ldb * 2 + P.K * 2 >= kLim) return false;
    return true;
}

template <int AL, int BL>
int launch_gemm(const GemmParams& P, const Variant& V, hipStream_t stream) {
    if (P.M <= 0 || P.N <= 0) return DLLM_OK;
    bool ring = ring_ok<AL, BL>(P);
    if (P.epi == EPI_GEGLU) return ring ? dllm_launch_gemm_ring(P, AL, stream, V.ring_stages) : DLLM_ERR_SHAPE;   // only the ring kernel pairs the columns
    if (V.force_ring && ring) {
        if (V.dma_mode < 0) return dllm_launch_gemm_ring(P, AL, stream, V.ring_stages);
        GemmParams Q = P;
        Q.sk_w = V.dma_mode + 1;   // (0 = the kernel's own default placement)
        return dllm_launch_gemm_ring(Q, AL, stream, V.ring_stages);
    }
    if constexpr (AL == A_K && BL == B_K) {
        if ((V.force_mfma32 || V.force_w4) && pipe_offsets_ok<AL, BL>(P) && (P.M % 256) == 0 && (P.N % 256) == 0 && (P.K % BK) == 0 && P.K >= BK && !P.out_f32 && P.bias == nullptr &&
            P.residual == nullptr && P.rg_bias == nullptr && P.epi == 0 && !P.accumulate && P.splitk <= 1 && (P.ldc & 7) == 0 &&
            (reinterpret_cast<uintptr_t>(P.C) & 15) == 0)
            return V.force_w4 ? dllm_launch_gemm_w4(P, stream) : dllm_launch_gemm_pipe32(P, stream);
    }
    ring = ring && V.force_tile == 0 && !V.no_ring;   // tile codes 128 / 256 / 257 / 259 keep selecting the older families (tests)
    const int64_t tiles256 = cdiv64(P.M, 256) * cdiv64(P.N, 256);
    bool glds_ok = V.use_glds && (P.K % BK) == 0 && P.K >= BK && !(AL == A_M && BL == B_K) && pipe_offsets_ok<AL, BL>(P);
    if (AL == A_CONV)  // LDS-DMA gather: plain geometry, a K tile inside one tap, pipelined kernel only
        glds_ok = glds_ok && V.glds_pipe && (P.cv.C % BK) == 0 && BL == B_K;
    // relative speeds: register-staged 128-tile 0.85 (two blocks per CU), ring 128-tile 1.0 (one block per CU), register-staged
    // 256-tile 1.0, LDS-DMA 256-tile 1.15
    const double e256 = tile_eff(P.M, P.N, 256, glds_ok ? 1.15 : 1.0);
    const double e128 = ring ? tile_eff(P.M, P.N, 128, 1.0, 1) : tile_eff(P.M, P.N, 128, 0.85);
    bool pick256 = e256 >= e128;
    if (P.splitk > 1) return ring ? dllm_launch_gemm_ring(P, AL, stream, V.ring_stages) : launch_gemm_t<AL, BL, 128>(P, stream);
    // Ring-eligible problems (forward linears, plain-geometry convs): three-way choice by estimated time = rounds of 256 blocks x
    // (fixed + K tiles x per-K-tile cost), the per-round figures measured on the UNet's shapes at batch 2 and 16 with weights
    // streamed from HBM (tools/unet_gemm_bench.py, profiles/r04_unet_gemm_b{2,16}.log; microseconds): ring 128 x 128: 3.2 + 0.57 k
    // (conv gather 0.71 k); pipelined 256 x 128: 14 + 1.13 k; pipelined 256 x 256: 27 + 1.5 k.  The ring kernel wins short
    // reductions and narrow outputs (K <= ~1000 at any size: 37 vs 65 us for [65536, 320, 320]), the 256-row tiles win deep ones.
    int choice = 0;   // 0: the older logic below; 1 ring; 2 pipelined 256 x 128; 3 pipelined 256 x 256
    if constexpr (BL == B_K && (AL == A_K || AL == A_CONV)) {
        // (grids beyond 16384 tiles of 128 x 128 -- the VAE's 512 x 512 convolutions at the training batch -- keep the older logic:
        // the model's constants were measured on the UNet's shapes, profiles/r04_train_ring_selection_ab.log)
        if (ring && glds_ok && V.glds_pipe && !V.force_n128 && cdiv64(P.M, 128) * cdiv64(P.N, 128) <= 16384) {
            const double kt = (double)(P.K / BK);
            const bool conv = AL == A_CONV;
            const int64_t tiles128 = cdiv64(P.M, 128) * cdiv64(P.N, 128);
            // (grids of more than one block per CU run the two-stage ring, two blocks per CU -- linears up to K = 2048, conv gathers at
            // any K: 2.3 + 0.48 k (conv 0.6 k) per round; the figures reproduce the batch-16 conv table to 1-3 %)
            const bool two_stage = V.ring_stages >= 0 && tiles128 > 256 && (P.K <= 32 * BK || conv);
            const double t_ring = (double)cdiv64(tiles128, 256) * (two_stage ? 2.3 + kt * (conv ? 0.6 : 0.48) : 3.2 + kt * (conv ? 0.71 : 0.57));
            const double t_n128 = (double)cdiv64(cdiv64(P.M, 256) * cdiv64(P.N, 128), 256) * (14.0 + kt * 1.13);
            const double t_256 = (double)cdiv64(tiles256, 256) * (27.0 + kt * 1.5);
            choice = (t_ring <= t_n128 && t_ring <= t_256) ? 1 : (t_n128 < t_256 ? 2 : 3);
        }
    }
    if constexpr (!(AL == A_M && BL == B_K)) {
        int tail = 0, w = 0, blocks = 0;
        if (P.ws != nullptr && V.streamk_ws && glds_ok && V.glds_pipe && V.force_tile == 0 && !V.force_n128 && P.dbg_noload == 0 &&
            streamk_plan_small(P.M, P.N, P.K, tail, w, blocks)) {
            constexpr int LDS = 2 * 2 * 256 * BK * 2;
            GemmParams Q = P;
            Q.sk_full = 0; Q.sk_tail = tail; Q.sk_w = w;
            static std::atomic<uint64_t> ldst_ok{0}, ldsts_ok{0};
            bool shift = false;
            if constexpr (AL == A_CONV) shift = (P.cv.up_shift | P.cv.even_only) != 0;
            if constexpr (AL == A_CONV) {
                if (shift) {
                    dllm_ensure_dyn_lds(&gemm_pipe_tail_kernel<A_CONVS, BL>, LDS, ldsts_ok);
                    hipLaunchKernelGGL((gemm_pipe_tail_kernel<A_CONVS, BL>), dim3((unsigned)blocks), dim3(512), LDS, stream, Q);
                }
            }
            if (!shift) {
                dllm_ensure_dyn_lds(&gemm_pipe_tail_kernel<AL, BL>, LDS, ldst_ok);
                hipLaunchKernelGGL((gemm_pipe_tail_kernel<AL, BL>), dim3((unsigned)blocks), dim3(512), LDS, stream, Q);
            }
            hipLaunchKernelGGL(gemm_pipe_fixup_kernel, dim3((unsigned)tail), dim3(512), 0, stream, Q);
            return dllm_check_launch();
        }
    }
    if constexpr (BL == B_K && (AL == A_K || AL == A_CONV)) {
        // narrow outputs (N = 320: 62 % of two 256-wide tiles, 83 % of three 128-wide ones): the pipelined kernel on 256 x 128 tiles
        if (choice == 1) return dllm_launch_gemm_ring(P, AL, stream, V.ring_stages);
        if (choice == 3) pick256 = true;
        bool n128 = V.force_n128 != 0 || choice == 2;
        if (choice == 0 && !n128 && V.force_tile == 0 && glds_ok && cdiv64(P.M, 256) * cdiv64(P.N, 128) >= 128) {  // at least half the CUs
            const double c256 = tile_cost_us(P.M, P.N, P.K, 256, 256, 256, 9.0, 1.9);
            const double cn = tile_cost_us(P.M, P.N, P.K, 256, 128, 256, 6.0, 1.3);
            const double c128 = tile_cost_us(P.M, P.N, P.K, 128, 128, 512, 6.0, 1.75);
            n128 = cn < c256 && cn < c128;
        }
        if (glds_ok && V.glds_pipe && n128) {
            constexpr int LDSN = 2 * (256 + 128) * BK * 2;
            static std::atomic<uint64_t> ldsn_ok{0}, ldsn_s_ok{0};
            const int64_t tiles = cdiv64(P.M, 256) * cdiv64(P.N, 128);
            if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
            if constexpr (AL == A_CONV) {
                if (P.cv.up_shift | P.cv.even_only) {   // shift addressing: its own instantiation (the plain gather is untouched)
                    dllm_ensure_dyn_lds(&gemm_pipe_kernel<A_CONVS, BL, 128>, LDSN, ldsn_s_ok);
                    hipLaunchKernelGGL((gemm_pipe_kernel<A_CONVS, BL, 128>), dim3((unsigned)tiles), dim3(512), LDSN, stream, P);
                    return dllm_check_launch();
                }
            }
            dllm_ensure_dyn_lds(&gemm_pipe_kernel<AL, BL, 128>, LDSN, ldsn_ok);
            hipLaunchKernelGGL((gemm_pipe_kernel<AL, BL, 128>), dim3((unsigned)tiles), dim3(512), LDSN, stream, P);
            return dllm_check_launch();
        }
    }
    if (V.force_tile == 256 || (V.force_tile == 0 && pick256)) {
        if constexpr (AL == A_CONV) {
            if constexpr (BL == B_K) {
                if (glds_ok) {
                    constexpr int LDS = 2 * 2 * 256 * BK * 2;
                    static std::atomic<uint64_t> lds_ok{0}, lds_s_ok{0};
                    if (P.cv.up_shift | P.cv.even_only) {
                        dllm_ensure_dyn_lds(&gemm_pipe_kernel<A_CONVS, BL>, LDS, lds_s_ok);
                        hipLaunchKernelGGL((gemm_pipe_kernel<A_CONVS, BL>), dim3((unsigned)tiles256), dim3(512), LDS, stream, P);
                        return dllm_check_launch();
                    }
                    dllm_ensure_dyn_lds(&gemm_pipe_kernel<AL, BL>, LDS, lds_ok);
                    hipLaunchKernelGGL((gemm_pipe_kernel<AL, BL>), dim3((unsigned)tiles256), dim3(512), LDS, stream, P);
                    return dllm_check_launch();
                }
            }
        }
        if constexpr (AL != A_CONV) {
            if (glds_ok) {
                constexpr int LDS = 2 * 2 * 256 * BK * 2;
                static std::atomic<uint64_t> lds_ok{0}, lds2_ok{0};
                dllm_ensure_dyn_lds(&gemm_glds_kernel<AL, BL>, LDS, lds_ok);
                if (V.glds_pipe) {
                    dllm_ensure_dyn_lds(&gemm_pipe_kernel<AL, BL>, LDS, lds2_ok);
                    int full = 0, tail = 0, w = 0, blocks = 0;
                    // XCD-synchronised persistent walk (opt-in, variant bit 24): grids of at least 4 rounds, workspace present
                    static std::atomic<uint64_t> lds4_ok{0};
                    const bool persist = V.persist && V.streamk_ws && P.ws != nullptr && P.splitk <= 1 && P.dbg_noload == 0 && tiles256 >= 1024;
                    // the four-wave kernel (gemm_w4.hip) takes the grids of at least one full round of 256 tiles (the LLM's linears; tile
                    // code 280: any grid); same tile order, same arithmetic, same epilogues
                    const bool w4m = !persist && dllm_w4m_eligible(P, AL, BL) && (V.w4m == 2 || (V.w4m == 1 && tiles256 >= 256));
                    auto launch_main = [&](const GemmParams& Q, int64_t ntiles) {
                        if (w4m) {
                            (void)dllm_launch_gemm_w4m(Q, AL, BL, ntiles, stream);
                        } else if (persist) {
                            dllm_ensure_dyn_lds(&gemm_pipe_persist_kernel<AL, BL>, LDS, lds4_ok);
                            hipLaunchKernelGGL((gemm_pipe_persist_kernel<AL, BL>), dim3((unsigned)dllm_num_cus()), dim3(512), LDS, stream, Q);
                        } else {
                            hipLaunchKernelGGL((gemm_pipe_kernel<AL, BL>), dim3((unsigned)ntiles), dim3(512), LDS, stream, Q);
                        }
                    };
                    if (P.ws != nullptr && V.streamk_ws && P.splitk <= 1 && P.dbg_noload == 0 && streamk_plan(P.M, P.N, P.K, full, tail, w, blocks)) {
                        // whole rounds, then the last partial round's K loops spread evenly over the CUs, then the fix-up
                        static std::atomic<uint64_t> lds3_ok{0};
                        dllm_ensure_dyn_lds(&gemm_pipe_tail_kernel<AL, BL>, LDS, lds3_ok);
                        GemmParams Q = P;
                        Q.sk_full = full; Q.sk_tail = tail; Q.sk_w = w;
                        launch_main(Q, full);
                        hipLaunchKernelGGL((gemm_pipe_tail_kernel<AL, BL>), dim3((unsigned)blocks), dim3(512), LDS, stream, Q);
                        hipLaunchKernelGGL(gemm_pipe_fixup_kernel, dim3((unsigned)tail), dim3(512), 0, stream, Q);
                        return dllm_check_launch();
                    }
                    launch_main(P, tiles256);
                    return dllm_check_launch();
                }
                hipLaunchKernelGGL((gemm_glds_kernel<AL, BL>), dim3((unsigned)tiles256), dim3(512), LDS, stream, P);
                return dllm_check_launch();
            }
        }
        return launch_gemm_t<AL, BL, 256>(P, stream);
    }
    if (ring) return dllm_launch_gemm_ring(P, AL, stream, V.ring_stages);
    return launch_gemm_t<AL, BL, 128>(P, stream);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" {

// layout_a: 0 = A[m][k] k-contiguous (lda = row pitch), 1 = A stored [K][lda] with m contiguous.
// layout_b: 0 = B given as [N][ldb] k-contiguous (nn.Linear weight), 1 = B stored [K][ldb] with n contiguous.
// epi: 0 none, 1 exact-erf GELU, 2 quick-GELU, 3 SiLU.  out_dtype: DLLM_BF16 / DLLM_F32.
int dllm_gemm_bf16_splitk(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                          int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int layout_a, int layout_b, int epi,
                          int out_dtype, int accumulate, float alpha, int splitk, float* workspace, int* counters, int variant,
                          void* stream);

int dllm_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int layout_a, int layout_b, int epi,
                   int out_dtype, int accumulate, float alpha, void* stream) {
    return dllm_gemm_bf16_splitk(A, B, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, layout_a, layout_b, epi, out_dtype,
                                 accumulate, alpha, 1, nullptr, nullptr, 0, stream);
}

// splitk > 1: workspace = fp32 [splitk][M][N] (caller-allocated); requires N % 4 == 0.  counters: optional int32[>= number of
// 128 x 128 output tiles], all zero on entry and left all zero: the reduction then happens inside the GEMM launch (the last K
// slice of a tile to arrive reduces it); NULL = a separate reduce kernel.  One counter array per stream in flight.
int dllm_gemm_bf16_splitk(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                          int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int layout_a, int layout_b, int epi,
                          int out_dtype, int accumulate, float alpha, int splitk, float* workspace, int* counters, int variant,
                          void* stream) {
    Variant V;
    if (parse_variant(variant, V) != DLLM_OK) return DLLM_ERR_SHAPE;
    if (splitk > 1 && (workspace == nullptr || (N & 3))) return DLLM_ERR_SHAPE;
    if (M < 0 || N < 0 || K < 0) return DLLM_ERR_SHAPE;
    if (M == 0 || N == 0) return DLLM_OK;
    if (!aligned16(A) || !aligned16(B)) return DLLM_ERR_ALIGN;
    if ((lda & 7) || (ldb & 7)) return DLLM_ERR_ALIGN;
    if (layout_a == A_K && (K & 7)) return DLLM_ERR_ALIGN;
    if (layout_b == B_K && (K & 7)) return DLLM_ERR_ALIGN;
    if (layout_a == A_M && (M & 7)) return DLLM_ERR_ALIGN;
    if (layout_b == B_N && (N & 7)) return DLLM_ERR_ALIGN;
    if (out_dtype != DLLM_BF16 && out_dtype != DLLM_F32) return DLLM_ERR_DTYPE;
    GemmParams P{};
    P.A = (const bf16*)A; P.B = (const bf16*)B; P.C = C; P.bias = (const bf16*)bias; P.residual = (const bf16*)residual;
    P.M = M; P.N = N; P.K = K; P.lda = lda; P.ldb = ldb; P.ldc = ldc; P.ldr = ldr;
    P.epi = epi; P.out_f32 = (out_dtype == DLLM_F32); P.accumulate = accumulate; P.alpha = alpha;
    P.splitk = splitk > 1 ? splitk : 1; P.ws = workspace; P.counters = splitk > 1 ? counters : nullptr; P.dbg_noload = V.dbg_noload;
    // grouped tile order: M rows per column group.  Measured (tools/gemm_groupm_sweep.py): 2-4 is 2-5 % faster than 8 for the
    // forward / input-gradient layouts (A = activations, M = 32768), 8 is best for the weight gradient; 16+ loses 10 %.
    P.group_m = V.group_m > 0 ? V.group_m : (layout_a == A_M ? 8 : 4);
    hipStream_t s = (hipStream_t)stream;
    if (layout_a == A_K && layout_b == B_K) return launch_gemm<A_K, B_K>(P, V, s);
    if (layout_a == A_K && layout_b == B_N) return launch_gemm<A_K, B_N>(P, V, s);
    if (layout_a == A_M && layout_b == B_N) return launch_gemm<A_M, B_N>(P, V, s);
    if (layout_a == A_M && layout_b == B_K) return launch_gemm<A_M, B_K>(P, V, s);
    return DLLM_ERR_SHAPE;
}

// Fused entry points: bits 8-9 of `group_m` choose the kernel family (0: the launcher's choice -- the four-wave kernel from one full round of
// tiles on; 1: the 8-wave kernel; 2: the four-wave kernel), bits 0-7 are GROUP_M (tools / tests; same results either way)
static inline bool fused_use_w4m(int group_m, int64_t tiles, const GemmParams& P, int la, int lb) {
    const int fam = (group_m >> 8) & 3;
    return fam != 1 && (fam == 2 || tiles >= 256) && dllm_w4m_eligible(P, la, lb);
}
// DreamLLMMLP (modeling_dreamllm.py:237) with the SwiGLU folded into the two GEMMs beside it (see gemm_epilogue_swiglu_*).
//   fwd: gu[M, 2F] = x Wgu^T (Wgu = packed [gate rows; up rows], [2F, K]) and act[M, F] = silu(gu[:, :F]) * gu[:, F:], one launch.
//   bwd: dgu[M, 2F] = glu_bwd(dy Wd, gu) with Wd [D, F] (the down projection's nn.Linear weight), d_act never stored.
// Shapes: M % 256 == 0, K (= hidden width, fwd) / D (bwd) a multiple of 64, F % 128 == 0 (fwd) / F % 256 == 0 (bwd); every pointer 16-byte
// aligned, every leading dimension a multiple of 8.  Anything else: DLLM_ERR_SHAPE / DLLM_ERR_ALIGN (the caller runs the unfused launches).
int dllm_gemm_swiglu_fwd(const void* x, const void* wgu, void* gu, void* act, int64_t M, int64_t F, int64_t K, int64_t ldx, int64_t ldw,
                         int64_t ldgu, int64_t ldact, int group_m, void* stream) {
    if (M <= 0 || F <= 0 || K < BK || (M % 256) || (F % 128) || (K % BK)) return DLLM_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(wgu) || !aligned16(gu) || !aligned16(act) || ((ldx | ldw | ldgu | ldact) & 7)) return DLLM_ERR_ALIGN;
    if (256 * ldx * 2 >= ((int64_t)1 << 31) - 4096 || (F + 256) * ldw * 2 >= ((int64_t)1 << 31) - 4096) return DLLM_ERR_SHAPE;   // 32-bit DMA offsets
    GemmParams P{};
    P.A = (const bf16*)x; P.B = (const bf16*)wgu; P.C = gu; P.aux_out = (bf16*)act;
    P.M = M; P.N = 2 * F; P.K = K; P.lda = ldx; P.ldb = ldw; P.ldc = ldgu; P.ld_aux_out = ldact; P.glu_F = F;
    P.epi = EPI_SWIGLU_FWD; P.alpha = 1.f; P.splitk = 1; P.group_m = (group_m & 0xff) > 0 ? (group_m & 0xff) : 4;
    const int64_t tiles = (M / 256) * (2 * F / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    if (fused_use_w4m(group_m, tiles, P, A_K, B_K)) return dllm_launch_gemm_w4m(P, A_K, B_K, tiles, (hipStream_t)stream);
    constexpr int LDS = 2 * 2 * 256 * BK * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_pipe_kernel<A_K, B_K>, LDS, lds_ok);
    hipLaunchKernelGGL((gemm_pipe_kernel<A_K, B_K>), dim3((unsigned)tiles), dim3(512), LDS, (hipStream_t)stream, P);
    return dllm_check_launch();
}
// DreamLLMAttention's q / k / v projections as ONE GEMM on the packed [(Hq + 2 Hkv) * 128, K] weight with apply_rotary_pos_emb
// (modeling_dreamllm.py:184-209,336-338) on the q and k heads in its epilogue (SURVEY §8(b2) `rope` epilogue).  rope_cols = (Hq + Hkv) * 128;
// cos / sin: fp32 [max_pos][64]; pos: int64 [M] or NULL (position = row % S).  head_dim 128 only; M % 256 == 0, N % 256 == 0, rope_cols % 256 == 0,
// K % 64 == 0.  Same results as dllm_gemm_bf16 + dllm_rope, bit for bit.
int dllm_gemm_rope_qkv(const void* x, const void* wqkv, void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* pos, int64_t M,
                       int64_t N, int64_t K, int64_t rope_cols, int S, int64_t ldx, int64_t ldw, int64_t ldo, int group_m, void* stream) {
    if (M <= 0 || N <= 0 || K < BK || (M % 256) || (N % 256) || (K % BK) || rope_cols <= 0 || rope_cols > N || (rope_cols % 256) || S <= 0) return DLLM_ERR_SHAPE;
    if (cos_tab == nullptr || sin_tab == nullptr) return DLLM_ERR_SHAPE;
    if (!aligned16(x) || !aligned16(wqkv) || !aligned16(qkv) || !aligned16(cos_tab) || !aligned16(sin_tab) || ((ldx | ldw | ldo) & 7)) return DLLM_ERR_ALIGN;
    if (256 * ldx * 2 >= ((int64_t)1 << 31) - 4096 || 256 * ldw * 2 >= ((int64_t)1 << 31) - 4096) return DLLM_ERR_SHAPE;   // 32-bit DMA offsets
    GemmParams P{};
    P.A = (const bf16*)x; P.B = (const bf16*)wqkv; P.C = qkv;
    P.M = M; P.N = N; P.K = K; P.lda = ldx; P.ldb = ldw; P.ldc = ldo;
    P.rope_cos = cos_tab; P.rope_sin = sin_tab; P.rope_pos = pos; P.rope_S = S; P.rope_cols = rope_cols;
    P.epi = EPI_ROPE_QKV; P.alpha = 1.f; P.splitk = 1; P.group_m = (group_m & 0xff) > 0 ? (group_m & 0xff) : 4;
    const int64_t tiles = (M / 256) * (N / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    // (the RoPE epilogue -- table rows gathered per output row -- costs the four-wave kernel more than its K loop gains: 1328 against 1362 TF on the
    // packed q|k|v projection, profiles/r06_w4m_prefill_fused_ab.log; it runs there only when asked for)
    if (((group_m >> 8) & 3) == 2 && fused_use_w4m(group_m, tiles, P, A_K, B_K)) return dllm_launch_gemm_w4m(P, A_K, B_K, tiles, (hipStream_t)stream);
    constexpr int LDS = 2 * 2 * 256 * BK * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_pipe_kernel<A_K, B_K>, LDS, lds_ok);
    hipLaunchKernelGGL((gemm_pipe_kernel<A_K, B_K>), dim3((unsigned)tiles), dim3(512), LDS, (hipStream_t)stream, P);
    return dllm_check_launch();
}
int dllm_gemm_swiglu_bwd(const void* dy, const void* wd, const void* gu, void* dgu, int64_t M, int64_t F, int64_t D, int64_t lddy,
                         int64_t ldw, int64_t ldgu, int64_t lddgu, int group_m, void* stream) {
    if (M <= 0 || F <= 0 || D < BK || (M % 256) || (F % 256) || (D % BK)) return DLLM_ERR_SHAPE;
    if (!aligned16(dy) || !aligned16(wd) || !aligned16(gu) || !aligned16(dgu) || ((lddy | ldw | ldgu | lddgu) & 7)) return DLLM_ERR_ALIGN;
    if (256 * lddy * 2 >= ((int64_t)1 << 31) - 4096 || (64 * ldw + F) * 2 >= ((int64_t)1 << 31) - 4096) return DLLM_ERR_SHAPE;   // 32-bit DMA offsets
    GemmParams P{};
    P.A = (const bf16*)dy; P.B = (const bf16*)wd; P.C = dgu; P.aux_in = (const bf16*)gu;
    P.M = M; P.N = F; P.K = D; P.lda = lddy; P.ldb = ldw; P.ldc = lddgu; P.ld_aux_in = ldgu; P.glu_F = F;
    P.epi = EPI_SWIGLU_BWD; P.alpha = 1.f; P.splitk = 1; P.group_m = (group_m & 0xff) > 0 ? (group_m & 0xff) : 4;
    const int64_t tiles = (M / 256) * (F / 256);
    if (tiles > 0x7fffffff) return DLLM_ERR_SHAPE;
    if (fused_use_w4m(group_m, tiles, P, A_K, B_N)) return dllm_launch_gemm_w4m(P, A_K, B_N, tiles, (hipStream_t)stream);
    constexpr int LDS = 2 * 2 * 256 * BK * 2;
    static std::atomic<uint64_t> lds_ok{0};
    dllm_ensure_dyn_lds(&gemm_pipe_kernel<A_K, B_N>, LDS, lds_ok);
    hipLaunchKernelGGL((gemm_pipe_kernel<A_K, B_N>), dim3((unsigned)tiles), dim3(512), LDS, (hipStream_t)stream, P);
    return dllm_check_launch();
}

// Stream-K tail (splitk <= 1 and workspace != NULL): the workspace must hold dllm_gemm_streamk_ws_bytes() bytes; the library then
// splits launches whose last round of 256 x 256 tiles would leave most of the chip idle (see gemm_pipe_tail_kernel).  Results are
// deterministic (fixed summation order) and differ from the unsplit launch only by fp32 re-association of the K sum.
int64_t dllm_gemm_streamk_ws_bytes(void) { return SK_WS_BYTES; }
// 1 when dllm_gemm_bf16_splitk(..., splitk = 1, workspace != NULL, ...) would take the stream-K tail path for this problem
int dllm_gemm_streamk_hint(int64_t M, int64_t N, int64_t K, int layout_a, int layout_b) {
    if (M <= 0 || N <= 0 || K < BK || (K % BK) != 0) return 0;
    if (layout_a == A_M && layout_b == B_K) return 0;
    {   // small grid, deep K (layout_a 2 = implicit-GEMM conv, K = KH*KW*C with C % 64 == 0 checked by the caller)
        int tail, w, blocks;
        if (streamk_plan_small(M, N, K, tail, w, blocks)) return 1;
    }
    if (layout_a == A_CONV) return 0;
    if (tile_eff(M, N, 256, 1.15) < tile_eff(M, N, 128, 0.85)) return 0;
    int full, tail, w, blocks;
    return streamk_plan(M, N, K, full, tail, w, blocks) ? 1 : 0;
}

// Split-K helper: number of K splits this library would like for an [M,N,K] problem (1 = none).  Small grids with a deep
// reduction (UNet at batch 2: M = 128..2048, K = 5760..23040) otherwise leave most of the 256 CUs idle.
// layout_a / layout_b as in dllm_gemm_bf16 (layout_a 2 = the implicit-GEMM conv gather): the ring-buffered kernel's cost model only
// applies where that kernel runs (k-contiguous or gathered A, k-contiguous B); the other layouts of a small grid run on the register-staged
// 128-tile kernel and take its rule (the hint used to assume the forward layout for every caller).
int dllm_gemm_splitk_hint(int64_t M, int64_t N, int64_t K, int layout_a, int layout_b) {
    if (M <= 0 || N <= 0 || (N & 3)) return 1;
    const int64_t tiles = cdiv64(M, 128) * cdiv64(N, 128);
    const int64_t ktiles = cdiv64(K, BK);
    const bool ring_layout = (layout_a == A_K || layout_a == A_CONV) && layout_b == B_K;
    if ((K % BK) == 0 && ring_layout) {
        // The ring-buffered kernel (one 128 x 128 block per CU, every K step overlapped with three stages of loads) needs far fewer
        // slices than the register-staged kernel did: a slice count is worth it only when the shorter K loop pays for the fp32 slab
        // round trip and the reduce launch.  Estimated time (us; per-round figures of launch_gemm's model, the reduce launch ~7 us,
        // slabs written + re-read at ~4 TB/s): measured on the UNet's shapes at batch 2 (profiles/r04_unet_gemm_b2.log) this picks
        // 1 for [8192, 320, 2880] (37 us; two slices: 51), 3 for [2048, 640, 5760] (66 -> 35), 5-6 for [512, 1280, 11520] (124 -> 39),
        // ~20 for [128, 1280, 11520] (124 -> 24), and 1 for every K <= 1280 (a split [512, 1280, 1280] is 2-4 us SLOWER).
        if (tiles >= 256 || ktiles < 8) return 1;
        const double ts = 0.65;
        int best = 1;
        double best_t = (double)cdiv64(tiles, 256) * (3.2 + (double)ktiles * ts);
        const int64_t smax = ktiles / 2 < 32 ? ktiles / 2 : 32;
        for (int64_t s = 2; s <= smax; ++s) {
            const int64_t per = cdiv64(ktiles, s);
            if ((s - 1) * per >= ktiles) continue;   // an empty last slice: s - 1 slices do the same work
            const double t = (double)cdiv64(tiles * s, 256) * (3.2 + (double)per * ts) + 7.0 + (double)s * (double)M * (double)N * 8.0 / 4.0e6;
            if (t < 0.97 * best_t) {
                best_t = t;
                best = (int)s;
            }
        }
        return best;
    }
    if (tiles > 256 || ktiles < 16) return 1;
    // register-staged 128-tile kernel (K % 64 != 0): aim at ~1.5 blocks per CU for tiny grids; grids of 128..256 tiles are split so
    // that two blocks share a CU: 512 slots / tiles
    int64_t want = tiles >= 128 ? 512 / tiles : cdiv64(384, tiles);
    int64_t maxs = ktiles / 8;                  // keep >= 8 K tiles (512 k) per split
    int64_t s = want < maxs ? want : maxs;
    if (s > 32) s = 32;
    return s < 2 ? 1 : (int)s;
}

// NHWC convolution as implicit GEMM: out[n,oh,ow,co] = sum_{kh,kw,ci} in[n,ih,iw,ci] * w[co,kh,kw,ci] (+bias, +residual).
// x: [NB,H,W,C] bf16, w: [CO, KH*KW*C] bf16 (k-contiguous), out: [NB,OH,OW,CO].
// up2: the logical input is the nearest-2x upsampling of x (Upsample2D + conv fused).
// even_only: transposed gather used for the dgrad of a stride-2 conv (logical stride 1 over a zero-stuffed grid).
// image_bias: optional bf16 [NB, CO] added per image (ResnetBlock2D time_emb_proj broadcast), before the activation.
int dllm_conv2d_nhwc_bf16_splitk(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                 const void* image_bias, int NB, int H, int W, int C, int OH, int OW, int CO, int KH, int KW,
                                 int stride, int pad, int up2, int even_only, int epi, int out_dtype, int splitk,
                                 float* workspace, int* counters, int variant, void* stream);

int dllm_conv2d_nhwc_bf16(const void* x, const void* w, void* out, const void* bias, const void* residual,
                          const void* image_bias, int NB, int H, int W, int C, int OH, int OW, int CO, int KH, int KW,
                          int stride, int pad, int up2, int even_only, int epi, int out_dtype, void* stream) {
    return dllm_conv2d_nhwc_bf16_splitk(x, w, out, bias, residual, image_bias, NB, H, W, C, OH, OW, CO, KH, KW, stride, pad, up2,
                                        even_only, epi, out_dtype, 1, nullptr, nullptr, 0, stream);
}

// splitk > 1: workspace = fp32 [splitk][NB*OH*OW][CO]; requires CO % 4 == 0; counters as in dllm_gemm_bf16_splitk.
int dllm_conv2d_nhwc_bf16_splitk(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                 const void* image_bias, int NB, int H, int W, int C, int OH, int OW, int CO, int KH, int KW,
                                 int stride, int pad, int up2, int even_only, int epi, int out_dtype, int splitk,
                                 float* workspace, int* counters, int variant, void* stream) {
    Variant V;
    if (parse_variant(variant, V) != DLLM_OK) return DLLM_ERR_SHAPE;
    if (splitk > 1 && (workspace == nullptr || (CO & 3))) return DLLM_ERR_SHAPE;
    if (NB < 0 || H <= 0 || W <= 0 || C <= 0 || CO <= 0 || OH <= 0 || OW <= 0) return DLLM_ERR_SHAPE;
    if (NB == 0) return DLLM_OK;
    if ((C & 7) != 0) return DLLM_ERR_ALIGN;
    if (!aligned16(x) || !aligned16(w)) return DLLM_ERR_ALIGN;
    if (!((KH == 3 && KW == 3) || (KH == 1 && KW == 1))) return DLLM_ERR_SHAPE;
    GemmParams P{};
    P.A = (const bf16*)x; P.B = (const bf16*)w; P.C = out; P.bias = (const bf16*)bias; P.residual = (const bf16*)residual;
    P.M = (int64_t)NB * OH * OW; P.N = CO; P.K = (int64_t)KH * KW * C;
    P.lda = C; P.ldb = P.K; P.ldc = CO; P.ldr = CO;
    P.epi = epi; P.out_f32 = (out_dtype == DLLM_F32); P.accumulate = 0; P.alpha = 1.0f;
    P.rg_bias = (const bf16*)image_bias; P.rg_rows = (int64_t)OH * OW;
    P.splitk = splitk > 1 ? splitk : 1; P.ws = workspace; P.counters = splitk > 1 ? counters : nullptr; P.dbg_noload = V.dbg_noload;
    P.group_m = V.group_m > 0 ? V.group_m : 4;  // output pixels are the M dimension (activations), as in the forward GEMMs
    P.cv = ConvGeom{H, W, C, OH, OW, KH, KW, stride, pad, up2, even_only};
    return launch_gemm<A_CONV, B_K>(P, V, (hipStream_t)stream);
}

}  // extern "C"
