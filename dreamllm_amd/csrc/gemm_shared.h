// Shared pieces of the bf16 MFMA GEMM family (gemm.hip: 256-row pipelined / register-staged kernels; gemm_ring.hip: the 128 x 128
// ring-buffered kernel for small grids): parameter block, LDS images, fragment reads, epilogues, LDS-DMA helpers.
#pragma once
#include <type_traits>

#include "common.h"

// parameter blocks: plain structs at global scope (shared by the translation units of the family)
struct ConvGeom {
    int H, W, C;     // physical input spatial dims and channels (NHWC)
    int OH, OW;      // output spatial dims
    int KH, KW;      // 3x3 or 1x1
    int stride;      // 1 or 2 (forward)
    int pad;         // 1 for 3x3, 0 for 1x1
    int up_shift;    // 1: logical input is the nearest-2x upsampled image (physical = logical >> 1)
    int even_only;   // 1: transposed (dgrad of a stride-2 conv): logical index must be even, physical = logical >> 1
};

struct GemmParams {
    const bf16* A;
    const bf16* B;
    void* C;
    const bf16* bias;      // [N] or null
    const bf16* residual;  // [M, ldr] or null (added after activation)
    const bf16* rg_bias;   // [M / rg_rows, N] or null: per-row-group bias (UNet time embedding per image), before activation
    int64_t rg_rows;
    int64_t M, N, K;
    int64_t lda, ldb, ldc, ldr;
    int epi;
    int out_f32;     // C dtype
    int accumulate;  // C += result
    float alpha;     // scale applied to the accumulator before bias
    int group_m;     // tiles per column group of the grouped tile order (pipe kernel; default 8)
    int dbg_noload;  // benchmark-only: skip the K-loop prefetches (wrong results) to expose the compute+barrier ceiling
    int splitk;      // > 1: blockIdx.y owns a K range and writes raw fp32 partials to ws[split][M][N]
    float* ws;
    int sk_full;     // stream-K tail: tiles of the grouped order covered by the whole-round launch (0 = every tile, no tail)
    int sk_tail;     //                tiles whose K loops the tail kernel spreads over the CUs
    int sk_w;        //                K tiles per tail block
    int* counters;   // split-K: one arrival counter per output tile (zero on entry, left zero): the last K slice reduces in-kernel
    ConvGeom cv;
    // fused SwiGLU epilogues of the pipelined 256 x 256 kernel (gemm.hip: EPI_SWIGLU_FWD / EPI_SWIGLU_BWD); glu_F = F (hidden width of the MLP)
    const bf16* aux_in;   // BWD: the forward's packed [M, 2F] gate|up buffer
    bf16* aux_out;        // FWD: act [M, F] = silu(gate) * up
    int64_t ld_aux_in, ld_aux_out, glu_F;
    // fused RoPE epilogue of the packed q|k|v projection (gemm.hip: EPI_ROPE_QKV): fp32 [max_pos][64] half-dim tables, positions or null
    const float* rope_cos;
    const float* rope_sin;
    const int64_t* rope_pos;   // [M] position ids, or null: position = row % rope_S
    int rope_S;
    int64_t rope_cols;         // leading output columns that are rotated (q heads + k heads; head_dim 128), the rest (v) is plain
};

namespace {

constexpr int BK = 64;
constexpr int A_K = 0, A_M = 1, A_CONV = 2;
constexpr int A_CONVS = 3;  // pipelined kernel only: conv gather with "shift" addressing (fused nearest-2x upsample / transposed stride-2)
constexpr int B_K = 0, B_N = 1;

constexpr int EPI_NONE = 0, EPI_GELU = 1, EPI_QUICK_GELU = 2, EPI_SILU = 3;
constexpr int EPI_GEGLU = 4;  // gemm_ring.hip only: out[M, F] = (x Wh^T + bh) * gelu(x Wg^T + bg), weight rows [hidden F | gate F]
// gemm.hip, pipelined 256 x 256 kernel only (DreamLLMMLP, modeling_dreamllm.py:237: down(silu(gate(x)) * up(x))):
constexpr int EPI_SWIGLU_FWD = 5;  // C = [M, 2F] gate|up (as the plain GEMM) AND aux_out = silu(gate) * up, one launch
constexpr int EPI_SWIGLU_BWD = 6;
constexpr int EPI_ROPE_QKV = 7;    // packed q|k|v projection with apply_rotary_pos_emb (modeling_dreamllm.py:184-209) on the q and k heads in the epilogue  // the down projection's input gradient d_act = dy Wd never leaves the block: C = [M, 2F] d(gate|up)

// ---- LDS images -------------------------------------------------------------------------------------------------
// k-contiguous tile: [128 rows][64 k] bf16, 128 B per row, 16-B chunk c stored at chunk c ^ ((row >> 1) & 7):
// conflict-free for ds_read_b128 fragment reads (16 distinct rows, same chunk) and for the staging ds_write_b128.
__device__ __forceinline__ int kc_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// m-contiguous tile: [64 k rows][128 m] bf16, 256 B per row, 32-B slot s stored at s ^ f(krow),
// f = (krow & 3) | ((krow >> 3) & 1) << 2: the 8 k rows one half-wave touches in a transpose read hit 8 distinct slots.
template <int T>
__device__ __forceinline__ int mc_off(int krow, int byte_in_row) {
    const int f = (krow & 3) | (((krow >> 3) & 1) << 2);
    return krow * (2 * T) + ((((byte_in_row >> 5) ^ f)) << 5) + (byte_in_row & 31);
}

__device__ __forceinline__ bf16x8 frag_kc(const char* tile, int row, int kk, int lane) {
    return *reinterpret_cast<const bf16x8*>(tile + kc_off(row, kk * 4 + (lane >> 4)));
}

template <int T>
__device__ __forceinline__ bf16x8 frag_mc(const char* tile, int mbase, int kk, int lane) {
    // lane (g = lane>>4, t = lane&15) receives m = mbase + t, k = kk*32 + g*8 + 0..7
    const int g = lane >> 4, t = lane & 15;
    const int k0 = kk * 32 + g * 8 + (t >> 2);
    const int bcol = mbase * 2 + (t & 3) * 8;
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + mc_off<T>(k0, bcol)));
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, tile + mc_off<T>(k0 + 4, bcol)));
    union {
        struct { short4v a, b; } s;
        bf16x8 v;
    } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// ---- epilogue shared by both kernels: lane holds C[m][n..n+3], m = mbase+i*16+(lane&15), n = nbase+j*16+(lane>>4)*4
// bias / per-image bias / activation / residual / dtype conversion for 4 consecutive outputs C[m][n..n+3]
__device__ __forceinline__ void epilogue_store4(const GemmParams& P, int64_t m, int64_t n, float (&v)[4], bool vec_ok) {
    const int nvalid = (int)min((int64_t)4, P.N - n);
    if (P.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += (float)P.bias[n + r];
    }
    if (P.rg_bias != nullptr) {
        const bf16* rb = P.rg_bias + (m / P.rg_rows) * P.N + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += (float)rb[r];
    }
    if (P.epi == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    } else if (P.epi == EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    } else if (P.epi == EPI_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    }
    if (vec_ok && nvalid == 4) {
        if (P.residual != nullptr) {
            bf16x4 rv = ld_bf16x4(P.residual + m * P.ldr + n);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
        }
        if (P.out_f32) {
            float* cp = reinterpret_cast<float*>(P.C) + m * P.ldc + n;
            f32x4 o = f32x4{v[0], v[1], v[2], v[3]};
            if (P.accumulate) o += *reinterpret_cast<f32x4*>(cp);
            *reinterpret_cast<f32x4*>(cp) = o;
        } else {
            bf16* cp = reinterpret_cast<bf16*>(P.C) + m * P.ldc + n;
            if (P.accumulate) {
                bf16x4 c = ld_bf16x4(cp);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += (float)c[r];
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16)v[r];
            st_bf16x4(cp, o);
        }
    } else {
        for (int r = 0; r < nvalid; ++r) {
            float x = v[r];
            if (P.residual != nullptr) x += (float)P.residual[m * P.ldr + n + r];
            if (P.out_f32) {
                float* cp = reinterpret_cast<float*>(P.C) + m * P.ldc + n + r;
                *cp = P.accumulate ? (*cp + x) : x;
            } else {
                bf16* cp = reinterpret_cast<bf16*>(P.C) + m * P.ldc + n + r;
                *cp = (bf16)(P.accumulate ? ((float)*cp + x) : x);
            }
        }
    }
}

// ---- epilogue shared by the kernels: lane holds C[m][n..n+3], m = mbase+i*16+(lane&15), n = nbase+j*16+(lane>>4)*4
template <int MI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& P, f32x4 (&acc)[MI][4], int64_t mbase, int64_t nbase, int lane,
                                              int split = 0, int tile = 0, int* lds_flag = nullptr) {
    const bool vec_ok = ((P.N & 3) == 0) && ((P.ldc & 3) == 0) && (P.residual == nullptr || (P.ldr & 3) == 0);
    if (P.dbg_noload == 3 && acc[0][0][0] != 12345.678f) return;  // benchmark-only: no C stores (the test keeps acc live)
    if (P.splitk > 1) {
        // raw fp32 partial slab of this K slice (N % 4 == 0 enforced by the host)
        const bool fused = P.counters != nullptr && lds_flag != nullptr;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = mbase + i * 16 + (lane & 15);
            if (m >= P.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = nbase + j * 16 + (lane >> 4) * 4;
                if (n >= P.N) continue;
                float* dst = P.ws + ((int64_t)split * P.M + m) * P.N + n;
                if (fused)  // write-through (sc1): the slab is visible at agent scope without a whole-L2 write-back
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(acc[i][j]) : "memory");
                else
                    *reinterpret_cast<f32x4*>(dst) = acc[i][j];
            }
        }
        if (!fused) return;  // the separate reduce kernel follows
        // In-kernel reduction: every K slice of a tile publishes its slab with write-through stores (the other slices run on other
        // XCDs, whose L2s are not coherent with this one; a release fence = whole-L2 write-back per block measured 1.5x SLOWER than
        // the separate reduce kernel) and takes a ticket once all its stores have completed; the slice that draws the last ticket
        // invalidates (agent-scope acquire), sums the slabs in slice order (deterministic: the order of splitk_reduce_kernel) and
        // applies the epilogue.  One launch less per split GEMM: ~140 per denoising step of the UNet at batch 2.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            *lds_flag = __hip_atomic_fetch_add(P.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*lds_flag != P.splitk - 1) return;
        if (threadIdx.x == 0) __hip_atomic_store(P.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the slabs are read with agent-coherent loads (sc0 sc1: they miss this XCD's caches) instead of an acquire fence: a
        // buffer_inv per reducing block empties the XCD's L2 under every other block of the launch (measured 109 -> 75 steps/s).
        // All loads of one slice are in flight together (MI * 4 requests per lane), slices are summed in order.
        f32x4 sum[MI][4];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < P.splitk; ++z) {
            f32x4 t[MI][4];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int64_t m = min(mbase + i * 16 + (lane & 15), P.M - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t n = min(nbase + j * 16 + (lane >> 4) * 4, P.N - 4);
                    const float* src = P.ws + ((int64_t)z * P.M + m) * P.N + n;
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t[i][j]) : "v"(src) : "memory");
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (i == 0 && j == 0)
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(t[0][0]) : : "memory");
                    else
                        asm volatile("" : "+v"(t[i][j]));
                    sum[i][j] += t[i][j];
                }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int64_t m = mbase + i * 16 + (lane & 15);
            if (m >= P.M) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = nbase + j * 16 + (lane >> 4) * 4;
                if (n >= P.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = sum[i][j][r] * P.alpha;
                epilogue_store4(P, m, n, v, vec_ok);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int64_t m = mbase + i * 16 + (lane & 15);
        if (m >= P.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = nbase + j * 16 + (lane >> 4) * 4;
            if (n >= P.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * P.alpha;
            epilogue_store4(P, m, n, v, vec_ok);
        }
    }
}

// ---- LDS-staged epilogue for full, 16-byte-aligned bf16 output tiles (wave tile 128 x 64) ------------------------------------
// The direct epilogue above stores 8 bytes per lane: one store instruction touches 16 rows x 32 bytes, and four instructions
// are needed to complete a 128-byte line.  Measured on the 256-tile kernels that costs ~20 us per output tile (268 MB of C at
// 1.6 TB/s; tools/gemm_ksweep.py with and without the stores).  Here each wave passes its tile through a private 8-KiB LDS
// region in two 64-row halves: ds_write_b64 in the accumulator layout (XOR-swizzled: chunk ^ 2*((row>>1)&7), conflict-free
// for the 16-lane write groups), ds_read_b128 row-contiguous, then 16-byte global stores of 8 rows x 128 contiguous bytes.
// Bias / activation / residual / accumulate are applied before the LDS write (single rounding, as in the direct epilogue).
__device__ __forceinline__ bool epilogue_lds_ok(const GemmParams& P, int64_t m0, int64_t n0, int TM, int TN) {
    return !P.out_f32 && P.splitk <= 1 && P.dbg_noload != 3 && (P.ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(P.C) & 15) == 0 &&
           m0 + TM <= P.M && n0 + TN <= P.N && (P.residual == nullptr || (P.ldr & 3) == 0);
}
__device__ __forceinline__ bool epilogue_lds_ok(const GemmParams& P, int64_t m0, int64_t n0, int T) {
    return epilogue_lds_ok(P, m0, n0, T, T);
}
// bias / per-image bias / activation for 4 consecutive outputs of a full tile
__device__ __forceinline__ void epilogue_bias_act4(const GemmParams& P, int64_t m, int64_t n, float (&v)[4]) {
    if (P.bias != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)P.bias[n + r];
    }
    if (P.rg_bias != nullptr) {
        const bf16* rb = P.rg_bias + (m / P.rg_rows) * P.N + n;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += (float)rb[r];
    }
    if (P.epi == EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf_f(v[r]);
    } else if (P.epi == EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = quick_gelu_f(v[r]);
    } else if (P.epi == EPI_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
    }
}
// wl: this wave's private 8-KiB LDS region (64 rows x 128 bytes); (mw, nw): global origin of the wave's (16 MI) x 64 tile.
// A tile that is ADDED to the product (residual, or C itself with `accumulate`) is first brought into the same region with
// row-contiguous 16-byte loads; each lane then folds its own 8-byte chunk in fp32 and overwrites it in place.
template <int MI>
__device__ __forceinline__ void gemm_epilogue_lds(const GemmParams& P, f32x4 (&acc)[MI][4], char* wl, int64_t mw, int64_t nw,
                                                  int lane) {
    bf16* C = reinterpret_cast<bf16*>(P.C);
    const bool bias_act = P.bias != nullptr || P.rg_bias != nullptr || P.epi != 0;
    // the addend staged through LDS: the residual if there is one, else C for `accumulate`
    const bf16* pre = P.residual != nullptr ? P.residual : (P.accumulate ? C : nullptr);
    const int64_t ldp = P.residual != nullptr ? P.ldr : P.ldc;
    const bool pre_lds = pre != nullptr && (ldp & 7) == 0 && (reinterpret_cast<uintptr_t>(pre) & 15) == 0;
    const bool res_direct = P.residual != nullptr && !pre_lds;            // unaligned residual: 8-byte loads in accumulator layout
    const bool acc_direct = P.accumulate && (P.residual != nullptr || !pre_lds);
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
        if (pre_lds) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + (lane >> 3), p = lane & 7;
                const bf16x8 v = ld_bf16x8(pre + (mw + half * 64 + row) * ldp + nw + p * 8);
                *reinterpret_cast<bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4)) = v;
            }
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii;
            const int r = ii * 16 + (lane & 15);
            const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] * P.alpha;
                const int64_t m = mw + i * 16 + (lane & 15), n = nw + j * 16 + (lane >> 4) * 4;
                if (bias_act) epilogue_bias_act4(P, m, n, v);
                bf16x4* slot = reinterpret_cast<bf16x4*>(wl + r * 128 + (((j * 4 + (lane >> 4)) ^ sw) << 3));
                if (pre_lds) {
                    const bf16x4 t = *slot;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
                }
                if (res_direct) {
                    const bf16x4 t = ld_bf16x4(P.residual + m * P.ldr + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
                }
                if (acc_direct) {
                    const bf16x4 t = ld_bf16x4(C + m * P.ldc + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)t[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)v[e];
                *slot = o;
            }
        }
        // wave-private region: only this wave's own LDS traffic has to be ordered (the compiler inserts the lgkmcnt waits)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
            st_bf16x8(C + (mw + half * 64 + row) * P.ldc + nw + p * 8, v);
        }
    }
}

// deterministic split-K reduction + epilogue: one thread per 4 outputs
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams P) {
    const int64_t n4 = P.N >> 2;
    const int64_t total = P.M * n4;
    const bool vec_ok = ((P.ldc & 3) == 0) && (P.residual == nullptr || (P.ldr & 3) == 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / n4, n = (i % n4) * 4;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < P.splitk; ++z) a += *reinterpret_cast<const f32x4*>(P.ws + ((int64_t)z * P.M + m) * P.N + n);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = a[r] * P.alpha;
        epilogue_store4(P, m, n, v, vec_ok);
    }
}

// direct-to-LDS copy of 16 bytes per lane (LDS destination = wave-uniform base + lane * 16)
#define GLDS16(gptr, lptr)                                                                                             \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),                           \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

__device__ __forceinline__ uint32_t lds_addr(const char* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ bf16x8 join_frag(u32x2 lo, u32x2 hi) {
    union {
        struct { u32x2 a, b; } s;
        bf16x8 v;
    } u;
    u.s.a = lo;
    u.s.b = hi;
    return u.v;
}

// Fragment reads as inline asm with counted waits (see gemm_pipe_kernel in gemm.hip for why)
template <bool MC>
struct FragR {
    u32x4 v;       // k-contiguous image: one ds_read_b128
    u32x2 lo, hi;  // m-contiguous image: two ds_read_b64_tr_b16
};
template <bool MC, int IDX, int KK>
__device__ __forceinline__ void fragr_issue(FragR<MC>& f, uint32_t base) {
    if constexpr (MC) {
        const uint32_t a = base ^ (uint32_t)(IDX << 5);  // 32-B slot (idx ^ f): bits 5..7 of the address hold f only
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a), "n"(KK * 16384));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a), "n"(KK * 16384 + 2048));
    } else {
        const uint32_t a = KK ? (base ^ 64u) : base;     // chunk (4*kk + g) ^ s = (g ^ s) ^ 4*kk
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.v) : "v"(a), "n"(IDX * 2048));
    }
}
template <int N, bool MC>
__device__ __forceinline__ void fragr_wait(FragR<MC>& f) {  // wait until at most N younger LDS operations are outstanding
    if constexpr (MC)
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f.lo), "+v"(f.hi) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f.v) : "n"(N) : "memory");
}
template <bool MC>
__device__ __forceinline__ void fragr_touch(FragR<MC>& f) {  // orders the consumers of f after the preceding wait
    if constexpr (MC)
        asm volatile("" : "+v"(f.lo), "+v"(f.hi));
    else
        asm volatile("" : "+v"(f.v));
}
template <bool MC>
__device__ __forceinline__ bf16x8 fragr_value(const FragR<MC>& f) {
    if constexpr (MC)
        return join_frag(f.lo, f.hi);
    else
        return __builtin_bit_cast(bf16x8, f.v);
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// 16 zero bytes: the source of LDS-DMA gathers whose tap falls outside the image
__device__ __attribute__((aligned(16))) bf16 g_zero_page[8];

}  // namespace

// gemm_ring.hip: 128 x 128 tiles, 8 waves, 4-stage LDS-DMA ring (A k-contiguous or conv gather with C % 64 == 0, B k-contiguous,
// K % 64 == 0); split-K through fp32 slabs + splitk_reduce_kernel.  layout_a: A_K or A_CONV (shift addressing chosen from P.cv).
__attribute__((visibility("hidden"))) int dllm_launch_gemm_ring(const GemmParams& P, int layout_a, hipStream_t stream, int two_stage = 0);
// gemm_mfma32.hip (experiment): the 256 x 256 pipelined kernel on v_mfma_f32_32x32x16_bf16; forward layout, full tiles, plain epilogue
__attribute__((visibility("hidden"))) int dllm_launch_gemm_pipe32(const GemmParams& P, hipStream_t stream);
// gemm_w4.hip (round 6 experiment): four waves, one per SIMD, wave tile 128 x 128 on MFMA 32x32x16, buffer-form LDS-DMA; same eligibility
__attribute__((visibility("hidden"))) int dllm_launch_gemm_w4(const GemmParams& P, hipStream_t stream);
// gemm_w4.hip: the four-wave 256 x 256 kernel on MFMA 16x16x32 (one wave per SIMD, LDS stage released half a tile early); layouts
// (A_K, B_K), (A_K, B_N), (A_M, B_N); every epilogue of gemm_pipe_kernel; ntiles = blocks of the grouped tile order to launch
__attribute__((visibility("hidden"))) bool dllm_w4m_eligible(const GemmParams& P, int layout_a, int layout_b);
__attribute__((visibility("hidden"))) int dllm_launch_gemm_w4m(const GemmParams& P, int layout_a, int layout_b, int64_t ntiles, hipStream_t stream);
