// Fused epilogues of the 256 x 256 LDS-DMA kernels (gemm.hip: gemm_pipe_kernel, 8 waves of 128 x 64; gemm_w4.hip: 4 waves of 128 x 128 = two
// such halves side by side) and the grouped tile order.  Included after gemm_shared.h.
#pragma once
#include "gemm_shared.h"

namespace {

// ---- fused SwiGLU epilogues (round 6) ----------------------------------------------------------------------------------------
// Both keep the arithmetic of the stand-alone kernels (elementwise.hip: glu_fwd_kernel / glu_bwd_kernel) on the SAME bf16-rounded
// operands, so fused and unfused paths give identical results; what disappears is a launch and its round trip through HBM:
//   FWD  (gate|up projection): the wave holds gate (acc[i][0..1]) and up (acc[i][2..3]) of 32 outputs x 128 rows; it stores the packed
//        [M, 2F] gate|up tile the backward reads AND act = silu(gate) * up [M, F] -- glu_fwd (a read of 2 x [M, F] and a launch) is gone;
//   BWD  (down projection's input gradient): d_act = dy Wd stays in the accumulators; gate / up tiles come in through the wave's LDS
//        region (row-contiguous 16-byte loads), d gate / d up go out the same way -- the [M, F] d_act tensor (written, then read) and
//        glu_bwd's launch are gone.
// Full tiles only (M % 256 == 0; FWD: F % 128 == 0; BWD: F % 256 == 0), 16-byte aligned rows: the entry points check.
// wl: this wave's LDS region, 16 KiB (two 64-row x 128-byte images, chunk swizzle of gemm_epilogue_lds).
template <int MI>
__device__ __forceinline__ void gemm_epilogue_swiglu_fwd(const GemmParams& P, f32x4 (&acc)[MI][4], char* wl, int64_t mw, int64_t n0, int wn,
                                                         int lane) {
    bf16* gu = reinterpret_cast<bf16*>(P.C);
    const int64_t h0 = (n0 >> 1) + (wn >> 6) * 32;   // first of the wave's 32 output columns
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
        // pass 1: the packed gate|up tile (64 rows x [32 gate | 32 up]) through the region, rows out as 2 x 64 contiguous bytes
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii;
            const int r = ii * 16 + (lane & 15);
            const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (bf16)(acc[i][j][e] * P.alpha);
                *reinterpret_cast<bf16x4*>(wl + r * 128 + (((j * 4 + (lane >> 4)) ^ sw) << 3)) = o;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
            st_bf16x8(gu + (mw + half * 64 + row) * P.ldc + ((p & 4) ? P.glu_F : 0) + h0 + (p & 3) * 8, v);
        }
        // pass 2: act from the ROUNDED gate / up (what glu_fwd_kernel reads back from memory), 64 rows x 64 bytes
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii;
            const int r = ii * 16 + (lane & 15);
            const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = (float)(bf16)(acc[i][j][e] * P.alpha), y = (float)(bf16)(acc[i][j + 2][e] * P.alpha);
                    o[e] = (bf16)swiglu_fwd_elem(x, y);
                }
                *reinterpret_cast<bf16x4*>(wl + 8192 + r * 128 + (((j * 4 + (lane >> 4)) ^ sw) << 3)) = o;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 16 + (lane >> 2), p = lane & 3;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + 8192 + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
            st_bf16x8(P.aux_out + (mw + half * 64 + row) * P.ld_aux_out + h0 + p * 8, v);
        }
    }
}

// EPI_ROPE_QKV: rope_kernel's arithmetic (elementwise.hip) on the bf16-ROUNDED projection, in the epilogue of the packed q|k|v GEMM.
// The wave holds x1 = dims [32 (wc & 1), +32) of head (wc >> 1) in acc[i][0..1] and x2 = the same dims + 64 in acc[i][2..3] (B rows re-mapped
// in pipe_tile); y1 = x1 cos - x2 sin, y2 = x2 cos + x1 sin with the fp32 table rows of the token's position.  Identical results to GEMM +
// dllm_rope; the separate launch and its read + write of the q and k heads are gone.
template <int MI>
__device__ __forceinline__ void gemm_epilogue_rope(const GemmParams& P, f32x4 (&acc)[MI][4], char* wl, int64_t mw, int64_t n0, int wn, int lane) {
    bf16* C = reinterpret_cast<bf16*>(P.C);
    const int wc = wn >> 6;
    const int64_t c0 = n0 + (wc >> 1) * 128 + (wc & 1) * 32;   // output column of the wave's first x1 dim
    const int d0 = (wc & 1) * 32 + (lane >> 4) * 4;             // the lane's first dim (of the 64 pair dims) for j = 0
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii;
            const int r = ii * 16 + (lane & 15);
            const int sw = ((r >> 1) & 7) << 1;
            const int64_t m = mw + i * 16 + (lane & 15);
            const int64_t p = P.rope_pos ? P.rope_pos[m] : (m % P.rope_S);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4 cc = *reinterpret_cast<const f32x4*>(P.rope_cos + p * 64 + d0 + j * 16);
                const f32x4 ss = *reinterpret_cast<const f32x4*>(P.rope_sin + p * 64 + d0 + j * 16);
                bf16x4 o1, o2;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y1, y2;
                    rope_pair((float)(bf16)(acc[i][j][e] * P.alpha), (float)(bf16)(acc[i][j + 2][e] * P.alpha), cc[e], ss[e], y1, y2);
                    o1[e] = (bf16)y1;
                    o2[e] = (bf16)y2;
                }
                *reinterpret_cast<bf16x4*>(wl + r * 128 + (((j * 4 + (lane >> 4)) ^ sw) << 3)) = o1;
                *reinterpret_cast<bf16x4*>(wl + r * 128 + ((((j + 2) * 4 + (lane >> 4)) ^ sw) << 3)) = o2;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(wl + row * 128 + ((p ^ ((row >> 1) & 7)) << 4));
            st_bf16x8(C + (mw + half * 64 + row) * P.ldc + c0 + ((p & 4) ? 64 : 0) + (p & 3) * 8, v);
        }
    }
}

template <int MI>
__device__ __forceinline__ void gemm_epilogue_swiglu_bwd(const GemmParams& P, f32x4 (&acc)[MI][4], char* wl, int64_t mw, int64_t nw, int lane) {
    bf16* dgu = reinterpret_cast<bf16*>(P.C);
    char* w0 = wl;            // gate in, d gate out
    char* w1 = wl + 8192;     // up in, d up out
#pragma unroll
    for (int half = 0; half < MI / 4; ++half) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const bf16* src = P.aux_in + (mw + half * 64 + row) * P.ld_aux_in + nw + p * 8;
            const int off = row * 128 + ((p ^ ((row >> 1) & 7)) << 4);
            *reinterpret_cast<bf16x8*>(w0 + off) = ld_bf16x8(src);
            *reinterpret_cast<bf16x8*>(w1 + off) = ld_bf16x8(src + P.glu_F);
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = half * 4 + ii;
            const int r = ii * 16 + (lane & 15);
            const int sw = ((r >> 1) & 7) << 1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = r * 128 + (((j * 4 + (lane >> 4)) ^ sw) << 3);
                const bf16x4 gv = *reinterpret_cast<const bf16x4*>(w0 + off), uv = *reinterpret_cast<const bf16x4*>(w1 + off);
                bf16x4 oa, ob;
#pragma unroll
                for (int e = 0; e < 4; ++e) {   // glu_bwd_kernel<0>, on d_act rounded to bf16 as the unfused path stores it
                    const float d = (float)(bf16)(acc[i][j][e] * P.alpha), x = (float)gv[e], y = (float)uv[e];
                    float dg, du;
                    swiglu_bwd_elem(d, x, y, dg, du);
                    oa[e] = (bf16)dg;
                    ob[e] = (bf16)du;
                }
                *reinterpret_cast<bf16x4*>(w0 + off) = oa;
                *reinterpret_cast<bf16x4*>(w1 + off) = ob;
            }
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + (lane >> 3), p = lane & 7;
            const int off = row * 128 + ((p ^ ((row >> 1) & 7)) << 4);
            bf16* dst = dgu + (mw + half * 64 + row) * P.ldc + nw + p * 8;
            st_bf16x8(dst, *reinterpret_cast<const bf16x8*>(w0 + off));
            st_bf16x8(dst + P.glu_F, *reinterpret_cast<const bf16x8*>(w1 + off));
        }
    }
}

// Tile `wgid` of the grouped (GROUP_M) tile order -> (pid_m, pid_n)
__device__ __forceinline__ void pipe_decode_tile(const GemmParams& P, int wgid, int num_pid_m, int num_pid_n, int& pid_m, int& pid_n) {
    const int GROUP_M = P.group_m > 0 ? P.group_m : 8;
    const int in_group = GROUP_M * num_pid_n;
    const int group_id = wgid / in_group;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(num_pid_m - first_m, GROUP_M);
    pid_m = first_m + (wgid % in_group) % gsz;
    pid_n = (wgid % in_group) / gsz;
}


}  // namespace
