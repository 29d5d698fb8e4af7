// Row normalisation kernels: DreamLLMRMSNorm (+ fused residual add) and LayerNorm, forward and backward.
// HBM-bound: one 64-lane wave owns one row, 16-byte (8 x bf16) loads, the row is held in registers so x is
// read exactly once; reductions are wave shuffles (no LDS).  Algorithmic traffic: 4 B/element forward
// (2 in + 2 out), 6 B/element with the fused residual (x, res in; h, y out => 8 B/elt).
//
// Reference semantics: omni/models/dreamllm/modeling_dreamllm.py:77-91 (DreamLLMRMSNorm.forward):
//   fp32 upcast -> x * rsqrt(mean(x^2) + eps) -> cast to input dtype -> weight * (.)   (two bf16 roundings).
#include "common.h"

namespace {

template <int MAXV>
struct RowRegs {
    float v[MAXV][8];
};

template <int MAXV>
__device__ __forceinline__ void load_row(const bf16* __restrict__ p, int nv, int lane, RowRegs<MAXV>& r) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
            bf16x8 t = ld_bf16x8(p + (int64_t)vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[i][j] = (float)t[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
        }
    }
}

// ---------------------------------------------------------------- RMSNorm forward
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                          const bf16* __restrict__ w, bf16* __restrict__ h_out,
                                                          bf16* __restrict__ y, float* __restrict__ rstd_out,
                                                          int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 3;
    RowRegs<MAXV> r;
    load_row<MAXV>(x + row * D, nv, lane, r);
    if (res != nullptr) {
        RowRegs<MAXV> q;
        load_row<MAXV>(res + row * D, nv, lane, q);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (bf16)(r.v[i][j] + q.v[i][j]);  // bf16 add rounds (reference: residual + hidden_states)
                r.v[i][j] = (float)o[j];
            }
            if (vi < nv) st_bf16x8(h_out + row * D + (int64_t)vi * 8, o);
        }
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += r.v[i][j] * r.v[i][j];
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    if (lane == 0 && rstd_out != nullptr) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
            bf16x8 wv = ld_bf16x8(w + (int64_t)vi * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bf16 t = (bf16)(r.v[i][j] * rstd);  // .to(input_dtype)
                o[j] = (bf16)((float)wv[j] * (float)t);   // weight * (.)
            }
            st_bf16x8(y + row * D + (int64_t)vi * 8, o);
        }
    }
}

// Block-per-row forward for D = 2048 * VPL and many rows (the training shapes): a row is spread over the 4 waves of a block
// (VPL 16-byte vectors per lane instead of 8 => ~40 VGPRs instead of 151, 8 blocks per CU in flight).  The wave-per-row kernel
// above stays in use for short inputs (decode steps), whose summation order the fused decode GEMV reproduces bit for bit.
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_fwd_block_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                                const bf16* __restrict__ w, bf16* __restrict__ h_out,
                                                                bf16* __restrict__ y, float* __restrict__ rstd_out, int64_t rows,
                                                                int D, float eps) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    float v[VPL][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int64_t off = row * D + (int64_t)(tid + 256 * i) * 8;
        const bf16x8 a = ld_bf16x8(x + off);
        if (res != nullptr) {
            const bf16x8 b = ld_bf16x8(res + off);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = (bf16)((float)a[j] + (float)b[j]);  // bf16 add rounds (reference: residual + hidden_states)
                v[i][j] = (float)o[j];
            }
            st_bf16x8(h_out + off, o);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = (float)a[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    if (tid == 0 && rstd_out != nullptr) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 wv = ld_bf16x8(w + (int64_t)(tid + 256 * i) * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bf16 t = (bf16)(v[i][j] * rstd);   // .to(input_dtype)
            o[j] = (bf16)((float)wv[j] * (float)t);  // weight * (.)
        }
        st_bf16x8(y + row * D + (int64_t)(tid + 256 * i) * 8, o);
    }
}

// ---------------------------------------------------------------- RMSNorm backward
// dx = rstd * (g - xhat * mean(g * xhat)) [+ dh_in],  g = dy * w, xhat = h * rstd;  dw_partial[wave] = sum_rows dy * xhat
template <int MAXV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h,
                                                          const bf16* __restrict__ w, const float* __restrict__ rstd_in,
                                                          const bf16* __restrict__ dh_in, bf16* __restrict__ dx,
                                                          float* __restrict__ dw_partial, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nv = D >> 3;
    RowRegs<MAXV> wreg, acc;
    load_row<MAXV>(w, nv, lane, wreg);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc.v[i][j] = 0.f;

    for (int64_t row = gw; row < rows; row += nwaves) {
        RowRegs<MAXV> g, xh;
        load_row<MAXV>(dy + row * D, nv, lane, g);
        load_row<MAXV>(h + row * D, nv, lane, xh);
        const float rstd = rstd_in[row];
        float c = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh.v[i][j] *= rstd;
                if (dw_partial != nullptr) acc.v[i][j] += g.v[i][j] * xh.v[i][j];
                g.v[i][j] *= wreg.v[i][j];
                c += g.v[i][j] * xh.v[i][j];
            }
        c = wave_sum(c) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
            if (vi < nv) {
                bf16x8 o;
                if (dh_in != nullptr) {
                    bf16x8 d = ld_bf16x8(dh_in + row * D + (int64_t)vi * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (g.v[i][j] - xh.v[i][j] * c) + (float)d[j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (g.v[i][j] - xh.v[i][j] * c));
                }
                st_bf16x8(dx + row * D + (int64_t)vi * 8, o);
            }
        }
    }
    if (dw_partial != nullptr) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
            if (vi < nv) {
                float* p = dw_partial + (int64_t)gw * D + (int64_t)vi * 8;
                *reinterpret_cast<f32x4*>(p) = f32x4{acc.v[i][0], acc.v[i][1], acc.v[i][2], acc.v[i][3]};
                *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc.v[i][4], acc.v[i][5], acc.v[i][6], acc.v[i][7]};
            }
        }
    }
}

// Block-per-row variant for D = 2048 * VPL (the LLM's 4096): a row is spread over the 4 waves of a block (VPL 16-byte vectors
// per lane instead of 8), so the kernel needs ~70 VGPRs instead of 256 and runs 8 blocks per CU -- the wave-per-row kernel above
// holds 4 rows of fp32 state per lane and is limited to one wave per SIMD (2.1 TB/s).  dw_partial row = blockIdx.x.
template <int VPL>
__global__ __launch_bounds__(256) void rmsnorm_bwd_block_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h,
                                                                const bf16* __restrict__ w, const float* __restrict__ rstd_in,
                                                                const bf16* __restrict__ dh_in, bf16* __restrict__ dx,
                                                                float* __restrict__ dw_partial, int64_t rows, int D) {
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float wv[VPL][8], acc[VPL][8];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const bf16x8 t = ld_bf16x8(w + (int64_t)(tid + 256 * i) * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            wv[i][j] = (float)t[j];
            acc[i][j] = 0.f;
        }
    }
    int par = 0;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x, par ^= 1) {
        float g[VPL][8], xh[VPL][8];
        const float rstd = rstd_in[row];
        float c = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const bf16x8 a = ld_bf16x8(dy + row * D + (int64_t)(tid + 256 * i) * 8);
            const bf16x8 b = ld_bf16x8(h + row * D + (int64_t)(tid + 256 * i) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh[i][j] = (float)b[j] * rstd;
                if (dw_partial != nullptr) acc[i][j] += (float)a[j] * xh[i][j];
                g[i][j] = (float)a[j] * wv[i][j];
                c += g[i][j] * xh[i][j];
            }
        }
        c = wave_sum(c);
        if (lane == 0) red[par][wave] = c;  // two slots: the next row's write cannot race this row's reads
        __syncthreads();
        c = (red[par][0] + red[par][1] + red[par][2] + red[par][3]) / (float)D;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int64_t off = row * D + (int64_t)(tid + 256 * i) * 8;
            bf16x8 o;
            if (dh_in != nullptr) {
                const bf16x8 d = ld_bf16x8(dh_in + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (g[i][j] - xh[i][j] * c) + (float)d[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (g[i][j] - xh[i][j] * c));
            }
            st_bf16x8(dx + off, o);
        }
    }
    if (dw_partial != nullptr) {
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            float* p = dw_partial + (int64_t)blockIdx.x * D + (int64_t)(tid + 256 * i) * 8;
            *reinterpret_cast<f32x4*>(p) = f32x4{acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
            *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[i][4], acc[i][5], acc[i][6], acc[i][7]};
        }
    }
}

// out[c] = sum_p partial[p][c]; 32 columns per block, 32 row groups (1024 threads) reduced through LDS in a fixed order.  Round 3:
// was 64 columns x 4 row groups per 256-thread block = 64 blocks with 512 dependent-address loads per thread for the LLM's
// [2048, 4096] partials (94 us, 65 calls per training step); 128 blocks x 1024 threads keep 8x more loads in flight.
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ partial, void* __restrict__ out,
                                                               int nparts, int D, int out_dtype) {
    __shared__ float red[32][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + col;
    float s0 = 0.f, s1 = 0.f;
    if (c < D) {
        int p = rg;
        for (; p + 32 < nparts; p += 64) {   // two independent chains per thread
            s0 += partial[(int64_t)p * D + c];
            s1 += partial[(int64_t)(p + 32) * D + c];
        }
        if (p < nparts) s0 += partial[(int64_t)p * D + c];
    }
    red[rg][col] = s0 + s1;
    __syncthreads();
    if (rg == 0 && c < D) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][col];
        if (out_dtype == DLLM_BF16)
            reinterpret_cast<bf16*>(out)[c] = (bf16)s;
        else
            reinterpret_cast<float*>(out)[c] = s;
    }
}

// ---------------------------------------------------------------- LayerNorm forward / backward
// y = (x - mean) * rstd * w + b, statistics in fp32 (torch.nn.LayerNorm semantics for bf16 inputs).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                            const bf16* __restrict__ b, bf16* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int64_t rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = D >> 3;
    RowRegs<MAXV> r;
    load_row<MAXV>(x + row * D, nv, lane, r);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r.v[i][j];
    const float mean = wave_sum(s) / (float)D;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = r.v[i][j] - mean;
                ss += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    if (lane == 0 && mean_out != nullptr) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nv) {
            bf16x8 wv = ld_bf16x8(w + (int64_t)vi * 8);
            bf16x8 bv = (b != nullptr) ? ld_bf16x8(b + (int64_t)vi * 8) : zero_bf16x8();
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((r.v[i][j] - mean) * rstd * (float)wv[j] + (float)bv[j]);
            st_bf16x8(y + row * D + (int64_t)vi * 8, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*w.  dw/db partials optional (frozen weights pass null).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                            const bf16* __restrict__ w, const float* __restrict__ mean_in,
                                                            const float* __restrict__ rstd_in, bf16* __restrict__ dx,
                                                            float* __restrict__ dw_partial, float* __restrict__ db_partial,
                                                            int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nv = D >> 3;
    RowRegs<MAXV> wreg, accw, accb;
    load_row<MAXV>(w, nv, lane, wreg);
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            accw.v[i][j] = 0.f;
            accb.v[i][j] = 0.f;
        }
    for (int64_t row = gw; row < rows; row += nwaves) {
        RowRegs<MAXV> g, xh;
        load_row<MAXV>(dy + row * D, nv, lane, g);
        load_row<MAXV>(x + row * D, nv, lane, xh);
        const float mean = mean_in[row], rstd = rstd_in[row];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                xh.v[i][j] = (vi < nv) ? (xh.v[i][j] - mean) * rstd : 0.f;
                if (dw_partial != nullptr) {
                    accw.v[i][j] += g.v[i][j] * xh.v[i][j];
                    accb.v[i][j] += g.v[i][j];
                }
                g.v[i][j] *= wreg.v[i][j];
                c1 += g.v[i][j];
                c2 += g.v[i][j] * xh.v[i][j];
            }
        }
        c1 = wave_sum(c1) / (float)D;
        c2 = wave_sum(c2) / (float)D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
            if (vi < nv) {
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (bf16)(rstd * (g.v[i][j] - c1 - xh.v[i][j] * c2));
                st_bf16x8(dx + row * D + (int64_t)vi * 8, o);
            }
        }
    }
    if (dw_partial != nullptr) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int vi = lane + i * 64;
            if (vi < nv) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dw_partial[(int64_t)gw * D + (int64_t)vi * 8 + j] = accw.v[i][j];
                    db_partial[(int64_t)gw * D + (int64_t)vi * 8 + j] = accb.v[i][j];
                }
            }
        }
    }
}

inline int pick_maxv(int D) {
    if (D % 8 != 0 || D <= 0) return -1;
    const int nv = D / 8;
    if (nv <= 128) return 2;
    if (nv <= 256) return 4;
    if (nv <= 512) return 8;
    if (nv <= 1024) return 16;
    return -1;
}

}  // namespace

#define DISPATCH_MAXV(mv, KERNEL, grid, block, stream, ...)                                   \
    switch (mv) {                                                                             \
        case 2: KERNEL<2><<<grid, block, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;       \
        case 4: KERNEL<4><<<grid, block, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;       \
        case 8: KERNEL<8><<<grid, block, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;       \
        case 16: KERNEL<16><<<grid, block, 0, (hipStream_t)stream>>>(__VA_ARGS__); break;     \
        default: return DLLM_ERR_SHAPE;                                                       \
    }

extern "C" {

// Number of fp32 partial rows the backward kernels write (workspace = nparts * D floats).
int dllm_norm_bwd_nparts(int64_t rows) {
    int64_t blocks = (rows + 3) / 4;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    return (int)(blocks * 4);
}

int dllm_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd, int64_t rows, int D,
                     float eps, void* stream) {
    const int mv = pick_maxv(D);
    if (mv < 0 || rows < 0) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    if (res != nullptr && h_out == nullptr) return DLLM_ERR_SHAPE;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (rows >= 256 && (D % 2048) == 0 && D <= 8192) {  // training shapes: block per row
        const dim3 g2((unsigned)rows);
        hipStream_t s = (hipStream_t)stream;
        const bf16 *xp = (const bf16*)x, *rp = (const bf16*)res, *wp = (const bf16*)w;
        switch (D / 2048) {
            case 1: hipLaunchKernelGGL(rmsnorm_fwd_block_kernel<1>, g2, block, 0, s, xp, rp, wp, (bf16*)h_out, (bf16*)y, rstd, rows, D, eps); break;
            case 2: hipLaunchKernelGGL(rmsnorm_fwd_block_kernel<2>, g2, block, 0, s, xp, rp, wp, (bf16*)h_out, (bf16*)y, rstd, rows, D, eps); break;
            case 3: hipLaunchKernelGGL(rmsnorm_fwd_block_kernel<3>, g2, block, 0, s, xp, rp, wp, (bf16*)h_out, (bf16*)y, rstd, rows, D, eps); break;
            default: hipLaunchKernelGGL(rmsnorm_fwd_block_kernel<4>, g2, block, 0, s, xp, rp, wp, (bf16*)h_out, (bf16*)y, rstd, rows, D, eps); break;
        }
        return dllm_check_launch();
    }
    DISPATCH_MAXV(mv, rmsnorm_fwd_kernel, grid, block, stream, (const bf16*)x,
                                          (const bf16*)res, (const bf16*)w, (bf16*)h_out, (bf16*)y, rstd, rows, D, eps);
    return dllm_check_launch();
}

// dw_partial: fp32 [dllm_norm_bwd_nparts(rows), D] workspace or NULL (frozen weight); dw_out: final [D] (dtype flag).
int dllm_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dh_in, void* dx,
                     float* dw_partial, void* dw_out, int dw_dtype, int64_t rows, int D, void* stream) {
    const int mv = pick_maxv(D);
    if (mv < 0 || rows < 0) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    const int nparts = dllm_norm_bwd_nparts(rows);
    dim3 grid(nparts / 4), block(256);
    if ((D % 2048) == 0 && D <= 8192) {  // block per row: one partial row per block, nparts blocks
        const dim3 g2(nparts);
        hipStream_t s = (hipStream_t)stream;
        const bf16 *dyp = (const bf16*)dy, *hp = (const bf16*)h, *wp = (const bf16*)w, *dhp = (const bf16*)dh_in;
        switch (D / 2048) {
            case 1: hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<1>, g2, block, 0, s, dyp, hp, wp, rstd, dhp, (bf16*)dx, dw_partial, rows, D); break;
            case 2: hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<2>, g2, block, 0, s, dyp, hp, wp, rstd, dhp, (bf16*)dx, dw_partial, rows, D); break;
            case 3: hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<3>, g2, block, 0, s, dyp, hp, wp, rstd, dhp, (bf16*)dx, dw_partial, rows, D); break;
            default: hipLaunchKernelGGL(rmsnorm_bwd_block_kernel<4>, g2, block, 0, s, dyp, hp, wp, rstd, dhp, (bf16*)dx, dw_partial, rows, D); break;
        }
    } else {
        DISPATCH_MAXV(mv, rmsnorm_bwd_kernel, grid, block, stream, (const bf16*)dy,
                                              (const bf16*)h, (const bf16*)w, rstd, (const bf16*)dh_in, (bf16*)dx, dw_partial,
                                              rows, D);
    }
    if (dw_partial != nullptr && dw_out != nullptr) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 31) / 32), dim3(1024), 0, (hipStream_t)stream, dw_partial, dw_out,
                           nparts, D, dw_dtype);
    }
    return dllm_check_launch();
}

int dllm_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows, int D,
                       float eps, void* stream) {
    const int mv = pick_maxv(D);
    if (mv < 0 || rows < 0) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    if ((mean == nullptr) != (rstd == nullptr)) return DLLM_ERR_SHAPE;
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    DISPATCH_MAXV(mv, layernorm_fwd_kernel, grid, block, stream, (const bf16*)x,
                                          (const bf16*)w, (const bf16*)b, (bf16*)y, mean, rstd, rows, D, eps);
    return dllm_check_launch();
}

int dllm_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                       float* dw_partial, float* db_partial, void* dw_out, void* db_out, int dw_dtype, int64_t rows, int D,
                       void* stream) {
    const int mv = pick_maxv(D);
    if (mv < 0 || rows < 0) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    if ((dw_partial == nullptr) != (db_partial == nullptr)) return DLLM_ERR_SHAPE;
    const int nparts = dllm_norm_bwd_nparts(rows);
    dim3 grid(nparts / 4), block(256);
    DISPATCH_MAXV(mv, layernorm_bwd_kernel, grid, block, stream, (const bf16*)dy,
                                          (const bf16*)x, (const bf16*)w, mean, rstd, (bf16*)dx, dw_partial, db_partial, rows,
                                          D);
    if (dw_partial != nullptr && dw_out != nullptr) {
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 31) / 32), dim3(1024), 0, (hipStream_t)stream, dw_partial, dw_out,
                           nparts, D, dw_dtype);
        hipLaunchKernelGGL(colsum_partials_kernel, dim3((D + 31) / 32), dim3(1024), 0, (hipStream_t)stream, db_partial, db_out,
                           nparts, D, dw_dtype);
    }
    return dllm_check_launch();
}

}  // extern "C"
