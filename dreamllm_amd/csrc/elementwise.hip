// HBM-bound elementwise / gather / loss / optimizer kernels of the DreamLLM hot path (gfx950).
// All use 16-byte (8 x bf16) accesses, fp32 math, one rounding on store.
#include "common.h"

namespace {

// ------------------------------------------------------------------ RoPE (modeling_dreamllm.py:176-209)
// x: [T tokens][NH heads][D] view (token stride ts, head stride hs, d contiguous), rotated in place:
//   y1 = x1*cos - x2*sin ; y2 = x2*cos + x1*sin   with (x1, x2) the two halves of the head dim ("rotate_half").
// cs: fp32 [max_pos][D/2] cos table, sn likewise; pos: int64 [T] position ids or null (=> token index % S).
// sign = -1 gives the backward (transpose rotation).
__global__ __launch_bounds__(256) void rope_kernel(bf16* __restrict__ x, const float* __restrict__ cs,
                                                   const float* __restrict__ sn, const int64_t* __restrict__ pos, int64_t T,
                                                   int S, int NH, int D, int64_t ts, int64_t hs, float sign) {
    const int half = D >> 1, vph = half >> 3;  // 8-element vectors per half
    const int64_t total = T * NH * vph;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i % vph);
        const int64_t th = i / vph;
        const int hh = (int)(th % NH);
        const int64_t tok = th / NH;
        const int64_t p = pos ? pos[tok] : (tok % S);
        bf16* p1 = x + tok * ts + (int64_t)hh * hs + j * 8;
        bf16* p2 = p1 + half;
        const bf16x8 a = ld_bf16x8(p1), b = ld_bf16x8(p2);
        const float* c = cs + p * half + j * 8;
        const float* s = sn + p * half + j * 8;
        bf16x8 o1, o2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cc = c[e], ss = s[e] * sign;
            float y1, y2;
            rope_pair((float)a[e], (float)b[e], cc, ss, y1, y2);
            o1[e] = (bf16)y1;
            o2[e] = (bf16)y2;
        }
        st_bf16x8(p1, o1);
        st_bf16x8(p2, o2);
    }
}

// ------------------------------------------------------------------ SwiGLU / GEGLU
// MODE 0: out = silu(a) * b   (DreamLLMMLP, modeling_dreamllm.py:237)     a = gate, b = up
// MODE 1: out = b_ * gelu(a)  with a = gate half, b = value half (diffusers GEGLU: hidden, gate = chunk(2); hidden*gelu(gate))
// NT: non-temporal loads / stores (bit 8 of the ABI's `mode`): the operands are [T, F] streams of 0.7 GB each at the LLM shape,
// read or written exactly once by this kernel
template <bool NT>
__device__ __forceinline__ bf16x8 ldv(const bf16* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
    else return ld_bf16x8(p);
}
template <bool NT>
__device__ __forceinline__ void stv(bf16* p, bf16x8 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<bf16x8*>(p));
    else st_bf16x8(p, v);
}

template <int MODE, bool NT = false>
__global__ __launch_bounds__(256) void glu_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                      bf16* __restrict__ out, int64_t M, int F, int64_t lda, int64_t ldb,
                                                      int64_t ldo) {
    // one block per row (grid-stride over rows), threads over the 16-byte vectors of the row: no 64-bit div/mod per vector
    const int vpr = F >> 3;
    for (int64_t r = blockIdx.x; r < M; r += gridDim.x)
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        const int c = v * 8;
        const bf16x8 av = ldv<NT>(a + r * lda + c), bv = ldv<NT>(b + r * ldb + c);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = (float)av[e], y = (float)bv[e];
            o[e] = (bf16)(MODE == 0 ? swiglu_fwd_elem(x, y) : gelu_erf_f(x) * y);
        }
        stv<NT>(out + r * ldo + c, o);
    }
}
template <int MODE, bool NT = false>
__global__ __launch_bounds__(256) void glu_bwd_kernel(const bf16* __restrict__ dout, const bf16* __restrict__ a,
                                                      const bf16* __restrict__ b, bf16* __restrict__ da, bf16* __restrict__ db,
                                                      bf16* __restrict__ act_out, int64_t M, int F, int64_t ldd, int64_t lda,
                                                      int64_t ldb, int64_t ldda, int64_t lddb, int64_t ldact) {
    const int vpr = F >> 3;
    for (int64_t r = blockIdx.x; r < M; r += gridDim.x)
    for (int v = threadIdx.x; v < vpr; v += blockDim.x) {
        const int c = v * 8;
        const bf16x8 dv = ldv<NT>(dout + r * ldd + c), av = ldv<NT>(a + r * lda + c), bv = ldv<NT>(b + r * ldb + c);
        bf16x8 oa, ob, oc;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)dv[e], x = (float)av[e], y = (float)bv[e];
            float act = 0.f;
            if constexpr (MODE == 0) {   // fixed operation order shared with the GEMM epilogues (common.h)
                float dg, du;
                swiglu_bwd_elem(d, x, y, dg, du);
                oa[e] = (bf16)dg;
                ob[e] = (bf16)du;
            } else {
                act = gelu_erf_f(x);
                const float dact = gelu_erf_grad_f(x);
                oa[e] = (bf16)(d * y * dact);
                ob[e] = (bf16)(d * act);
            }
            // the forward product, recomputed in the same pass for the down-projection's weight gradient: bit-identical to
            // glu_fwd_kernel (same silu_f / gelu_erf_f expression, one rounding)
            oc[e] = (bf16)(MODE == 0 ? swiglu_fwd_elem(x, y) : act * y);
        }
        stv<NT>(da + r * ldda + c, oa);
        stv<NT>(db + r * lddb + c, ob);
        if (act_out != nullptr) stv<NT>(act_out + r * ldact + c, oc);
    }
}

// ------------------------------------------------------------------ row gather / scatter (embedding + multimodal splice)
// out[i,:] = table[idx[i],:]
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16* __restrict__ table, const int64_t* __restrict__ idx,
                                                          bf16* __restrict__ out, int64_t n, int D, int64_t ld_t,
                                                          int64_t ld_o) {
    const int vpr = D >> 3;
    const int64_t total = n * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vpr;
        const int c = (int)(i % vpr) * 8;
        st_bf16x8(out + r * ld_o + c, ld_bf16x8(table + idx[r] * ld_t + c));
    }
}
// dst[idx[i],:] = src[i,:]   (idx unique)
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16* __restrict__ src, const int64_t* __restrict__ idx,
                                                           bf16* __restrict__ dst, int64_t n, int D, int64_t ld_s,
                                                           int64_t ld_d) {
    const int vpr = D >> 3;
    const int64_t total = n * vpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / vpr;
        const int c = (int)(i % vpr) * 8;
        st_bf16x8(dst + idx[r] * ld_d + c, ld_bf16x8(src + r * ld_s + c));
    }
}
// Embedding backward, deterministic: rows sorted by token id; seg_start[u]..seg_start[u+1] are the positions (into
// `order`; order == null: the rows themselves) of unique id uid[u].  dtable[uid[u],:] = sum_{j in segment} dy[order[j],:]  (fp32 accumulation).
// FIN / FOUT: fp32 rows in / out (round 6: a segment of ~10^4 positions -- one block, ~25 GB/s -- took 2.7 ms of every training step; the
// caller now cuts long segments into chunks, sums the chunks into fp32 partial rows here and the partial rows of a token in a second
// call: dllm_segment_sum_rows_ex).
template <bool FIN>
__device__ __forceinline__ void seg_load8(const void* base, int64_t row, int64_t ld, int v, float (&d)[8]) {
    if constexpr (FIN) {
        const float* p = reinterpret_cast<const float*>(base) + row * ld + v * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            d[e] = a[e];
            d[e + 4] = b[e];
        }
    } else {
        const bf16x8 t = ld_bf16x8(reinterpret_cast<const bf16*>(base) + row * ld + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[e] = (float)t[e];
    }
}
template <bool FIN, bool FOUT>
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const void* __restrict__ dy, const int64_t* __restrict__ order,
                                                               const int64_t* __restrict__ seg_start,
                                                               const int64_t* __restrict__ uid, void* __restrict__ dtable,
                                                               int64_t nuniq, int D, int64_t ld_dy, int64_t ld_t) {
    const int vpr = D >> 3;
    for (int64_t u = blockIdx.x; u < nuniq; u += gridDim.x) {
        const int64_t s0 = seg_start[u], s1 = seg_start[u + 1];
        const int64_t orow = uid ? uid[u] : u;
        for (int v = threadIdx.x; v < vpr; v += 256) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int64_t j = s0;
            // 8 rows at a time: 8 independent index loads, then 8 independent row loads in flight, summed in the original order (deterministic)
            for (; j + 8 <= s1; j += 8) {
                int64_t idx[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) idx[q] = order ? order[j + q] : j + q;
                float d[8][8];
#pragma unroll
                for (int q = 0; q < 8; ++q) seg_load8<FIN>(dy, idx[q], ld_dy, v, d[q]);
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] += d[q][e];
            }
            for (; j < s1; ++j) {
                float d[8];
                seg_load8<FIN>(dy, order ? order[j] : j, ld_dy, v, d);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += d[e];
            }
            if constexpr (FOUT) {
                float* p = reinterpret_cast<float*>(dtable) + orow * ld_t + v * 8;
                *reinterpret_cast<f32x4*>(p) = f32x4{acc[0], acc[1], acc[2], acc[3]};
                *reinterpret_cast<f32x4*>(p + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
            } else {
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (bf16)acc[e];
                st_bf16x8(reinterpret_cast<bf16*>(dtable) + orow * ld_t + v * 8, o);
            }
        }
    }
}

// Row softmax of fp32 scores -> bf16 probabilities (the single-head 512-wide VAE mid-block attention, whose head_dim is outside
// the flash kernels: scores come from the GEMM kernel in fp32, probabilities go back into it as the A operand of P V).
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, bf16* __restrict__ y, int cols, int64_t ld_x,
                                                           int64_t ld_y) {
    __shared__ float scratch[4];
    const float* row = x + (int64_t)blockIdx.x * ld_x;
    bf16* out = y + (int64_t)blockIdx.x * ld_y;
    float mx = -INFINITY;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        if (c + 3 < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        } else {
            for (int e = c; e < cols; ++e) mx = fmaxf(mx, row[e]);
        }
    }
    mx = block_max<4>(mx, scratch);
    float se = 0.f;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        if (c + 3 < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            se += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
        } else {
            for (int e = c; e < cols; ++e) se += __expf(row[e] - mx);
        }
    }
    se = block_sum<4>(se, scratch);
    const float inv = 1.0f / se;
    for (int c = threadIdx.x * 4; c < cols; c += 1024) {
        if (c + 3 < cols) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)(__expf(v[e] - mx) * inv);
            st_bf16x4(out + c, o);
        } else {
            for (int e = c; e < cols; ++e) out[e] = (bf16)(__expf(row[e] - mx) * inv);
        }
    }
}

// ------------------------------------------------------------------ softmax cross-entropy over fp32 logits
// (modeling_dreamllm.py:1453-1470: logits.float(), CrossEntropyLoss(reduction="none"), mean over labels != -100).
// One block per row.  loss_row[r] = lse - logit[label] (0 for ignored rows);  dlogits (bf16, optional) =
// (softmax - onehot) * gscale for valid rows, 0 for ignored rows, where gscale = dloss / n_valid is read from a
// device scalar so no host sync is needed.
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels,
                                                         float* __restrict__ loss_row, bf16* __restrict__ dlogits,
                                                         const float* __restrict__ gscale_ptr, int V, int64_t ld_l,
                                                         int64_t ld_d) {
    __shared__ float scratch[4];
    const int64_t r = blockIdx.x;
    const float* row = logits + r * ld_l;
    const int64_t label = labels[r];
    const bool valid = label >= 0 && label < V;
    if (!valid && dlogits == nullptr) {
        if (threadIdx.x == 0) loss_row[r] = 0.f;
        return;
    }
    float mx = -INFINITY;
    if (valid || true) {
        for (int c = threadIdx.x * 4; c < V; c += 1024) {
            if (c + 3 < V) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
                mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            } else {
                for (int e = c; e < V; ++e) mx = fmaxf(mx, row[e]);
            }
        }
    }
    mx = block_max<4>(mx, scratch);
    float se = 0.f;
    for (int c = threadIdx.x * 4; c < V; c += 1024) {
        if (c + 3 < V) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
            se += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
        } else {
            for (int e = c; e < V; ++e) se += __expf(row[e] - mx);
        }
    }
    se = block_sum<4>(se, scratch);
    const float lse = mx + logf(se);
    if (threadIdx.x == 0) loss_row[r] = valid ? (lse - row[label]) : 0.f;
    if (dlogits != nullptr) {
        const float gs = valid ? gscale_ptr[0] : 0.f;
        bf16* drow = dlogits + r * ld_d;
        for (int c = threadIdx.x * 4; c < V; c += 1024) {
            if (c + 3 < V) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(row + c);
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float p = __expf(v[e] - lse);
                    if (c + e == label) p -= 1.f;
                    o[e] = (bf16)(p * gs);
                }
                st_bf16x4(drow + c, o);
            } else {
                for (int e = c; e < V; ++e) {
                    float p = __expf(row[e] - lse);
                    if (e == label) p -= 1.f;
                    drow[e] = (bf16)(p * gs);
                }
            }
        }
    }
}

// ------------------------------------------------------------------ AdamW (decoupled weight decay), fp32 math
// p, g: bf16 or fp32 by flag; m, v: same dtype as given by state_f32.  Matches torch.optim.AdamW(fused) update:
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
template <typename PT, typename ST>
__global__ __launch_bounds__(256) void adamw_kernel(PT* __restrict__ p, const PT* __restrict__ g, ST* __restrict__ m,
                                                    ST* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float bc2, float gscale,
                                                    const float* __restrict__ gscale_dev) {
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];  // e.g. the clip coefficient, kept on device (no host sync)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        float pf = (float)p[i];
        const float gf = (float)g[i] * gscale;
        float mf = (float)m[i], vf = (float)v[i];
        pf *= 1.f - lr * wd;
        mf = b1 * mf + (1.f - b1) * gf;
        vf = b2 * vf + (1.f - b2) * gf * gf;
        const float denom = sqrtf(vf) / sqrtf(bc2) + eps;
        pf -= (lr / bc1) * mf / denom;
        p[i] = (PT)pf;
        m[i] = (ST)mf;
        v[i] = (ST)vf;
    }
}

// bf16 parameters / gradients / moments, 16-byte aligned, n % 8 == 0 (every weight matrix): 8 elements per thread per access,
// all four streams requested before the arithmetic, non-temporal (each array is touched once per step: 14 B / parameter,
// 94.6 GB for the 7B model -- nothing of it is worth a cache line).  Same arithmetic, element for element, as adamw_kernel.
__global__ __launch_bounds__(256) void adamw_vec8_kernel(bf16* __restrict__ p, const bf16* __restrict__ g, bf16* __restrict__ m,
                                                         bf16* __restrict__ v, int64_t n8, float lr, float b1, float b2, float eps,
                                                         float wd, float bc1, float bc2, float gscale,
                                                         const float* __restrict__ gscale_dev) {
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];
    const float decay = 1.f - lr * wd;
    bf16x8* pv = reinterpret_cast<bf16x8*>(p);
    const bf16x8* gv = reinterpret_cast<const bf16x8*>(g);
    bf16x8* mv = reinterpret_cast<bf16x8*>(m);
    bf16x8* vv = reinterpret_cast<bf16x8*>(v);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        bf16x8 P = __builtin_nontemporal_load(pv + i);
        const bf16x8 G = __builtin_nontemporal_load(gv + i);
        bf16x8 M = __builtin_nontemporal_load(mv + i), V = __builtin_nontemporal_load(vv + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float pf = (float)P[e];
            const float gf = (float)G[e] * gscale;
            float mf = (float)M[e], vf = (float)V[e];
            pf *= decay;
            mf = b1 * mf + (1.f - b1) * gf;
            vf = b2 * vf + (1.f - b2) * gf * gf;
            const float denom = sqrtf(vf) / sqrtf(bc2) + eps;
            pf -= (lr / bc1) * mf / denom;
            P[e] = (bf16)pf;
            M[e] = (bf16)mf;
            V[e] = (bf16)vf;
        }
        __builtin_nontemporal_store(P, pv + i);
        __builtin_nontemporal_store(M, mv + i);
        __builtin_nontemporal_store(V, vv + i);
    }
}

// ---- multi-tensor forms (round 4): the 7B step ran 295 AdamW + 295 sum-of-squares launches (one per parameter tensor) ------------
// One launch serves up to MT_MAX tensors: the table of pointers / sizes travels as a kernel argument (no device-side table, no
// host-to-device copy), block b owns chunk b of the concatenated chunk space (MT_CHUNK elements of ONE tensor per chunk), found by
// a scan of the table's chunk prefix.  Same arithmetic, element for element, as adamw_vec8_kernel / sumsq_kernel's per-element
// terms; the sum-of-squares partial of a chunk is a fixed-order block reduction, one float per chunk, summed later in chunk order
// (dllm_reduce_sum_f32): deterministic, identical on every data-parallel replica.
constexpr int MT_MAX = 48;
constexpr int MT_CHUNK = 256 * 8 * 16;   // elements per block: 16 vectors of 8 per thread (458 KiB of AdamW traffic)
struct MtAdamTable {
    bf16* p[MT_MAX];
    const bf16* g[MT_MAX];
    bf16* m[MT_MAX];
    bf16* v[MT_MAX];
    int64_t n8[MT_MAX];         // vectors of 8 elements
    int chunk_end[MT_MAX];      // exclusive prefix end of each tensor's chunks inside this launch
    int count;
};
struct MtSumsqTable {
    const bf16* x[MT_MAX];
    int64_t n8[MT_MAX];
    int chunk_end[MT_MAX];
    int count;
};

__global__ __launch_bounds__(256) void adamw_multi_kernel(MtAdamTable T, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                          float bc2, float gscale, const float* __restrict__ gscale_dev) {
    int t = 0;
    while (t < T.count - 1 && (int)blockIdx.x >= T.chunk_end[t]) ++t;
    const int c0 = t == 0 ? 0 : T.chunk_end[t - 1];
    const int64_t base = (int64_t)((int)blockIdx.x - c0) * (MT_CHUNK / 8);
    const int64_t n8 = T.n8[t];
    if (gscale_dev != nullptr) gscale *= gscale_dev[0];
    const float decay = 1.f - lr * wd;
    bf16x8* pv = reinterpret_cast<bf16x8*>(T.p[t]);
    const bf16x8* gv = reinterpret_cast<const bf16x8*>(T.g[t]);
    bf16x8* mv = reinterpret_cast<bf16x8*>(T.m[t]);
    bf16x8* vv = reinterpret_cast<bf16x8*>(T.v[t]);
    const int64_t end = min(n8, base + MT_CHUNK / 8);
    for (int64_t i = base + threadIdx.x; i < end; i += 256) {
        bf16x8 P = __builtin_nontemporal_load(pv + i);
        const bf16x8 G = __builtin_nontemporal_load(gv + i);
        bf16x8 M = __builtin_nontemporal_load(mv + i), V = __builtin_nontemporal_load(vv + i);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float pf = (float)P[e];
            const float gf = (float)G[e] * gscale;
            float mf = (float)M[e], vf = (float)V[e];
            pf *= decay;
            mf = b1 * mf + (1.f - b1) * gf;
            vf = b2 * vf + (1.f - b2) * gf * gf;
            const float denom = sqrtf(vf) / sqrtf(bc2) + eps;
            pf -= (lr / bc1) * mf / denom;
            P[e] = (bf16)pf;
            M[e] = (bf16)mf;
            V[e] = (bf16)vf;
        }
        __builtin_nontemporal_store(P, pv + i);
        __builtin_nontemporal_store(M, mv + i);
        __builtin_nontemporal_store(V, vv + i);
    }
}

__global__ __launch_bounds__(256) void sumsq_multi_kernel(MtSumsqTable T, float* __restrict__ partials) {
    __shared__ float scratch[4];
    int t = 0;
    while (t < T.count - 1 && (int)blockIdx.x >= T.chunk_end[t]) ++t;
    const int c0 = t == 0 ? 0 : T.chunk_end[t - 1];
    const int64_t base = (int64_t)((int)blockIdx.x - c0) * (MT_CHUNK / 8);
    const int64_t end = min(T.n8[t], base + MT_CHUNK / 8);
    const bf16x8* xv = reinterpret_cast<const bf16x8*>(T.x[t]);
    float s = 0.f;
    for (int64_t i = base + threadIdx.x; i < end; i += 1024) {   // four 16-byte loads in flight per thread
        bf16x8 a = xv[i], b = zero_bf16x8(), c = zero_bf16x8(), d = zero_bf16x8();
        if (i + 256 < end) b = xv[i + 256];
        if (i + 512 < end) c = xv[i + 512];
        if (i + 768 < end) d = xv[i + 768];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float fa = (float)a[e], fb = (float)b[e], fc = (float)c[e], fd = (float)d[e];
            s += fa * fa + fb * fb + fc * fc + fd * fd;
        }
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// sum of squares of a bf16/fp32 buffer (grad-norm clipping) as one partial per block: no atomics => bit-identical on every
// DDP replica.  HBM-bound (2 B/elt): 1024-thread blocks, 16-byte loads, 4 of them in flight per thread (64 KiB per CU).
template <typename T>
__global__ __launch_bounds__(1024) void sumsq_kernel(const T* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float scratch[16];
    constexpr int VE = 16 / (int)sizeof(T);  // elements per 16-byte vector
    typedef T vec_t __attribute__((ext_vector_type(VE)));
    float s = 0.f;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const int64_t nv = aligned ? n / VE : 0;
    const vec_t* xv = reinterpret_cast<const vec_t*>(x);
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const vec_t a = xv[i], b = xv[i + stride], c = xv[i + 2 * stride], d = xv[i + 3 * stride];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float fa = (float)a[e], fb = (float)b[e], fc = (float)c[e], fd = (float)d[e];
            s += fa * fa + fb * fb + fc * fc + fd * fd;
        }
    }
    for (; i < nv; i += stride) {
        const vec_t a = xv[i];
#pragma unroll
        for (int e = 0; e < VE; ++e) s += (float)a[e] * (float)a[e];
    }
    for (int64_t j = nv * VE + (int64_t)blockIdx.x * 1024 + threadIdx.x; j < n; j += stride) {  // tail / unaligned buffer
        const float v = (float)x[j];
        s += v * v;
    }
    s = block_sum<16>(s, scratch);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// out[0] = sum(in[0..n)) in a fixed order (single block): final stage of the deterministic reductions
__global__ __launch_bounds__(256) void reduce_sum_f32_kernel(const float* __restrict__ in, int64_t n, float* __restrict__ out) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += in[i];
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

// mean((a - b)^2): per-block partial sums partials[block] = sum (a-b)^2 over the block's grid-stride elements (the caller adds them up in
// index order: dllm_reduce_sum_f32 -- no float atomics, bit-identical run to run) ; dgrad: da = 2 (a - b) * gscale[0] / n
__global__ __launch_bounds__(256) void mse_kernel(const bf16* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                  float* __restrict__ partials) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = (float)a[i] - b[i];
        s += d * d;
    }
    s = block_sum<4>(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void mse_bwd_kernel(const bf16* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                      const float* __restrict__ gscale, bf16* __restrict__ da) {
    const float gs = 2.f * gscale[0] / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        da[i] = (bf16)(((float)a[i] - b[i]) * gs);
}

// ------------------------------------------------------------------ fused classifier-free guidance + DDIM(eta=0) step
// One thread per latent pixel (4 channels, NHWC).  Replaces modeling_plugins.py:824-833 (chunk, CFG combine,
// scheduler.step) and the next iteration's torch.cat([latents]*2) + cast (:811-812):
//   eps = e_u + s (e_c - e_u);  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t)  [epsilon]  or  sqrt(a_t) x - sqrt(1-a_t) v  [v-pred]
//   x_prev = sqrt(a_prev) x0 + sqrt(1-a_prev) eps
// pred: bf16 [2B][P][4] (uncond batch first), lat: fp32 [B][P][4] in/out, next_in: bf16 [2B][P][8] (channels 4..7 zero,
// the conv_in operand of the next step) or null.
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const bf16* __restrict__ pred, float* __restrict__ lat,
                                                       bf16* __restrict__ next_in, int64_t BP, int64_t total_half, float gs,
                                                       float sa, float s1a, float sap, float s1ap, int vpred) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < BP; i += (int64_t)gridDim.x * 256) {
        const bf16x4 eu = ld_bf16x4(pred + i * 4), ec = ld_bf16x4(pred + (total_half + i) * 4);
        f32x4 x = *reinterpret_cast<f32x4*>(lat + i * 4);
        bf16x8 o = zero_bf16x8();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float u = (float)eu[c];
            float e = u + gs * ((float)ec[c] - u);
            float x0;
            if (vpred) {
                x0 = sa * x[c] - s1a * e;
                e = sa * e + s1a * x[c];
            } else {
                x0 = (x[c] - s1a * e) / sa;
            }
            x[c] = sap * x0 + s1ap * e;
            o[c] = (bf16)x[c];
        }
        *reinterpret_cast<f32x4*>(lat + i * 4) = x;
        if (next_in != nullptr) {
            st_bf16x8(next_in + i * 8, o);
            st_bf16x8(next_in + (total_half + i) * 8, o);
        }
    }
}

// ------------------------------------------------------------------ add / broadcast add / stand-alone activations
// out[i] = a[i] + b[i % period]  (period == n: plain add).  Used for residual adds outside a GEMM epilogue, the CLIP
// position embedding and the per-(image,channel) time-embedding add of the UNet ResBlock (with row_period).
__global__ __launch_bounds__(256) void add_bcast_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                        bf16* __restrict__ out, int64_t nvec, int64_t period_vec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 x = ld_bf16x8(a + i * 8), y = ld_bf16x8(b + (i % period_vec) * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)x[e] + (float)y[e]);
        st_bf16x8(out + i * 8, o);
    }
}
// out[n, r, c] = a[n, r, c] + b[n, c]   (rows_per_group rows share one b row): UNet time-embedding add on NHWC.
__global__ __launch_bounds__(256) void add_rowgroup_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b,
                                                           bf16* __restrict__ out, int64_t nvec, int vec_per_row,
                                                           int64_t rows_per_group) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / vec_per_row;
        const int c = (int)(i % vec_per_row);
        const int64_t grp = row / rows_per_group;
        const bf16x8 x = ld_bf16x8(a + i * 8), y = ld_bf16x8(b + (grp * vec_per_row + c) * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)x[e] + (float)y[e]);
        st_bf16x8(out + i * 8, o);
    }
}
template <int MODE>
__device__ __forceinline__ float act_f(float x) {
    return MODE == 1 ? gelu_erf_f(x) : (MODE == 2 ? quick_gelu_f(x) : silu_f(x));
}
template <int MODE>
__device__ __forceinline__ float act_grad_f(float x) {
    if (MODE == 1) return gelu_erf_grad_f(x);
    if (MODE == 2) {
        const float s = sigmoid_f(1.702f * x);
        return s + 1.702f * x * s * (1.f - s);
    }
    const float s = sigmoid_f(x);
    return s * (1.f + x * (1.f - s));
}
template <int MODE>
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = ld_bf16x8(x + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)act_f<MODE>((float)v[e]);
        st_bf16x8(out + i * 8, o);
    }
}
template <int MODE>
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                      bf16* __restrict__ dx, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const bf16x8 v = ld_bf16x8(x + i * 8), d = ld_bf16x8(dy + i * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (bf16)((float)d[e] * act_grad_f<MODE>((float)v[e]));
        st_bf16x8(dx + i * 8, o);
    }
}

// ------------------------------------------------------------------ transpose-read semantics probe (test infrastructure)
__global__ void probe_tr16_kernel(const short* __restrict__ in, short* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
    __syncthreads();
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(short4v, lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = r[j];
}
__global__ void probe_mfma_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, float* __restrict__ out) {
    // a: [16][32] row-major (row, k); b: [16][32] (col, k).  lane supplies row/col lane&15, k = (lane>>4)*8..
    const int lane = threadIdx.x;
    bf16x8 fa = ld_bf16x8(a + (lane & 15) * 32 + (lane >> 4) * 8);
    bf16x8 fb = ld_bf16x8(b + (lane & 15) * 32 + (lane >> 4) * 8);
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}

inline int grid_for(int64_t total, int per_block = 256) {
    int64_t b = (total + per_block - 1) / per_block;
    if (b > 8192) b = 8192;  // 256 CUs x 8 blocks x 4: grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" {

int dllm_rope(void* x, const float* cos_tab, const float* sin_tab, const int64_t* pos, int64_t T, int S, int NH, int D,
              int64_t tok_stride, int64_t head_stride, int backward, void* stream) {
    if (T < 0 || NH <= 0 || (D % 16) != 0 || S <= 0) return DLLM_ERR_SHAPE;
    if ((tok_stride | head_stride) & 7) return DLLM_ERR_ALIGN;
    if (T == 0) return DLLM_OK;
    const int64_t total = T * NH * (D / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (bf16*)x, cos_tab, sin_tab, pos,
                       T, S, NH, D, tok_stride, head_stride, backward ? -1.0f : 1.0f);
    return dllm_check_launch();
}

// mode 0: SwiGLU out = silu(a)*b ; mode 1: GEGLU out = gelu(a)*b
int dllm_glu_fwd(const void* a, const void* b, void* out, int64_t M, int F, int64_t lda, int64_t ldb, int64_t ldo, int mode,
                 void* stream) {
    if (M < 0 || F <= 0 || (F & 7) || ((lda | ldb | ldo) & 7)) return DLLM_ERR_SHAPE;
    if (M == 0) return DLLM_OK;
    const unsigned g = (unsigned)(M < (1 << 20) ? M : (1 << 20));
    const int vpr = F / 8, nthr = vpr >= 256 ? 256 : ((vpr + 63) / 64) * 64;  // a row's vectors on 64..256 threads
    if (mode == 0x100)
        hipLaunchKernelGGL((glu_fwd_kernel<0, true>), dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b,
                           (bf16*)out, M, F, lda, ldb, ldo);
    else if (mode == 0)
        hipLaunchKernelGGL(glu_fwd_kernel<0>, dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b,
                           (bf16*)out, M, F, lda, ldb, ldo);
    else
        hipLaunchKernelGGL(glu_fwd_kernel<1>, dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b,
                           (bf16*)out, M, F, lda, ldb, ldo);
    return dllm_check_launch();
}
int dllm_glu_bwd(const void* dout, const void* a, const void* b, void* da, void* db, void* act_out, int64_t M, int F, int64_t ldd,
                 int64_t lda, int64_t ldb, int64_t ldda, int64_t lddb, int64_t ldact, int mode, void* stream) {
    if (M < 0 || F <= 0 || (F & 7) || ((ldd | lda | ldb | ldda | lddb) & 7)) return DLLM_ERR_SHAPE;
    if (act_out != nullptr && (ldact & 7)) return DLLM_ERR_SHAPE;
    if (M == 0) return DLLM_OK;
    const unsigned g = (unsigned)(M < (1 << 20) ? M : (1 << 20));
    const int vpr = F / 8, nthr = vpr >= 256 ? 256 : ((vpr + 63) / 64) * 64;
    if (mode == 0x100)
        hipLaunchKernelGGL((glu_bwd_kernel<0, true>), dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)dout, (const bf16*)a,
                           (const bf16*)b, (bf16*)da, (bf16*)db, (bf16*)act_out, M, F, ldd, lda, ldb, ldda, lddb, ldact);
    else if (mode == 0)
        hipLaunchKernelGGL(glu_bwd_kernel<0>, dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)dout, (const bf16*)a,
                           (const bf16*)b, (bf16*)da, (bf16*)db, (bf16*)act_out, M, F, ldd, lda, ldb, ldda, lddb, ldact);
    else
        hipLaunchKernelGGL(glu_bwd_kernel<1>, dim3(g), dim3(nthr), 0, (hipStream_t)stream, (const bf16*)dout, (const bf16*)a,
                           (const bf16*)b, (bf16*)da, (bf16*)db, (bf16*)act_out, M, F, ldd, lda, ldb, ldda, lddb, ldact);
    return dllm_check_launch();
}

int dllm_gather_rows(const void* table, const int64_t* idx, void* out, int64_t n, int D, int64_t ld_t, int64_t ld_o,
                     void* stream) {
    if (n < 0 || D <= 0 || (D & 7) || ((ld_t | ld_o) & 7)) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)table,
                       idx, (bf16*)out, n, D, ld_t, ld_o);
    return dllm_check_launch();
}
int dllm_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t n, int D, int64_t ld_s, int64_t ld_d,
                      void* stream) {
    if (n < 0 || D <= 0 || (D & 7) || ((ld_s | ld_d) & 7)) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for(n * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16*)src,
                       idx, (bf16*)dst, n, D, ld_s, ld_d);
    return dllm_check_launch();
}
int dllm_segment_sum_rows_ex(const void* dy, const int64_t* order, const int64_t* seg_start, const int64_t* uid, void* dtable,
                             int64_t nuniq, int D, int64_t ld_dy, int64_t ld_t, int in_dtype, int out_dtype, void* stream) {
    if (nuniq < 0 || D <= 0 || (D & 7) || ((ld_dy | ld_t) & 7) || seg_start == nullptr) return DLLM_ERR_SHAPE;
    if ((in_dtype != DLLM_BF16 && in_dtype != DLLM_F32) || (out_dtype != DLLM_BF16 && out_dtype != DLLM_F32)) return DLLM_ERR_SHAPE;
    if (nuniq == 0) return DLLM_OK;
    const int g = (int)(nuniq < 4096 ? nuniq : 4096);
    hipStream_t s = (hipStream_t)stream;
    const bool fi = in_dtype == DLLM_F32, fo = out_dtype == DLLM_F32;
    if (!fi && !fo) hipLaunchKernelGGL((segment_sum_rows_kernel<false, false>), dim3(g), dim3(256), 0, s, dy, order, seg_start, uid, dtable, nuniq, D, ld_dy, ld_t);
    else if (!fi && fo) hipLaunchKernelGGL((segment_sum_rows_kernel<false, true>), dim3(g), dim3(256), 0, s, dy, order, seg_start, uid, dtable, nuniq, D, ld_dy, ld_t);
    else if (fi && !fo) hipLaunchKernelGGL((segment_sum_rows_kernel<true, false>), dim3(g), dim3(256), 0, s, dy, order, seg_start, uid, dtable, nuniq, D, ld_dy, ld_t);
    else hipLaunchKernelGGL((segment_sum_rows_kernel<true, true>), dim3(g), dim3(256), 0, s, dy, order, seg_start, uid, dtable, nuniq, D, ld_dy, ld_t);
    return dllm_check_launch();
}
int dllm_segment_sum_rows(const void* dy, const int64_t* order, const int64_t* seg_start, const int64_t* uid, void* dtable,
                          int64_t nuniq, int D, int64_t ld_dy, int64_t ld_t, void* stream) {
    if (order == nullptr || uid == nullptr) return DLLM_ERR_SHAPE;
    return dllm_segment_sum_rows_ex(dy, order, seg_start, uid, dtable, nuniq, D, ld_dy, ld_t, DLLM_BF16, DLLM_BF16, stream);
}

// softmax over the last dimension: x fp32 [rows][cols] (row pitch ld_x, multiple of 4) -> y bf16 (row pitch ld_y, multiple of 4)
int dllm_softmax_rows(const float* x, void* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y, void* stream) {
    if (rows < 0 || cols <= 0 || (ld_x & 3) || (ld_y & 3)) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, cols, ld_x, ld_y);
    return dllm_check_launch();
}

int dllm_cross_entropy(const float* logits, const int64_t* labels, float* loss_row, void* dlogits, const float* gscale,
                       int64_t rows, int V, int64_t ld_logits, int64_t ld_dlogits, void* stream) {
    if (rows < 0 || V <= 0 || (ld_logits & 3)) return DLLM_ERR_SHAPE;
    if (dlogits != nullptr && ((ld_dlogits & 3) || gscale == nullptr)) return DLLM_ERR_SHAPE;
    if (rows == 0) return DLLM_OK;
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, logits, labels, loss_row,
                       (bf16*)dlogits, gscale, V, ld_logits, ld_dlogits);
    return dllm_check_launch();
}

// grad_scale_dev: optional device scalar multiplied into grad_scale (gradient-clipping coefficient without a host sync).
int dllm_adamw(void* p, const void* g, void* m, void* v, int64_t n, int param_dtype, int state_dtype, float lr, float beta1,
               float beta2, float eps, float weight_decay, int step, float grad_scale, const float* grad_scale_dev,
               void* stream) {
    if (n < 0 || step < 1) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    const int gsz = grid_for(n);
    hipStream_t s = (hipStream_t)stream;
    const bool vec8 = (n & 7) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                        reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (param_dtype == DLLM_BF16 && state_dtype == DLLM_BF16 && vec8)
        hipLaunchKernelGGL(adamw_vec8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (bf16*)p, (const bf16*)g, (bf16*)m, (bf16*)v,
                           n / 8, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    else if (param_dtype == DLLM_BF16 && state_dtype == DLLM_BF16)
        hipLaunchKernelGGL((adamw_kernel<bf16, bf16>), dim3(gsz), dim3(256), 0, s, (bf16*)p, (const bf16*)g, (bf16*)m, (bf16*)v,
                           n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    else if (param_dtype == DLLM_BF16 && state_dtype == DLLM_F32)
        hipLaunchKernelGGL((adamw_kernel<bf16, float>), dim3(gsz), dim3(256), 0, s, (bf16*)p, (const bf16*)g, (float*)m,
                           (float*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    else if (param_dtype == DLLM_F32 && state_dtype == DLLM_F32)
        hipLaunchKernelGGL((adamw_kernel<float, float>), dim3(gsz), dim3(256), 0, s, (float*)p, (const float*)g, (float*)m,
                           (float*)v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    else
        return DLLM_ERR_DTYPE;
    return dllm_check_launch();
}

// partials[0..DLLM_SUMSQ_PARTS) = per-block partial sums of x^2 (unused slots must be pre-zeroed by the caller); combine
// all partial buffers with dllm_reduce_sum_f32.  Deterministic (no atomics): DDP replicas compute identical clip factors.
int dllm_sumsq(const void* x, int64_t n, int dtype, float* out, void* stream) {
    if (n < 0) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    int64_t g = cdiv64(n, 8 * 1024);
    if (g > 256) g = 256;
    if (dtype == DLLM_BF16)
        hipLaunchKernelGGL(sumsq_kernel<bf16>, dim3((unsigned)g), dim3(1024), 0, (hipStream_t)stream, (const bf16*)x, n, out);
    else if (dtype == DLLM_F32)
        hipLaunchKernelGGL(sumsq_kernel<float>, dim3((unsigned)g), dim3(1024), 0, (hipStream_t)stream, (const float*)x, n, out);
    else
        return DLLM_ERR_DTYPE;
    return dllm_check_launch();
}

// Multi-tensor AdamW: `count` bf16 parameter tensors with bf16 moments sharing every hyper-parameter and the step number, each
// 16-byte aligned with n[i] % 8 == 0 (every weight matrix; callers send the rest through dllm_adamw).  p / g / m / v / n are HOST
// arrays of `count` entries; the library issues ceil(count / 48) launches (7 for the 295 trainable tensors of the 7B stage-II step,
// instead of 295).  Same update, element for element, as dllm_adamw.
int dllm_adamw_multi(void* const* p, const void* const* g, void* const* m, void* const* v, const int64_t* n, int count, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     const float* grad_scale_dev, void* stream) {
    if (count < 0 || step < 1) return DLLM_ERR_SHAPE;
    if (count == 0) return DLLM_OK;
    if (p == nullptr || g == nullptr || m == nullptr || v == nullptr || n == nullptr) return DLLM_ERR_SHAPE;
    const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
    for (int i = 0; i < count; ++i) {
        if (n[i] < 0 || (n[i] & 7)) return DLLM_ERR_ALIGN;
        if ((reinterpret_cast<uintptr_t>(p[i]) | reinterpret_cast<uintptr_t>(g[i]) | reinterpret_cast<uintptr_t>(m[i]) |
             reinterpret_cast<uintptr_t>(v[i])) & 15)
            return DLLM_ERR_ALIGN;
    }
    for (int i = 0; i < count;) {  // `i` is the consumed index: empty tensors are skipped without being revisited by the next batch
        MtAdamTable T{};
        int chunks = 0, k = 0;
        for (; i < count && k < MT_MAX; ++i) {
            if (n[i] == 0) continue;
            T.p[k] = (bf16*)p[i]; T.g[k] = (const bf16*)g[i]; T.m[k] = (bf16*)m[i]; T.v[k] = (bf16*)v[i];
            T.n8[k] = n[i] / 8;
            chunks += (int)cdiv64(n[i], MT_CHUNK);
            T.chunk_end[k] = chunks;
            ++k;
        }
        T.count = k;
        if (k == 0) continue;
        hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, T, lr, beta1, beta2, eps,
                           weight_decay, bc1, bc2, grad_scale, grad_scale_dev);
    }
    return dllm_check_launch();
}

// chunks (= partial sums) dllm_sumsq_multi writes for these sizes
int64_t dllm_sumsq_multi_parts(const int64_t* n, int count) {
    int64_t c = 0;
    for (int i = 0; i < count; ++i) c += cdiv64(n[i] > 0 ? n[i] : 0, MT_CHUNK);
    return c;
}

// Sum of squares of `count` bf16 tensors (16-byte aligned, n[i] % 8 == 0): one fp32 partial per chunk of 32768 elements into
// partials[0 .. dllm_sumsq_multi_parts), in tensor-then-chunk order; ceil(count / 48) launches.  Combine with dllm_reduce_sum_f32.
int dllm_sumsq_multi(const void* const* x, const int64_t* n, int count, float* partials, void* stream) {
    if (count < 0) return DLLM_ERR_SHAPE;
    if (count == 0) return DLLM_OK;
    if (x == nullptr || n == nullptr || partials == nullptr) return DLLM_ERR_SHAPE;
    for (int i = 0; i < count; ++i)
        if (n[i] < 0 || (n[i] & 7) || (reinterpret_cast<uintptr_t>(x[i]) & 15)) return DLLM_ERR_ALIGN;
    int64_t done = 0;
    for (int i = 0; i < count;) {  // consumed index, as in dllm_adamw_multi
        MtSumsqTable T{};
        int chunks = 0, k = 0;
        for (; i < count && k < MT_MAX; ++i) {
            if (n[i] == 0) continue;
            T.x[k] = (const bf16*)x[i];
            T.n8[k] = n[i] / 8;
            chunks += (int)cdiv64(n[i], MT_CHUNK);
            T.chunk_end[k] = chunks;
            ++k;
        }
        T.count = k;
        if (k == 0) continue;
        hipLaunchKernelGGL(sumsq_multi_kernel, dim3((unsigned)chunks), dim3(256), 0, (hipStream_t)stream, T, partials + done);
        done += chunks;
    }
    return dllm_check_launch();
}

int dllm_reduce_sum_f32(const float* in, int64_t n, float* out, void* stream) {
    if (n < 0) return DLLM_ERR_SHAPE;
    hipLaunchKernelGGL(reduce_sum_f32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, in, n, out);
    return dllm_check_launch();
}

// partials[0 .. nparts) := per-block sums of (pred - target)^2 (pred bf16, target fp32), 1 <= nparts <= 1024 blocks of 256 threads walking
// the n elements grid-stride; sum(partials) in index order (dllm_reduce_sum_f32) is the deterministic total
int dllm_mse_sum(const void* pred, const float* target, int64_t n, float* partials, int nparts, void* stream) {
    if (n < 0 || nparts < 1 || nparts > 1024) return DLLM_ERR_SHAPE;
    hipLaunchKernelGGL(mse_kernel, dim3(nparts), dim3(256), 0, (hipStream_t)stream, (const bf16*)pred, target, n, partials);
    return dllm_check_launch();
}
int dllm_mse_bwd(const void* pred, const float* target, int64_t n, const float* gscale, void* dpred, void* stream) {
    if (n < 0) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)pred, target, n,
                       gscale, (bf16*)dpred);
    return dllm_check_launch();
}

// out = a + b[i % period] ; n and period multiples of 8 (period == n => plain add)
int dllm_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t period, void* stream) {
    if (n < 0 || (n & 7) || period <= 0 || (period & 7)) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    hipLaunchKernelGGL(add_bcast_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       (const bf16*)b, (bf16*)out, n / 8, period / 8);
    return dllm_check_launch();
}
// out[g, r, :] = a[g, r, :] + b[g, :]  with a viewed as [groups][rows_per_group][C]
int dllm_add_rowgroup(const void* a, const void* b, void* out, int64_t groups, int64_t rows_per_group, int C, void* stream) {
    if (groups < 0 || rows_per_group <= 0 || C <= 0 || (C & 7)) return DLLM_ERR_SHAPE;
    const int64_t n = groups * rows_per_group * C;
    if (n == 0) return DLLM_OK;
    hipLaunchKernelGGL(add_rowgroup_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a,
                       (const bf16*)b, (bf16*)out, n / 8, C / 8, rows_per_group);
    return dllm_check_launch();
}
// mode: 1 exact GELU, 2 quick-GELU, 3 SiLU
int dllm_act_fwd(const void* x, void* out, int64_t n, int mode, void* stream) {
    if (n < 0 || (n & 7) || mode < 1 || mode > 3) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    const int g = grid_for(n / 8);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL(act_fwd_kernel<1>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)out, n / 8);
    else if (mode == 2) hipLaunchKernelGGL(act_fwd_kernel<2>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)out, n / 8);
    else hipLaunchKernelGGL(act_fwd_kernel<3>, dim3(g), dim3(256), 0, s, (const bf16*)x, (bf16*)out, n / 8);
    return dllm_check_launch();
}
int dllm_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int mode, void* stream) {
    if (n < 0 || (n & 7) || mode < 1 || mode > 3) return DLLM_ERR_SHAPE;
    if (n == 0) return DLLM_OK;
    const int g = grid_for(n / 8);
    hipStream_t s = (hipStream_t)stream;
    if (mode == 1) hipLaunchKernelGGL(act_bwd_kernel<1>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n / 8);
    else if (mode == 2) hipLaunchKernelGGL(act_bwd_kernel<2>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n / 8);
    else hipLaunchKernelGGL(act_bwd_kernel<3>, dim3(g), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (bf16*)dx, n / 8);
    return dllm_check_launch();
}

// n_half = B * pixels (latent pixels of the conditional half); see cfg_ddim_kernel for layouts.
int dllm_cfg_ddim_step(const void* pred, float* latents, void* next_in, int64_t n_half, int64_t unused, float guidance,
                       float sqrt_at, float sqrt_1mat, float sqrt_aprev, float sqrt_1maprev, int v_prediction, void* stream) {
    if (n_half < 0) return DLLM_ERR_SHAPE;
    if (n_half == 0) return DLLM_OK;
    (void)unused;
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n_half)), dim3(256), 0, (hipStream_t)stream, (const bf16*)pred, latents,
                       (bf16*)next_in, n_half, n_half, guidance, sqrt_at, sqrt_1mat, sqrt_aprev, sqrt_1maprev, v_prediction);
    return dllm_check_launch();
}

// test-only probes of the gfx950 fragment conventions the kernels rely on
int dllm_probe_tr16(const void* in256, void* out256, void* stream) {
    hipLaunchKernelGGL(probe_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const short*)in256, (short*)out256);
    return dllm_check_launch();
}
int dllm_probe_mfma16(const void* a, const void* b, float* out256, void* stream) {
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, out256);
    return dllm_check_launch();
}

}  // extern "C"
