/* dreamllm_hip.h -- C ABI of libdreamllm_hip.so, the gfx950 (MI355X / CDNA4) compute library behind the DreamLLM hot path.
 *
 * The reference (RunpeiDong/DreamLLM, "Omni") is 100 % Python: it has NO FFI / operator boundary of its own -- the seam
 * is the set of PyTorch call sites listed per function below (paths relative to /root/reference/, "[ext]" = arithmetic
 * that lives in the pinned third-party packages diffusers==0.24.0 / transformers==4.35.2, pyproject.toml:74,82).  This
 * header is therefore the boundary WE define; the Python operator layer above it
 * (dreamllm_amd/ops.py, ctypes) and the module layer above that (dreamllm_amd/modeling_*.py) mirror the reference's
 * class / plugin API one-to-one.  INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions (every entry point):
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless stated; bf16 = IEEE bfloat16 bits;
 *   - returns 0 (DLLM_OK) or a negative code: -1 bad shape, -2 unsupported dtype, -3 launch failure, -4 misaligned
 *     pointer / stride (activations and row pitches must be multiples of 8 elements = 16 bytes);
 *   - never allocates, frees or synchronises; asynchronous on `stream` (a hipStream_t passed as void*); workspaces are
 *     caller-allocated; stateless and re-entrant (callable from autograd worker threads);
 *   - there is no CPU fallback.
 */
#ifndef DREAMLLM_HIP_H
#define DREAMLLM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLLM_OK 0
#define DLLM_ERR_SHAPE (-1)
#define DLLM_ERR_DTYPE (-2)
#define DLLM_ERR_LAUNCH (-3)
#define DLLM_ERR_ALIGN (-4)
#define DLLM_BF16 0
#define DLLM_F32 1

/* ---------------------------------------------------------------------------------------------------- normalisation
 * DreamLLMRMSNorm.forward, omni/models/dreamllm/modeling_dreamllm.py:77-91 (fp32 statistics, cast, then weight *), with
 * an optional fused residual add in front (decoder residual stream, modeling_dreamllm.py:632,638).
 * x,res,w,h_out,y: bf16; rstd: fp32[rows] (saved for backward).  res/h_out may be NULL together. */
int dllm_rmsnorm_fwd(const void* x, const void* res, const void* w, void* h_out, void* y, float* rstd, int64_t rows, int D,
                     float eps, void* stream);
/* number of fp32 partial rows the norm backward kernels write: dw_partial = [nparts][D] floats */
int dllm_norm_bwd_nparts(int64_t rows);
/* autograd of the above: dx = rstd*(g - xhat*mean(g*xhat)) [+ dh_in]; dw_out[D] (dtype flag) via dw_partial, or NULLs. */
int dllm_rmsnorm_bwd(const void* dy, const void* h, const void* w, const float* rstd, const void* dh_in, void* dx,
                     float* dw_partial, void* dw_out, int dw_dtype, int64_t rows, int D, void* stream);
/* torch.nn.LayerNorm as used by CLIPVisionModel [ext] (modeling_plugins.py:214-219,321-323), the optional
 * post_layernorm (modeling_plugins.py:234-238) and BasicTransformerBlock.norm{1,2,3} of the UNet [ext]. */
int dllm_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* mean, float* rstd, int64_t rows, int D,
                       float eps, void* stream);
int dllm_layernorm_bwd(const void* dy, const void* x, const void* w, const float* mean, const float* rstd, void* dx,
                       float* dw_partial, float* db_partial, void* dw_out, void* db_out, int dw_dtype, int64_t rows, int D,
                       void* stream);
/* GroupNorm(32)(+SiLU) on NHWC activations: ResnetBlock2D.norm1/norm2, Transformer2DModel.norm, conv_norm_out of
 * UNet2DConditionModel / AutoencoderKL [ext] (call sites modeling_plugins.py:511,556,815-821,842).
 * x,y: bf16 [NB,HW,C]; mean,rstd: fp32 [NB,G]; ab: fp32 [NB,C,2] scratch; part: dllm_groupnorm_ws_floats() floats. */
int64_t dllm_groupnorm_ws_floats(int NB, int HW, int C);
int dllm_groupnorm_fwd(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, float* ab,
                       float* part, int NB, int HW, int C, int G, float eps, int act, void* stream);
/* input gradient only (affine frozen: modeling_plugins.py:405-407); c1,c2: fp32 [NB,G] scratch */
/* Single-launch GroupNorm(+SiLU) spread over 4 blocks per (image, group) for tiny batches (the UNet inside the denoising loop at batch
 * 2: 64 (image, group) pairs on 256 CUs).  `sync`: caller-owned int32[>= NB*G*32], zero before the first use, left consistent by every
 * launch (one buffer per stream in flight).  Eligible when NB*HW*C <= 2^23, HW >= 1024, HW % 4 == 0, NB*G*4 <= 512; otherwise
 * DLLM_ERR_SHAPE and the caller uses dllm_groupnorm_fwd.  Same results as dllm_groupnorm_fwd up to fp32 summation order.  Word 31 of
 * an (image, group) slot counts how often a block gave up waiting for its partners (~2 ms) and computed the slice's statistics alone
 * (never wrong, but a different fp32 summation order): 0 on a GPU this process has to itself. */
int dllm_groupnorm_fwd_split(const void* x, const void* gamma, const void* beta, void* y, float* mean, float* rstd, int* sync,
                             int NB, int HW, int C, int G, float eps, int act, void* stream);
int dllm_groupnorm_bwd(const void* dy, const void* x, const void* gamma, const void* beta, const float* mean, const float* rstd,
                       void* dx, float* c1, float* c2, float* part, int NB, int HW, int C, int G, int act, void* stream);

/* ---------------------------------------------------------------------------------------------------- contractions
 * bf16 MFMA GEMM, fp32 accumulate: C[M,N] = epi(alpha * A*B + bias) + residual.  Replaces every nn.Linear on the path:
 * q/k/v/o_proj modeling_dreamllm.py:273-276,336-338,395; gate/up/down_proj :219-221,237; lm_head :1216,1452; the
 * projectors omni/models/projector/mlp_projector.py:19,23-27,39-50; and the linears inside CLIP / UNet / VAE [ext].
 * layout_a: 0 = A[m][k] k-contiguous (lda = row pitch); 1 = A stored [K][lda], m contiguous (x^T without a transpose).
 * layout_b: 0 = B given as [N][ldb] k-contiguous (nn.Linear weight [out,in]); 1 = B stored [K][ldb], n contiguous.
 *   forward y = x W^T : (0,0);  dgrad dx = dy W : (0,1);  wgrad dW = dy^T x : (1,1).
 * epi: 0 none, 1 exact-erf GELU, 2 quick-GELU (CLIP), 3 SiLU.  out_dtype DLLM_BF16 | DLLM_F32.  accumulate: C += .
 * epi 4 = GEGLU (diffusers `GEGLU.forward` [ext]: hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate)), forward layout (0,0)
 *   only: B is the [2F, K] projection weight (rows [hidden F | gate F]), bias [2F] or NULL, N = F, C = [M, F] bf16; K % 64 == 0 and
 *   F % 64 == 0, no residual / accumulate / split-K (DLLM_ERR_SHAPE otherwise).  The hidden and gate columns of an output meet
 *   inside one wave of the ring-buffered kernel, so the [M, 2F] projection is never written. */
int dllm_gemm_bf16(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                   int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int layout_a, int layout_b, int epi,
                   int out_dtype, int accumulate, float alpha, void* stream);
/* Split-K variants for small grids with deep reductions (UNet at batch 2): splitk > 1 writes fp32 partials to the caller's
 * workspace [splitk][M][N]; the reduction is deterministic (slice order).  `counters` (optional): int32[>= ceil(M/128)*ceil(N/128)],
 * all zero on entry and left all zero -- the last K slice of a tile to arrive then reduces it INSIDE the GEMM launch (agent-scope
 * release/acquire around a ticket); NULL = a separate reduce kernel follows.  Use one counter array per stream in flight.
 * dllm_gemm_splitk_hint suggests splitk for a problem in the given layouts (layout_a 2 = the NHWC conv gather of dllm_conv2d_nhwc_bf16). */
int dllm_gemm_splitk_hint(int64_t M, int64_t N, int64_t K, int layout_a, int layout_b);
/* Stream-K tail of the 256 x 256 pipelined kernel: dllm_gemm_bf16_splitk(..., splitk = 1, workspace != NULL, ...) with a caller-owned
 * workspace of dllm_gemm_streamk_ws_bytes() bytes lets launches whose LAST round of tiles would leave most of the 256 CUs idle (the
 * MLP weight gradients: 1376 and 688 tiles = 5.4 / 2.7 rounds) run their whole rounds as usual and spread the K loops of the
 * remaining tiles evenly over the CUs (fp32 partial slabs in the workspace, summed in fixed K order by a fix-up launch: deterministic).
 * dllm_gemm_streamk_hint returns 1 when a problem would take that path (so the caller only hands over the workspace then).
 * Replaces nothing in the reference (torch.nn.Linear backward -> cuBLAS picks its own split): a scheduling choice of this library. */
int64_t dllm_gemm_streamk_ws_bytes(void);
int dllm_gemm_streamk_hint(int64_t M, int64_t N, int64_t K, int layout_a, int layout_b);
/* `variant` selects the kernel PER CALL (the library holds no mutable state; every entry point is re-entrant and may be called
 * from any thread on any stream): low 16 bits = tile family -- 0 automatic (what the product passes), 128 / 256 register-staged
 * tiles, 257 plain LDS-DMA 256-tile kernel, 259 software-pipelined 8-wave LDS-DMA kernel (the automatic choice for eligible shapes of
 * less than one round of 256 tiles, and for fp32 outputs / ragged edge tiles / NHWC convs), 280 the four-wave kernel of round 6 (one wave
 * per SIMD, 128 x 128 wave tiles, the LDS stage released half a tile early: the automatic choice from 256 full tiles on, bf16 output;
 * same bits as 259; ineligible calls fall back to 259);
 * 264 the ring-buffered 128 x 128 kernel for small grids (forward linears / NHWC convs with K % 64 == 0; the automatic choice
 * wherever the register-staged 128-tile kernel used to run; ineligible calls fall back to the automatic choice), 267 / 268 the same
 * kernel forced to its four-stage (one block per CU) / two-stage (two blocks per CU) form; 262 the 256 x 128 pipelined tile; 266 the
 * MFMA 32x32x16 experiment (forward layout, plain epilogue);
 * bits 16-23 = GROUP_M of the grouped tile order (0 = per-layout default); bit 24 = XCD-synchronised persistent walk; bit 25 = never
 * choose the ring-buffered kernel automatically (round 3's selection, an A/B knob); bit 26 = "with splitk <= 1 my workspace is a
 * stream-K workspace of dllm_gemm_streamk_ws_bytes() bytes" -- without it a workspace passed with splitk <= 1 is IGNORED (round 4,
 * round 3 treated any non-NULL workspace as 128 MiB of slab space; a caller re-using its smaller split-K buffer with
 * splitk = 1 would have been written out of bounds); bit 27 = the ring-buffered kernel never takes its two-stage form (A/B knob);
 * bit 28 = never choose the four-wave kernel automatically (round 6's A/B knob: the 8-wave kernel everywhere, as before).
 * 261 = the first (MFMA 32x32x16) form of the four-wave kernel, kept as an experiment.  Tests pass 128 / 256 / 257 / 259 / 262 / 264 / 280 to cover
 * every kernel family.  Anything else returns DLLM_ERR_SHAPE.  (A build with -DDLLM_BENCH_MODES additionally accepts the
 * wrong-result diagnostic modes 258 / 260 / 263 / 265 and the ring kernel's timing diagnostics 269-278 (s_memtime stamps of one block written to
 * the workspace, request-placement variants, no-DMA / no-MFMA ablations: tools/ring_timeline.py) used by tools/; the shipped library does not
 * contain them.) */
int dllm_gemm_bf16_splitk(const void* A, const void* B, void* C, const void* bias, const void* residual, int64_t M, int64_t N,
                          int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int64_t ldr, int layout_a, int layout_b, int epi,
                          int out_dtype, int accumulate, float alpha, int splitk, float* workspace, int* counters, int variant,
                          void* stream);
/* DreamLLMMLP.forward, modeling_dreamllm.py:237 `down_proj(act_fn(gate_proj(x)) * up_proj(x))`, with the SwiGLU folded into the GEMMs
 * beside it (SURVEY §8(b2) `silu_mul` epilogue; round 6).  wgu = the packed [2F, K] weight (gate rows, then up rows).
 *   fwd: gu[M, 2F] = x wgu^T  AND  act[M, F] = silu(gu[:, :F]) * gu[:, F:]  in one launch (replaces GEMM + dllm_glu_fwd);
 *   bwd: dgu[M, 2F] = d(gate|up) of the product given dy [M, D] and the down projection's weight wd [D, F]: d_act = dy wd stays in the
 *        accumulators (replaces the input-gradient GEMM + dllm_glu_bwd).
 * Same arithmetic on the same bf16-rounded operands as the unfused pair: identical results.  M % 256 == 0, K / D % 64 == 0,
 * F % 128 == 0 (fwd) / F % 256 == 0 (bwd), 16-byte aligned pointers, leading dimensions % 8 == 0; otherwise DLLM_ERR_SHAPE / _ALIGN and
 * the caller uses the unfused launches.  group_m: bits 0-7 GROUP_M of the grouped tile order (0 = default); bits 8-9 the kernel family
 * (0: the library's choice -- the four-wave kernel from one full round of 256 tiles on; 1: the 8-wave kernel; 2: the four-wave kernel;
 * identical results). */
int dllm_gemm_swiglu_fwd(const void* x, const void* wgu, void* gu, void* act, int64_t M, int64_t F, int64_t K, int64_t ldx, int64_t ldw,
                         int64_t ldgu, int64_t ldact, int group_m, void* stream);
int dllm_gemm_swiglu_bwd(const void* dy, const void* wd, const void* gu, void* dgu, int64_t M, int64_t F, int64_t D, int64_t lddy,
                         int64_t ldw, int64_t ldgu, int64_t lddgu, int group_m, void* stream);
/* DreamLLMAttention.forward, modeling_dreamllm.py:336-338 + apply_rotary_pos_emb :184-209 (SURVEY §8(b2) `rope` epilogue; round 6): the packed
 * q|k|v projection qkv[M, N] = x wqkv^T with the rotary embedding applied to its first rope_cols = (Hq + Hkv) * 128 columns (head_dim 128) in the
 * GEMM's epilogue.  cos_tab / sin_tab: fp32 [max_pos][64]; pos: int64 [M] position ids or NULL (position = row % S).  Same arithmetic on the
 * same bf16-rounded projection as dllm_gemm_bf16 + dllm_rope: identical results.  M % 256 == 0, N % 256 == 0, rope_cols % 256 == 0, K % 64 == 0,
 * 16-byte aligned pointers, leading dimensions % 8 == 0; otherwise DLLM_ERR_SHAPE / _ALIGN (the caller runs the two launches). */
int dllm_gemm_rope_qkv(const void* x, const void* wqkv, void* qkv, const float* cos_tab, const float* sin_tab, const int64_t* pos, int64_t M,
                       int64_t N, int64_t K, int64_t rope_cols, int S, int64_t ldx, int64_t ldw, int64_t ldo, int group_m, void* stream);
int dllm_conv2d_nhwc_bf16_splitk(const void* x, const void* w, void* out, const void* bias, const void* residual,
                                 const void* image_bias, int NB, int H, int W, int C, int OH, int OW, int CO, int KH, int KW,
                                 int stride, int pad, int up2, int even_only, int epi, int out_dtype, int splitk,
                                 float* workspace, int* counters, int variant, void* stream);
/* NHWC convolution (3x3 / 1x1) as implicit GEMM: ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D, Upsample2D,
 * conv_in/conv_out of UNet2DConditionModel and AutoencoderKL [ext] (call sites modeling_plugins.py:511,556,815-821,842).
 * x [NB,H,W,C] bf16, w [CO][KH*KW*C] bf16 (k = (kh,kw,ci)), out [NB,OH,OW,CO]; image_bias [NB,CO] = per-image
 * time-embedding add; up2: nearest-2x upsample fused into the gather; even_only: transposed gather (dgrad of stride 2). */
int dllm_conv2d_nhwc_bf16(const void* x, const void* w, void* out, const void* bias, const void* residual,
                          const void* image_bias, int NB, int H, int W, int C, int OH, int OW, int CO, int KH, int KW,
                          int stride, int pad, int up2, int even_only, int epi, int out_dtype, void* stream);
/* [NB,2H,2W,C] -> [NB,H,W,C] 2x2 sums: backward of F.interpolate(nearest, x2) in Upsample2D [ext]. */
int dllm_sumpool2_nhwc(const void* in, void* out, int NB, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------------- attention
 * Flash attention forward: replaces eager attention modeling_dreamllm.py:357-379 and flash_attn_func /
 * flash_attn_varlen_func :532-549 (causal, dropout 0, scale 1/sqrt(Dh), fp32 softmax; a padded batch is described by one
 * contiguous valid span per row -- seqstart[b] (0 if NULL) and seqlens[b] -- instead of _upad_input / pad_input :553-583:
 * keys outside the span are masked, query rows outside it come back as zeros; with Sq != Sk (KV cache) seqstart masks the
 * leading keys and every query is valid), CLIP self-attention and the UNet self/cross attention [ext].
 * q,o: [B,Sq,H,D] views (element strides sb,ss,sh; d contiguous); k,v: [B,Sk,Hkv,D] views sharing one stride set;
 * D in {64,128}; H % Hkv == 0 (GQA, repeat_kv :242-251); seqlens / seqstart int32[B] or NULL; lse fp32 [B,H,Sq] or NULL.
 * `causal`: bit 0 = causal mask; bits 1-2 select the forward kernel per call: 0 automatic (Sq >= 512: the round-5 "ping-pong"
 * 256-query kernel of csrc/attn_fwd_pp.hip -- the two waves of a SIMD one segment apart, MFMA 32x32x16, register-resident K / V
 * fragments -- when one (batch, head) key axis spans < 1 GiB, else the 8-wave pipelined kernel; below 512 the 4-wave 128-query
 * kernel); 1 / 2 / 3 force the 4-wave / 8-wave pipelined / ping-pong kernel -- the tests run every shape through all three. */
int dllm_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, const int* seqlens, const int* seqstart, int B,
                  int H, int Hkv, int Sq, int Sk, int D, int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh,
                  int64_t o_sb, int64_t o_ss, int64_t o_sh, float scale, int causal, void* stream);
/* its autograd: dq/dk/dv (dk,dv share strides; may alias slices of one packed dQKV buffer); delta: fp32 [3,B,H,Sq] workspace
   (planes delta, -delta, -lse/scale).  Bits 1-2 of `causal` select the kernels as in dllm_attn_fwd: 1 the 4-wave kernels, 2 the
   8-wave pipelined ones; 3 and automatic (axes >= 512) additionally take the ping-pong dQ kernel of csrc/attn_bwd_pp.hip at D = 128
   when the key axis spans < 1 GiB and dq is 16-byte aligned with strides that are multiples of 8 elements (dK / dV stay on the 8-wave
   kernels, as does D = 64). */
int dllm_attn_bwd(const void* dout, const void* q, const void* k, const void* v, const void* o, const float* lse, float* delta,
                  void* dq, void* dk, void* dv, const int* seqlens, const int* seqstart, int B, int H, int Hkv, int Sq, int Sk, int D,
                  int64_t q_sb, int64_t q_ss, int64_t q_sh, int64_t k_sb, int64_t k_ss, int64_t k_sh, int64_t o_sb, int64_t o_ss, int64_t o_sh,
                  int64_t dq_sb, int64_t dq_ss, int64_t dq_sh, int64_t dk_sb, int64_t dk_ss, int64_t dk_sh, float scale,
                  int causal, void* stream);

/* ---------------------------------------------------------------------------------------------------- elementwise
 * apply_rotary_pos_emb + rotate_half, modeling_dreamllm.py:176-209, in place on a [T,NH,D] view; cos/sin fp32
 * [max_pos][D/2]; pos int64[T] or NULL (=> token index % S); backward = 1 applies the transpose rotation. */
int dllm_rope(void* x, const float* cos_tab, const float* sin_tab, const int64_t* pos, int64_t T, int S, int NH, int D,
              int64_t tok_stride, int64_t head_stride, int backward, void* stream);
/* mode 0: silu(a)*b = DreamLLMMLP act_fn(gate)*up, modeling_dreamllm.py:237;  mode 1: gelu(a)*b = diffusers GEGLU [ext] */
int dllm_glu_fwd(const void* a, const void* b, void* out, int64_t M, int F, int64_t lda, int64_t ldb, int64_t ldo, int mode,
                 void* stream);
/* backward; act_out (nullable): the forward product silu(a)*b / a*gelu(b) re-emitted in the same pass (the decoder layer does not
 * store it: it is the A operand of the down-projection's weight gradient) */
int dllm_glu_bwd(const void* dout, const void* a, const void* b, void* da, void* db, void* act_out, int64_t M, int F, int64_t ldd,
                 int64_t lda, int64_t ldb, int64_t ldda, int64_t lddb, int64_t ldact, int mode, void* stream);
/* out = a + b[i % period]: residual adds outside a GEMM epilogue, CLIP position embedding [ext] */
int dllm_add_bcast(const void* a, const void* b, void* out, int64_t n, int64_t period, void* stream);
int dllm_add_rowgroup(const void* a, const void* b, void* out, int64_t groups, int64_t rows_per_group, int C, void* stream);
/* stand-alone activations: 1 nn.GELU (MLPProjector, mlp_projector.py:40-44), 2 quick-GELU, 3 SiLU (time embedding [ext]) */
int dllm_act_fwd(const void* x, void* out, int64_t n, int mode, void* stream);
int dllm_act_bwd(const void* dy, const void* x, void* dx, int64_t n, int mode, void* stream);

/* ---------------------------------------------------------------------------------------------------- gather / splice
 * embed_tokens lookup modeling_dreamllm.py:1066-1067 and the dream-state gather :1399-1418 */
int dllm_gather_rows(const void* table, const int64_t* idx, void* out, int64_t n, int D, int64_t ld_t, int64_t ld_o,
                     void* stream);
/* the multimodal splice (dream queries :1081-1099, image features :1104-1141) as one index scatter: dst[idx[i]] = src[i] */
int dllm_scatter_rows(const void* src, const int64_t* idx, void* dst, int64_t n, int D, int64_t ld_s, int64_t ld_d,
                      void* stream);
/* deterministic nn.Embedding backward over id-sorted rows (modeling_dreamllm.py:1066: the gradient of embed_tokens):
 * dtable[uid[u], :] = sum over j in [seg_start[u], seg_start[u+1]) of dy[order[j], :], fp32 accumulation in the order of j. */
int dllm_segment_sum_rows(const void* dy, const int64_t* order, const int64_t* seg_start, const int64_t* uid, void* dtable,
                          int64_t nuniq, int D, int64_t ld_dy, int64_t ld_t, void* stream);
/* the same with fp32 rows in and / or out (in_dtype / out_dtype: DLLM_BF16 | DLLM_F32), order == NULL (segments of consecutive rows) and
 * uid == NULL (segment u -> row u): a token with ~10^4 positions in a batch (one work-group walking them: 2.7 ms) is summed in two calls --
 * chunks of its segment into fp32 partial rows, then the partial rows of each token into the table. */
int dllm_segment_sum_rows_ex(const void* dy, const int64_t* order, const int64_t* seg_start, const int64_t* uid, void* dtable,
                             int64_t nuniq, int D, int64_t ld_dy, int64_t ld_t, int in_dtype, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------- losses
 * logits.float() + CrossEntropyLoss(reduction="none") + masked mean, modeling_dreamllm.py:1453-1470: per-row loss and
 * (optionally) bf16 dlogits scaled by the DEVICE scalar gscale (= dloss / n_valid), so the step has no host sync. */
int dllm_cross_entropy(const float* logits, const int64_t* labels, float* loss_row, void* dlogits, const float* gscale,
                       int64_t rows, int V, int64_t ld_logits, int64_t ld_dlogits, void* stream);
/* softmax(dim=-1) of fp32 scores -> bf16 probabilities: the single-head, 512-wide mid-block attention of AutoencoderKL [ext]
 * (modeling_plugins.py:511,842), whose head_dim is outside the flash kernels: scores and P V run on dllm_gemm_bf16. */
int dllm_softmax_rows(const float* x, void* y, int64_t rows, int cols, int64_t ld_x, int64_t ld_y, void* stream);
/* F.mse_loss(model_pred.float(), target.float()), modeling_plugins.py:559: partials[b] = sum over block b's elements of (pred - target)^2
   (nparts <= 1024 blocks; the total is dllm_reduce_sum_f32 over the partials: fixed order, no float atomics) ; and its grad */
int dllm_mse_sum(const void* pred, const float* target, int64_t n, float* partials, int nparts, void* stream);
int dllm_mse_bwd(const void* pred, const float* target, int64_t n, const float* gscale, void* dpred, void* stream);

/* ---------------------------------------------------------------------------------------------------- optimizer step
 * torch.optim.AdamW as built by omni/train/trainer.py:392-465 and stepped at :799-821 (clip_grad_norm_ :807):
 * fused update with optional device-side clip coefficient; ||g||^2 is reduced deterministically (per-block partials +
 * fixed-order final sum) so that data-parallel replicas derive bit-identical clip factors. */
int dllm_adamw(void* p, const void* g, void* m, void* v, int64_t n, int param_dtype, int state_dtype, float lr, float beta1,
               float beta2, float eps, float weight_decay, int step, float grad_scale, const float* grad_scale_dev,
               void* stream);
int dllm_sumsq(const void* x, int64_t n, int dtype, float* partials256, void* stream); /* 256 per-block partials, no atomics */
/* Multi-tensor forms (round 4): p / g / m / v / x / n are HOST arrays of `count` entries (bf16 tensors, 16-byte aligned, n % 8 == 0,
 * bf16 moments, shared hyper-parameters and step); the tables travel as kernel arguments, 48 tensors per launch: the 295 + 295
 * per-tensor launches of the 7B step become 7 + 7.  dllm_sumsq_multi writes one fp32 partial per 32768-element chunk (fixed
 * order; dllm_sumsq_multi_parts of them) for dllm_reduce_sum_f32. */
int dllm_adamw_multi(void* const* p, const void* const* g, void* const* m, void* const* v, const int64_t* n, int count, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale,
                     const float* grad_scale_dev, void* stream);
int64_t dllm_sumsq_multi_parts(const int64_t* n, int count);
int dllm_sumsq_multi(const void* const* x, const int64_t* n, int count, float* partials, void* stream);
int dllm_reduce_sum_f32(const float* in, int64_t n, float* out, void* stream);        /* fixed-order final reduction */

/* ---------------------------------------------------------------------------------------------------- denoising loop
 * One launch for modeling_plugins.py:824-833 (noise_pred.chunk(2), CFG combine, scheduler.step with deterministic DDIM)
 * plus the next iteration's torch.cat([latents]*2) (:811): pred bf16 [2B,P,4] NHWC, latents fp32 [B,P,4] in place,
 * next_in bf16 [2B,P,8] (channel-padded conv_in operand) or NULL. */
int dllm_cfg_ddim_step(const void* pred, float* latents, void* next_in, int64_t n_half, int64_t unused, float guidance,
                       float sqrt_at, float sqrt_1mat, float sqrt_aprev, float sqrt_1maprev, int v_prediction, void* stream);

/* ---------------------------------------------------------------------------------------------------- greedy decode
 * Token step of the KV-cache decode loop (omni/eval/language_eval/modeling_dreamllm.py:76-97): every nn.Linear of
 * modeling_dreamllm.py:212-239,254-400,1452 degenerates to y[M<=8][N] = x W^T, HBM-bound on W.
 * dllm_gemv_bf16: fp32 accumulate, optional fused residual, bf16 or fp32 (logits) output.  M <= 4 with M * K * 2 <= 60 KiB: x staged
 * once per block in LDS, two output rows per wave, v_dot2c_f32_bf16 (round 4); otherwise one wave per output row.
 * dllm_attn_decode: softmax(q K^T * scale) V for ONE query token per (b, h) over the cache [B][S_max][Hkv][D]; the valid
 * length kv_len[b] is read from device memory so that the launch is step-invariant (hipGraph replay).  ws: fp32 workspace of
 * dllm_attn_decode_ws_floats(B, H, D, nsplit) elements (split-KV partial softmax states).  kv_start (int32 [B] on device, or
 * NULL): first valid cache slot of a LEFT-padded prompt (padding_side="left" callers: omni/eval/vqa/vqa_inference.py:276,
 * omni/eval/text2img/ddp_sample_coco.py:64); slots [kv_start[b], kv_len[b]) are attended. */
int dllm_gemv_bf16(const void* x, const void* W, void* y, const void* residual, int M, int64_t N, int64_t K, int64_t ldx,
                   int64_t ldw, int64_t ldy, int64_t ldr, int out_dtype, void* stream);
/* fused forms that cut the token step from 17 to 7 launches per layer: RMSNorm folded into the GEMV (same roundings as
 * dllm_rmsnorm_fwd), q/k/v in one launch, gate/up + SwiGLU in one launch; RoPE (q in place, k) + KV-cache append in one. */
int dllm_gemv_fused(const void* x, const void* norm_w, float eps, const void* W0, const void* W1, const void* W2, void* y0, void* y1,
                    void* y2, const void* residual, int M, int64_t N0, int64_t N1, int64_t N2, int64_t K, int64_t ldx, int64_t ldw,
                    int64_t ldy0, int64_t ldy1, int64_t ldy2, int64_t ldr, int swiglu, int out_dtype, void* stream);
int dllm_rope_append(void* q, const void* k, const void* v, void* kcache, void* vcache, const float* cos_tab, const float* sin_tab,
                     const int64_t* pos, const int* kv_len, int B, int H, int Hkv, int D, int64_t q_sb, int64_t kv_sb, int64_t c_sb,
                     int64_t c_ss, int64_t c_sh, void* stream);
int64_t dllm_attn_decode_ws_floats(int B, int H, int D, int nsplit);
int dllm_attn_decode(const void* q, const void* kcache, const void* vcache, const int* kv_len, const int* kv_start, void* out,
                     float* ws, int B, int H, int Hkv, int D, int64_t q_sb, int64_t q_sh, int64_t c_sb, int64_t c_ss, int64_t c_sh, int64_t o_sb,
                     int64_t o_sh, float scale, int nsplit, void* stream);
/* dllm_rope_append + dllm_attn_decode in ONE launch pair (round 4; 6 launches per layer and token instead of 7): q [B][H][D]
 * UN-rotated (read only), k_new / v_new [B][Hkv][D] (batch pitch kv_sb, k un-rotated) = the step's key / value.  The attention
 * kernel rotates q in registers (rounded to bf16 as dllm_rope_append stores it), rotates k_new with rotary position pos[b], writes
 * it and v_new to cache slot kv_len[b] - 1 and attends to them in the same launch (apply_rotary_pos_emb
 * modeling_dreamllm.py:184-209 + the cache update :340-345 + the 1-token attention :346-400).  counters: int32 [B * H], all zero
 * on entry and left all zero, or NULL: with counters the split that finishes last merges the split-KV partial states inside the
 * launch (agent-scope stores / loads around a ticket; fixed split order = the combine kernel's bits), so the combine launch
 * disappears too: 5 launches per layer and token. */
int dllm_attn_decode_rope(const void* q, const void* k_new, const void* v_new, void* kcache, void* vcache, const float* cos_tab,
                          const float* sin_tab, const int64_t* pos, const int* kv_len, const int* kv_start, void* out, float* ws,
                          int* counters, int B, int H, int Hkv, int D, int64_t q_sb, int64_t q_sh, int64_t kv_sb, int64_t c_sb,
                          int64_t c_ss, int64_t c_sh, int64_t o_sb, int64_t o_sh, float scale, int nsplit, void* stream);
/* The o projection of a token step fed directly with the split-KV partials of dllm_attn_decode[_rope] called with out = NULL (round 6):
 * y[M, N] = merge(ws) W^T (+ residual), merge = the combine of the split states (the bits dllm_attn_decode writes to `out`), computed while
 * every block stages x: one launch per layer and token fewer.  ws fp32 [M * H][nsplit][D + 2]; W [N][ldw], K = H * D; M <= 4. */
int dllm_gemv_attn_combine(const float* ws, const void* W, void* y, const void* residual, int M, int H, int D, int nsplit, int64_t N,
                           int64_t ldw, int64_t ldy, int64_t ldr, int out_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------- test probes
 * hardware-convention probes used by tests/test_kernels_gpu.py (ds_read_b64_tr_b16 and MFMA 16x16x32 fragment layouts) */
int dllm_probe_tr16(const void* in256, void* out256, void* stream);
int dllm_probe_mfma16(const void* a, const void* b, float* out256, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DREAMLLM_HIP_H */
