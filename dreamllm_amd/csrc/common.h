// Shared device helpers for the dreamllm_amd HIP kernels (gfx950 / CDNA4 only).
// Wave = 64 lanes; bf16 storage, fp32 math.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DLLM_OK 0
#define DLLM_ERR_SHAPE (-1)
#define DLLM_ERR_DTYPE (-2)
#define DLLM_ERR_LAUNCH (-3)
#define DLLM_ERR_ALIGN (-4)

// dtype enum shared with include/dreamllm_hip.h
#define DLLM_BF16 0
#define DLLM_F32 1

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }

// 16-byte global load/store of 8 bf16.
__device__ __forceinline__ bf16x8 ld_bf16x8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st_bf16x8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ bf16x4 ld_bf16x4(const bf16* p) { return *reinterpret_cast<const bf16x4*>(p); }
__device__ __forceinline__ void st_bf16x4(bf16* p, bf16x4 v) { *reinterpret_cast<bf16x4*>(p) = v; }

__device__ __forceinline__ bf16x8 zero_bf16x8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

// Full-wave (64 lane) reductions through cross-lane shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block reduction for blocks of NW waves; scratch must hold NW floats (LDS).
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    return r;
}
template <int NW>
__device__ __forceinline__ float block_max(float v, float* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    float r = scratch[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// rotate_half pair of apply_rotary_pos_emb (modeling_dreamllm.py:176-209): y1 = x1 c - x2 s, y2 = x2 c + x1 s, in ONE fixed operation
// order (a rounded product, then a fused multiply-add) so that every kernel that inlines it -- rope_kernel, the q|k|v GEMM's RoPE
// epilogue -- produces the same bits (left to -ffast-math, the contraction differed between the two and one element in 2e5 moved an ulp)
__device__ __forceinline__ void rope_pair(float x1, float x2, float c, float s, float& y1, float& y2) {
    const float t1 = x2 * s, t2 = x1 * s;
    y1 = __builtin_fmaf(x1, c, -t1);
    y2 = __builtin_fmaf(x2, c, t2);
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
// SwiGLU forward product silu(gate) * up with the product kept apart from silu's division (same reason as swiglu_bwd_elem below)
__device__ __forceinline__ float swiglu_fwd_elem(float x, float y) {
#pragma clang fp reassociate(off) contract(off)
    const float a = silu_f(x);
    return a * y;
}
// SwiGLU backward of one element in ONE fixed operation order: d(gate) = (d y) * [sg (1 + x (1 - sg))], d(up) = d * (x sg), sg = sigmoid(x).
// Shared by glu_bwd_kernel<0> and the SwiGLU-backward epilogue of the GEMM kernels (gemm_epilogues.h): left to -ffast-math the three call sites
// (the element-wise kernel, the 8-wave GEMM, the four-wave GEMM) re-associated the triple product / contracted the inner sum differently and a
// training step's gradients moved by an ulp with the kernel family (found by test_training_step_has_the_same_bits_on_either_gemm_kernel_family).
__device__ __forceinline__ void swiglu_bwd_elem(float d, float x, float y, float& dgate, float& dup) {
#pragma clang fp reassociate(off) contract(off)
    const float sg = sigmoid_f(x);
    const float act = x * sg;
    const float om = 1.f - sg;
    const float t = __builtin_fmaf(x, om, 1.f);
    const float dact = sg * t;
    const float dyv = d * y;
    dgate = dyv * dact;
    dup = d * act;
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_f(1.702f * x); }

static inline int dllm_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DLLM_OK : DLLM_ERR_LAUNCH;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// One-time, per-device opt-in to more than 64 KiB of dynamic LDS for a kernel.  Thread-safe (autograd's backward thread and
// the main thread both launch) and per device (one bit per ordinal): the only process-level state of this library, and it is
// idempotent -- two racing threads both set the same attribute.
#include <atomic>
template <typename K>
static inline void dllm_ensure_dyn_lds(K kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done.fetch_or(bit, std::memory_order_release);
}

// Number of compute units of the current device (256 on MI355X), cached per device ordinal; idempotent like the helper above.
static inline int dllm_num_cus() {
    static std::atomic<int> cached[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    int v = cached[dev & 63].load(std::memory_order_acquire);
    if (v > 0) return v;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    n = (n / 8) * 8;  // whole blocks per XCD
    if (n < 8) n = 8;
    cached[dev & 63].store(n, std::memory_order_release);
    return n;
}
