// rg_common.h -- shared device/host helpers of librnagan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include "../../include/rnagan_hip.h"

// ---------------------------------------------------------------------------------------------
// error handling (no exceptions cross the C ABI)
// ---------------------------------------------------------------------------------------------
void rg_set_error(const char* fmt, ...);

#define RG_REQUIRE(cond, code, ...)     \
  do {                                  \
    if (!(cond)) {                      \
      rg_set_error(__VA_ARGS__);        \
      return (code);                    \
    }                                   \
  } while (0)

#define RG_LAUNCH_CHECK(name)                                              \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      rg_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return RG_EHIP;                                                      \
    }                                                                      \
  } while (0)

#ifdef RG_HALF_F16
#define RG_H16 RG_F16      /* the dtype code of this build's 16-bit storage type (see below) */
#else
#define RG_H16 RG_BF16
#endif
static inline hipStream_t rg_stream(void* s) { return (hipStream_t)s; }
static inline size_t rg_dtype_size(int dtype) { return dtype == RG_H16 ? 2 : 4; }
static inline bool rg_is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int rg_ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline size_t rg_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// The 16-bit storage type of this BUILD of the library (raw 16 bits in memory; arithmetic, statistics and accumulation are
// fp32 everywhere).  Default: bf16 (librnagan_hip.so; callers pass RG_BF16).  With -DRG_HALF_F16 the same sources build
// librnagan_hip_f16.so, whose 16-bit type is IEEE fp16 (callers pass RG_F16; BASELINE.json configs[3]: the same kernels on
// v_mfma_f32_*_f16, which take the same cycles as the bf16 forms -- MI355X_MICROARCH, matrix cores).  Everything that touches
// the stored bits goes through these helpers and the RG_MFMA_* macros of rg_gather.h.  Round-to-nearest-even by the
// compiler's native conversions (v_cvt_pk_bf16_f32 / v_cvt_f16_f32: NaN-preserving; fp16 overflow gives infinity).
// ---------------------------------------------------------------------------------------------
struct h16_t { uint16_t bits; };

#ifdef RG_HALF_F16
#define RG_H16_NAME "f16"
#define RG_H16_ONE 0x3c00      /* 1.0 */
__device__ __forceinline__ float h16_to_f32(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ uint16_t f32_to_h16(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(uint16_t, h);
}
// the low / high element of a packed pair
__device__ __forceinline__ float h16lo_to_f32(uint32_t pair) { return h16_to_f32((uint16_t)pair); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t pair) { return h16_to_f32((uint16_t)(pair >> 16)); }
#else
#define RG_H16_NAME "bf16"
#define RG_H16_ONE 0x3f80      /* 1.0 */
__device__ __forceinline__ float h16_to_f32(uint16_t b) {
  return __uint_as_float(((uint32_t)b) << 16);
}
__device__ __forceinline__ uint16_t f32_to_h16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ float h16lo_to_f32(uint32_t pair) { return __uint_as_float(pair << 16); }
__device__ __forceinline__ float h16hi_to_f32(uint32_t pair) { return __uint_as_float(pair & 0xffff0000u); }
#endif

// matrix instructions on the library's 16-bit type (8 elements per lane and operand; the trailing cbsz / abid / blgp arguments of
// the builtins are always 0 here and are accepted for call-site compatibility)
typedef __attribute__((ext_vector_type(16))) float rg_f32x16;
typedef __attribute__((ext_vector_type(4))) float rg_f32x4;
#ifdef RG_HALF_F16
typedef __attribute__((ext_vector_type(8))) _Float16 rg_h16x8;
#define RG_MFMA_H16_ASM_16x16x32 "v_mfma_f32_16x16x32_f16"
__device__ __forceinline__ rg_f32x16 rg_mfma_h16_32x32x16(rg_h16x8 a, rg_h16x8 b, rg_f32x16 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ rg_f32x4 rg_mfma_h16_16x16x32(rg_h16x8 a, rg_h16x8 b, rg_f32x4 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
#else
typedef __attribute__((ext_vector_type(8))) __bf16 rg_h16x8;
#define RG_MFMA_H16_ASM_16x16x32 "v_mfma_f32_16x16x32_bf16"
__device__ __forceinline__ rg_f32x16 rg_mfma_h16_32x32x16(rg_h16x8 a, rg_h16x8 b, rg_f32x16 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ rg_f32x4 rg_mfma_h16_16x16x32(rg_h16x8 a, rg_h16x8 b, rg_f32x4 c, int, int, int) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#endif

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int dtype = RG_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float round(float v) { return v; }
};
template <> struct Elem<h16_t> {
  static constexpr int dtype = RG_H16;
  __device__ static __forceinline__ float ld(const h16_t* p) { return h16_to_f32(p->bits); }
  __device__ static __forceinline__ void st(h16_t* p, float v) { p->bits = f32_to_h16(v); }
  __device__ static __forceinline__ float round(float v) { return h16_to_f32(f32_to_h16(v)); }
};

// vector load/store of VEC (1 or 4) consecutive elements as floats
template <typename T, int VEC> struct Vec;
template <> struct Vec<float, 1> {
  __device__ static __forceinline__ void ld(const float* p, float* o) { o[0] = p[0]; }
  __device__ static __forceinline__ void st(float* p, const float* v) { p[0] = v[0]; }
};
template <> struct Vec<float, 4> {
  __device__ static __forceinline__ void ld(const float* p, float* o) {
    float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
  }
  __device__ static __forceinline__ void st(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec<h16_t, 1> {
  __device__ static __forceinline__ void ld(const h16_t* p, float* o) { o[0] = h16_to_f32(p->bits); }
  __device__ static __forceinline__ void st(h16_t* p, const float* v) { p->bits = f32_to_h16(v[0]); }
};
template <> struct Vec<h16_t, 4> {
  __device__ static __forceinline__ void ld(const h16_t* p, float* o) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    o[0] = h16lo_to_f32(t.x); o[1] = h16hi_to_f32(t.x);
    o[2] = h16lo_to_f32(t.y); o[3] = h16hi_to_f32(t.y);
  }
  __device__ static __forceinline__ void st(h16_t* p, const float* v) {
    uint2 t;
    t.x = (uint32_t)f32_to_h16(v[0]) | ((uint32_t)f32_to_h16(v[1]) << 16);
    t.y = (uint32_t)f32_to_h16(v[2]) | ((uint32_t)f32_to_h16(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = t;
  }
};

template <> struct Vec<float, 8> {
  __device__ static __forceinline__ void ld(const float* p, float* o) { Vec<float, 4>::ld(p, o); Vec<float, 4>::ld(p + 4, o + 4); }
  __device__ static __forceinline__ void st(float* p, const float* v) { Vec<float, 4>::st(p, v); Vec<float, 4>::st(p + 4, v + 4); }
};
template <> struct Vec<h16_t, 8> {
  __device__ static __forceinline__ void ld(const h16_t* p, float* o) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    o[0] = h16lo_to_f32(t.x); o[1] = h16hi_to_f32(t.x);
    o[2] = h16lo_to_f32(t.y); o[3] = h16hi_to_f32(t.y);
    o[4] = h16lo_to_f32(t.z); o[5] = h16hi_to_f32(t.z);
    o[6] = h16lo_to_f32(t.w); o[7] = h16hi_to_f32(t.w);
  }
  __device__ static __forceinline__ void st(h16_t* p, const float* v) {
    uint4 t;
    t.x = (uint32_t)f32_to_h16(v[0]) | ((uint32_t)f32_to_h16(v[1]) << 16);
    t.y = (uint32_t)f32_to_h16(v[2]) | ((uint32_t)f32_to_h16(v[3]) << 16);
    t.z = (uint32_t)f32_to_h16(v[4]) | ((uint32_t)f32_to_h16(v[5]) << 16);
    t.w = (uint32_t)f32_to_h16(v[6]) | ((uint32_t)f32_to_h16(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = t;
  }
};

// VEC consecutive elements AS LOADED (a native vector: one load instruction, no arithmetic behind it); cvt() widens to floats.
// The row kernels of rg_bn.hip issue the loads of several rows / operands first and convert when they compute: with the
// conversion inside the load helper (Vec<>::ld above) hipcc serialised the loads of the multi-operand passes -- load, load,
// s_waitcnt vmcnt(0), convert, load, ... in the ISA of the BatchNorm-backward apply.
typedef __attribute__((ext_vector_type(4))) unsigned rg_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned rg_u32x2;
template <typename T, int VEC> struct RawVec;
template <> struct RawVec<float, 1> {
  float r;
  __device__ __forceinline__ void ld(const float* p) { r = p[0]; }
  __device__ __forceinline__ void cvt(float* o) const { o[0] = r; }
};
template <> struct RawVec<float, 4> {
  rg_f32x4 r;
  __device__ __forceinline__ void ld(const float* p) { r = *reinterpret_cast<const rg_f32x4*>(p); }
  __device__ __forceinline__ void cvt(float* o) const { o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
};
template <> struct RawVec<float, 8> {
  rg_f32x4 r0, r1;
  __device__ __forceinline__ void ld(const float* p) {
    r0 = *reinterpret_cast<const rg_f32x4*>(p); r1 = *reinterpret_cast<const rg_f32x4*>(p + 4);
  }
  __device__ __forceinline__ void cvt(float* o) const {
    o[0] = r0.x; o[1] = r0.y; o[2] = r0.z; o[3] = r0.w; o[4] = r1.x; o[5] = r1.y; o[6] = r1.z; o[7] = r1.w;
  }
};
template <> struct RawVec<h16_t, 1> {
  uint16_t r;
  __device__ __forceinline__ void ld(const h16_t* p) { r = p->bits; }
  __device__ __forceinline__ void cvt(float* o) const { o[0] = h16_to_f32(r); }
};
template <> struct RawVec<h16_t, 4> {
  rg_u32x2 r;
  __device__ __forceinline__ void ld(const h16_t* p) { r = *reinterpret_cast<const rg_u32x2*>(p); }
  __device__ __forceinline__ void cvt(float* o) const {
    o[0] = h16lo_to_f32(r.x); o[1] = h16hi_to_f32(r.x);
    o[2] = h16lo_to_f32(r.y); o[3] = h16hi_to_f32(r.y);
  }
};
template <> struct RawVec<h16_t, 8> {
  rg_u32x4 r;
  __device__ __forceinline__ void ld(const h16_t* p) { r = *reinterpret_cast<const rg_u32x4*>(p); }
  __device__ __forceinline__ void cvt(float* o) const {
    o[0] = h16lo_to_f32(r.x); o[1] = h16hi_to_f32(r.x);
    o[2] = h16lo_to_f32(r.y); o[3] = h16hi_to_f32(r.y);
    o[4] = h16lo_to_f32(r.z); o[5] = h16hi_to_f32(r.z);
    o[6] = h16lo_to_f32(r.w); o[7] = h16hi_to_f32(r.w);
  }
};

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float lrelu_mask(float v, float slope) { return v > 0.f ? 1.f : slope; }

// 64-lane wave reduction (sum)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// block-wide sum for blockDim.x == 256; result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem4[w] = v;
  __syncthreads();
  return smem4[0] + smem4[1] + smem4[2] + smem4[3];
}

// ---- resize-convolution geometry (DCGANUpGenerator): bilinear x2 (align_corners=False) + ReflectionPad2d(1)
// padded coordinate i in [0, 2L+2) -> upsampled coordinate (reflection without repeating the edge)
__device__ __forceinline__ int up_reflect(int i, int L2) {
  int u = i - 1;
  return u < 0 ? -u : (u >= L2 ? 2 * L2 - 2 - u : u);
}
// bilinear x2: u = 2q   -> 0.25 x[q-1] + 0.75 x[q]   (x[-1] := x[0])
//              u = 2q+1 -> 0.75 x[q]   + 0.25 x[q+1] (x[L]  := x[L-1]);  out = (1 - l1) x[i0] + l1 x[i1]
__device__ __forceinline__ void up_taps(int u, int L, int& i0, int& i1, float& l1) {
  int q = u >> 1;
  if (u & 1) { i0 = q; i1 = min(q + 1, L - 1); l1 = 0.25f; }
  else       { i0 = max(q - 1, 0); i1 = q; l1 = 0.75f; }
}

// ---- the Adam update of one element (torch.optim.Adam, single-tensor path), shared by every kernel that applies it (rg_misc.hip's
// streaming kernels, the generator layer-0 and conv weight-gradient kernels that step their tensor in the epilogue): ONE
// expression, so that a tensor stepped by any of them comes out bit-identical.  hyper[0..7] = b1, b2, 1 - b1, 1 - b2, eps,
// lr / bc1, 1 / sqrt(bc2), weight decay (rg_adam_hyper_dev); hyper[8] = 1 / loss scale: the factor every kernel applies to the
// gradient it reads before the update (1 except in the fp16 build's loss-scaled backward; a multiplication by 1.0f is exact, so the
// unscaled paths are bit for bit what they were).
__device__ __forceinline__ void rg_adam_upd(float& pp, float gg, float& mm, float& vv, float b2, float omb1, float omb2, float eps,
                                            float step_size, float inv_sqrt_bc2, float wd) {
  if (wd != 0.f) gg += wd * pp;               // torch.optim.Adam weight_decay (L2 on the gradient); betaVAE training
  // m = b1*m + (1-b1)g ; v = b2*v + (1-b2)g^2 ; denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m/denom
  // (1-beta) is rounded from double like torch's python-side `1 - beta2`; lerp form for m as torch
  mm = mm + omb1 * (gg - mm);
  vv = b2 * vv + omb2 * gg * gg;
  float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
  pp -= step_size * (mm / denom);
}
struct RgAdamHyper {
  float b1, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd, ginv;
  __device__ __forceinline__ void load(const float* __restrict__ h) {
    b1 = h[0]; b2 = h[1]; omb1 = h[2]; omb2 = h[3]; eps = h[4]; step_size = h[5]; inv_sqrt_bc2 = h[6]; wd = h[7]; ginv = h[8];
  }
  __device__ __forceinline__ void upd(float& pp, float gg, float& mm, float& vv) const {
    rg_adam_upd(pp, gg * ginv, mm, vv, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd);
  }
};

// dtype dispatch helper for host code
#define RG_DISPATCH_DTYPE(dtype, T, ...)                                  \
  if ((dtype) == RG_F32) { using T = float; __VA_ARGS__ }                 \
  else if ((dtype) == RG_H16) { using T = h16_t; __VA_ARGS__ }          \
  else { rg_set_error("bad dtype %d", (int)(dtype)); return RG_EINVAL; }
