// rg_misc.hip -- image-side pointwise ops, scalar reductions, discriminator head, latent prep,
// Adam.  All HBM-bound streaming kernels: 16-byte accesses, grid-stride, deterministic two-stage sums.
#include "rg_common.h"

namespace {

constexpr int RED_BLOCKS = 1024;

inline unsigned grid_for(size_t n, int per_thread = 1) {
  size_t b = (n + (size_t)256 * per_thread - 1) / ((size_t)256 * per_thread);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// ---------------------------------------------------------------------------------- pointwise fp32
template <class F>
__global__ __launch_bounds__(256) void ew4_kernel(F f, size_t n) {
  size_t n4 = n / 4;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) f.vec(i * 4);
  // tail
  size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) f.one(t);
}

struct TanhBwd {
  const float* gy; const float* y; float* gz;
  __device__ void vec(size_t i) const {
    float4 g = *(const float4*)(gy + i), v = *(const float4*)(y + i);
    *(float4*)(gz + i) = make_float4(g.x * (1.f - v.x * v.x), g.y * (1.f - v.y * v.y), g.z * (1.f - v.z * v.z),
                                     g.w * (1.f - v.w * v.w));
  }
  __device__ void one(size_t i) const { gz[i] = gy[i] * (1.f - y[i] * y[i]); }
};
struct Interp {
  const float* r; const float* f; float* o; float eps;
  __device__ void vec(size_t i) const {
    float4 a = *(const float4*)(r + i), b = *(const float4*)(f + i);
    float e = eps, e1 = 1.f - eps;
    *(float4*)(o + i) = make_float4(e * a.x + e1 * b.x, e * a.y + e1 * b.y, e * a.z + e1 * b.z, e * a.w + e1 * b.w);
  }
  __device__ void one(size_t i) const { o[i] = eps * r[i] + (1.f - eps) * f[i]; }
};
struct InterpDev {
  const float* r; const float* f; float* o; const float* eps;
  __device__ void vec(size_t i) const {
    float4 a = *(const float4*)(r + i), b = *(const float4*)(f + i);
    float e = eps[0], e1 = 1.f - e;
    *(float4*)(o + i) = make_float4(e * a.x + e1 * b.x, e * a.y + e1 * b.y, e * a.z + e1 * b.z, e * a.w + e1 * b.w);
  }
  __device__ void one(size_t i) const { float e = eps[0]; o[i] = e * r[i] + (1.f - e) * f[i]; }
};
struct ScaleBy {
  const float* x; const float* coef; float* o;
  __device__ void vec(size_t i) const {
    float c = coef[0];
    float4 a = *(const float4*)(x + i);
    *(float4*)(o + i) = make_float4(a.x * c, a.y * c, a.z * c, a.w * c);
  }
  __device__ void one(size_t i) const { o[i] = x[i] * coef[0]; }
};
struct Clamp {
  float* p; float lo, hi;
  __device__ void vec(size_t i) const {
    float4 a = *(float4*)(p + i);
    *(float4*)(p + i) = make_float4(fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi), fminf(fmaxf(a.z, lo), hi),
                                    fminf(fmaxf(a.w, lo), hi));
  }
  __device__ void one(size_t i) const { p[i] = fminf(fmaxf(p[i], lo), hi); }
};
struct Adam {
  float* p; const float* g; float* m; float* v; float b1, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd = 0.f;
  float ginv = 1.f;                    // 1 / loss scale (hyper[8]; rg_common.h)
  __device__ __forceinline__ void upd(float& pp, float gg, float& mm, float& vv) const {
    rg_adam_upd(pp, gg * ginv, mm, vv, b2, omb1, omb2, eps, step_size, inv_sqrt_bc2, wd);      // rg_common.h: the one expression
  }
  __device__ void vec(size_t i) const {
    float4 P = *(float4*)(p + i), G = *(const float4*)(g + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
    upd(P.x, G.x, M.x, V.x); upd(P.y, G.y, M.y, V.y); upd(P.z, G.z, M.z, V.z); upd(P.w, G.w, M.w, V.w);
    *(float4*)(p + i) = P; *(float4*)(m + i) = M; *(float4*)(v + i) = V;
  }
  __device__ void one(size_t i) const { upd(p[i], g[i], m[i], v[i]); }
};

// Adam with the step-dependent constants in device memory.  WIRE: the gradient is read as bf16 from the all-reduced
// wire buffer of a data-parallel run instead of g; SHADOW: the rounded updated parameters are also written to a bf16
// buffer (the GEMM operand image of the tap-major conv weights).  The 7 constants are loaded once per thread.
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld4(const float* p) {
  nt_f4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nt_st4(float* p, float4 v) {
  nt_f4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}
template <bool WIRE, bool SHADOW>
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       const float* __restrict__ hyper, uint16_t* __restrict__ shadow,
                                                       const uint16_t* __restrict__ gw, size_t n) {
  const Adam a{p, g, m, v, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7], hyper[8]};
  const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
    const size_t i = q * 4;
    // m, v and g are streamed (touched once per step): non-temporal so that they do not evict the activations and
    // weight images the next kernels re-read from L2 / Infinity Cache
    float4 P = *(float4*)(p + i), M = nt_ld4(m + i), V = nt_ld4(v + i), G;
    if (WIRE) {
      const uint2 w = *(const uint2*)(gw + i);
      G = make_float4(h16lo_to_f32(w.x), h16hi_to_f32(w.x), h16lo_to_f32(w.y),
                      h16hi_to_f32(w.y));
    } else {
      G = nt_ld4(g + i);
    }
    a.upd(P.x, G.x, M.x, V.x); a.upd(P.y, G.y, M.y, V.y); a.upd(P.z, G.z, M.z, V.z); a.upd(P.w, G.w, M.w, V.w);
    *(float4*)(p + i) = P; nt_st4(m + i, M); nt_st4(v + i, V);
    if (SHADOW)
      *(uint2*)(shadow + i) = make_uint2((uint32_t)f32_to_h16(P.x) | ((uint32_t)f32_to_h16(P.y) << 16),
                                         (uint32_t)f32_to_h16(P.z) | ((uint32_t)f32_to_h16(P.w) << 16));
  }
  const size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // tail
  if (t < n) {
    a.upd(p[t], WIRE ? h16_to_f32(gw[t]) : g[t], m[t], v[t]);
    if (SHADOW) shadow[t] = f32_to_h16(p[t]);
  }
}

// ---- Adam over a flat buffer cut into SEGMENTS, some of whose gradients are still split-K partial slabs of their weight-
// gradient launch ([nsplit][n] fp32, rg_conv_wgrad_slabs): the step sums them itself, in slab order, instead of reading a
// reduced gradient -- the reduction launches of a backward pass (5-6 per pass at ~12 us: bandwidth-bound, 67 MB of slabs each)
// and the write + re-read of the reduced gradient disappear.  One launch for the whole buffer: every workgroup walks the
// segments in order.  A slab segment is processed by groups of SL threads per 16-byte column (SL = 1 / 4 / 16 chosen from
// nsplit: lane l sums slabs l, l + SL, ... with 8 loads in flight, the SL partial sums are combined through LDS in lane order:
// a fixed summation order, deterministic); the group's first thread applies Adam.
constexpr int ADAM_MAX_SEGS = 24;
struct AdamSeg { unsigned long long off, n; const float* slab; int nsplit; int s16; };      // s16: the slabs are bf16
struct AdamSegs { int nseg; int pad; AdamSeg s[ADAM_MAX_SEGS]; };

// one 4-element column piece of slab z: fp32 (16 bytes) or bf16 (8 bytes, widened)
template <bool S16>
__device__ __forceinline__ float4 adam_slab_ld(const float* __restrict__ slab, size_t z, size_t n, size_t q) {
  if (S16) {
    typedef unsigned nt_u2 __attribute__((ext_vector_type(2)));
    const nt_u2 w = __builtin_nontemporal_load(reinterpret_cast<const nt_u2*>(reinterpret_cast<const uint16_t*>(slab) + z * n + q * 4));
    return make_float4(h16lo_to_f32(w.x), h16hi_to_f32(w.x), h16lo_to_f32(w.y),
                       h16hi_to_f32(w.y));
  }
  return nt_ld4(slab + z * n + q * 4);
}

template <bool SHADOW, int SL, bool S16>
__device__ __forceinline__ void adam_slab_segment(const Adam& a, float* __restrict__ p, float* __restrict__ m,
                                                  float* __restrict__ v, uint16_t* __restrict__ shadow,
                                                  const float* __restrict__ slab, int nsplit, size_t n, float4 (*sm)[64]) {
  // n is a multiple of 4 (conv weights: O * 16 * I); a block takes 256 / SL columns per trip
  constexpr int COLS = 256 / SL;
  const int col = threadIdx.x % COLS, l = threadIdx.x / COLS;
  const size_t n4 = n / 4;
  const size_t trips = (n4 + COLS - 1) / COLS;
  for (size_t tb = blockIdx.x; tb < trips; tb += gridDim.x) {
    const size_t q = tb * COLS + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < n4) {
      int z = l;
      for (; z + 7 * SL < nsplit; z += 8 * SL) {           // 8 independent loads, added in slab order
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = adam_slab_ld<S16>(slab, (size_t)(z + k * SL), n, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s.x += t[k].x; s.y += t[k].y; s.z += t[k].z; s.w += t[k].w; }
      }
      for (; z < nsplit; z += SL) {
        const float4 t = adam_slab_ld<S16>(slab, (size_t)z, n, q);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
    if (SL > 1) {
      __syncthreads();                                     // the previous trip's readers are done
      sm[l][col] = s;
      __syncthreads();
      if (l == 0) {
#pragma unroll
        for (int k = 1; k < SL; ++k) { const float4 t = sm[k][col]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
      }
    }
    if (l == 0 && q < n4) {
      const size_t i = q * 4;
      float4 P = *(float4*)(p + i), M = nt_ld4(m + i), V = nt_ld4(v + i);
      a.upd(P.x, s.x, M.x, V.x); a.upd(P.y, s.y, M.y, V.y); a.upd(P.z, s.z, M.z, V.z); a.upd(P.w, s.w, M.w, V.w);
      *(float4*)(p + i) = P; nt_st4(m + i, M); nt_st4(v + i, V);
      if (SHADOW)
        *(uint2*)(shadow + i) = make_uint2((uint32_t)f32_to_h16(P.x) | ((uint32_t)f32_to_h16(P.y) << 16),
                                           (uint32_t)f32_to_h16(P.z) | ((uint32_t)f32_to_h16(P.w) << 16));
    }
  }
}

template <bool SHADOW>
__global__ __launch_bounds__(256) void adam_segs_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v,
                                                        const float* __restrict__ hyper, uint16_t* __restrict__ shadow,
                                                        AdamSegs t) {
  __shared__ float4 sm[16][64];                            // [slab lane][column]: SL = 4 uses [4][64], SL = 16 [16][16]
  const Adam a{p, g, m, v, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7], hyper[8]};
  for (int si = 0; si < t.nseg; ++si) {
    const AdamSeg sg = t.s[si];
    if (sg.nsplit < 0) continue;                           // stepped elsewhere
    float* ps = p + sg.off; float* ms = m + sg.off; float* vs = v + sg.off;
    uint16_t* sh = SHADOW ? shadow + sg.off : nullptr;
    if (sg.slab) {
      if (sg.s16) {
        if (sg.nsplit <= 4) adam_slab_segment<SHADOW, 1, true>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, sm);
        else if (sg.nsplit <= 32) adam_slab_segment<SHADOW, 4, true>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, sm);
        else adam_slab_segment<SHADOW, 16, true>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, reinterpret_cast<float4(*)[64]>(sm));
      } else {
        if (sg.nsplit <= 4) adam_slab_segment<SHADOW, 1, false>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, sm);
        else if (sg.nsplit <= 32) adam_slab_segment<SHADOW, 4, false>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, sm);
        else adam_slab_segment<SHADOW, 16, false>(a, ps, ms, vs, sh, sg.slab, sg.nsplit, sg.n, reinterpret_cast<float4(*)[64]>(sm));
      }
      continue;
    }
    const float* gs = g + sg.off;
    const size_t n4 = sg.n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
      const size_t i = q * 4;
      float4 P = *(float4*)(ps + i), M = nt_ld4(ms + i), V = nt_ld4(vs + i), G = nt_ld4(gs + i);
      a.upd(P.x, G.x, M.x, V.x); a.upd(P.y, G.y, M.y, V.y); a.upd(P.z, G.z, M.z, V.z); a.upd(P.w, G.w, M.w, V.w);
      *(float4*)(ps + i) = P; nt_st4(ms + i, M); nt_st4(vs + i, V);
      if (SHADOW)
        *(uint2*)(sh + i) = make_uint2((uint32_t)f32_to_h16(P.x) | ((uint32_t)f32_to_h16(P.y) << 16),
                                       (uint32_t)f32_to_h16(P.z) | ((uint32_t)f32_to_h16(P.w) << 16));
    }
    const size_t tl = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // tail (a segment that is not a multiple of 4)
    if (tl < sg.n) {
      a.upd(ps[tl], gs[tl], ms[tl], vs[tl]);
      if (SHADOW) sh[tl] = f32_to_h16(ps[tl]);
    }
  }
}

// ---- the data-parallel counterpart of adam_segs_kernel: the flat fp32 gradient goes onto the bf16 WIRE buffer of the all-reduce
// (rg_cast_pad's job) with the same segment table -- a slab segment is summed here (slab order, fp32) and rounded once onto the
// wire, so the reduction launch of every split layer and the fp32 gradient it wrote for this pass to read back disappear; a
// skipped segment (nsplit = -1) was put on the wire by its weight-gradient launch (rg_conv_wgrad_wire).
template <int SL, bool S16>
__device__ __forceinline__ void wire_slab_segment(uint16_t* __restrict__ wire, const float* __restrict__ slab, int nsplit,
                                                  size_t n, float4 (*sm)[64]) {
  constexpr int COLS = 256 / SL;
  const int col = threadIdx.x % COLS, l = threadIdx.x / COLS;
  const size_t n4 = n / 4;
  const size_t trips = (n4 + COLS - 1) / COLS;
  for (size_t tb = blockIdx.x; tb < trips; tb += gridDim.x) {
    const size_t q = tb * COLS + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < n4) {
      int z = l;
      for (; z + 7 * SL < nsplit; z += 8 * SL) {
        float4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = adam_slab_ld<S16>(slab, (size_t)(z + k * SL), n, q);
#pragma unroll
        for (int k = 0; k < 8; ++k) { s.x += t[k].x; s.y += t[k].y; s.z += t[k].z; s.w += t[k].w; }
      }
      for (; z < nsplit; z += SL) {
        const float4 t = adam_slab_ld<S16>(slab, (size_t)z, n, q);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
    }
    if (SL > 1) {
      __syncthreads();
      sm[l][col] = s;
      __syncthreads();
      if (l == 0) {
#pragma unroll
        for (int k = 1; k < SL; ++k) { const float4 t = sm[k][col]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
      }
    }
    if (l == 0 && q < n4)
      *(uint2*)(wire + q * 4) = make_uint2((uint32_t)f32_to_h16(s.x) | ((uint32_t)f32_to_h16(s.y) << 16),
                                           (uint32_t)f32_to_h16(s.z) | ((uint32_t)f32_to_h16(s.w) << 16));
  }
}

__global__ __launch_bounds__(256) void wire_segs_kernel(const float* __restrict__ g, uint16_t* __restrict__ wire, AdamSegs t) {
  __shared__ float4 sm[16][64];
  for (int si = 0; si < t.nseg; ++si) {
    const AdamSeg sg = t.s[si];
    if (sg.nsplit < 0) continue;                           // already on the wire
    uint16_t* ws = wire + sg.off;
    if (sg.slab) {
      if (sg.s16) {
        if (sg.nsplit <= 4) wire_slab_segment<1, true>(ws, sg.slab, sg.nsplit, sg.n, sm);
        else if (sg.nsplit <= 32) wire_slab_segment<4, true>(ws, sg.slab, sg.nsplit, sg.n, sm);
        else wire_slab_segment<16, true>(ws, sg.slab, sg.nsplit, sg.n, reinterpret_cast<float4(*)[64]>(sm));
      } else {
        if (sg.nsplit <= 4) wire_slab_segment<1, false>(ws, sg.slab, sg.nsplit, sg.n, sm);
        else if (sg.nsplit <= 32) wire_slab_segment<4, false>(ws, sg.slab, sg.nsplit, sg.n, sm);
        else wire_slab_segment<16, false>(ws, sg.slab, sg.nsplit, sg.n, reinterpret_cast<float4(*)[64]>(sm));
      }
      continue;
    }
    const float* gs = g + sg.off;
    const size_t n4 = sg.n / 4, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
      const float4 G = nt_ld4(gs + q * 4);
      *(uint2*)(ws + q * 4) = make_uint2((uint32_t)f32_to_h16(G.x) | ((uint32_t)f32_to_h16(G.y) << 16),
                                         (uint32_t)f32_to_h16(G.z) | ((uint32_t)f32_to_h16(G.w) << 16));
    }
    const size_t tl = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tl < sg.n) ws[tl] = f32_to_h16(gs[tl]);
  }
}

// element-per-thread form for ranges that do not start on a 16-byte boundary (the small bias / BatchNorm ranges between the
// nn.Linear weights that rg_linear_wgrad_adam steps itself): fp32 gradient, no shadow
__global__ __launch_bounds__(256) void adam_dev_scalar_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              const float* __restrict__ hyper, size_t n) {
  const Adam a{p, g, m, v, hyper[0], hyper[1], hyper[2], hyper[3], hyper[4], hyper[5], hyper[6], hyper[7], hyper[8]};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a.upd(p[i], g[i], m[i], v[i]);
}

__global__ void adam_hyper_kernel(int* step_dev, double lr, double b1, double b2, double eps, double wd, double ginv, float* hyper) {
  hyper[7] = (float)wd;
  hyper[8] = (float)ginv;
  int step = *step_dev + 1;
  *step_dev = step;
  double bc1 = 1.0 - pow(b1, (double)step);
  double bc2 = 1.0 - pow(b2, (double)step);
  hyper[0] = (float)b1; hyper[1] = (float)b2; hyper[2] = (float)(1.0 - b1); hyper[3] = (float)(1.0 - b2);
  hyper[4] = (float)eps; hyper[5] = (float)(lr / bc1); hyper[6] = (float)(1.0 / sqrt(bc2));
}

// ---------------------------------------------------------------------------------- reductions
// stage 1: each block writes one partial (double accumulation across a thread's strided elements is
// avoided: fp32 per-thread sums over <= n/(blocks*256) items, then tree) ; stage 2: one block.
template <class F>
__global__ __launch_bounds__(256) void reduce1_kernel(F f, size_t n, float* partial) {
  __shared__ float sm[4];
  float s = 0.f;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += f(i);
  float t = block_sum_256(s, sm);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}
template <class Fin>
__global__ __launch_bounds__(256) void reduce2_kernel(Fin fin, const float* partial, int nb) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  float t = block_sum_256(s, sm);
  if (threadIdx.x == 0) fin(t);
}
struct SqF {
  const float* x;
  __device__ float operator()(size_t i) const { float v = x[i]; return v * v; }
};
struct StoreFin {
  float* out;
  __device__ void operator()(float t) const { out[0] = t; }
};

// out[c] (+)= sum over n,hw of g[n][c][hw]: one block-row per (c, chunk)
__global__ __launch_bounds__(256) void nchw_chan_partial_kernel(const float* g, float* partial, int N, int C, int HW,
                                                                int chunks) {
  __shared__ float sm[4];
  int c = blockIdx.x, ch = blockIdx.y;
  size_t per = ((size_t)N * HW + chunks - 1) / chunks;
  size_t b = (size_t)ch * per, e = b + per;
  size_t tot = (size_t)N * HW;
  if (e > tot) e = tot;
  float s = 0.f;
  for (size_t i = b + threadIdx.x; i < e; i += 256) {
    size_t n = i / HW, hw = i - n * HW;
    s += g[(n * C + c) * (size_t)HW + hw];
  }
  float t = block_sum_256(s, sm);
  if (threadIdx.x == 0) partial[(size_t)c * chunks + ch] = t;
}
__global__ void nchw_chan_final_kernel(const float* partial, float* out, int C, int chunks, int accumulate) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < chunks; ++i) s += partial[(size_t)c * chunks + i];
  out[c] = accumulate ? out[c] + s : s;
}

// in_inv = 1 / (the scale the gradient whose squared norm is sq carries), out_scale = the scale the tangent direction is to
// carry (both 1 outside the fp16 build's loss-scaled penalty step; powers of two, so exact)
__global__ void gp_coef_kernel(const float* sq, float* loss, float* coef, float lambd, float in_inv, float out_scale) {
  float nrm = sqrtf(sq[0]) * in_inv;
  loss[0] = (nrm - 1.f) * (nrm - 1.f);
  coef[0] = lambd * 2.f * (nrm - 1.f) / nrm * (in_inv * out_scale);
}

__global__ __launch_bounds__(256) void mean_diff_kernel(const float* a, const float* b, float* out, int n, float sign) {
  __shared__ float sm[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += b ? (a[i] - b[i]) : a[i];
  float t = block_sum_256(s, sm);
  if (threadIdx.x == 0) out[0] = sign * t / (float)n;
}

// latent prep: per column e: v = u+z ; mean, unbiased std over the N rows ; out = (v-mean)/std
// block = 64 columns x 4 row groups; a thread keeps its <= 16 rows in registers (one pass over memory), the
// row groups are combined through LDS.  N > 64 takes the plain three-pass kernel.
__global__ __launch_bounds__(256) void latent_prep_kernel(const float* __restrict__ u, const float* __restrict__ z,
                                                          float* __restrict__ out, int N, int E) {
  __shared__ float sm[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + col;
  const bool ok = e < E;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int n = rg + 4 * k;
    v[k] = 0.f;
    if (ok && n < N) v[k] = u[(size_t)n * E + e] + z[(size_t)n * E + e];
    s += v[k];
  }
  sm[rg][col] = s;
  __syncthreads();
  const float mu = (sm[0][col] + sm[1][col] + sm[2][col] + sm[3][col]) / (float)N;
  __syncthreads();
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float d = v[k] - mu;
    if (rg + 4 * k < N) ss += d * d;
  }
  sm[rg][col] = ss;
  __syncthreads();
  const float sd = sqrtf((sm[0][col] + sm[1][col] + sm[2][col] + sm[3][col]) / (float)(N - 1));   // N == 1 -> NaN, as torch.std
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int n = rg + 4 * k;
    if (ok && n < N) out[(size_t)n * E + e] = (v[k] - mu) / sd;
  }
}
__global__ void latent_prep_big_kernel(const float* u, const float* z, float* out, int N, int E) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += u[(size_t)n * E + e] + z[(size_t)n * E + e];
  float mu = s / (float)N;
  float ss = 0.f;
  for (int n = 0; n < N; ++n) {
    float d = u[(size_t)n * E + e] + z[(size_t)n * E + e] - mu;
    ss += d * d;
  }
  float sd = sqrtf(ss / (float)(N - 1));
  for (int n = 0; n < N; ++n) out[(size_t)n * E + e] = (u[(size_t)n * E + e] + z[(size_t)n * E + e] - mu) / sd;
}

// ---------------------------------------------------------------------------------- head
// h[n] = sum_j a[n][j] * wq[j], j = tap*C + c, wq[j] = round_T(w[c*16 + tap])
template <typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* a, const float* w, float* h, float* out, int C,
                                                       float slope) {
  // thread = channel: its 16 taps are one contiguous 64-byte weight row, and for a fixed tap the block reads
  // consecutive channels of the activation
  __shared__ float sm[4];
  const int n = blockIdx.x;
  const T* an = a + (size_t)n * 16 * C;
  float s = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    float wv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) Vec<float, 4>::ld(w + (size_t)c * 16 + 4 * q, wv + 4 * q);
#pragma unroll
    for (int tap = 0; tap < 16; ++tap) s += Elem<T>::ld(an + (size_t)tap * C + c) * Elem<T>::round(wv[tap]);
  }
  float t = block_sum_256(s, sm);
  if (threadIdx.x == 0) { h[n] = t; out[n] = lrelu_f(t, slope); }
}
__global__ void head_grad_kernel(const float* h, float* gh, int N, float coef, float slope) {
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) gh[n] = coef * lrelu_mask(h[n], slope);
}
template <typename T>
__global__ void head_bwd_data_kernel(const float* gh, const float* w, T* ga, int N, int C) {
  size_t J = (size_t)16 * C, tot = (size_t)N * J;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
    size_t n = i / J;
    int j = (int)(i - n * J);
    int tap = j / C, c = j - tap * C;
    Elem<T>::st(ga + i, gh[n] * Elem<T>::round(w[c * 16 + tap]));
  }
}
template <typename T>
__global__ void head_wgrad_kernel(const float* gh, const T* a, float* dw, int N, int C, int accumulate) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  int J = 16 * C;
  if (j >= J) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s += gh[n] * Elem<T>::ld(a + (size_t)n * J + j);
  int tap = j / C, c = j - tap * C;
  float* d = dw + c * 16 + tap;
  *d = accumulate ? *d + s : s;
}

// bf16 form: the batch is split over 8 thread groups (a thread: 8 consecutive j = one 16-byte load per sample, N / 8 independent
// loads in flight instead of a serial walk over the batch: 24 -> 5 us at N = 64, C = 2048), combined through LDS in a fixed order
__global__ __launch_bounds__(256) void head_wgrad_bf16_kernel(const float* __restrict__ gh, const uint16_t* __restrict__ a,
                                                              float* __restrict__ dw, int N, int C, int accumulate) {
  __shared__ float sm[8][256 + 8];
  const int t = threadIdx.x, jg = t & 31, ng = t >> 5;
  const int J = 16 * C, j0 = blockIdx.x * 256 + jg * 8;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (j0 < J) {
#pragma unroll 4
    for (int n = ng; n < N; n += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(a + (size_t)n * J + j0);
      const float g = gh[n];
      const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] += g * h16lo_to_f32(d[k]);
        acc[2 * k + 1] += g * h16hi_to_f32(d[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) sm[ng][jg * 8 + k] = acc[k];
  __syncthreads();
  const int j = blockIdx.x * 256 + t;
  if (j < J) {
    float s = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) s += sm[g8][t];
    const int tap = j / C, c = j - tap * C;
    float* d = dw + c * 16 + tap;
    *d = accumulate ? *d + s : s;
  }
}

__global__ void widen_bf16_kernel(const uint16_t* src, float* dst, size_t n) {
  size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    uint2 v = reinterpret_cast<const uint2*>(src)[i];
    reinterpret_cast<float4*>(dst)[i] = make_float4(h16lo_to_f32(v.x), h16hi_to_f32(v.x),
                                                    h16lo_to_f32(v.y), h16hi_to_f32(v.y));
  }
  size_t t = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) dst[t] = h16_to_f32(src[t]);
}

template <typename T>
__global__ void cast_pad_kernel(const float* src, T* dst, int M, int K, int ldd) {
  size_t tot = (size_t)M * ldd;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
    size_t m = i / ldd;
    int k = (int)(i - m * ldd);
    Elem<T>::st(dst + i, k < K ? src[m * K + k] : 0.f);
  }
}

// unpadded case (K == ldd: the data-parallel gradient compression, whole weight matrices): 8 elements per thread, two
// 16-byte loads and one 16-byte store
__global__ __launch_bounds__(256) void cast_bf16x8_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = *reinterpret_cast<const float4*>(src + 8 * i), b = *reinterpret_cast<const float4*>(src + 8 * i + 4);
    uint4 o;
    o.x = (uint32_t)f32_to_h16(a.x) | ((uint32_t)f32_to_h16(a.y) << 16);
    o.y = (uint32_t)f32_to_h16(a.z) | ((uint32_t)f32_to_h16(a.w) << 16);
    o.z = (uint32_t)f32_to_h16(b.x) | ((uint32_t)f32_to_h16(b.y) << 16);
    o.w = (uint32_t)f32_to_h16(b.z) | ((uint32_t)f32_to_h16(b.w) << 16);
    *reinterpret_cast<uint4*>(dst + 8 * i) = o;
  }
}

}  // namespace

#define EW_LAUNCH(name, functor, n, st)                                                        \
  do {                                                                                         \
    if ((n) == 0) return RG_OK;                                                                \
    hipLaunchKernelGGL((ew4_kernel<decltype(functor)>), dim3(grid_for((n), 4)), dim3(256), 0, st, functor, n); \
    RG_LAUNCH_CHECK(name);                                                                     \
    return RG_OK;                                                                              \
  } while (0)

static inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int rg_tanh_bwd(const float* gy, const float* y, float* gz, size_t n, void* stream) {
  RG_REQUIRE(gy && y && gz && aligned16(gy) && aligned16(y) && aligned16(gz), RG_EINVAL, "tanh_bwd: bad args");
  TanhBwd f{gy, y, gz};
  EW_LAUNCH("tanh_bwd", f, n, rg_stream(stream));
}
extern "C" int rg_interp(const float* real, const float* fake, float* out, size_t n, float eps, void* stream) {
  RG_REQUIRE(real && fake && out && aligned16(real) && aligned16(fake) && aligned16(out), RG_EINVAL, "interp: bad args");
  Interp f{real, fake, out, eps};
  EW_LAUNCH("interp", f, n, rg_stream(stream));
}
extern "C" int rg_scale_by(const float* x, const float* coef, float* out, size_t n, void* stream) {
  RG_REQUIRE(x && coef && out && aligned16(x) && aligned16(out), RG_EINVAL, "scale_by: bad args");
  ScaleBy f{x, coef, out};
  EW_LAUNCH("scale_by", f, n, rg_stream(stream));
}
extern "C" int rg_clamp(float* p, size_t n, float lo, float hi, void* stream) {
  RG_REQUIRE(p && aligned16(p), RG_EINVAL, "clamp: bad args");
  Clamp f{p, lo, hi};
  EW_LAUNCH("clamp", f, n, rg_stream(stream));
}
extern "C" int rg_adam_step(float* p, const float* g, float* m, float* v, size_t n, int step, double lr, double beta1,
                            double beta2, double eps, void* stream) {
  RG_REQUIRE(p && g && m && v && step >= 1, RG_EINVAL, "adam_step: bad args");
  RG_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), RG_EINVAL, "adam_step: 16-byte alignment");
  double bc1 = 1.0 - pow(beta1, (double)step);
  double bc2 = 1.0 - pow(beta2, (double)step);
  Adam f{p, g, m, v, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
         (float)(lr / bc1), (float)(1.0 / sqrt(bc2))};
  EW_LAUNCH("adam_step", f, n, rg_stream(stream));
}

extern "C" int rg_adam_step_dev(float* p, const float* g, float* m, float* v, size_t n, const float* hyper,
                                void* shadow_bf16, const void* grad_bf16, void* stream) {
  RG_REQUIRE(p && (g || grad_bf16) && m && v && hyper, RG_EINVAL, "adam_step_dev: bad args");
  if (n == 0) return RG_OK;
  if (!(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) && !shadow_bf16 && !grad_bf16 && g &&
      ((((uintptr_t)p ^ (uintptr_t)g) | ((uintptr_t)p ^ (uintptr_t)m) | ((uintptr_t)p ^ (uintptr_t)v)) & 3) == 0) {
    hipLaunchKernelGGL(adam_dev_scalar_kernel, dim3(grid_for(n, 1)), dim3(256), 0, rg_stream(stream), p, g, m, v, hyper, n);
    RG_LAUNCH_CHECK("adam_step_dev(scalar)");
    return RG_OK;
  }
  RG_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && ((uintptr_t)shadow_bf16 & 7) == 0 &&
                 ((uintptr_t)grad_bf16 & 7) == 0,
             RG_EINVAL, "adam_step_dev: alignment");
  uint16_t* sh = (uint16_t*)shadow_bf16;
  const uint16_t* gw = (const uint16_t*)grad_bf16;
  const dim3 grid(grid_for(n, 4)), block(256);
  hipStream_t st = rg_stream(stream);
  if (gw && sh) hipLaunchKernelGGL((adam_dev_kernel<true, true>), grid, block, 0, st, p, g, m, v, hyper, sh, gw, n);
  else if (gw) hipLaunchKernelGGL((adam_dev_kernel<true, false>), grid, block, 0, st, p, g, m, v, hyper, sh, gw, n);
  else if (sh) hipLaunchKernelGGL((adam_dev_kernel<false, true>), grid, block, 0, st, p, g, m, v, hyper, sh, gw, n);
  else hipLaunchKernelGGL((adam_dev_kernel<false, false>), grid, block, 0, st, p, g, m, v, hyper, sh, gw, n);
  RG_LAUNCH_CHECK("adam_step_dev");
  return RG_OK;
}
// Adam over [p, p + n) cut into nseg consecutive segments (seg_off / seg_n in elements, covering the range in order; every
// seg_off a multiple of 4); segment i with seg_slab[i] != NULL takes its gradient as the sum of seg_nsplit[i] fp32 slabs of
// seg_n[i] elements each (what rg_conv_wgrad_slabs left), the others read g.  See adam_segs_kernel.
extern "C" int rg_adam_step_slabs(float* p, const float* g, float* m, float* v, size_t n, const float* hyper,
                                  void* shadow_bf16, int nseg, const unsigned long long* seg_off,
                                  const unsigned long long* seg_n, const void* const* seg_slab, const int* seg_nsplit,
                                  const int* seg_dtype, void* stream) {
  RG_REQUIRE(p && g && m && v && hyper && seg_off && seg_n && seg_slab && seg_nsplit && seg_dtype, RG_EINVAL,
             "adam_step_slabs: bad args");
  RG_REQUIRE(nseg >= 1 && nseg <= ADAM_MAX_SEGS, RG_EINVAL, "adam_step_slabs: 1 .. %d segments", ADAM_MAX_SEGS);
  RG_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v) && ((uintptr_t)shadow_bf16 & 7) == 0, RG_EINVAL,
             "adam_step_slabs: alignment");
  AdamSegs t{};
  t.nseg = nseg;
  unsigned long long pos = 0;
  for (int i = 0; i < nseg; ++i) {
    RG_REQUIRE(seg_off[i] == pos && seg_off[i] % 4 == 0, RG_EINVAL, "adam_step_slabs: segments must tile the range in order, "
               "each starting on a multiple of 4 elements (segment %d)", i);
    RG_REQUIRE(!seg_slab[i] || (seg_nsplit[i] >= 1 && seg_n[i] % 4 == 0 && aligned16(seg_slab[i]) &&
                                (seg_dtype[i] == RG_F32 || seg_dtype[i] == RG_H16)), RG_EINVAL,
               "adam_step_slabs: slab segment %d", i);
    // nsplit = -1 without a slab: the segment is SKIPPED (its tensor is stepped by another launch: rg_conv_wgrad_adam)
    t.s[i] = AdamSeg{seg_off[i], seg_n[i], (const float*)seg_slab[i], seg_slab[i] ? seg_nsplit[i] : (seg_nsplit[i] < 0 ? -1 : 0),
                     seg_slab[i] && seg_dtype[i] == RG_H16 ? 1 : 0};
    pos += seg_n[i];
  }
  RG_REQUIRE(pos == n, RG_EINVAL, "adam_step_slabs: the segments cover %llu of %zu elements", pos, n);
  const dim3 grid(grid_for(n, 4)), block(256);
  hipStream_t st = rg_stream(stream);
  if (shadow_bf16) hipLaunchKernelGGL((adam_segs_kernel<true>), grid, block, 0, st, p, g, m, v, hyper, (uint16_t*)shadow_bf16, t);
  else hipLaunchKernelGGL((adam_segs_kernel<false>), grid, block, 0, st, p, g, m, v, hyper, (uint16_t*)nullptr, t);
  RG_LAUNCH_CHECK("adam_step_slabs");
  return RG_OK;
}
// The gradient's way onto the bf16 wire of a data-parallel all-reduce with the segment table of rg_adam_step_slabs (see
// wire_segs_kernel): g fp32 [n], wire bf16 [n]; plain segments are rounded, slab segments summed and rounded once, segments with
// nsplit = -1 left as they are.
extern "C" int rg_grad_to_wire(const float* g, void* wire_bf16, size_t n, int nseg, const unsigned long long* seg_off,
                               const unsigned long long* seg_n, const void* const* seg_slab, const int* seg_nsplit,
                               const int* seg_dtype, void* stream) {
  RG_REQUIRE(g && wire_bf16 && seg_off && seg_n && seg_slab && seg_nsplit && seg_dtype, RG_EINVAL, "grad_to_wire: bad args");
  RG_REQUIRE(nseg >= 1 && nseg <= ADAM_MAX_SEGS, RG_EINVAL, "grad_to_wire: 1 .. %d segments", ADAM_MAX_SEGS);
  RG_REQUIRE(aligned16(g) && ((uintptr_t)wire_bf16 & 7) == 0, RG_EINVAL, "grad_to_wire: alignment");
  AdamSegs t{};
  t.nseg = nseg;
  unsigned long long pos = 0;
  for (int i = 0; i < nseg; ++i) {
    RG_REQUIRE(seg_off[i] == pos && seg_off[i] % 4 == 0, RG_EINVAL, "grad_to_wire: segments must tile the range in order, each "
               "starting on a multiple of 4 elements (segment %d)", i);
    RG_REQUIRE(!seg_slab[i] || (seg_nsplit[i] >= 1 && seg_n[i] % 4 == 0 && aligned16(seg_slab[i]) &&
                                (seg_dtype[i] == RG_F32 || seg_dtype[i] == RG_H16)), RG_EINVAL, "grad_to_wire: slab segment %d", i);
    t.s[i] = AdamSeg{seg_off[i], seg_n[i], (const float*)seg_slab[i], seg_slab[i] ? seg_nsplit[i] : (seg_nsplit[i] < 0 ? -1 : 0),
                     seg_slab[i] && seg_dtype[i] == RG_H16 ? 1 : 0};
    pos += seg_n[i];
  }
  RG_REQUIRE(pos == n, RG_EINVAL, "grad_to_wire: the segments cover %llu of %zu elements", pos, n);
  hipLaunchKernelGGL(wire_segs_kernel, dim3(grid_for(n, 4)), dim3(256), 0, rg_stream(stream), g, (uint16_t*)wire_bf16, t);
  RG_LAUNCH_CHECK("grad_to_wire");
  return RG_OK;
}
extern "C" int rg_adam_hyper_dev2(int* step_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                                  double grad_scale_inv, float* hyper, void* stream) {
  RG_REQUIRE(step_dev && hyper && grad_scale_inv > 0.0, RG_EINVAL, "adam_hyper_dev: bad args");
  hipLaunchKernelGGL(adam_hyper_kernel, dim3(1), dim3(1), 0, rg_stream(stream), step_dev, lr, beta1, beta2, eps, weight_decay,
                     grad_scale_inv, hyper);
  RG_LAUNCH_CHECK("adam_hyper_dev");
  return RG_OK;
}
extern "C" int rg_adam_hyper_dev(int* step_dev, double lr, double beta1, double beta2, double eps, double weight_decay,
                                 float* hyper, void* stream) {
  return rg_adam_hyper_dev2(step_dev, lr, beta1, beta2, eps, weight_decay, 1.0, hyper, stream);
}
extern "C" int rg_interp_dev(const float* real, const float* fake, float* out, size_t n, const float* eps,
                             void* stream) {
  RG_REQUIRE(real && fake && out && eps && aligned16(real) && aligned16(fake) && aligned16(out), RG_EINVAL,
             "interp_dev: bad args");
  InterpDev f{real, fake, out, eps};
  EW_LAUNCH("interp_dev", f, n, rg_stream(stream));
}

extern "C" size_t rg_reduce_workspace_bytes(size_t n) { (void)n; return RED_BLOCKS * sizeof(float); }

extern "C" int rg_sqnorm(const float* x, float* out, size_t n, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(x && out, RG_EINVAL, "sqnorm: bad args");
  RG_REQUIRE(ws && ws_bytes >= RED_BLOCKS * sizeof(float), RG_EWORKSPACE, "sqnorm: workspace too small");
  hipStream_t st = rg_stream(stream);
  int nb = (int)((n + 255) / 256);
  if (nb > RED_BLOCKS) nb = RED_BLOCKS;
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL((reduce1_kernel<SqF>), dim3(nb), dim3(256), 0, st, SqF{x}, n, (float*)ws);
  RG_LAUNCH_CHECK("sqnorm");
  hipLaunchKernelGGL((reduce2_kernel<StoreFin>), dim3(1), dim3(256), 0, st, StoreFin{out}, (const float*)ws, nb);
  RG_LAUNCH_CHECK("sqnorm");
  return RG_OK;
}

extern "C" int rg_nchw_chan_sum(const float* g, float* out, int N, int C, int HW, int accumulate, void* ws,
                                size_t ws_bytes, void* stream) {
  RG_REQUIRE(g && out && N > 0 && C > 0 && HW > 0, RG_EINVAL, "nchw_chan_sum: bad args");
  int chunks = 256;
  RG_REQUIRE(ws && ws_bytes >= (size_t)C * chunks * sizeof(float), RG_EWORKSPACE, "nchw_chan_sum: workspace too small");
  hipStream_t st = rg_stream(stream);
  hipLaunchKernelGGL(nchw_chan_partial_kernel, dim3(C, chunks), dim3(256), 0, st, g, (float*)ws, N, C, HW, chunks);
  RG_LAUNCH_CHECK("nchw_chan_sum");
  hipLaunchKernelGGL(nchw_chan_final_kernel, dim3((C + 63) / 64), dim3(64), 0, st, (const float*)ws, out, C, chunks,
                     accumulate);
  RG_LAUNCH_CHECK("nchw_chan_sum");
  return RG_OK;
}

extern "C" int rg_gp_coef_scaled(const float* sq, float* loss, float* coef, float lambd, float in_scale, float out_scale,
                                 void* stream) {
  RG_REQUIRE(sq && loss && coef && in_scale > 0.f && out_scale > 0.f, RG_EINVAL, "gp_coef: bad args");
  hipLaunchKernelGGL(gp_coef_kernel, dim3(1), dim3(1), 0, rg_stream(stream), sq, loss, coef, lambd, 1.f / in_scale, out_scale);
  RG_LAUNCH_CHECK("gp_coef");
  return RG_OK;
}
extern "C" int rg_gp_coef(const float* sq, float* loss, float* coef, float lambd, void* stream) {
  RG_REQUIRE(sq && loss && coef, RG_EINVAL, "gp_coef: bad args");
  hipLaunchKernelGGL(gp_coef_kernel, dim3(1), dim3(1), 0, rg_stream(stream), sq, loss, coef, lambd, 1.f, 1.f);
  RG_LAUNCH_CHECK("gp_coef");
  return RG_OK;
}
extern "C" int rg_mean_diff(const float* a, const float* b, float* out, int n, float sign, void* stream) {
  RG_REQUIRE(a && out && n > 0, RG_EINVAL, "mean_diff: bad args");
  hipLaunchKernelGGL(mean_diff_kernel, dim3(1), dim3(256), 0, rg_stream(stream), a, b, out, n, sign);
  RG_LAUNCH_CHECK("mean_diff");
  return RG_OK;
}
// split form of latent_prep for synchronised statistics: column sums of v = u + z, then (v - mean) / std with the
// all-reduced sums and the global row count (unbiased std, as torch.std)
__global__ void latent_stats_kernel(const float* u, const float* z, float* s, float* ss, int N, int E) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float a = 0.f, b = 0.f;
  for (int n = 0; n < N; ++n) { float v = u[(size_t)n * E + e] + z[(size_t)n * E + e]; a += v; b += v * v; }
  s[e] = a; ss[e] = b;
}
__global__ void latent_apply_kernel(const float* u, const float* z, const float* s, const float* ss, float* out, int N,
                                    int E, float nt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * E) return;
  int e = (int)(i % E);
  float mu = s[e] / nt;
  float var = fmaxf(ss[e] - nt * mu * mu, 0.f) / (nt - 1.f);
  out[i] = (u[i] + z[i] - mu) / sqrtf(var);
}
extern "C" int rg_latent_stats(const float* u, const float* z, float* s, float* ss, int N, int E, void* stream) {
  RG_REQUIRE(u && z && s && ss && N > 0 && E > 0, RG_EINVAL, "latent_stats: bad args");
  hipLaunchKernelGGL(latent_stats_kernel, dim3((E + 63) / 64), dim3(64), 0, rg_stream(stream), u, z, s, ss, N, E);
  RG_LAUNCH_CHECK("latent_stats");
  return RG_OK;
}
extern "C" int rg_latent_apply(const float* u, const float* z, const float* s, const float* ss, float* out, int N, int E,
                               int N_total, void* stream) {
  RG_REQUIRE(u && z && s && ss && out && N > 0 && E > 0 && N_total >= N, RG_EINVAL, "latent_apply: bad args");
  size_t tot = (size_t)N * E;
  hipLaunchKernelGGL(latent_apply_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, rg_stream(stream), u, z, s, ss,
                     out, N, E, (float)N_total);
  RG_LAUNCH_CHECK("latent_apply");
  return RG_OK;
}

extern "C" int rg_latent_prep(const float* u, const float* z, float* out, int N, int E, void* stream) {
  RG_REQUIRE(u && z && out && N > 0 && E > 0, RG_EINVAL, "latent_prep: bad args");
  if (N <= 64)
    hipLaunchKernelGGL(latent_prep_kernel, dim3((E + 63) / 64), dim3(256), 0, rg_stream(stream), u, z, out, N, E);
  else
    hipLaunchKernelGGL(latent_prep_big_kernel, dim3((E + 63) / 64), dim3(64), 0, rg_stream(stream), u, z, out, N, E);
  RG_LAUNCH_CHECK("latent_prep");
  return RG_OK;
}

extern "C" int rg_head_fwd(const void* a, const float* w, float* h, float* out, int N, int C, float slope, int dtype,
                           void* stream) {
  RG_REQUIRE(a && w && h && out && N > 0 && C > 0, RG_EINVAL, "head_fwd: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((head_fwd_kernel<T>), dim3(N), dim3(256), 0, rg_stream(stream), (const T*)a, w, h, out, C, slope);
    RG_LAUNCH_CHECK("head_fwd");
    return RG_OK;
  })
}
extern "C" int rg_head_grad(const float* h, float* gh, int N, float coef, float slope, void* stream) {
  RG_REQUIRE(h && gh && N > 0, RG_EINVAL, "head_grad: bad args");
  hipLaunchKernelGGL(head_grad_kernel, dim3((N + 255) / 256), dim3(256), 0, rg_stream(stream), h, gh, N, coef, slope);
  RG_LAUNCH_CHECK("head_grad");
  return RG_OK;
}
extern "C" int rg_head_bwd_data(const float* gh, const float* w, void* ga, int N, int C, int dtype, void* stream) {
  RG_REQUIRE(gh && w && ga && N > 0 && C > 0, RG_EINVAL, "head_bwd_data: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((head_bwd_data_kernel<T>), dim3(grid_for((size_t)N * 16 * C)), dim3(256), 0, rg_stream(stream),
                       gh, w, (T*)ga, N, C);
    RG_LAUNCH_CHECK("head_bwd_data");
    return RG_OK;
  })
}
extern "C" int rg_head_wgrad(const float* gh, const void* a, float* dw, int N, int C, int dtype, int accumulate,
                             void* stream) {
  RG_REQUIRE(gh && a && dw && N > 0 && C > 0, RG_EINVAL, "head_wgrad: bad args");
  if (dtype == RG_H16 && (((uintptr_t)a) & 15) == 0 && C % 8 == 0) {
    hipLaunchKernelGGL(head_wgrad_bf16_kernel, dim3((16 * C + 255) / 256), dim3(256), 0, rg_stream(stream), gh,
                       (const uint16_t*)a, dw, N, C, accumulate);
    RG_LAUNCH_CHECK("head_wgrad");
    return RG_OK;
  }
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((head_wgrad_kernel<T>), dim3((16 * C + 255) / 256), dim3(256), 0, rg_stream(stream), gh,
                       (const T*)a, dw, N, C, accumulate);
    RG_LAUNCH_CHECK("head_wgrad");
    return RG_OK;
  })
}

// images NCHW fp32 in [-1,1] -> NHWC fp32 in [0,1]: x*0.5 + 0.5 (transforms.Normalize(-mean/std, 1/std) with
// mean = std = 0.5 followed by permute(0,2,3,1), src/gan_utils.py:236-241)
__global__ void export_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int HW) {
  size_t tot = (size_t)N * C * HW;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C);
    size_t t = i / C;
    size_t p = t % HW, n = t / HW;
    y[i] = (x[(n * C + c) * HW + p] - (-1.f)) / 2.f;       // (x - (-mean/std)) / (1/std)
  }
}
extern "C" int rg_export_images_nhwc(const float* x_nchw, float* y_nhwc, int N, int C, int H, int W, void* stream) {
  RG_REQUIRE(x_nchw && y_nhwc && N > 0 && C > 0 && H > 0 && W > 0, RG_EINVAL, "export_images_nhwc: bad args");
  size_t tot = (size_t)N * C * H * W;
  hipLaunchKernelGGL(export_nhwc_kernel, dim3(grid_for(tot)), dim3(256), 0, rg_stream(stream), x_nchw, y_nhwc, N, C, H * W);
  RG_LAUNCH_CHECK("export_images_nhwc");
  return RG_OK;
}

// uint8 tile -> float / 255 -> (x - mean) / std, element order unchanged; the same three fp32 operations in the same
// order as ToTensor + Normalize on the host, so the result is bit-identical to the reference's input transform.
// 16 elements per thread: one 16-byte load, four 16-byte stores.
__global__ void u8_to_norm_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, size_t n, float mean,
                                  float stdv) {
  const size_t n16 = n >> 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
    const uint4 v = reinterpret_cast<const uint4*>(src)[i];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float4* d = reinterpret_cast<float4*>(dst) + i * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 o;
      o.x = ((float)(w[k] & 0xffu) / 255.0f - mean) / stdv;
      o.y = ((float)((w[k] >> 8) & 0xffu) / 255.0f - mean) / stdv;
      o.z = ((float)((w[k] >> 16) & 0xffu) / 255.0f - mean) / stdv;
      o.w = ((float)(w[k] >> 24) / 255.0f - mean) / stdv;
      d[k] = o;
    }
  }
  // tail (n % 16 elements), first threads of block 0
  const size_t tail0 = n16 << 4;
  if (blockIdx.x == 0 && tail0 + threadIdx.x < n) dst[tail0 + threadIdx.x] = ((float)src[tail0 + threadIdx.x] / 255.0f - mean) / stdv;
}
extern "C" int rg_u8_to_norm(const void* src_u8, float* dst, size_t n, float mean, float stdv, void* stream) {
  RG_REQUIRE(src_u8 && dst && stdv != 0.f, RG_EINVAL, "u8_to_norm: bad args");
  RG_REQUIRE(((uintptr_t)src_u8 & 15) == 0 && ((uintptr_t)dst & 15) == 0, RG_EINVAL, "u8_to_norm: 16-byte aligned buffers required");
  if (n == 0) return RG_OK;
  hipLaunchKernelGGL(u8_to_norm_kernel, dim3(grid_for(n, 16)), dim3(256), 0, rg_stream(stream), (const uint8_t*)src_u8,
                     dst, n, mean, stdv);
  RG_LAUNCH_CHECK("u8_to_norm");
  return RG_OK;
}

// fp32 -> OCP fp8 e4m3 (v_cvt_pk_fp8_f32), dst[i] = fp8(src[i] * mul): weights / latents of the fp8 inference path
__global__ void cast_fp8_kernel(const float* __restrict__ src, uint8_t* __restrict__ dst, size_t n, float mul) {
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * mul, v.y * mul, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * mul, v.w * mul, w, true);
    reinterpret_cast<int*>(dst)[i] = w;
  }
  const size_t t0 = n4 << 2;
  if (blockIdx.x == 0 && t0 + threadIdx.x < n) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(src[t0 + threadIdx.x] * mul, 0.f, 0, false);
    dst[t0 + threadIdx.x] = (uint8_t)(w & 0xff);
  }
}
extern "C" int rg_cast_fp8(const float* src, void* dst, size_t n, float mul, void* stream) {
  RG_REQUIRE(src && dst && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 3) == 0, RG_EINVAL, "cast_fp8: bad args");
  if (n == 0) return RG_OK;
  hipLaunchKernelGGL(cast_fp8_kernel, dim3(grid_for(n, 4)), dim3(256), 0, rg_stream(stream), src, (uint8_t*)dst, n, mul);
  RG_LAUNCH_CHECK("cast_fp8");
  return RG_OK;
}
// hardware check of the fp8 MFMA operand map this library assumes (exact small integers): lane l of
// v_mfma_f32_16x16x32_fp8_fp8 holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15] in byte j
__global__ void selftest_fp8_kernel(int* out) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  auto enc = [](int v) -> unsigned {          // small integers -4..4 as e4m3 bytes
    const int a = v < 0 ? -v : v;
    const unsigned m = a == 0 ? 0x00u : a == 1 ? 0x38u : a == 2 ? 0x40u : a == 3 ? 0x44u : 0x48u;
    return m | (v < 0 ? 0x80u : 0u);
  };
  unsigned long long fa = 0, fb = 0;
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * q + j;
    fa |= (unsigned long long)enc((r * 3 + k * 5) % 7 - 3) << (8 * j);            // A[r][k]
    fb |= (unsigned long long)enc((r * 2 + k * 7 + 1) % 5 - 2) << (8 * j);        // B[k][col r]  (asymmetric in (col, k))
  }
  typedef __attribute__((ext_vector_type(4))) float f4;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8((long)fa, (long)fb, acc, 0, 0, 0);
  int bad = 0;
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * q + i, col = r;
    float ref = 0.f;
    for (int k = 0; k < 32; ++k) ref += (float)((row * 3 + k * 5) % 7 - 3) * (float)((col * 2 + k * 7 + 1) % 5 - 2);
    if (ref != acc[i]) ++bad;
  }
  // round trip of the converter: small integers and a few representable values come back exactly
  const float probes[8] = {0.f, 1.f, -2.f, 3.f, 0.5f, -0.875f, 448.f, 0.015625f};
  const unsigned want[8] = {0x00u, 0x38u, 0xC0u, 0x44u, 0x30u, 0xB6u, 0x7Eu, 0x08u};
  if (lane < 8) {
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(probes[lane], 0.f, 0, false);
    if ((unsigned)(w & 0xff) != want[lane]) atomicAdd(&out[1], 1);
  }
  atomicAdd(&out[0], bad);
}
extern "C" int rg_selftest_fp8(int* detail, void* stream) {
  RG_REQUIRE(detail, RG_EINVAL, "selftest_fp8: bad args");
  hipLaunchKernelGGL(selftest_fp8_kernel, dim3(1), dim3(64), 0, rg_stream(stream), detail);
  RG_LAUNCH_CHECK("selftest_fp8");
  return RG_OK;
}

extern "C" int rg_widen_bf16(const void* src, float* dst, size_t n, void* stream) {
  RG_REQUIRE(src && dst, RG_EINVAL, "widen_bf16: bad args");
  if (n == 0) return RG_OK;
  hipLaunchKernelGGL(widen_bf16_kernel, dim3(grid_for(n, 4)), dim3(256), 0, rg_stream(stream), (const uint16_t*)src,
                     dst, n);
  RG_LAUNCH_CHECK("widen_bf16");
  return RG_OK;
}

extern "C" int rg_cast_pad(const float* src, void* dst, int M, int K, int ldd, int dtype, void* stream) {
  RG_REQUIRE(src && dst && M > 0 && K > 0 && ldd >= K, RG_EINVAL, "cast_pad: bad args");
  const size_t tot = (size_t)M * ldd;
  if (dtype == RG_H16 && K == ldd && tot % 8 == 0 && tot >= 4096 && aligned16(src) && aligned16(dst)) {
    const size_t n8 = tot / 8, want = (n8 + 255) / 256;
    hipLaunchKernelGGL(cast_bf16x8_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, rg_stream(stream), src,
                       (uint16_t*)dst, n8);
    RG_LAUNCH_CHECK("cast_pad");
    return RG_OK;
  }
  RG_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((cast_pad_kernel<T>), dim3(grid_for((size_t)M * ldd)), dim3(256), 0, rg_stream(stream), src,
                       (T*)dst, M, K, ldd);
    RG_LAUNCH_CHECK("cast_pad");
    return RG_OK;
  })
}

