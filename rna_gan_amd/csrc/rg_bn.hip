// rg_bn.hip -- train-mode BatchNorm2d + LeakyReLU on NHWC [M][C] tensors: forward, backward,
// forward-mode tangent and the joint (double) backward of the gradient penalty; pointwise helpers.
// All HBM-bound: one coalesced pass per tensor, 8/16-byte vector accesses, per-channel sums through
// a deterministic two-stage column reduction (partials in the caller's workspace).
#include "rg_common.h"

namespace {

constexpr int CR_TX = 32;  // channel-vector lanes per block
constexpr int CR_TY = 8;   // row lanes per block

struct CRPlan { int gx, gy, rows_per_block; };

static CRPlan cr_plan(int M, int C, int vec) {
  CRPlan p;
  int cvec = (C + vec - 1) / vec;
  p.gx = (cvec + CR_TX - 1) / CR_TX;
  // enough row-chunks to fill 256 CUs x 4-8 blocks, but few enough that the finishing pass stays tiny
  int want = (1536 + p.gx - 1) / p.gx;
  int maxg = (M + 4 * CR_TY - 1) / (4 * CR_TY);
  p.gy = want < maxg ? want : maxg;
  if (p.gy < 1) p.gy = 1;
  if (p.gy > 512) p.gy = 512;
  p.rows_per_block = (M + p.gy - 1) / p.gy;
  p.gy = (M + p.rows_per_block - 1) / p.rows_per_block;
  return p;
}
static inline int vec_of(int C) { return (C % 4 == 0) ? 4 : 1; }

template <int NQ, int VEC, class F>
__global__ __launch_bounds__(256) void colreduce_kernel(F f, int M, int C, int rows_per_block, float* partial) {
  __shared__ float sm[CR_TY][NQ][CR_TX * VEC];
  const int tx = threadIdx.x % CR_TX, ty = threadIdx.x / CR_TX;
  const int c = (blockIdx.x * CR_TX + tx) * VEC;
  float acc[NQ][VEC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  if (c < C)
    for (int r = r0 + ty; r < r1; r += CR_TY) f(r, c, acc);
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) sm[ty][q][tx * VEC + v] = acc[q][v];
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < CR_TY; ++t) s += sm[t][q][tx * VEC + v];
        partial[((size_t)blockIdx.y * NQ + q) * C + c + v] = s;
      }
  }
}

// finishing pass: block = 32 channels x 8 partial-row lanes; each lane strides over the G partial rows
// (independent loads, pipelined), LDS tree over the 8 lanes, lane 0 applies the finisher.
template <int NQ, class Fin>
__global__ __launch_bounds__(256) void colfinish_kernel(Fin fin, const float* partial, int C, int G) {
  __shared__ float sm[8][NQ][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + tx;
  float s[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) s[q] = 0.f;
  if (c < C) {
#pragma unroll 4
    for (int g = ty; g < G; g += 8)
#pragma unroll
      for (int q = 0; q < NQ; ++q) s[q] += partial[((size_t)g * NQ + q) * C + c];
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) sm[ty][q][tx] = s[q];
  __syncthreads();
  if (ty == 0 && c < C) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += sm[k][q][tx];
      s[q] = t;
    }
    fin(c, s);
  }
}

template <int NQ, template <int> class F, class Fin, class... Args>
int col_reduce(const char* name, int M, int C, void* ws, size_t ws_bytes, hipStream_t st, Fin fin, Args... args) {
  int vec = vec_of(C);
  CRPlan p = cr_plan(M, C, vec);
  size_t need = (size_t)p.gy * NQ * C * sizeof(float);
  RG_REQUIRE(ws && ws_bytes >= need, RG_EWORKSPACE, "%s: workspace %zu < %zu", name, ws_bytes, need);
  float* partial = (float*)ws;
  if (vec == 4) {
    F<4> f{args...};
    hipLaunchKernelGGL((colreduce_kernel<NQ, 4, F<4>>), dim3(p.gx, p.gy), dim3(256), 0, st, f, M, C, p.rows_per_block,
                       partial);
  } else {
    F<1> f{args...};
    hipLaunchKernelGGL((colreduce_kernel<NQ, 1, F<1>>), dim3(p.gx, p.gy), dim3(256), 0, st, f, M, C, p.rows_per_block,
                       partial);
  }
  RG_LAUNCH_CHECK(name);
  hipLaunchKernelGGL((colfinish_kernel<NQ, Fin>), dim3((C + 31) / 32), dim3(256), 0, st, fin, partial, C, p.gy);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

template <int VEC, class F>
__global__ __launch_bounds__(256) void pointwise_kernel(F f, size_t nvec, int cvec) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    size_t row = i / cvec;
    int c = (int)(i - row * cvec) * VEC;
    f(row, c);
  }
}

template <template <int> class F, class... Args>
int pointwise(const char* name, int M, int C, hipStream_t st, Args... args) {
  int vec = vec_of(C);
  int cvec = C / vec;
  size_t nvec = (size_t)M * cvec;
  if (nvec == 0) return RG_OK;
  size_t blocks = (nvec + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec == 4) {
    F<4> f{args...};
    hipLaunchKernelGGL((pointwise_kernel<4, F<4>>), dim3((unsigned)blocks), dim3(256), 0, st, f, nvec, cvec);
  } else {
    F<1> f{args...};
    hipLaunchKernelGGL((pointwise_kernel<1, F<1>>), dim3((unsigned)blocks), dim3(256), 0, st, f, nvec, cvec);
  }
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

// per-channel parameter bundle
struct BNC {
  const float* mean; const float* invstd; const float* gamma; const float* beta; float slope;
};

// ------------------------------------------------------------------------------------------ stats
template <typename T> struct StatsF {
  template <int VEC> struct K {
    const T* z; int C;
    __device__ void operator()(int r, int c, float (*acc)[VEC]) const {
      float v[VEC];
      Vec<T, VEC>::ld(z + (size_t)r * C + c, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) { acc[0][i] += v[i]; acc[1][i] += v[i] * v[i]; }
    }
  };
};
struct Store2Fin {
  float* a; float* b;
  __device__ void operator()(int c, const float* s) const { a[c] = s[0]; b[c] = s[1]; }
};

// ------------------------------------------------------------------------------------------ bn_act
template <typename T> struct BnActF {
  template <int VEC> struct K {
    const T* z; T* a; BNC p; int C;
    __device__ void operator()(size_t r, int c) const {
      float v[VEC], o[VEC];
      Vec<T, VEC>::ld(z + r * C + c, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float y = (v[i] - p.mean[c + i]) * (p.invstd[c + i] * p.gamma[c + i]) + p.beta[c + i];
        o[i] = lrelu_f(y, p.slope);
      }
      Vec<T, VEC>::st(a + r * C + c, o);
    }
  };
};

// ------------------------------------------------------------------------------------------ bwd
template <typename T> struct BwdRedF {
  template <int VEC> struct K {
    const T* z; const T* ga; BNC p; int C;
    __device__ void operator()(int r, int c, float (*acc)[VEC]) const {
      float v[VEC], g[VEC];
      Vec<T, VEC>::ld(z + (size_t)r * C + c, v);
      Vec<T, VEC>::ld(ga + (size_t)r * C + c, g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float xh = (v[i] - p.mean[c + i]) * p.invstd[c + i];
        float y = xh * p.gamma[c + i] + p.beta[c + i];
        float gy = g[i] * lrelu_mask(y, p.slope);
        acc[0][i] += gy; acc[1][i] += gy * xh;
      }
    }
  };
};
struct BwdFin {
  float* s_gy; float* s_gyxh; float* dgamma; float* dbeta; int accumulate;
  __device__ void operator()(int c, const float* s) const {
    s_gy[c] = s[0]; s_gyxh[c] = s[1];
    if (dgamma) {
      if (accumulate) { dgamma[c] += s[1]; dbeta[c] += s[0]; }
      else { dgamma[c] = s[1]; dbeta[c] = s[0]; }
    }
  }
};
template <typename T> struct BwdApplyF {
  template <int VEC> struct K {
    const T* z; const T* ga; T* gz; BNC p; const float* s_gy; const float* s_gyxh; float inv_m; int C;
    __device__ void operator()(size_t r, int c) const {
      float v[VEC], g[VEC], o[VEC];
      Vec<T, VEC>::ld(z + r * C + c, v);
      Vec<T, VEC>::ld(ga + r * C + c, g);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float is = p.invstd[c + i], gm = p.gamma[c + i];
        float xh = (v[i] - p.mean[c + i]) * is;
        float y = xh * gm + p.beta[c + i];
        float gy = g[i] * lrelu_mask(y, p.slope);
        o[i] = (gm * is) * (gy - s_gy[c + i] * inv_m - xh * (s_gyxh[c + i] * inv_m));
      }
      Vec<T, VEC>::st(gz + r * C + c, o);
    }
  };
};

// ------------------------------------------------------------------------------------------ tangent
template <typename T> struct TanRedF {
  template <int VEC> struct K {
    const T* z; const T* zt; BNC p; int C;
    __device__ void operator()(int r, int c, float (*acc)[VEC]) const {
      float v[VEC], t[VEC];
      Vec<T, VEC>::ld(z + (size_t)r * C + c, v);
      Vec<T, VEC>::ld(zt + (size_t)r * C + c, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float xh = (v[i] - p.mean[c + i]) * p.invstd[c + i];
        acc[0][i] += t[i]; acc[1][i] += xh * t[i];
      }
    }
  };
};
template <typename T> struct TanApplyF {
  template <int VEC> struct K {
    const T* z; const T* zt; T* at; BNC p; const float* s_zt; const float* s_xhzt; float inv_m; int C;
    __device__ void operator()(size_t r, int c) const {
      float v[VEC], t[VEC], o[VEC];
      Vec<T, VEC>::ld(z + r * C + c, v);
      Vec<T, VEC>::ld(zt + r * C + c, t);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float is = p.invstd[c + i], gm = p.gamma[c + i];
        float xh = (v[i] - p.mean[c + i]) * is;
        float y = xh * gm + p.beta[c + i];
        float yt = (gm * is) * (t[i] - s_zt[c + i] * inv_m - xh * (s_xhzt[c + i] * inv_m));
        o[i] = yt * lrelu_mask(y, p.slope);
      }
      Vec<T, VEC>::st(at + r * C + c, o);
    }
  };
};

// ------------------------------------------------------------------------------------------ double bwd
template <typename T> struct DblRedF {
  template <int VEC> struct K {
    const T* z; const T* qa; const T* zt; const T* ga1; BNC p; int C;
    __device__ void operator()(int r, int c, float (*acc)[VEC]) const {
      float v[VEC], t[VEC], g[VEC], q[VEC];
      Vec<T, VEC>::ld(z + (size_t)r * C + c, v);
      Vec<T, VEC>::ld(zt + (size_t)r * C + c, t);
      Vec<T, VEC>::ld(ga1 + (size_t)r * C + c, g);
      if (qa) Vec<T, VEC>::ld(qa + (size_t)r * C + c, q);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float xh = (v[i] - p.mean[c + i]) * p.invstd[c + i];
        float y = xh * p.gamma[c + i] + p.beta[c + i];
        float mk = lrelu_mask(y, p.slope);
        acc[0][i] += g[i] * mk * t[i];
        if (qa) { float qy = q[i] * mk; acc[1][i] += qy; acc[2][i] += qy * xh; }
      }
    }
  };
};
// per-channel coefficients for the apply pass: coef[0]=A-3bc, [1]=c, [2]=b, [3]=s_qy/m, [4]=s_qyxh/m
struct DblFin {
  const float* s_gy; const float* s_gyxh; const float* s_zt; const float* s_xhzt; const float* invstd;
  float* coef; float* dgamma; float* dbeta; int accumulate; float m; int C;
  __device__ void operator()(int c, const float* s) const {
    float inv_m = 1.f / m;
    float b = s_gyxh[c] * inv_m, cc = s_xhzt[c] * inv_m;
    float A = s[0] * inv_m - (s_gy[c] * inv_m) * (s_zt[c] * inv_m);
    coef[0 * C + c] = A - 3.f * b * cc;
    coef[1 * C + c] = cc;
    coef[2 * C + c] = b;
    coef[3 * C + c] = s[1] * inv_m;
    coef[4 * C + c] = s[2] * inv_m;
    float dg = m * invstd[c] * (A - b * cc) + s[2];
    float db = s[1];
    if (accumulate) { dgamma[c] += dg; dbeta[c] += db; }
    else { dgamma[c] = dg; dbeta[c] = db; }
  }
};
template <typename T> struct DblApplyF {
  template <int VEC> struct K {
    const T* z; const T* qa; const T* zt; const T* ga1; T* pz; BNC p; const float* s_gy; const float* s_zt;
    const float* coef; float inv_m; int C;
    __device__ void operator()(size_t r, int c) const {
      float v[VEC], t[VEC], g[VEC], q[VEC], o[VEC];
      Vec<T, VEC>::ld(z + r * C + c, v);
      Vec<T, VEC>::ld(zt + r * C + c, t);
      Vec<T, VEC>::ld(ga1 + r * C + c, g);
      if (qa) Vec<T, VEC>::ld(qa + r * C + c, q);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        int ch = c + i;
        float is = p.invstd[ch], gm = p.gamma[ch];
        float xh = (v[i] - p.mean[ch]) * is;
        float y = xh * gm + p.beta[ch];
        float mk = lrelu_mask(y, p.slope);
        float gy = g[i] * mk;
        float r0 = xh * coef[0 * C + ch] + coef[1 * C + ch] * (gy - s_gy[ch] * inv_m) +
                   coef[2 * C + ch] * (t[i] - s_zt[ch] * inv_m);
        float out = -(gm * is * is) * r0;
        if (qa) {
          float qy = q[i] * mk;
          out += (gm * is) * (qy - coef[3 * C + ch] - xh * coef[4 * C + ch]);
        }
        o[i] = out;
      }
      Vec<T, VEC>::st(pz + r * C + c, o);
    }
  };
};

// ------------------------------------------------------------------------------------------ misc
template <typename T> struct ColSumF {
  template <int VEC> struct K {
    const T* g; int C;
    __device__ void operator()(int r, int c, float (*acc)[VEC]) const {
      float v[VEC];
      Vec<T, VEC>::ld(g + (size_t)r * C + c, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[0][i] += v[i];
    }
  };
};
struct AccFin {
  float* out; int accumulate;
  __device__ void operator()(int c, const float* s) const { out[c] = accumulate ? out[c] + s[0] : s[0]; }
};

template <typename T> struct LreluBwdF {
  template <int VEC> struct K {
    const T* g; const T* a; T* out; float slope;
    __device__ void operator()(size_t r, int c) const {   // called with C = VEC*cvec = row length
      (void)c;
    }
  };
};

template <typename T, int VEC>
__global__ void lrelu_bwd_kernel(const T* g, const T* a, T* out, size_t nvec, float slope) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    float gv[VEC], av[VEC], o[VEC];
    Vec<T, VEC>::ld(g + i * VEC, gv);
    Vec<T, VEC>::ld(a + i * VEC, av);
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = gv[k] * lrelu_mask(av[k], slope);
    Vec<T, VEC>::st(out + i * VEC, o);
  }
}

__global__ void bn_finalize_kernel(const float* sum, const float* sumsq, int M, int C, float eps, float momentum,
                                   float* mean, float* invstd, float* rmean, float* rvar, int64_t* nbt) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  float m = (float)M;
  float mu = sum[c] / m;
  float var = fmaxf(sumsq[c] / m - mu * mu, 0.f);
  mean[c] = mu;
  invstd[c] = rsqrtf(var + eps);
  if (rmean) {
    float unb = var * (m / fmaxf(m - 1.f, 1.f));
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

}  // namespace

// helper aliases: template-template adaptors (F<VEC>) for col_reduce / pointwise
#define RG_TT(NAME, OUTER) \
  template <int V> using NAME = typename OUTER::template K<V>;

extern "C" size_t rg_colreduce_workspace_bytes(int M, int C, int nq) {
  if (M <= 0 || C <= 0) return 0;
  CRPlan p = cr_plan(M, C, vec_of(C));
  // partials + 5 per-channel coefficient rows used by rg_bn_double_bwd
  return rg_align_up((size_t)p.gy * (nq < 1 ? 1 : nq) * C * sizeof(float), 256) + (size_t)5 * C * sizeof(float);
}

namespace {
template <typename T> struct Impl {
  RG_TT(StatsK, StatsF<T>)
  RG_TT(BnActK, BnActF<T>)
  RG_TT(BwdRedK, BwdRedF<T>)
  RG_TT(BwdApplyK, BwdApplyF<T>)
  RG_TT(TanRedK, TanRedF<T>)
  RG_TT(TanApplyK, TanApplyF<T>)
  RG_TT(DblRedK, DblRedF<T>)
  RG_TT(DblApplyK, DblApplyF<T>)
  RG_TT(ColSumK, ColSumF<T>)
};
}  // namespace

extern "C" int rg_bn_stats(const void* z, float* sum, float* sumsq, int M, int C, int dtype, void* ws, size_t ws_bytes,
                           void* stream) {
  RG_REQUIRE(z && sum && sumsq && M > 0 && C > 0, RG_EINVAL, "bn_stats: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    return (col_reduce<2, Impl<T>::template StatsK>("bn_stats", M, C, ws, ws_bytes, rg_stream(stream),
                                                     Store2Fin{sum, sumsq}, (const T*)z, C));
  })
}

extern "C" int rg_bn_finalize(const float* sum, const float* sumsq, int M, int C, float eps, float momentum,
                              float* mean, float* invstd, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, void* stream) {
  RG_REQUIRE(sum && sumsq && mean && invstd && M > 0 && C > 0, RG_EINVAL, "bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, rg_stream(stream), sum, sumsq, M, C, eps,
                     momentum, mean, invstd, running_mean, running_var, running_mean ? num_batches_tracked : nullptr);
  RG_LAUNCH_CHECK("bn_finalize");
  return RG_OK;
}

extern "C" int rg_bn_act(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                         void* a, int M, int C, float slope, int dtype, void* stream) {
  RG_REQUIRE(z && a && mean && invstd && gamma && beta && M > 0 && C > 0, RG_EINVAL, "bn_act: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (pointwise<Impl<T>::template BnActK>("bn_act", M, C, rg_stream(stream), (const T*)z, (T*)a, p, C));
  })
}

extern "C" int rg_bn_act_bwd(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, void* gz, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta,
                             int accumulate, int M, int C, float slope, int dtype, void* ws, size_t ws_bytes,
                             void* stream) {
  RG_REQUIRE(z && ga && gz && s_gy && s_gyxh && M > 0 && C > 0, RG_EINVAL, "bn_act_bwd: bad args");
  RG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), RG_EINVAL, "bn_act_bwd: dgamma/dbeta must come together");
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc = col_reduce<2, Impl<T>::template BwdRedK>("bn_act_bwd", M, C, ws, ws_bytes, st,
                                                       BwdFin{s_gy, s_gyxh, dgamma, dbeta, accumulate}, (const T*)z,
                                                       (const T*)ga, p, C);
    if (rc) return rc;
    return (pointwise<Impl<T>::template BwdApplyK>("bn_act_bwd", M, C, st, (const T*)z, (const T*)ga, (T*)gz, p,
                                                    (const float*)s_gy, (const float*)s_gyxh, 1.f / (float)M, C));
  })
}

extern "C" int rg_bn_tangent(const void* z, const void* zt, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, void* at, float* s_zt, float* s_xhzt, int M, int C, float slope,
                             int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && at && s_zt && s_xhzt && M > 0 && C > 0, RG_EINVAL, "bn_tangent: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc = col_reduce<2, Impl<T>::template TanRedK>("bn_tangent", M, C, ws, ws_bytes, st, Store2Fin{s_zt, s_xhzt},
                                                       (const T*)z, (const T*)zt, p, C);
    if (rc) return rc;
    return (pointwise<Impl<T>::template TanApplyK>("bn_tangent", M, C, st, (const T*)z, (const T*)zt, (T*)at, p,
                                                    (const float*)s_zt, (const float*)s_xhzt, 1.f / (float)M, C));
  })
}

extern "C" int rg_bn_double_bwd(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, const float* s_gy,
                                const float* s_gyxh, const float* s_zt, const float* s_xhzt, void* pz, float* dgamma,
                                float* dbeta, int accumulate, int M, int C, float slope, int dtype, void* ws,
                                size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && ga1 && pz && dgamma && dbeta && s_gy && s_gyxh && s_zt && s_xhzt && M > 0 && C > 0, RG_EINVAL,
             "bn_double_bwd: bad args");
  size_t need = rg_colreduce_workspace_bytes(M, C, 3);
  RG_REQUIRE(ws && ws_bytes >= need, RG_EWORKSPACE, "bn_double_bwd: workspace %zu < %zu", ws_bytes, need);
  float* coef = (float*)((char*)ws + need - (size_t)5 * C * sizeof(float));
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc = col_reduce<3, Impl<T>::template DblRedK>(
        "bn_double_bwd", M, C, ws, ws_bytes - (size_t)5 * C * sizeof(float), st,
        DblFin{s_gy, s_gyxh, s_zt, s_xhzt, invstd, coef, dgamma, dbeta, accumulate, (float)M, C}, (const T*)z,
        (const T*)qa, (const T*)zt, (const T*)ga1, p, C);
    if (rc) return rc;
    return (pointwise<Impl<T>::template DblApplyK>("bn_double_bwd", M, C, st, (const T*)z, (const T*)qa, (const T*)zt,
                                                    (const T*)ga1, (T*)pz, p, s_gy, s_zt, (const float*)coef,
                                                    1.f / (float)M, C));
  })
}

extern "C" int rg_col_sum(const void* g, float* out, int M, int C, int dtype, int accumulate, void* ws, size_t ws_bytes,
                          void* stream) {
  RG_REQUIRE(g && out && M > 0 && C > 0, RG_EINVAL, "col_sum: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    return (col_reduce<1, Impl<T>::template ColSumK>("col_sum", M, C, ws, ws_bytes, rg_stream(stream),
                                                      AccFin{out, accumulate}, (const T*)g, C));
  })
}

extern "C" int rg_lrelu_bwd(const void* g, const void* a, void* out, size_t n, float slope, int dtype, void* stream) {
  RG_REQUIRE(g && a && out, RG_EINVAL, "lrelu_bwd: bad args");
  if (n == 0) return RG_OK;
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    if (n % 4 == 0) {
      size_t nvec = n / 4, blocks = (nvec + 255) / 256;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((lrelu_bwd_kernel<T, 4>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)g, (const T*)a,
                         (T*)out, nvec, slope);
    } else {
      size_t blocks = (n + 255) / 256;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((lrelu_bwd_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)g, (const T*)a,
                         (T*)out, n, slope);
    }
    RG_LAUNCH_CHECK("lrelu_bwd");
    return RG_OK;
  })
}
