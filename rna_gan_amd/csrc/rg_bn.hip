// rg_bn.hip -- train-mode BatchNorm2d + LeakyReLU on NHWC [M][C] tensors: forward, backward,
// forward-mode tangent and the joint (double) backward of the gradient penalty; pointwise helpers.
// All HBM-bound streaming kernels built on ONE skeleton ("row loop"): a thread owns a fixed vector of
// VEC consecutive channels (16-byte accesses for bf16, VEC = 8), loads the per-channel parameters into
// registers ONCE, then walks rows with a stride of TY; a wave covers whole 128..512-byte row segments.
// Per-channel sums use a deterministic two-stage reduction (block partials in the caller's workspace,
// then a small finishing kernel).
#include "rg_common.h"
#include "rg_internal.h"
#include <stdlib.h>
#include <string.h>

namespace {

struct Plan { int vec, tx, gx, gy, rows_per_block; };

template <typename T>
static Plan make_plan(int M, int C, int target_blocks, int gy_cap = 512) {
  Plan p;
  const int vmax = sizeof(T) == 2 ? 8 : 4;
  p.vec = (C % vmax == 0) ? vmax : (C % 4 == 0 ? 4 : 1);
  int cvec = (C + p.vec - 1) / p.vec;
  p.tx = cvec >= 32 ? 32 : (cvec >= 16 ? 16 : 8);
  p.gx = (cvec + p.tx - 1) / p.tx;
  int ty = 256 / p.tx;
  int want = (target_blocks + p.gx - 1) / p.gx;
  int maxg = (M + 4 * ty - 1) / (4 * ty);
  p.gy = want < maxg ? want : maxg;
  if (p.gy < 1) p.gy = 1;
  if (p.gy > gy_cap) p.gy = gy_cap;
  p.rows_per_block = (M + p.gy - 1) / p.gy;
  p.gy = (M + p.rows_per_block - 1) / p.rows_per_block;
  return p;
}

// upper bound of gy over both element types (workspace sizing)
static int max_gy(int M, int C) {
  Plan a = make_plan<float>(M, C, 1536), b = make_plan<h16_t>(M, C, 1536);
  return a.gy > b.gy ? a.gy : b.gy;
}

// ---- batch GROUPS (gridDim.z): the same pass over several equally sized row blocks [g*M, (g+1)*M) of one tensor, each with
// its own statistics -- the D step runs D(real) and D(fake) as one double batch and BatchNorm has to see the two halves as
// two forward calls.  A functor / finisher that supports groups has shift(g); others ignore the group index (gridDim.z = 1).
template <class F> __device__ __forceinline__ auto shift_group(F& f, int g, int) -> decltype(f.shift(g), void()) { f.shift(g); }
template <class F> __device__ __forceinline__ void shift_group(F&, int, long) {}

// ---- the order in which a pass walks its row blocks (blockIdx.y is dispatched in ascending order): rev = from the END of the
// tensor.  What a pass reads first should be what the kernel in front of it touched last -- the memory-side cache holds a fraction
// of a 67-134 MB activation tensor, not all of it.  Measured on the benchmarked iteration (option bn_rev): the
// REDUCTIONS from the end (their operand was just written, its tail last) and the apply passes behind them from the front (the
// reduction read the head last): -0.07 ms; every other combination, an eight-range interleave matching the conv launches' XCD
// order and the conv launches themselves from the end: 0 ... +0.16 ms.
__device__ __forceinline__ int row_block_order(int rev) {
  return rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y;
}

// ---- reduction skeleton: F has  init(c)  and  row(r, c, acc[NQ][VEC])
template <int NQ, int VEC, int TX, class F>
__global__ __launch_bounds__(256) void rowreduce_kernel(F f, int M, int C, int rows_per_block, float* partial, int rev) {
  constexpr int TY = 256 / TX;
  shift_group(f, (int)blockIdx.z, 0);
  partial += (size_t)blockIdx.z * gridDim.y * NQ * C;
  __shared__ float sm[TY][NQ][TX * VEC];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c = (blockIdx.x * TX + tx) * VEC;
  float acc[NQ][VEC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
  // rev (option bn_rev bit 2): the row blocks are walked from the end of the tensor (the producer wrote its last rows last);
  // partial row `by` holds the same rows' sums either way, so the finisher's result does not change by a bit
  const int by = row_block_order(rev);
  const int r0 = by * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  if (c < C) {
    f.init(c);
    constexpr int U = F::U;                                       // rows in flight per thread (see rowapply_kernel)
    int r = r0 + ty;
    for (; r + (U - 1) * TY < r1; r += U * TY) {
      typename F::In in[U];
#pragma unroll
      for (int k = 0; k < U; ++k) f.load(r + k * TY, c, in[k]);
#pragma unroll
      for (int k = 0; k < U; ++k) f.accum(in[k], acc);
    }
    for (; r < r1; r += TY) f.row(r, c, acc);
  }
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
        for (int t = 0; t < TY; ++t) s += sm[t][q][tx * VEC + v];
        partial[((size_t)by * NQ + q) * C + c + v] = s;
      }
  }
}

// finishing pass: block = 32 channels x 8 partial-row lanes
// Partial-row layout with groups: [nblk][groups][G / nblk] rows (nblk = 1: group-major; nblk = 4: the transposed conv's
// statistics, whose rows are ordered by output-parity class first and row tile second, so a batch half is the first / second
// half of EVERY class block).
__device__ __forceinline__ size_t group_row(int r, int grp, int groups, int Gb) { return (size_t)((r / Gb) * groups + grp) * Gb + r % Gb; }

template <int NQ, class Fin>
__global__ __launch_bounds__(256) void colfinish_kernel(Fin fin, const float* partial, int C, int G, int groups = 1,
                                                        int nblk = 1) {
  // block = CF_CH channels x CF_RL partial-row lanes: C / 8 blocks (the pass is latency-bound -- a few hundred KB read by a
  // grid that used to be C / 32 blocks with 24 dependent loads per thread; 8-10 us per launch, 48 launches per iteration)
  constexpr int CF_CH = 8, CF_RL = 32;
  __shared__ float sm[CF_RL][NQ][CF_CH];
  const int tx = threadIdx.x % CF_CH, ty = threadIdx.x / CF_CH;
  const int c = blockIdx.x * CF_CH + tx;
  // groups are finished one after the other by the same thread: a finisher that updates shared state (running statistics,
  // accumulated parameter gradients) sees them in order, exactly as consecutive calls would
  for (int grp = 0; grp < groups; ++grp) {
    const int Gb = G / nblk;
    float s[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q] = 0.f;
    if (c < C) {
#pragma unroll 4
      for (int g = ty; g < G; g += CF_RL)
#pragma unroll
        for (int q = 0; q < NQ; ++q) s[q] += partial[(group_row(g, grp, groups, Gb) * NQ + q) * C + c];
    }
    if (grp) __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) sm[ty][q][tx] = s[q];
    __syncthreads();
    if (ty == 0 && c < C) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < CF_RL; ++k) t += sm[k][q][tx];
        s[q] = t;
      }
      Fin fg = fin;
      shift_group(fg, grp, 0);
      fg(c, s);
    }
  }
}

// finishing pass for MANY partial rows (conv-epilogue statistics: one row per (row tile, wave row)): block = 8 channels,
// 256 threads stride over the partial rows with 32-byte loads, then wave shuffles + LDS.  Fixed order -> deterministic.
// gridDim.y slices the partial rows (each slice finished by its own block: fin gets c + slice*C as the channel index,
// i.e. a Store2Fin pointed at a [slices][C] staging area produces the input of a second, small colfinish pass).
template <int NQ, class Fin>
__global__ __launch_bounds__(256) void colfinish_wide_kernel(Fin fin, const float* __restrict__ partial, int C, int G,
                                                             int nblk = 1) {
  __shared__ float sm[4][NQ][8];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c0 = blockIdx.x * 8;
  const int per = (G + gridDim.y - 1) / gridDim.y;
  const int g0 = blockIdx.y * per, g1 = min(G, g0 + per);
  const int grp = (int)blockIdx.z, groups = (int)gridDim.z, Gb = G / nblk;       // batch group (see shift_group)
  shift_group(fin, grp, 0);
  float acc[NQ][8];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < 8; ++v) acc[q][v] = 0.f;
  for (int g = g0 + t; g < g1; g += 256)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      float v[8];
      Vec<float, 8>::ld(partial + (group_row(g, grp, groups, Gb) * NQ + q) * C + c0, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[q][k] += v[k];
    }
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < 8; ++v) {
      float w = wave_sum(acc[q][v]);
      if (lane == 0) sm[wave][q][v] = w;
    }
  __syncthreads();
  if (t < 8) {
    float sv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sv[q] = sm[0][q][t] + sm[1][q][t] + sm[2][q][t] + sm[3][q][t];
    fin(c0 + t + (int)blockIdx.y * C, sv);
  }
}
// level-1 finisher of the sliced form: row `slice` of a [slices][2][C] staging area (colfinish_kernel's input layout)
struct SliceFin {
  float* out; int C; int slices = 0;
  __device__ void shift(int g) { out += (size_t)g * slices * 2 * C; }
  __device__ void operator()(int cs, const float* s) const {
    const int slice = cs / C, c = cs - slice * C;
    out[((size_t)slice * 2 + 0) * C + c] = s[0];
    out[((size_t)slice * 2 + 1) * C + c] = s[1];
  }
};

// ---- pointwise skeleton: F has  init(c),  load(r, c, In&)  and  finish(r, c, const In&)  (row = load + finish).
// F::U rows are in flight per thread: all their loads are issued before the first store, so that a thread keeps
// U x (inputs) x 16 bytes outstanding -- with one row at a time the kernels sat at ~3.7 TB/s, latency-bound (the
// compiler cannot hoist the next row's loads over the previous row's store: the pointers may alias).
template <int VEC, int TX, class F>
__global__ __launch_bounds__(256) void rowapply_kernel(F f, int M, int C, int rows_per_block, int rev) {
  constexpr int TY = 256 / TX, U = F::U;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int c = (blockIdx.x * TX + tx) * VEC;
  if (c >= C) return;
  // rev: the row blocks are walked from the END of the tensor -- the pass in front of this one (a reduction over the same
  // operands, or the kernel that wrote them) touched the last rows last, so they are what the memory-side cache still holds
  const int by = row_block_order(rev);
  const int r0 = by * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  shift_group(f, (int)blockIdx.z, 0);
  f.init(c);
  int r = r0 + ty;
  for (; r + (U - 1) * TY < r1; r += U * TY) {
    typename F::In in[U];
#pragma unroll
    for (int k = 0; k < U; ++k) f.load(r + k * TY, c, in[k]);
#pragma unroll
    for (int k = 0; k < U; ++k) f.finish(r + k * TY, c, in[k]);
  }
  for (; r < r1; r += TY) f.row(r, c);
}

template <int NQ, typename T, template <typename, int> class F, class Fin, class... Args>
int row_reduce_g(const char* name, int groups, int M, int C, void* ws, size_t ws_bytes, hipStream_t st, Fin fin, Args... args) {
  Plan p = make_plan<T>(M, C, 1536 / groups);
  size_t need = (size_t)groups * p.gy * NQ * C * sizeof(float);
  RG_REQUIRE(ws && ws_bytes >= need, RG_EWORKSPACE, "%s: workspace %zu < %zu", name, ws_bytes, need);
  float* partial = (float*)ws;
  dim3 grid(p.gx, p.gy, groups);
  const int rev = (rg_option("bn_rev", RG_BN_REV_DEFAULT) >> 2) & 1;
#define RG_RR(V, X)                                                                                          \
  do {                                                                                                       \
    F<T, V> f{args...};                                                                                      \
    hipLaunchKernelGGL((rowreduce_kernel<NQ, V, X, F<T, V>>), grid, dim3(256), 0, st, f, M, C, p.rows_per_block, partial, rev); \
  } while (0)
  if (p.vec == 8) { if (p.tx == 32) RG_RR(8, 32); else if (p.tx == 16) RG_RR(8, 16); else RG_RR(8, 8); }
  else if (p.vec == 4) { if (p.tx == 32) RG_RR(4, 32); else if (p.tx == 16) RG_RR(4, 16); else RG_RR(4, 8); }
  else { if (p.tx == 32) RG_RR(1, 32); else if (p.tx == 16) RG_RR(1, 16); else RG_RR(1, 8); }
#undef RG_RR
  RG_LAUNCH_CHECK(name);
  hipLaunchKernelGGL((colfinish_kernel<NQ, Fin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, partial, C, p.gy, groups);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}
template <int NQ, typename T, template <typename, int> class F, class Fin, class... Args>
int row_reduce(const char* name, int M, int C, void* ws, size_t ws_bytes, hipStream_t st, Fin fin, Args... args) {
  return row_reduce_g<NQ, T, F, Fin, Args...>(name, 1, M, C, ws, ws_bytes, st, fin, args...);
}

template <typename T, template <typename, int> class F, class... Args>
int row_apply_g(const char* name, int groups, int M, int C, hipStream_t st, Args... args) {
  Plan p = make_plan<T>(M, C, 4096 / groups, 8192);      // no partial rows behind an apply pass: as many blocks as the target asks
  dim3 grid(p.gx, p.gy, groups);
  // option bn_rev (bit 0: the passes behind a reduction over the same operands -- backward / tangent / double-backward applies;
  // bit 1: the forward apply behind the kernel that wrote its input): walk the rows from the end (see rowapply_kernel)
  const bool fwd = strncmp(name, "bn_forward", 10) == 0 || strcmp(name, "bn_act") == 0;
  const int rev = (rg_option("bn_rev", RG_BN_REV_DEFAULT) >> (fwd ? 1 : 0)) & 1;
#define RG_RA(V, X)                                                                                   \
  do {                                                                                                \
    F<T, V> f{args...};                                                                               \
    hipLaunchKernelGGL((rowapply_kernel<V, X, F<T, V>>), grid, dim3(256), 0, st, f, M, C, p.rows_per_block, rev); \
  } while (0)
  if (p.vec == 8) { if (p.tx == 32) RG_RA(8, 32); else if (p.tx == 16) RG_RA(8, 16); else RG_RA(8, 8); }
  else if (p.vec == 4) { if (p.tx == 32) RG_RA(4, 32); else if (p.tx == 16) RG_RA(4, 16); else RG_RA(4, 8); }
  else { if (p.tx == 32) RG_RA(1, 32); else if (p.tx == 16) RG_RA(1, 16); else RG_RA(1, 8); }
#undef RG_RA
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}
template <typename T, template <typename, int> class F, class... Args>
int row_apply(const char* name, int M, int C, hipStream_t st, Args... args) {
  return row_apply_g<T, F, Args...>(name, 1, M, C, st, args...);
}

// ---- single-launch form for SMALL tensors (the deep 4x4 / 8x8 layers): a block owns VEC channels for ALL rows, so
// the per-channel reduction, its finisher and the pointwise pass need no other block:
//   reduce over the rows -> wave shuffles + LDS -> fin(c, sums) -> block barrier -> pointwise pass.
// One launch instead of three (row reduce, column finish, row apply); opt-in (RNAGAN_BN_FUSED=1), see fused_small_ok.
template <int NQ, int VEC, class RF, class Fin, class AF>
__global__ __launch_bounds__(256) void fused_rows_kernel(RF rf, Fin fin, AF af, int M, int C) {
  __shared__ float sm[4][NQ][VEC];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = blockIdx.x * VEC;
  float acc[NQ][VEC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[q][v] = 0.f;
  rf.init(c);
#pragma unroll 8
  for (int r = t; r < M; r += 256) rf.row(r, c, acc);      // independent loads: keep several rows in flight
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float w = wave_sum(acc[q][v]);
      if (lane == 0) sm[wave][q][v] = w;
    }
  __syncthreads();
  if (t < VEC) {
    float sv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) sv[q] = sm[0][q][t] + sm[1][q][t] + sm[2][q][t] + sm[3][q][t];
    fin(c + t, sv);
  }
  __threadfence_block();
  __syncthreads();
  af.init(c);
#pragma unroll 8
  for (int r = t; r < M; r += 256) af.row(r, c);
}

static bool fused_small_ok(int M, int C, int vec) {
  // (read on every call so that a test can switch it; getenv is cheap next to a launch)
  int on;
  // measured: slower than the three coalesced launches (15.8 vs 17.6 us forward at [1024][2048] but 27 vs 19 us backward,
  // and 2x slower at [4096][1024]: a block's 16-byte column slices are strided by the row pitch) -> off by default
  { const char* e = getenv("RNAGAN_BN_FUSED"); on = e ? atoi(e) : 0; }
  // up to the [64, 8, 8, 1024] layer; wider tensors have too few channel groups per row to fill the chip this way
  return on && C % vec == 0 && C / vec >= 32 && (size_t)M * C <= ((size_t)9 << 19) && M >= 64;
}
template <int NQ, int VEC, class RF, class Fin, class AF>
static int fused_rows(const char* name, int M, int C, hipStream_t st, RF rf, Fin fin, AF af) {
  hipLaunchKernelGGL((fused_rows_kernel<NQ, VEC, RF, Fin, AF>), dim3(C / VEC), dim3(256), 0, st, rf, fin, af, M, C);
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}
template <typename T> struct VecOf { static constexpr int value = sizeof(T) == 2 ? 8 : 4; };

// per-channel parameter bundle (device pointers) and its register image for VEC channels
struct BNC {
  const float* mean; const float* invstd; const float* gamma; const float* beta; float slope;
};
template <int VEC> struct BNR {
  float mean[VEC], rstd[VEC], gam[VEC], bet[VEC];
  __device__ __forceinline__ void load(const BNC& p, int c) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      mean[i] = p.mean[c + i]; rstd[i] = p.invstd[c + i]; gam[i] = p.gamma[c + i]; bet[i] = p.beta[c + i];
    }
  }
};

// ------------------------------------------------------------------------------------------ stats
template <typename T, int VEC> struct StatsF {
  const T* z; int C; size_t gs = 0;                        // gs: elements per batch group
  __device__ void shift(int g) { z += g * gs; }
  __device__ void init(int) {}
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v; };
  __device__ __forceinline__ void load(int r, int c, In& in) const { in.v.ld(z + (size_t)r * C + c); }
  __device__ __forceinline__ void accum(const In& raw, float (*acc)[VEC]) const {
    struct { float v[VEC]; } in;
    raw.v.cvt(in.v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { acc[0][i] += in.v[i]; acc[1][i] += in.v[i] * in.v[i]; }
  }
  __device__ void row(int r, int c, float (*acc)[VEC]) const { In in; load(r, c, in); accum(in, acc); }
};
struct Store2Fin {
  float* a; float* b;
  __device__ void operator()(int c, const float* s) const { a[c] = s[0]; b[c] = s[1]; }
};

// finisher of the fused statistics pass: batch mean / inverse std and the running-statistics update of
// nn.BatchNorm2d in train mode, straight from the column sums (same arithmetic as bn_finalize_kernel)
struct StatsFinalizeFin {
  float m, eps, momentum;
  float* mean; float* invstd; float* rmean; float* rvar; int64_t* nbt; int gC = 0;       // gC: channels (group stride of mean / invstd)
  __device__ void shift(int g) { mean += g * gC; invstd += g * gC; }
  __device__ void operator()(int c, const float* s) const {
    if (c == 0 && nbt) *nbt += 1;
    // E[x^2] - E[x]^2 formed in double: the subtraction itself must not add to the cancellation when |mean| >> std
    const double mud = (double)s[0] / (double)m;
    const float mu = (float)mud;
    const float var = (float)fmax((double)s[1] / (double)m - mud * mud, 0.0);
    mean[c] = mu;
    invstd[c] = rsqrtf(var + eps);
    if (rmean) {
      float unb = var * (m / fmaxf(m - 1.f, 1.f));
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
    }
  }
};

// ------------------------------------------------------------------------------------------ bn_act
template <typename T, int VEC> struct BnActF {
  const T* z; T* a; BNC p; int C; size_t gs = 0;
  BNR<VEC> q;
  __device__ void shift(int g) { z += g * gs; a += g * gs; p.mean += g * C; p.invstd += g * C; }
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v; };
  __device__ void init(int c) { q.load(p, c); }
  __device__ __forceinline__ void load(int r, int c, In& in) const { in.v.ld(z + (size_t)r * C + c); }
  __device__ __forceinline__ void finish(int r, int c, const In& raw) const {
    struct { float v[VEC]; } in;
    raw.v.cvt(in.v);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = lrelu_f((in.v[i] - q.mean[i]) * (q.rstd[i] * q.gam[i]) + q.bet[i], p.slope);
    Vec<T, VEC>::st(a + (size_t)r * C + c, o);
  }
  __device__ void row(int r, int c) const { In in; load(r, c, in); finish(r, c, in); }
};

// ------------------------------------------------------------------------------------------ bwd
template <typename T, int VEC> struct BwdRedF {
  const T* z; const T* ga; BNC p; int C; size_t gs = 0;
  BNR<VEC> q;
  __device__ void shift(int g) { z += g * gs; ga += g * gs; p.mean += g * C; p.invstd += g * C; }
  __device__ void init(int c) { q.load(p, c); }
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v, g; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.g.ld(ga + (size_t)r * C + c);
  }
  __device__ __forceinline__ void accum(const In& raw, float (*acc)[VEC]) const {
    struct { float v[VEC], g[VEC]; } in;
    raw.v.cvt(in.v);
    raw.g.cvt(in.g);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float xh = (in.v[i] - q.mean[i]) * q.rstd[i];
      float gy = in.g[i] * lrelu_mask(xh * q.gam[i] + q.bet[i], p.slope);
      acc[0][i] += gy; acc[1][i] += gy * xh;
    }
  }
  __device__ void row(int r, int c, float (*acc)[VEC]) const { In in; load(r, c, in); accum(in, acc); }
};
struct BwdFin {
  float* s_gy; float* s_gyxh; float* dgamma; float* dbeta; int accumulate; int gC = 0;
  __device__ void shift(int g) { s_gy += g * gC; s_gyxh += g * gC; if (g) accumulate = 1; }     // later groups add to the first one's
  __device__ void operator()(int c, const float* s) const {
    s_gy[c] = s[0]; s_gyxh[c] = s[1];
    if (dgamma) {
      if (accumulate) { dgamma[c] += s[1]; dbeta[c] += s[0]; }
      else { dgamma[c] = s[1]; dbeta[c] = s[0]; }
    }
  }
};
template <typename T, int VEC> struct BwdApplyF {
  const T* z; const T* ga; T* gz; BNC p; const float* s_gy; const float* s_gyxh; float inv_m; int C; size_t gs = 0;
  BNR<VEC> q; float m1[VEC], m2[VEC];
  __device__ void shift(int g) {
    z += g * gs; ga += g * gs; gz += g * gs; p.mean += g * C; p.invstd += g * C; s_gy += g * C; s_gyxh += g * C;
  }
  __device__ void init(int c) {
    q.load(p, c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { m1[i] = s_gy[c + i] * inv_m; m2[i] = s_gyxh[c + i] * inv_m; }
  }
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v, g; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.g.ld(ga + (size_t)r * C + c);
  }
  __device__ __forceinline__ void finish(int r, int c, const In& raw) const {
    struct { float v[VEC], g[VEC]; } in;
    raw.v.cvt(in.v);
    raw.g.cvt(in.g);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float xh = (in.v[i] - q.mean[i]) * q.rstd[i];
      float gy = in.g[i] * lrelu_mask(xh * q.gam[i] + q.bet[i], p.slope);
      o[i] = (q.gam[i] * q.rstd[i]) * (gy - m1[i] - xh * m2[i]);
    }
    Vec<T, VEC>::st(gz + (size_t)r * C + c, o);
  }
  __device__ void row(int r, int c) const { In in; load(r, c, in); finish(r, c, in); }
};

// ------------------------------------------------------------------------------------------ tangent
template <typename T, int VEC> struct TanRedF {
  const T* z; const T* zt; BNC p; int C;
  BNR<VEC> q;
  __device__ void init(int c) { q.load(p, c); }
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v, t; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.t.ld(zt + (size_t)r * C + c);
  }
  __device__ __forceinline__ void accum(const In& raw, float (*acc)[VEC]) const {
    struct { float v[VEC], t[VEC]; } in;
    raw.v.cvt(in.v);
    raw.t.cvt(in.t);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float xh = (in.v[i] - q.mean[i]) * q.rstd[i];
      acc[0][i] += in.t[i]; acc[1][i] += xh * in.t[i];
    }
  }
  __device__ void row(int r, int c, float (*acc)[VEC]) const { In in; load(r, c, in); accum(in, acc); }
};
template <typename T, int VEC> struct TanApplyF {
  const T* z; const T* zt; T* at; BNC p; const float* s_zt; const float* s_xhzt; float inv_m; int C;
  BNR<VEC> q; float m1[VEC], m2[VEC];
  __device__ void init(int c) {
    q.load(p, c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) { m1[i] = s_zt[c + i] * inv_m; m2[i] = s_xhzt[c + i] * inv_m; }
  }
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v, t; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.t.ld(zt + (size_t)r * C + c);
  }
  __device__ __forceinline__ void finish(int r, int c, const In& raw) const {
    struct { float v[VEC], t[VEC]; } in;
    raw.v.cvt(in.v);
    raw.t.cvt(in.t);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float xh = (in.v[i] - q.mean[i]) * q.rstd[i];
      float yt = (q.gam[i] * q.rstd[i]) * (in.t[i] - m1[i] - xh * m2[i]);
      o[i] = yt * lrelu_mask(xh * q.gam[i] + q.bet[i], p.slope);
    }
    Vec<T, VEC>::st(at + (size_t)r * C + c, o);
  }
  __device__ void row(int r, int c) const { In in; load(r, c, in); finish(r, c, in); }
};

// ------------------------------------------------------------------------------------------ double bwd
template <typename T, int VEC> struct DblRedF {
  const T* z; const T* qa; const T* zt; const T* ga1; BNC p; int C;
  BNR<VEC> q;
  __device__ void init(int c) { q.load(p, c); }
  static constexpr int U = 2;
  struct In { RawVec<T, VEC> v, t, g, qq; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.t.ld(zt + (size_t)r * C + c);
    in.g.ld(ga1 + (size_t)r * C + c);
    if (qa) in.qq.ld(qa + (size_t)r * C + c);
  }
  __device__ __forceinline__ void accum(const In& raw, float (*acc)[VEC]) const {
    struct { float v[VEC], t[VEC], g[VEC], qq[VEC]; } in;
    raw.v.cvt(in.v);
    raw.t.cvt(in.t);
    raw.g.cvt(in.g);
    if (qa) raw.qq.cvt(in.qq);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float xh = (in.v[i] - q.mean[i]) * q.rstd[i];
      float mk = lrelu_mask(xh * q.gam[i] + q.bet[i], p.slope);
      acc[0][i] += in.g[i] * mk * in.t[i];
      if (qa) { float qy = in.qq[i] * mk; acc[1][i] += qy; acc[2][i] += qy * xh; }
    }
  }
  __device__ void row(int r, int c, float (*acc)[VEC]) const { In in; load(r, c, in); accum(in, acc); }
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
template <typename T, int VEC> struct DblApplyF {
  const T* z; const T* qa; const T* zt; const T* ga1; T* pz; BNC p; const float* s_gy; const float* s_zt;
  const float* coef; float inv_m; int C;
  BNR<VEC> q; float k0[VEC], k1[VEC], k2[VEC], k3[VEC], k4[VEC], mgy[VEC], mzt[VEC];
  __device__ void init(int c) {
    q.load(p, c);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      k0[i] = coef[0 * C + c + i]; k1[i] = coef[1 * C + c + i]; k2[i] = coef[2 * C + c + i];
      k3[i] = coef[3 * C + c + i]; k4[i] = coef[4 * C + c + i];
      mgy[i] = s_gy[c + i] * inv_m; mzt[i] = s_zt[c + i] * inv_m;
    }
  }
  static constexpr int U = 2;
  struct In { RawVec<T, VEC> v, t, g, qq; };
  __device__ __forceinline__ void load(int r, int c, In& in) const {
    in.v.ld(z + (size_t)r * C + c);
    in.t.ld(zt + (size_t)r * C + c);
    in.g.ld(ga1 + (size_t)r * C + c);
    if (qa) in.qq.ld(qa + (size_t)r * C + c);
  }
  __device__ __forceinline__ void finish(int r, int c, const In& raw) const {
    struct { float v[VEC], t[VEC], g[VEC], qq[VEC]; } in;
    raw.v.cvt(in.v);
    raw.t.cvt(in.t);
    raw.g.cvt(in.g);
    if (qa) raw.qq.cvt(in.qq);
    float o[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float is = q.rstd[i], gm = q.gam[i];
      float xh = (in.v[i] - q.mean[i]) * is;
      float mk = lrelu_mask(xh * gm + q.bet[i], p.slope);
      float gy = in.g[i] * mk;
      float r0 = xh * k0[i] + k1[i] * (gy - mgy[i]) + k2[i] * (in.t[i] - mzt[i]);
      float out = -(gm * is * is) * r0;
      if (qa) out += (gm * is) * (in.qq[i] * mk - k3[i] - xh * k4[i]);
      o[i] = out;
    }
    Vec<T, VEC>::st(pz + (size_t)r * C + c, o);
  }
  __device__ void row(int r, int c) const { In in; load(r, c, in); finish(r, c, in); }
};

// ---- split forms (synchronised statistics): raw sums out / finish from all-reduced sums
struct Store3Fin {
  float* raw; int C;
  __device__ void operator()(int c, const float* s) const { raw[c] = s[0]; raw[C + c] = s[1]; raw[2 * C + c] = s[2]; }
};
// DblFin with the coefficient means taken over the GLOBAL batch (sums in `s`, m_total rows) while the parameter
// gradients stay rank-local contributions (they are summed by the gradient all-reduce):
//   dgamma_local = sum_local(qy xh) + m_local/sigma (A - b c),  dbeta_local = sum_local(qy)
struct DblFinSync {
  const float* s_gy; const float* s_gyxh; const float* s_zt; const float* s_xhzt; const float* invstd;
  const float* raw_local; float* coef; float* dgamma; float* dbeta; int accumulate; float m_total, m_local; int C;
  __device__ void operator()(int c, const float* s) const {
    float inv_m = 1.f / m_total;
    float b = s_gyxh[c] * inv_m, cc = s_xhzt[c] * inv_m;
    float A = s[0] * inv_m - (s_gy[c] * inv_m) * (s_zt[c] * inv_m);
    coef[0 * C + c] = A - 3.f * b * cc;
    coef[1 * C + c] = cc;
    coef[2 * C + c] = b;
    coef[3 * C + c] = s[1] * inv_m;
    coef[4 * C + c] = s[2] * inv_m;
    float dg = m_local * invstd[c] * (A - b * cc) + raw_local[2 * C + c];
    float db = raw_local[C + c];
    if (accumulate) { dgamma[c] += dg; dbeta[c] += db; }
    else { dgamma[c] = dg; dbeta[c] = db; }
  }
};

// ------------------------------------------------------------------------------------------ misc
template <typename T, int VEC> struct ColSumF {
  const T* g; int C;
  __device__ void init(int) {}
  static constexpr int U = 4;
  struct In { RawVec<T, VEC> v; };
  __device__ __forceinline__ void load(int r, int c, In& in) const { in.v.ld(g + (size_t)r * C + c); }
  __device__ __forceinline__ void accum(const In& raw, float (*acc)[VEC]) const {
    struct { float v[VEC]; } in;
    raw.v.cvt(in.v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc[0][i] += in.v[i];
  }
  __device__ void row(int r, int c, float (*acc)[VEC]) const { In in; load(r, c, in); accum(in, acc); }
};
struct AccFin {
  float* out; int accumulate;
  __device__ void operator()(int c, const float* s) const { out[c] = accumulate ? out[c] + s[0] : s[0]; }
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
  const double mud = (double)sum[c] / (double)m;
  const float mu = (float)mud;
  const float var = (float)fmax((double)sumsq[c] / (double)m - mud * mud, 0.0);
  mean[c] = mu;
  invstd[c] = rsqrtf(var + eps);
  if (rmean) {
    float unb = var * (m / fmaxf(m - 1.f, 1.f));
    rmean[c] = (1.f - momentum) * rmean[c] + momentum * mu;
    rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
  }
}

}  // namespace

extern "C" size_t rg_colreduce_workspace_bytes(int M, int C, int nq) {
  if (M <= 0 || C <= 0) return 0;
  // partials + 5 per-channel coefficient rows used by rg_bn_double_bwd
  return rg_align_up((size_t)max_gy(M, C) * (nq < 1 ? 1 : nq) * C * sizeof(float), 256) + (size_t)5 * C * sizeof(float);
}

extern "C" int rg_bn_stats(const void* z, float* sum, float* sumsq, int M, int C, int dtype, void* ws, size_t ws_bytes,
                           void* stream) {
  RG_REQUIRE(z && sum && sumsq && M > 0 && C > 0, RG_EINVAL, "bn_stats: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<2, T, StatsF>("bn_stats", M, C, ws, ws_bytes, rg_stream(stream), Store2Fin{sum, sumsq},
                                     (const T*)z, C));
  })
}

extern "C" int rg_bn_stats_finalize(const void* z, int M, int C, float eps, float momentum, float* mean, float* invstd,
                                    float* running_mean, float* running_var, int64_t* num_batches_tracked, int dtype,
                                    void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && mean && invstd && M > 0 && C > 0, RG_EINVAL, "bn_stats_finalize: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<2, T, StatsF>("bn_stats_finalize", M, C, ws, ws_bytes, rg_stream(stream), fin, (const T*)z, C));
  })
}

extern "C" int rg_bn_forward(const void* z, int M, int C, float eps, float momentum, const float* gamma, const float* beta,
                             float slope, float* mean, float* invstd, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, void* a, int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && a && mean && invstd && gamma && beta && M > 0 && C > 0, RG_EINVAL, "bn_forward: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr};
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    constexpr int V = VecOf<T>::value;
    if (fused_small_ok(M, C, V))
      return fused_rows<2, V>("bn_forward(fused)", M, C, st, StatsF<T, V>{(const T*)z, C}, fin,
                              BnActF<T, V>{(const T*)z, (T*)a, p, C});
    int rc = row_reduce<2, T, StatsF>("bn_forward", M, C, ws, ws_bytes, st, fin, (const T*)z, C);
    if (rc) return rc;
    return (row_apply<T, BnActF>("bn_forward", M, C, st, (const T*)z, (T*)a, p, C));
  })
}

extern "C" int rg_bn_forward_partials(const float* partial, int G, const void* z, int M, int C, float eps, float momentum,
                                      const float* gamma, const float* beta, float slope, float* mean, float* invstd,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, void* a,
                                      int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(partial && G > 0 && z && a && mean && invstd && gamma && beta && M > 0 && C > 0, RG_EINVAL,
             "bn_forward_partials: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr};
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  // many partial rows (big layers: one per row tile and wave row): two levels, 32 slices first
  constexpr int SLICES = 32;
  if (G > 512 && C % 8 == 0 && ws && ws_bytes >= (size_t)SLICES * 2 * C * sizeof(float)) {
    float* stage = (float*)ws;
    hipLaunchKernelGGL((colfinish_wide_kernel<2, SliceFin>), dim3(C / 8, SLICES), dim3(256), 0, st, SliceFin{stage, C},
                       partial, C, G);
    RG_LAUNCH_CHECK("bn_forward_partials");
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, stage, C,
                       SLICES);
  } else {
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, partial, C, G);
  }
  RG_LAUNCH_CHECK("bn_forward_partials");
  RG_DISPATCH_DTYPE(dtype, T, { return (row_apply<T, BnActF>("bn_forward_partials", M, C, st, (const T*)z, (T*)a, p, C)); })
}

// Statistics only, from conv-epilogue column sums: mean / invstd (+ running statistics) as rg_bn_forward_partials computes
// them, without the normalisation pass (its consumer applies BatchNorm itself: rg_last_up_pre).
extern "C" int rg_bn_finalize_partials(const float* partial, int G, int M, int C, float eps, float momentum, float* mean,
                                       float* invstd, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(partial && G > 0 && mean && invstd && M > 0 && C > 0, RG_EINVAL, "bn_finalize_partials: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr};
  hipStream_t st = rg_stream(stream);
  constexpr int SLICES = 32;
  if (G > 512 && C % 8 == 0 && ws && ws_bytes >= (size_t)SLICES * 2 * C * sizeof(float)) {
    float* stage = (float*)ws;
    hipLaunchKernelGGL((colfinish_wide_kernel<2, SliceFin>), dim3(C / 8, SLICES), dim3(256), 0, st, SliceFin{stage, C},
                       partial, C, G);
    RG_LAUNCH_CHECK("bn_finalize_partials");
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, stage, C, SLICES,
                       1);
  } else {
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, partial, C, G, 1);
  }
  RG_LAUNCH_CHECK("bn_finalize_partials");
  return RG_OK;
}

// Two batch groups in one call: z / a are [2*M][C] (group-major), mean / invstd [2][C]; statistics, normalisation and the
// running-statistics update exactly as two consecutive rg_bn_forward(_partials) calls on the halves (first half first).
// partial (may be NULL): conv-epilogue column sums, 2*G rows [.][2][C] laid out [nblk][2 halves][G / nblk]: nblk = 1 for
// rg_conv_down (rows in row-tile order: the first G rows are the first half), nblk = 4 for rg_conv_up (class-major rows).
static int finalize_partials_g2(const char* name, const float* partial, int G, int nblk, int C, StatsFinalizeFin fin, void* ws,
                                size_t ws_bytes, hipStream_t st) {
  constexpr int SLICES = 32;
  RG_REQUIRE(nblk >= 1 && G % nblk == 0, RG_EINVAL, "%s: partial rows %d not a multiple of %d blocks", name, G, nblk);
  if (G > 512 && C % 8 == 0 && ws && ws_bytes >= (size_t)2 * SLICES * 2 * C * sizeof(float)) {
    float* stage = (float*)ws;
    hipLaunchKernelGGL((colfinish_wide_kernel<2, SliceFin>), dim3(C / 8, SLICES, 2), dim3(256), 0, st,
                       SliceFin{stage, C, SLICES}, partial, C, G, nblk);
    RG_LAUNCH_CHECK(name);
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, stage, C, SLICES,
                       2, 1);
  } else {
    hipLaunchKernelGGL((colfinish_kernel<2, StatsFinalizeFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, partial, C, G, 2,
                       nblk);
  }
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

// rg_bn_finalize_partials for two batch groups (mean / invstd [2][C]); nblk as in rg_bn_forward_g2.
extern "C" int rg_bn_finalize_partials_g2(const float* partial, int G, int nblk, int M, int C, float eps, float momentum,
                                          float* mean, float* invstd, float* running_mean, float* running_var,
                                          int64_t* num_batches_tracked, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(partial && G > 0 && mean && invstd && M > 0 && C > 0, RG_EINVAL, "bn_finalize_partials_g2: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr, C};
  return finalize_partials_g2("bn_finalize_partials_g2", partial, G, nblk, C, fin, ws, ws_bytes, rg_stream(stream));
}

extern "C" int rg_bn_forward_g2(const float* partial, int G, int nblk, const void* z, int M, int C, float eps, float momentum,
                                const float* gamma, const float* beta, float slope, float* mean, float* invstd,
                                float* running_mean, float* running_var, int64_t* num_batches_tracked, void* a, int dtype,
                                void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && a && mean && invstd && gamma && beta && M > 0 && C > 0 && (!partial || G > 0), RG_EINVAL,
             "bn_forward_g2: bad args");
  StatsFinalizeFin fin{(float)M, eps, momentum, mean, invstd, running_mean, running_var,
                       running_mean ? num_batches_tracked : nullptr, C};
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  const size_t gs = (size_t)M * C;
  if (partial) {
    int rc = finalize_partials_g2("bn_forward_g2", partial, G, nblk, C, fin, ws, ws_bytes, st);
    if (rc) return rc;
    RG_DISPATCH_DTYPE(dtype, T, {
      return (row_apply_g<T, BnActF>("bn_forward_g2", 2, M, C, st, (const T*)z, (T*)a, p, C, gs));
    })
  }
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc = row_reduce_g<2, T, StatsF>("bn_forward_g2", 2, M, C, ws, ws_bytes, st, fin, (const T*)z, C, gs);
    if (rc) return rc;
    return (row_apply_g<T, BnActF>("bn_forward_g2", 2, M, C, st, (const T*)z, (T*)a, p, C, gs));
  })
}

// Backward of two batch groups (see rg_bn_forward_g2): gz as two rg_bn_act_bwd calls on the halves would give it, s_gy /
// s_gyxh [2][C] per half, dgamma / dbeta = the sum over both halves (written, or added when `accumulate`).
extern "C" int rg_bn_act_bwd_g2(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, void* gz, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta,
                                int accumulate, int M, int C, float slope, int dtype, void* ws, size_t ws_bytes,
                                void* stream) {
  RG_REQUIRE(z && ga && gz && s_gy && s_gyxh && M > 0 && C > 0, RG_EINVAL, "bn_act_bwd_g2: bad args");
  RG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), RG_EINVAL, "bn_act_bwd_g2: dgamma/dbeta must come together");
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  const size_t gs = (size_t)M * C;
  RG_DISPATCH_DTYPE(dtype, T, {
    int rc = row_reduce_g<2, T, BwdRedF>("bn_act_bwd_g2", 2, M, C, ws, ws_bytes, st,
                                         BwdFin{s_gy, s_gyxh, dgamma, dbeta, accumulate, C}, (const T*)z, (const T*)ga, p, C,
                                         gs);
    if (rc) return rc;
    return (row_apply_g<T, BwdApplyF>("bn_act_bwd_g2", 2, M, C, st, (const T*)z, (const T*)ga, (T*)gz, p,
                                      (const float*)s_gy, (const float*)s_gyxh, 1.f / (float)M, C, gs));
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
    return (row_apply<T, BnActF>("bn_act", M, C, rg_stream(stream), (const T*)z, (T*)a, p, C));
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
    constexpr int V = VecOf<T>::value;
    if (fused_small_ok(M, C, V))
      return fused_rows<2, V>("bn_act_bwd(fused)", M, C, st, BwdRedF<T, V>{(const T*)z, (const T*)ga, p, C},
                              BwdFin{s_gy, s_gyxh, dgamma, dbeta, accumulate},
                              BwdApplyF<T, V>{(const T*)z, (const T*)ga, (T*)gz, p, (const float*)s_gy,
                                              (const float*)s_gyxh, 1.f / (float)M, C});
    int rc = row_reduce<2, T, BwdRedF>("bn_act_bwd", M, C, ws, ws_bytes, st,
                                       BwdFin{s_gy, s_gyxh, dgamma, dbeta, accumulate}, (const T*)z, (const T*)ga, p, C);
    if (rc) return rc;
    return (row_apply<T, BwdApplyF>("bn_act_bwd", M, C, st, (const T*)z, (const T*)ga, (T*)gz, p, (const float*)s_gy,
                                    (const float*)s_gyxh, 1.f / (float)M, C));
  })
}

// rg_bn_act_bwd[_g2] with the two sums taken from the partial rows a data-gradient conv's epilogue wrote (rg_conv_*_bnbwd):
// G rows per batch group [.][2][C], laid out [nblk][groups][G / nblk] (nblk = 1: rg_conv_down_bnbwd's row-tile order; 4:
// rg_conv_up_bnbwd's class-major order) -> finisher (s_gy, s_gyxh, dgamma, dbeta) + the pointwise pass; the reduction pass
// over (z, ga) is gone.
extern "C" int rg_bn_act_bwd_partials(const float* partial, int G, int nblk, const void* z, const void* ga, const float* mean,
                                      const float* invstd, const float* gamma, const float* beta, void* gz, float* s_gy,
                                      float* s_gyxh, float* dgamma, float* dbeta, int accumulate, int M, int C, int groups,
                                      float slope, int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(partial && G > 0 && z && ga && gz && s_gy && s_gyxh && M > 0 && C > 0 && (groups == 1 || groups == 2) && nblk >= 1 &&
                 G % nblk == 0, RG_EINVAL, "bn_act_bwd_partials: bad args");
  RG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), RG_EINVAL, "bn_act_bwd_partials: dgamma/dbeta must come together");
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  BwdFin fin{s_gy, s_gyxh, dgamma, dbeta, accumulate, C};
  constexpr int SLICES = 32;
  if (G > 512 && C % 8 == 0 && ws && ws_bytes >= (size_t)groups * SLICES * 2 * C * sizeof(float)) {
    float* stage = (float*)ws;
    hipLaunchKernelGGL((colfinish_wide_kernel<2, SliceFin>), dim3(C / 8, SLICES, groups), dim3(256), 0, st,
                       SliceFin{stage, C, SLICES}, partial, C, G, nblk);
    RG_LAUNCH_CHECK("bn_act_bwd_partials");
    hipLaunchKernelGGL((colfinish_kernel<2, BwdFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, stage, C, SLICES, groups, 1);
  } else {
    hipLaunchKernelGGL((colfinish_kernel<2, BwdFin>), dim3((C + 7) / 8), dim3(256), 0, st, fin, partial, C, G, groups, nblk);
  }
  RG_LAUNCH_CHECK("bn_act_bwd_partials");
  const size_t gs = (size_t)M * C;
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_apply_g<T, BwdApplyF>("bn_act_bwd_partials", groups, M, C, st, (const T*)z, (const T*)ga, (T*)gz, p,
                                      (const float*)s_gy, (const float*)s_gyxh, 1.f / (float)M, C, gs));
  })
}

extern "C" int rg_bn_tangent(const void* z, const void* zt, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, void* at, float* s_zt, float* s_xhzt, int M, int C, float slope,
                             int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && at && s_zt && s_xhzt && M > 0 && C > 0, RG_EINVAL, "bn_tangent: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    constexpr int V = VecOf<T>::value;
    if (fused_small_ok(M, C, V))
      return fused_rows<2, V>("bn_tangent(fused)", M, C, st, TanRedF<T, V>{(const T*)z, (const T*)zt, p, C},
                              Store2Fin{s_zt, s_xhzt},
                              TanApplyF<T, V>{(const T*)z, (const T*)zt, (T*)at, p, (const float*)s_zt,
                                              (const float*)s_xhzt, 1.f / (float)M, C});
    int rc = row_reduce<2, T, TanRedF>("bn_tangent", M, C, ws, ws_bytes, st, Store2Fin{s_zt, s_xhzt}, (const T*)z,
                                       (const T*)zt, p, C);
    if (rc) return rc;
    return (row_apply<T, TanApplyF>("bn_tangent", M, C, st, (const T*)z, (const T*)zt, (T*)at, p, (const float*)s_zt,
                                    (const float*)s_xhzt, 1.f / (float)M, C));
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
    constexpr int V = VecOf<T>::value;
    if (fused_small_ok(M, C, V))
      return fused_rows<3, V>("bn_double_bwd(fused)", M, C, st,
                              DblRedF<T, V>{(const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1, p, C},
                              DblFin{s_gy, s_gyxh, s_zt, s_xhzt, invstd, coef, dgamma, dbeta, accumulate, (float)M, C},
                              DblApplyF<T, V>{(const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1, (T*)pz, p, s_gy,
                                              s_zt, (const float*)coef, 1.f / (float)M, C});
    int rc = row_reduce<3, T, DblRedF>("bn_double_bwd", M, C, ws, ws_bytes - (size_t)5 * C * sizeof(float), st,
                                       DblFin{s_gy, s_gyxh, s_zt, s_xhzt, invstd, coef, dgamma, dbeta, accumulate,
                                              (float)M, C},
                                       (const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1, p, C);
    if (rc) return rc;
    return (row_apply<T, DblApplyF>("bn_double_bwd", M, C, st, (const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1,
                                    (T*)pz, p, s_gy, s_zt, (const float*)coef, 1.f / (float)M, C));
  })
}

/* ---- split forms for synchronised (global-batch) BatchNorm statistics: *_sums writes the RANK-LOCAL column sums,
 * the caller all-reduces them, *_apply finishes with the global sums and the global row count M_total. ---- */
extern "C" int rg_bn_bwd_sums(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* s_gy, float* s_gyxh, int M, int C, float slope, int dtype, void* ws,
                              size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && ga && s_gy && s_gyxh && M > 0 && C > 0, RG_EINVAL, "bn_bwd_sums: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<2, T, BwdRedF>("bn_bwd_sums", M, C, ws, ws_bytes, rg_stream(stream), Store2Fin{s_gy, s_gyxh},
                                      (const T*)z, (const T*)ga, p, C));
  })
}
extern "C" int rg_bn_bwd_apply(const void* z, const void* ga, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, const float* s_gy, const float* s_gyxh, void* gz, int M, int C,
                               int M_total, float slope, int dtype, void* stream) {
  RG_REQUIRE(z && ga && gz && s_gy && s_gyxh && M > 0 && C > 0 && M_total >= M, RG_EINVAL, "bn_bwd_apply: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_apply<T, BwdApplyF>("bn_bwd_apply", M, C, rg_stream(stream), (const T*)z, (const T*)ga, (T*)gz, p, s_gy,
                                    s_gyxh, 1.f / (float)M_total, C));
  })
}
extern "C" int rg_bn_tangent_sums(const void* z, const void* zt, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, float* s_zt, float* s_xhzt, int M, int C,
                                  float slope, int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && s_zt && s_xhzt && M > 0 && C > 0, RG_EINVAL, "bn_tangent_sums: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<2, T, TanRedF>("bn_tangent_sums", M, C, ws, ws_bytes, rg_stream(stream), Store2Fin{s_zt, s_xhzt},
                                      (const T*)z, (const T*)zt, p, C));
  })
}
extern "C" int rg_bn_tangent_apply(const void* z, const void* zt, const float* mean, const float* invstd,
                                   const float* gamma, const float* beta, const float* s_zt, const float* s_xhzt, void* at,
                                   int M, int C, int M_total, float slope, int dtype, void* stream) {
  RG_REQUIRE(z && zt && at && s_zt && s_xhzt && M > 0 && C > 0 && M_total >= M, RG_EINVAL, "bn_tangent_apply: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_apply<T, TanApplyF>("bn_tangent_apply", M, C, rg_stream(stream), (const T*)z, (const T*)zt, (T*)at, p,
                                    s_zt, s_xhzt, 1.f / (float)M_total, C));
  })
}
extern "C" int rg_bn_dbl_sums(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean,
                              const float* invstd, const float* gamma, const float* beta, float* raw3, int M, int C,
                              float slope, int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && ga1 && raw3 && M > 0 && C > 0, RG_EINVAL, "bn_dbl_sums: bad args");
  BNC p{mean, invstd, gamma, beta, slope};
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<3, T, DblRedF>("bn_dbl_sums", M, C, ws, ws_bytes, rg_stream(stream), Store3Fin{raw3, C},
                                      (const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1, p, C));
  })
}
extern "C" int rg_bn_dbl_apply(const void* z, const void* qa, const void* zt, const void* ga1, const float* mean,
                               const float* invstd, const float* gamma, const float* beta, const float* s_gy,
                               const float* s_gyxh, const float* s_zt, const float* s_xhzt, const float* raw3_global,
                               const float* raw3_local, void* pz, float* dgamma, float* dbeta, int accumulate, int M,
                               int C, int M_total, float slope, int dtype, void* ws, size_t ws_bytes, void* stream) {
  RG_REQUIRE(z && zt && ga1 && pz && dgamma && dbeta && s_gy && s_gyxh && s_zt && s_xhzt && raw3_global && raw3_local &&
                 M > 0 && C > 0 && M_total >= M, RG_EINVAL, "bn_dbl_apply: bad args");
  RG_REQUIRE(ws && ws_bytes >= (size_t)5 * C * sizeof(float), RG_EWORKSPACE, "bn_dbl_apply: workspace too small");
  float* coef = (float*)ws;
  BNC p{mean, invstd, gamma, beta, slope};
  hipStream_t st = rg_stream(stream);
  // the all-reduced sums are one "partial row" [3][C]: colfinish with G = 1 runs the finisher per channel
  hipLaunchKernelGGL((colfinish_kernel<3, DblFinSync>), dim3((C + 7) / 8), dim3(256), 0, st,
                     DblFinSync{s_gy, s_gyxh, s_zt, s_xhzt, invstd, raw3_local, coef, dgamma, dbeta, accumulate,
                                (float)M_total, (float)M, C},
                     raw3_global, C, 1);
  RG_LAUNCH_CHECK("bn_dbl_apply");
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_apply<T, DblApplyF>("bn_dbl_apply", M, C, st, (const T*)z, (const T*)qa, (const T*)zt, (const T*)ga1,
                                    (T*)pz, p, s_gy, s_zt, (const float*)coef, 1.f / (float)M_total, C));
  })
}

extern "C" int rg_col_sum(const void* g, float* out, int M, int C, int dtype, int accumulate, void* ws, size_t ws_bytes,
                          void* stream) {
  RG_REQUIRE(g && out && M > 0 && C > 0, RG_EINVAL, "col_sum: bad args");
  RG_DISPATCH_DTYPE(dtype, T, {
    return (row_reduce<1, T, ColSumF>("col_sum", M, C, ws, ws_bytes, rg_stream(stream), AccFin{out, accumulate},
                                      (const T*)g, C));
  })
}

extern "C" int rg_lrelu_bwd(const void* g, const void* a, void* out, size_t n, float slope, int dtype, void* stream) {
  RG_REQUIRE(g && a && out, RG_EINVAL, "lrelu_bwd: bad args");
  if (n == 0) return RG_OK;
  hipStream_t st = rg_stream(stream);
  RG_DISPATCH_DTYPE(dtype, T, {
    constexpr int VM = sizeof(T) == 2 ? 8 : 4;
    if (n % VM == 0) {
      size_t nvec = n / VM, blocks = (nvec + 255) / 256;
      if (blocks > 8192) blocks = 8192;
      hipLaunchKernelGGL((lrelu_bwd_kernel<T, VM>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)g, (const T*)a,
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
