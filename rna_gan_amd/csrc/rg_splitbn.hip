// rg_splitbn.hip -- split-K slab reduction FUSED with the train-mode BatchNorm pass that consumes it (bf16 path).
//
// The deep conv layers at batch 64 run split-K: 256 workgroups leave 67 MB of fp32 partial slabs [split][M][C] behind
// whatever the split factor.  Round 2 then ran, per layer, reduce_slabs_bf16 (slabs -> z) + a statistics
// pass (read z) + a column finisher + the apply pass (read z, write a): four launches, z read twice, for tensors of 4-16 MB
// whose passes are latency-bound (5-12 us each).  Here ONE launch does all of it:
//
//   phase 1  each workgroup owns a [rows x 128 columns] block: z = bf16(sum_s slab[s]) in fixed split order, kept packed in
//            registers and written once; per-column partial sums of the block (deterministic tree) -> `part`
//   hand-off the workgroups of one 128-column SLICE meet at a counter (CDNA guide, Guideline 16 recipe R1: partials stored
//            write-through (sc1), every storing wave drains, one lane adds to the slice's arrival counter; the last
//            arriver resets the counter and bumps the slice's generation word; the others poll that ONE word relaxed with
//            s_sleep, then ONE agent-scope acquire, vmcnt(0), barrier).  No memset node: the sync words are a small
//            caller-owned buffer zeroed once at allocation and left zero by every launch (self-resetting).
//   phase 2  every workgroup sums the slice's partial rows in fixed order (same result in all of them) -> batch mean /
//            inverse std (StatsFinalizeFin's arithmetic); the slice's first workgroup also writes mean / invstd and updates
//            the running statistics, batch groups in order (a double batch = two forward calls)
//   phase 3  a = lrelu(gamma * (z - mean) * invstd + beta) from the registers, written once
//
// and the backward twin: ga = bf16(sum_s slab[s]) (the data gradient arriving from the layer above), gy = ga * lrelu'(y),
// the two backward sums, hand-off, gz = gamma * invstd * (gy - mean(gy) - xhat * mean(gy * xhat)); dgamma / dbeta.
// HBM traffic = slabs once + each output once + z once (backward): the minimum.
//
// Requirements: the whole grid is co-resident (<= 256 workgroups of 1024 threads, <= 128 VGPRs: one per CU always fits),
// C % 128 == 0, rows per block in {64, 128, 256}, at most 64 row blocks per batch group.  Every spin is bounded; a timeout
// sets sync[SB_ERR].
#include "rg_internal.h"

namespace {

constexpr int SB_COLS = 128;          // columns per slice (16 threads x 8)
constexpr int SB_TY = 64;             // row lanes per block (1024 threads / 16 column threads)
constexpr int SB_ERR = 0;             // sync[0]: error word (timeouts); slice s uses sync[16 + 16*s + {0: arrivals, 1: generation}]

typedef unsigned __attribute__((address_space(1))) gu32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SbnArgs {
  const float* slab; size_t slab_stride; int nsplit;
  int slab16;               // the slabs are bf16 (RG_H16 slab_dtype: conv8_kernel's slab16 form), else fp32
  const uint16_t* zin;      // backward: the layer's stored pre-activation z
  uint16_t* out0;           // forward: z ; backward: ga (may be null)
  uint16_t* out1;           // forward: a ; backward: gz
  int M;                    // rows per batch group
  int C, groups, rbpg;      // rbpg: row blocks per group
  const float* gamma; const float* beta; float slope, eps, momentum;
  float* mean; float* invstd;             // [groups][C]   forward: written ; backward: read
  float* rmean; float* rvar; long long* nbt;
  float* s_gy; float* s_gyxh;             // backward outputs [groups][C]
  float* dgamma; float* dbeta; int accumulate;
  float* part;              // [slices][groups][rbpg][2][128]
  unsigned* sync;
};

// every word another workgroup reads in this launch: a GLOBAL (address space 1) agent-scope access -- sc1, never flat
__device__ __forceinline__ gu32* as_global(const void* p) { return (gu32*)(unsigned*)const_cast<void*>(p); }
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(as_global(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __uint_as_float(__hip_atomic_load(as_global(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// All workgroups of a slice meet here after their partial rows are stored (sc1).  Returns after the slice's partials are
// readable by every thread of this workgroup.
__device__ __forceinline__ void slice_rendezvous(unsigned* sync, int slice, unsigned members) {
  gu32* arrive = as_global(sync + 16 + 16 * slice);
  gu32* gen = arrive + 1;
  // the generation this launch starts from: read BEFORE arriving (it can only change after every member has arrived)
  unsigned g0 = 0;
  if (threadIdx.x == 0) g0 = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // EVERY storing wave drains its sc1 stores (R1)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == members - 1) {
      // last arriver: leave the counter zero for the next launch, then release the others
      __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned spins = 0;
      while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 26)) {                          // bounded: a lost member must not hang the GPU
          __hip_atomic_store(as_global(sync + SB_ERR), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");       // ONE acquire after the match
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

__device__ __forceinline__ void unpack8(const uint4& t, float* o) {
  o[0] = h16lo_to_f32(t.x); o[1] = h16hi_to_f32(t.x);
  o[2] = h16lo_to_f32(t.y); o[3] = h16hi_to_f32(t.y);
  o[4] = h16lo_to_f32(t.z); o[5] = h16hi_to_f32(t.z);
  o[6] = h16lo_to_f32(t.w); o[7] = h16hi_to_f32(t.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 t;
  t.x = (uint32_t)f32_to_h16(v[0]) | ((uint32_t)f32_to_h16(v[1]) << 16);
  t.y = (uint32_t)f32_to_h16(v[2]) | ((uint32_t)f32_to_h16(v[3]) << 16);
  t.z = (uint32_t)f32_to_h16(v[4]) | ((uint32_t)f32_to_h16(v[5]) << 16);
  t.w = (uint32_t)f32_to_h16(v[6]) | ((uint32_t)f32_to_h16(v[7]) << 16);
  return t;
}

// the NS slab pieces of (row, c .. c+7): loads issued together (a thread keeps 16 x 16 B in flight: rows are processed in
// chunks of U = 8 / NS so that the next rows' loads are not held back by the previous rows' stores)
// S16: the slabs are bf16 (conv8_kernel with G2Args::slab16): ONE 16-byte load per piece, widened when summed.
template <int NS, bool S16> struct SlabPiece;
template <int NS> struct SlabPiece<NS, false> { float4 v[NS][2]; };
template <int NS> struct SlabPiece<NS, true> { uint4 v[NS]; };
template <int NS>
__device__ __forceinline__ void slab_load(const float* __restrict__ p, size_t off, size_t stride, SlabPiece<NS, false>& q) {
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    q.v[s][0] = *reinterpret_cast<const float4*>(p + off + s * stride);
    q.v[s][1] = *reinterpret_cast<const float4*>(p + off + s * stride + 4);
  }
}
template <int NS>
__device__ __forceinline__ void slab_load(const float* __restrict__ p, size_t off, size_t stride, SlabPiece<NS, true>& q) {
  const uint16_t* h = reinterpret_cast<const uint16_t*>(p);
#pragma unroll
  for (int s = 0; s < NS; ++s) q.v[s] = *reinterpret_cast<const uint4*>(h + off + s * stride);
}
// their sum in fixed order s = 0 .. NS-1 (the order reduce_slabs_bf16_kernel uses)
template <int NS>
__device__ __forceinline__ void slab_sum(const SlabPiece<NS, false>& q, float* acc) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    acc[0] += q.v[s][0].x; acc[1] += q.v[s][0].y; acc[2] += q.v[s][0].z; acc[3] += q.v[s][0].w;
    acc[4] += q.v[s][1].x; acc[5] += q.v[s][1].y; acc[6] += q.v[s][1].z; acc[7] += q.v[s][1].w;
  }
}
template <int NS>
__device__ __forceinline__ void slab_sum(const SlabPiece<NS, true>& q, float* acc) {
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    float v[8];
    unpack8(q.v[s], v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += v[i];
  }
}

// ---- block geometry: 1024 threads = 16 waves (4 per SIMD: enough loads in flight to stream the slabs at HBM speed with one
// workgroup per CU); thread -> 8 columns (tx = t % 16) of row lane ty = t / 16 (64 row lanes, 4 of them per wave)
constexpr int SB_THREADS = 1024;
constexpr int SB_WAVES = SB_THREADS / 64;

// block partial of two per-thread column sums -> part row (sc1).  Deterministic: the 4 row lanes of a wave by shuffles,
// the 16 waves through LDS in order.
__device__ __forceinline__ void block_partial(float (*sm)[2][SB_COLS], float* s1, float* s2, float* prow) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tx = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s1[i] += __shfl_xor(s1[i], 16, 64); s1[i] += __shfl_xor(s1[i], 32, 64);
    s2[i] += __shfl_xor(s2[i], 16, 64); s2[i] += __shfl_xor(s2[i], 32, 64);
  }
  if (lane < 16) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { sm[wave][0][tx * 8 + i] = s1[i]; sm[wave][1][tx * 8 + i] = s2[i]; }
  }
  __syncthreads();
  if (threadIdx.x < 2 * SB_COLS) {
    const int q = threadIdx.x / SB_COLS, col = threadIdx.x % SB_COLS;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SB_WAVES; ++k) t += sm[k][q][col];
    st_sc1(prow + q * SB_COLS + col, t);
  }
}

// totals of one batch group of this slice -> tot[q * 128 + col] (LDS).  The group's rbpg partial rows (256 floats each) are
// read by ALL threads at once with 16-byte sc1 loads (row lane = t / 64 takes rows rl, rl + 16, ...), then the 16 row
// lanes are summed in order: fixed order -> every workgroup of the group gets the same bits.
__device__ __forceinline__ void slice_totals(const float* pgrp, int rbpg, float (*lanes)[2 * SB_COLS], float* tot) {
  const int q4 = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)pgrp, 0, rbpg * 2 * SB_COLS * 4, 0x00020000);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  u32x4 v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {        // rbpg <= 64: at most 4 rows per lane; rows past the end read 0 (buffer range check)
    const int r = rl + 16 * k;
    v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, r < rbpg ? (r * 2 * SB_COLS + q4 * 4) * 4 : 0x7ffffff0, 0, 16);   // sc1
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    acc.x += __uint_as_float(v[k].x); acc.y += __uint_as_float(v[k].y);
    acc.z += __uint_as_float(v[k].z); acc.w += __uint_as_float(v[k].w);
  }
  *reinterpret_cast<float4*>(&lanes[rl][q4 * 4]) = acc;
  __syncthreads();
  if (threadIdx.x < 2 * SB_COLS) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < SB_WAVES; ++k) t += lanes[k][threadIdx.x];
    tot[threadIdx.x] = t;
  }
  __syncthreads();
}

template <int RPT, int NS, bool S16>
__global__ __launch_bounds__(SB_THREADS) void slab_bn_fwd_kernel(SbnArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[SB_WAVES][2][SB_COLS];     // 16 KB: block partials, then the row lanes of phase 2
  __shared__ float tot[2 * SB_COLS];
  float (*lanes)[2 * SB_COLS] = reinterpret_cast<float (*)[2 * SB_COLS]>(&sm[0][0][0]);
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int slice = blockIdx.x, grp = blockIdx.y / a.rbpg, rb = blockIdx.y - grp * a.rbpg;
  const int c = slice * SB_COLS + tx * 8;
  const size_t row0 = (size_t)grp * a.M + (size_t)rb * (RPT * SB_TY) + ty;
  uint4 zp[RPT];
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  constexpr int U = (8 / NS) < RPT ? (8 / NS) : RPT;
#pragma unroll
  for (int k0 = 0; k0 < RPT; k0 += U) {
    SlabPiece<NS, S16> q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) slab_load<NS>(a.slab, (row0 + (size_t)(k0 + u) * SB_TY) * a.C + c, a.slab_stride, q[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u;
      float acc[8];
      slab_sum<NS>(q[u], acc);
      zp[k] = pack8(acc);
      *reinterpret_cast<uint4*>(a.out0 + (row0 + (size_t)k * SB_TY) * a.C + c) = zp[k];
      float v[8];
      unpack8(zp[k], v);                                // statistics of the ROUNDED values (= what is stored)
#pragma unroll
      for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
    }
  }
  float* pslice = a.part + (size_t)slice * a.groups * a.rbpg * 2 * SB_COLS;
  block_partial(sm, s1, s2, pslice + ((size_t)grp * a.rbpg + rb) * 2 * SB_COLS);
  slice_rendezvous(a.sync, slice, (unsigned)(a.groups * a.rbpg));

  // ---- phase 2: statistics of this block's group (every block of the group computes the same values)
  const float m = (float)a.M;
  if (rb == 0 && grp == 0) {
    // the slice's first block publishes mean / invstd of EVERY group and updates the running statistics, groups in
    // order: a double batch is two consecutive forward calls (the reference's D(real) then D(fake))
    const int col = threadIdx.x % SB_COLS;
    float rm = 0.f, rv = 0.f;
    if (a.rmean && threadIdx.x < SB_COLS) { rm = a.rmean[slice * SB_COLS + col]; rv = a.rvar[slice * SB_COLS + col]; }
    for (int g = 0; g < a.groups; ++g) {
      slice_totals(pslice + (size_t)g * a.rbpg * 2 * SB_COLS, a.rbpg, lanes, tot);
      if (threadIdx.x < SB_COLS) {
        const double mud = (double)tot[col] / (double)m;
        const float mu = (float)mud;
        const float var = (float)fmax((double)tot[SB_COLS + col] / (double)m - mud * mud, 0.0);
        a.mean[(size_t)g * a.C + slice * SB_COLS + col] = mu;
        a.invstd[(size_t)g * a.C + slice * SB_COLS + col] = rsqrtf(var + a.eps);
        if (a.rmean) {
          const float unb = var * (m / fmaxf(m - 1.f, 1.f));
          rm = (1.f - a.momentum) * rm + a.momentum * mu;
          rv = (1.f - a.momentum) * rv + a.momentum * unb;
        }
      }
      __syncthreads();
    }
    if (a.rmean && threadIdx.x < SB_COLS) { a.rmean[slice * SB_COLS + col] = rm; a.rvar[slice * SB_COLS + col] = rv; }
    if (a.nbt && slice == 0 && threadIdx.x == 0) *a.nbt += a.groups;
  }
  if (!(rb == 0 && grp == 0 && a.groups == 1))
    slice_totals(pslice + (size_t)grp * a.rbpg * 2 * SB_COLS, a.rbpg, lanes, tot);

  // ---- phase 3: apply from the registers
  float sc[8], sh[8], bet[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const double mud = (double)tot[tx * 8 + i] / (double)m;
    const float mu = (float)mud;
    const float var = (float)fmax((double)tot[SB_COLS + tx * 8 + i] / (double)m - mud * mud, 0.0);
    const float rstd = rsqrtf(var + a.eps);
    sc[i] = rstd * a.gamma[c + i];
    sh[i] = mu;
    bet[i] = a.beta[c + i];
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    float v[8], o[8];
    unpack8(zp[k], v);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = lrelu_f((v[i] - sh[i]) * sc[i] + bet[i], a.slope);      // BnActF's arithmetic
    *reinterpret_cast<uint4*>(a.out1 + (row0 + (size_t)k * SB_TY) * a.C + c) = pack8(o);
  }
}

// MODE 0: BatchNorm + LeakyReLU backward (the slabs are ga, the data gradient arriving at the block).
// MODE 1: forward-mode tangent of the same block (the slabs are zt = conv(tangent of the layer below)): sums of zt and
//         xhat * zt, at = lrelu'(y) * gamma * invstd * (zt - mean(zt) - xhat * mean(xhat * zt))  (rg_bn_tangent's arithmetic);
//         out0 = zt is always written (the penalty's joint reverse pass reads it).
template <int RPT, int NS, int MODE, bool S16>
__global__ __launch_bounds__(SB_THREADS) void slab_bn_bwd_kernel(SbnArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[SB_WAVES][2][SB_COLS];
  __shared__ float tot[2 * SB_COLS];
  float (*lanes)[2 * SB_COLS] = reinterpret_cast<float (*)[2 * SB_COLS]>(&sm[0][0][0]);
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
  const int slice = blockIdx.x, grp = blockIdx.y / a.rbpg, rb = blockIdx.y - grp * a.rbpg;
  const int c = slice * SB_COLS + tx * 8;
  const size_t row0 = (size_t)grp * a.M + (size_t)rb * (RPT * SB_TY) + ty;
  float mu[8], rstd[8], gam[8], bet[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mu[i] = a.mean[(size_t)grp * a.C + c + i]; rstd[i] = a.invstd[(size_t)grp * a.C + c + i];
    gam[i] = a.gamma[c + i]; bet[i] = a.beta[c + i];
  }
  uint4 gp[RPT], zp[RPT];
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  constexpr int U = (8 / NS) < RPT ? (8 / NS) : RPT;
#pragma unroll
  for (int k0 = 0; k0 < RPT; k0 += U) {
    SlabPiece<NS, S16> q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t off = (row0 + (size_t)(k0 + u) * SB_TY) * a.C + c;
      zp[k0 + u] = *reinterpret_cast<const uint4*>(a.zin + off);
      slab_load<NS>(a.slab, off, a.slab_stride, q[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u;
      float acc[8];
      slab_sum<NS>(q[u], acc);
      gp[k] = pack8(acc);
      if (a.out0) *reinterpret_cast<uint4*>(a.out0 + (row0 + (size_t)k * SB_TY) * a.C + c) = gp[k];
      float g[8], v[8];
      unpack8(gp[k], g); unpack8(zp[k], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[i] - mu[i]) * rstd[i];
        if (MODE == 0) {                                    // BwdRedF's arithmetic
          const float gy = g[i] * lrelu_mask(xh * gam[i] + bet[i], a.slope);
          s1[i] += gy; s2[i] += gy * xh;
        } else {                                            // TanRedF's
          s1[i] += g[i]; s2[i] += xh * g[i];
        }
      }
    }
  }
  float* pslice = a.part + (size_t)slice * a.groups * a.rbpg * 2 * SB_COLS;
  block_partial(sm, s1, s2, pslice + ((size_t)grp * a.rbpg + rb) * 2 * SB_COLS);
  slice_rendezvous(a.sync, slice, (unsigned)(a.groups * a.rbpg));

  if (rb == 0 && grp == 0) {
    // sums of every group out; parameter gradients = the groups' sums added in order (BwdFin: later groups accumulate)
    const int col = threadIdx.x % SB_COLS;
    float dg = 0.f, db = 0.f;
    for (int g = 0; g < a.groups; ++g) {
      slice_totals(pslice + (size_t)g * a.rbpg * 2 * SB_COLS, a.rbpg, lanes, tot);
      if (threadIdx.x < SB_COLS) {
        const size_t o = (size_t)g * a.C + slice * SB_COLS + col;
        a.s_gy[o] = tot[col]; a.s_gyxh[o] = tot[SB_COLS + col];
        if (g == 0) { dg = tot[SB_COLS + col]; db = tot[col]; } else { dg += tot[SB_COLS + col]; db += tot[col]; }
      }
      __syncthreads();
    }
    if (MODE == 0 && a.dgamma && threadIdx.x < SB_COLS) {
      const int o = slice * SB_COLS + col;
      if (a.accumulate) { a.dgamma[o] += dg; a.dbeta[o] += db; } else { a.dgamma[o] = dg; a.dbeta[o] = db; }
    }
  }
  if (!(rb == 0 && grp == 0 && a.groups == 1))
    slice_totals(pslice + (size_t)grp * a.rbpg * 2 * SB_COLS, a.rbpg, lanes, tot);

  const float inv_m = 1.f / (float)a.M;
  float m1[8], m2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { m1[i] = tot[tx * 8 + i] * inv_m; m2[i] = tot[SB_COLS + tx * 8 + i] * inv_m; }
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    float g[8], v[8], o[8];
    unpack8(gp[k], g); unpack8(zp[k], v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh = (v[i] - mu[i]) * rstd[i];
      const float mk = lrelu_mask(xh * gam[i] + bet[i], a.slope);
      if (MODE == 0) {                                      // BwdApplyF's arithmetic
        const float gy = g[i] * mk;
        o[i] = (gam[i] * rstd[i]) * (gy - m1[i] - xh * m2[i]);
      } else {                                              // TanApplyF's
        o[i] = (gam[i] * rstd[i]) * (g[i] - m1[i] - xh * m2[i]) * mk;
      }
    }
    *reinterpret_cast<uint4*>(a.out1 + (row0 + (size_t)k * SB_TY) * a.C + c) = pack8(o);
  }
}

struct SbnPlan { int rpt, rbpg, slices; };

// rows per block (64 row lanes x RPT) so that the grid is about one block per CU; false: this shape has no fused form
static bool sbn_plan(int M, int C, int groups, int nsplit, SbnPlan* p) {
  if (C % SB_COLS || M <= 0 || groups < 1 || groups > 2) return false;
  if (nsplit != 2 && nsplit != 4 && nsplit != 8) return false;
  const int slices = C / SB_COLS;
  const long long rows = (long long)M * groups;
  int best = 0;
  for (int rpt = 1; rpt <= 4 && !best; rpt *= 2) {
    const int rpb = rpt * SB_TY;
    if (M % rpb || M / rpb > 64) continue;             // (a group's partial rows are summed 16 lanes x 4 rows)
    if (rpt * nsplit > 8) continue;                    // register budget of a 1024-thread block (128 VGPRs, no spills)
    // co-residency bound of the hand-off: one 1024-thread block per CU (<= 128 VGPRs) always fits; two would too
    if (rows / rpb * slices <= 256) best = rpt;
  }
  if (!best) return false;
  p->rpt = best; p->rbpg = M / (best * SB_TY); p->slices = slices;
  return true;
}

template <int KIND, int RPT>       // KIND 0 forward, 1 backward, 2 tangent
static void sbn_launch_ns(const SbnArgs& a, dim3 grid, hipStream_t st) {
#define SBN_GO(NS)                                                                                          \
  do {                                                                                                      \
    if (KIND == 0) {                                                                                         \
      if (a.slab16) hipLaunchKernelGGL((slab_bn_fwd_kernel<RPT, NS, true>), grid, dim3(SB_THREADS), 0, st, a);  \
      else hipLaunchKernelGGL((slab_bn_fwd_kernel<RPT, NS, false>), grid, dim3(SB_THREADS), 0, st, a);          \
    } else if (a.slab16) {                                                                                   \
      hipLaunchKernelGGL((slab_bn_bwd_kernel<RPT, NS, KIND == 2 ? 1 : 0, true>), grid, dim3(SB_THREADS), 0, st, a);  \
    } else {                                                                                                 \
      hipLaunchKernelGGL((slab_bn_bwd_kernel<RPT, NS, KIND == 2 ? 1 : 0, false>), grid, dim3(SB_THREADS), 0, st, a); \
    }                                                                                                        \
  } while (0)
  if constexpr (RPT == 1) { if (a.nsplit == 2) SBN_GO(2); else if (a.nsplit == 4) SBN_GO(4); else SBN_GO(8); }
  else if constexpr (RPT == 2) { if (a.nsplit == 2) SBN_GO(2); else SBN_GO(4); }
  else SBN_GO(2);
#undef SBN_GO
}

template <int KIND>
static int sbn_launch(const char* name, SbnArgs& a, const SbnPlan& p, hipStream_t st) {
  a.rbpg = p.rbpg;
  dim3 grid((unsigned)p.slices, (unsigned)(a.groups * p.rbpg));
  switch (p.rpt) {
    case 1: sbn_launch_ns<KIND, 1>(a, grid, st); break;
    case 2: sbn_launch_ns<KIND, 2>(a, grid, st); break;
    default: sbn_launch_ns<KIND, 4>(a, grid, st); break;
  }
  RG_LAUNCH_CHECK(name);
  return RG_OK;
}

}  // namespace

// ---- C ABI -----------------------------------------------------------------------------------------------------------
extern "C" int rg_slab_bn_supported(long long M, int C, int groups, int nsplit) {
  SbnPlan p;
  return M < (1ll << 30) && sbn_plan((int)M, C, groups, nsplit, &p) ? 1 : 0;
}
extern "C" size_t rg_slab_bn_scratch_bytes(long long M, int C, int groups) {
  // partial rows: at most (rows / 64) x C x 2 floats; sized for the smallest block
  return (size_t)((M * groups + 63) / 64) * (size_t)C * 2 * sizeof(float);
}
extern "C" size_t rg_slab_bn_sync_words(void) { return 16 + 16 * 64; }

extern "C" int rg_bn_forward_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, void* z, void* a_out, long long M, int C,
                                   int groups, float eps, float momentum, const float* gamma, const float* beta, float slope,
                                   float* mean, float* invstd, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, void* scratch, size_t scratch_bytes, void* sync,
                                   void* stream) {
  RG_REQUIRE(slab && z && a_out && gamma && beta && mean && invstd && scratch && sync, RG_EINVAL, "bn_forward_slabs: null");
  SbnPlan p;
  RG_REQUIRE(M < (1ll << 30) && sbn_plan((int)M, C, groups, nsplit, &p), RG_EUNSUPPORTED,
             "bn_forward_slabs: M=%lld C=%d groups=%d nsplit=%d has no fused form (rg_slab_bn_supported)", M, C, groups, nsplit);
  RG_REQUIRE(p.slices <= 64, RG_EUNSUPPORTED, "bn_forward_slabs: C > 8192");
  RG_REQUIRE(scratch_bytes >= (size_t)p.slices * groups * p.rbpg * 2 * SB_COLS * sizeof(float), RG_EWORKSPACE,
             "bn_forward_slabs: scratch too small");
  SbnArgs a{};
  a.slab = (const float*)slab; a.slab_stride = slab_stride; a.nsplit = nsplit; a.slab16 = slab_dtype == RG_H16 ? 1 : 0;
  a.out0 = (uint16_t*)z; a.out1 = (uint16_t*)a_out; a.M = (int)M; a.C = C; a.groups = groups;
  a.gamma = gamma; a.beta = beta; a.slope = slope; a.eps = eps; a.momentum = momentum;
  a.mean = mean; a.invstd = invstd; a.rmean = running_mean; a.rvar = running_var; a.nbt = num_batches_tracked;
  a.part = (float*)scratch; a.sync = (unsigned*)sync;
  return sbn_launch<0>("bn_forward_slabs", a, p, rg_stream(stream));
}

extern "C" int rg_bn_act_bwd_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, const void* z, void* ga_out, void* gz,
                                   long long M, int C, int groups, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float slope, float* s_gy, float* s_gyxh, float* dgamma, float* dbeta,
                                   int accumulate, void* scratch, size_t scratch_bytes, void* sync, void* stream) {
  RG_REQUIRE(slab && z && gz && mean && invstd && gamma && beta && s_gy && s_gyxh && scratch && sync, RG_EINVAL,
             "bn_act_bwd_slabs: null");
  RG_REQUIRE((dgamma == nullptr) == (dbeta == nullptr), RG_EINVAL, "bn_act_bwd_slabs: dgamma and dbeta go together");
  SbnPlan p;
  RG_REQUIRE(M < (1ll << 30) && sbn_plan((int)M, C, groups, nsplit, &p), RG_EUNSUPPORTED,
             "bn_act_bwd_slabs: M=%lld C=%d groups=%d nsplit=%d has no fused form (rg_slab_bn_supported)", M, C, groups, nsplit);
  RG_REQUIRE(p.slices <= 64, RG_EUNSUPPORTED, "bn_act_bwd_slabs: C > 8192");
  RG_REQUIRE(scratch_bytes >= (size_t)p.slices * groups * p.rbpg * 2 * SB_COLS * sizeof(float), RG_EWORKSPACE,
             "bn_act_bwd_slabs: scratch too small");
  SbnArgs a{};
  a.slab = (const float*)slab; a.slab_stride = slab_stride; a.nsplit = nsplit; a.slab16 = slab_dtype == RG_H16 ? 1 : 0;
  a.zin = (const uint16_t*)z; a.out0 = (uint16_t*)ga_out; a.out1 = (uint16_t*)gz; a.M = (int)M; a.C = C; a.groups = groups;
  a.gamma = gamma; a.beta = beta; a.slope = slope;
  a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd);
  a.s_gy = s_gy; a.s_gyxh = s_gyxh; a.dgamma = dgamma; a.dbeta = dbeta; a.accumulate = accumulate;
  a.part = (float*)scratch; a.sync = (unsigned*)sync;
  return sbn_launch<1>("bn_act_bwd_slabs", a, p, rg_stream(stream));
}

extern "C" int rg_bn_tangent_slabs(const void* slab, int nsplit, size_t slab_stride, int slab_dtype, const void* z, void* zt_out, void* at,
                                   long long M, int C, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float slope, float* s_zt, float* s_xhzt, void* scratch,
                                   size_t scratch_bytes, void* sync, void* stream) {
  RG_REQUIRE(slab && z && zt_out && at && mean && invstd && gamma && beta && s_zt && s_xhzt && scratch && sync, RG_EINVAL,
             "bn_tangent_slabs: null");
  SbnPlan p;
  RG_REQUIRE(M < (1ll << 30) && sbn_plan((int)M, C, 1, nsplit, &p), RG_EUNSUPPORTED,
             "bn_tangent_slabs: M=%lld C=%d nsplit=%d has no fused form (rg_slab_bn_supported)", M, C, nsplit);
  RG_REQUIRE(p.slices <= 64, RG_EUNSUPPORTED, "bn_tangent_slabs: C > 8192");
  RG_REQUIRE(scratch_bytes >= (size_t)p.slices * p.rbpg * 2 * SB_COLS * sizeof(float), RG_EWORKSPACE,
             "bn_tangent_slabs: scratch too small");
  SbnArgs a{};
  a.slab = (const float*)slab; a.slab_stride = slab_stride; a.nsplit = nsplit; a.slab16 = slab_dtype == RG_H16 ? 1 : 0;
  a.zin = (const uint16_t*)z; a.out0 = (uint16_t*)zt_out; a.out1 = (uint16_t*)at; a.M = (int)M; a.C = C; a.groups = 1;
  a.gamma = gamma; a.beta = beta; a.slope = slope;
  a.mean = const_cast<float*>(mean); a.invstd = const_cast<float*>(invstd);
  a.s_gy = s_zt; a.s_gyxh = s_xhzt;
  a.part = (float*)scratch; a.sync = (unsigned*)sync;
  return sbn_launch<2>("bn_tangent_slabs", a, p, rg_stream(stream));
}
