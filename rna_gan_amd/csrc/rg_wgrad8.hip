// rg_wgrad8.hip -- weight gradient of the stride-2 4x4 conv layers on the 8-wave ping-pong pipeline of rg_conv8.hip.
//
//   dW[o][tap][i] = sum_p low[p][o] * high[src(p, tap)][i]        (p = output pixel, src = its input pixel under the tap)
//
// GEMM view: C[O][16*I] = L^T H with the contraction index (pixels) as the SLOW axis of both NHWC operands, so both are
// staged as [pixel][channel] LDS images (256-byte rows, 16-byte chunks XOR-swizzled with f(row) = ((row&3)<<2) |
// ((row>>2)&3)) and read with ds_read_b64_tr_b16 (hardware transpose) into k-contiguous MFMA fragments -- the layout of
// wgrad_dma_kernel (rg_mfma.hip), whose lane maps rg_selftest_layouts checks on the device.
//
// Pipeline = conv8_kernel's: block tile 256 (o) x 256 (tap, i) with 8 waves of 128 x 64, four 16 KB half-tiles per
// 64-pixel k-tile (A0/A1 = the two 128-channel halves of `low`, B0/B1 = two 128-column halves of the gathered `high`),
// one half-tile DMA'd per phase 6 phases ahead of its first read, counted vmcnt, waves 4-7 one barrier behind waves 0-3.
// Two (low, high) segments are summed in one launch (D step: real + fake batch; GP step: primal + tangent).  Split-K
// over pixels into fp32 slabs (fixed-order reduction by rg_reduce_slabs) or, with one split, straight into dW.
#include "rg_gather.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int V> using ic = std::integral_constant<int, V>;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct W8Args {
  const uint16_t* low[2];
  const uint16_t* high[2];
  unsigned low_bytes, high_bytes;
  int Kseg[2];           // pixels per segment (Kseg[1] = 0: one segment); Kseg[0] % 64 == 0 when there are two
  float* out;            // nsplit > 1: slab [nsplit][O][16*I]; nsplit == 1: dW itself
  int O, I;
  int lgWo, lgHo, Hh, Wh;
  int tiles_c, tiles_o, nsplit;
  int kt_per_split;      // k-tiles (64 pixels) per split, even
  int accumulate;        // nsplit == 1 only: add to dW
  // ADAM form (nsplit == 1 only): the tile is not written -- the epilogue applies the optimizer step to the tensor's master,
  // moments and bf16 operand image, which share dW's tap-major layout [O][16][I] (rg_conv_wgrad_adam)
  float* ap; float* am; float* av; uint16_t* ash; const float* hyper;
  int slab16;            // nsplit > 1 only: the partial tiles are stored as bf16 [nsplit][O][16*I] (each partial sum rounded once;
                         // the consumer -- rg_adam_step_slabs -- adds them in fp32): half the slab bytes written and re-read
  unsigned low_plane, high_plane;   // wgrad8_kernel<.., NP > 0> (rg_conv8f.hip): bytes between the bf16 planes of low / of high
};

// the epilogue's store loop for bf16 slabs: `rows` rows of the 128-column fp32 LDS image go to `tile` (this slab's element
// [o0][first column of the piece]), 16 threads x 8 columns per row and 32 rows per pass: 16-byte stores, 256 contiguous bytes per row
__device__ __forceinline__ void w8_store_slab16(const float* cs, uint16_t* tile, long long ldw, int rows, int t) {
  const int c8 = (t & 15) * 8, r32 = t >> 4;
#pragma unroll 4
  for (int p = 0; p < rows / 32; ++p) {
    const int row = r32 + 32 * p;
    const float4 a = *reinterpret_cast<const float4*>(cs + row * 128 + c8);
    const float4 b = *reinterpret_cast<const float4*>(cs + row * 128 + c8 + 4);
    uint4 o;
    o.x = (uint32_t)f32_to_h16(a.x) | ((uint32_t)f32_to_h16(a.y) << 16);
    o.y = (uint32_t)f32_to_h16(a.z) | ((uint32_t)f32_to_h16(a.w) << 16);
    o.z = (uint32_t)f32_to_h16(b.x) | ((uint32_t)f32_to_h16(b.y) << 16);
    o.w = (uint32_t)f32_to_h16(b.z) | ((uint32_t)f32_to_h16(b.w) << 16);
    *reinterpret_cast<uint4*>(tile + (long long)row * ldw + c8) = o;
  }
}

__device__ __forceinline__ int w8_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// NP > 0 (rg_conv8f.hip only): fp32 operands as K-concatenated bf16 planes, exactly as conv8_kernel's NP (rg_conv8.hip): low and
// high are plane-major [3][pixels][channels] buffers, flat k-tile q -> plane pair q % NP (hh hm mh hl lh mm), pixel tile q / NP.
// MF: 32 = v_mfma_f32_32x32x16 (16 pixels per k-step, 32-channel operand tiles), 16 = v_mfma_f32_16x16x32 (32 pixels per k-step,
// 16-channel tiles: the four 16-lane groups of a transposed read take the four pixel octets of the SAME 16 channels).  Same LDS
// images, DMA stream, waits and accumulator footprint; chosen by option wgrad8_mfma.
template <bool ADAM, int NP = 0, int MF = 32>
__global__ __launch_bounds__(512, 2) void wgrad8_kernel(W8Args g) {
  constexpr int NAT = MF == 32 ? 2 : 4;      // A (low channel) tiles per quadrant row (64 channels)
  constexpr int NBT = MF == 32 ? 1 : 2;      // B (column) tiles per quadrant column (32 columns)
  constexpr int NKS = MF == 32 ? 4 : 2;      // k-steps per 64-pixel k-tile
  constexpr int KSB = MF == 32 ? 4096 : 8192;   // LDS bytes per k-step (16 / 32 pixel rows of 256 bytes)
  constexpr int HT = 64 * 256;                     // bytes per half-tile: 64 pixels x 128 channels bf16
  constexpr int STAGE = 4 * HT;                    // [B0][B1][A0][A1]
  constexpr int OFF_B = 0, OFF_A = 2 * HT;
  constexpr int LDS_BYTES = 2 * STAGE;             // 128 KB
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_BYTES / 16];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  // XCD-aware work order (blocks b and b + 8 share an L2): each XCD walks a contiguous range of work items ordered
  // split-major, so the column tiles that read the SAME pixels run back to back on one L2
  const int ntiles = g.tiles_o * g.tiles_c;
  const int total = ntiles * g.nsplit;
  int wid = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = wid & 7, j = wid >> 3;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int zs = wid / ntiles;
  const int tid = wid - zs * ntiles;
  const int tile_o = tid / g.tiles_c, tile_c = tid - tile_o * g.tiles_c;
  const int o0 = tile_o * 256, c0 = tile_c * 256;
  const int Ktot = g.Kseg[0] + g.Kseg[1];
  const int nkt_all = ((Ktot + 63) >> 6) * (NP > 0 ? NP : 1);
  const int kt_begin = zs * g.kt_per_split;
  const int nkt = min(nkt_all, kt_begin + g.kt_per_split) - kt_begin;     // host: > 0 and even
  constexpr unsigned OOB = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rsL0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[0], 0, g.low_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[0], 0, g.high_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[1], 0, g.low_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[1], 0, g.high_bytes, 0x00020000);

  // ---- DMA lane assignment: one block-wide instruction covers 32 pixel rows x 256 B; this lane owns the 16-byte
  // physical chunk pc of rows lrow and lrow + 32, and fetches logical chunk lc = pc ^ f(row) (f is the same for both rows)
  const int lrow = wave * 4 + (lane >> 4);
  const int lc = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (wave & 3));
  const int Wo = 1 << g.lgWo, Ho = 1 << g.lgHo;
  int a_off[2];                       // byte offset of (row lrow, channel chunk) inside `low` for A half h; + 32 rows = + 64*O bytes
  int b_ci[2], b_kh[2], b_kw[2];      // B half h: channel offset and tap of this lane's 8 columns
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    a_off[h] = (lrow * g.O + o0 + h * 128 + lc * 8) * 2;
    const int col = c0 + h * 128 + lc * 8;
    const int tap = col / g.I;
    b_ci[h] = (col - tap * g.I) * 2;
    b_kh[h] = tap >> 2;
    b_kw[h] = tap & 3;
  }
  char* const ldsb = reinterpret_cast<char*>(lds);

  // pixel geometry of this lane's two rows of k-tile ktr (decoded when B0 of that k-tile is issued, reused for B1)
  int px_base[2], px_h[2], px_w[2];
  bool px_ok[2];
  int px_plane = 0;                   // NP > 0: byte offset of the `high` plane of the k-tile decoded last
  // flat k-tile -> (first pixel, byte offsets of the low / high planes it multiplies)
  auto flat_tile = [&](int ktr, int& lo_off, int& hi_off) -> int {
    int q = kt_begin + ktr;
    lo_off = 0; hi_off = 0;
    if constexpr (NP > 0) {
      const int pp = q % NP;
      q = q / NP;
      lo_off = ((0x120100 >> (4 * pp)) & 15) * (int)g.low_plane;
      hi_off = ((0x102010 >> (4 * pp)) & 15) * (int)g.high_plane;
    }
    return q * 64;
  };
  auto decode_pixels = [&](int ktr) {
    int lo_off;
    const int p0 = flat_tile(ktr, lo_off, px_plane);
    const bool seg1 = p0 >= g.Kseg[0];
    const int pb = seg1 ? p0 - g.Kseg[0] : p0;
    const int kend = (ktr < nkt) ? (seg1 ? g.Kseg[1] : g.Kseg[0]) : 0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int p = pb + jj * 32 + lrow;
      px_ok[jj] = p < kend;
      const int wo = p & (Wo - 1), ho = (p >> g.lgWo) & (Ho - 1), n = p >> (g.lgWo + g.lgHo);
      px_h[jj] = 2 * ho - 1;
      px_w[jj] = 2 * wo - 1;
      px_base[jj] = n * g.Hh;
    }
  };
  auto issue_a = [&](auto S, auto H, int ktr) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    int lo_off, hi_off;
    const int p0 = flat_tile(ktr, lo_off, hi_off);
    const bool seg1 = p0 >= g.Kseg[0];
    const int pb = seg1 ? p0 - g.Kseg[0] : p0;
    const int kend = (ktr < nkt) ? (seg1 ? g.Kseg[1] : g.Kseg[0]) : 0;
    const __amdgpu_buffer_rsrc_t rs = seg1 ? rsL1 : rsL0;
    const int so = pb * g.O * 2 + lo_off;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const bool ok = pb + jj * 32 + lrow < kend;
      const unsigned vo = ok ? (unsigned)(a_off[h] + jj * 64 * g.O + so) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(ldsb + s * STAGE + OFF_A + h * HT + jj * 8192 + wave * 1024),
                                               16, vo, 0, 0, 0);
    }
  };
  auto issue_b = [&](auto S, auto H, int ktr) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    if (h == 0) decode_pixels(ktr);
    int lo_off, hi_off;
    const bool seg1 = flat_tile(ktr, lo_off, hi_off) >= g.Kseg[0];
    const __amdgpu_buffer_rsrc_t rs = seg1 ? rsH1 : rsH0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int hi = px_h[jj] + b_kh[h], wi = px_w[jj] + b_kw[h];
      const bool v = px_ok[jj] && (unsigned)hi < (unsigned)g.Hh && (unsigned)wi < (unsigned)g.Wh;
      const unsigned vo = v ? (unsigned)(((px_base[jj] + hi) * g.Wh + wi) * g.I * 2 + b_ci[h] + px_plane) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(ldsb + s * STAGE + OFF_B + h * HT + jj * 8192 + wave * 1024),
                                               16, vo, 0, 0, 0);
    }
  };

  // ---- fragment reads: 32 channels x 16 pixels per MFMA operand = two transposed 64-bit reads (pixels 8fh+q .. +3
  // and +4) of this lane's 4-channel column group; channel block cb32 of the half-tile enters the swizzled chunk index
  // by XOR, so every (block, lo/hi) pair has its own address register; k-step (16 pixels) and half-tile are immediates
  const int wm = wave >> 2, wn = wave & 3;
  const int grp = lane >> 4, idx = lane & 15;
  const int q = idx >> 2, p4 = idx & 3, fh = grp >> 1, cb = grp & 1;
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  // transposed-read address of operand tile `tile` (32 channels: MF 32; 16 channels: MF 16) of a 128-channel half-tile, pixel
  // rows +0..3 (hi = 0) / +4..7 (hi = 1) of this lane group's pixel octet: 8 fh (two octets per 16-pixel k-step, two channel
  // blocks cb per tile) or 8 grp (four octets per 32-pixel k-step).  Swizzle f(row) = ((row & 3) << 2) | ((row >> 2) & 3).
  auto frag_addr = [&](int tile, int hi) -> unsigned {
    const int oct = MF == 32 ? fh : grp;
    const int prow = 8 * oct + q + 4 * hi;
    const int ca = MF == 32 ? tile * 4 + 2 * cb + (p4 >> 1) : tile * 2 + (p4 >> 1);
    const int swz = (q << 2) | ((2 * oct + hi) & 3);
    return (unsigned)(prow * 256 + ((ca ^ swz) << 4) + (p4 & 1) * 8);
  };
  unsigned aA[2][NAT][2], bA[2][NBT][2];            // [stage][tile][lo/hi]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
      for (int ti = 0; ti < NAT; ++ti) aA[s][ti][hl] = lds_base + s * STAGE + OFF_A + frag_addr(wm * NAT + ti, hl);
#pragma unroll
      for (int tj = 0; tj < NBT; ++tj) bA[s][tj][hl] = lds_base + s * STAGE + OFF_B + frag_addr(wn * NBT + tj, hl);
    }
  using acc_t = std::conditional_t<MF == 32, f32x16_t, rg_f32x4>;
  constexpr int NACC = NAT * NBT, ACC_R = MF == 32 ? 16 : 4;
  acc_t acc[2][2][NACC];                            // [quadrant row i][quadrant column j][A tile * NBT + B tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < NACC; ++s)
#pragma unroll
        for (int r = 0; r < ACC_R; ++r) acc[i][j][s][r] = 0.f;
  u32x2_t aR[16];                                   // [(A tile * NKS + k-step) * 2 + lo/hi]
  u32x2_t bS[3][8];                                 // three rotating B sets, [(B tile * NKS + k-step) * 2 + lo/hi]

#define W8_DSR(dst, addr, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define W8_READ_A(S, H)                                                                                   \
  do {                                                                                                    \
    _Pragma("unroll") for (int ks_ = 0; ks_ < NKS; ++ks_) _Pragma("unroll") for (int ti_ = 0; ti_ < NAT; ++ti_) { \
      W8_DSR(aR[(ti_ * NKS + ks_) * 2], aA[S][ti_][0], (H) * HT + ks_ * KSB);                             \
      W8_DSR(aR[(ti_ * NKS + ks_) * 2 + 1], aA[S][ti_][1], (H) * HT + ks_ * KSB);                         \
    }                                                                                                     \
  } while (0)
#define W8_READ_B(S, H, SET)                                                                              \
  do {                                                                                                    \
    _Pragma("unroll") for (int ks_ = 0; ks_ < NKS; ++ks_) _Pragma("unroll") for (int tj_ = 0; tj_ < NBT; ++tj_) { \
      W8_DSR(bS[SET][(tj_ * NKS + ks_) * 2], bA[S][tj_][0], (H) * HT + ks_ * KSB);                        \
      W8_DSR(bS[SET][(tj_ * NKS + ks_) * 2 + 1], bA[S][tj_][1], (H) * HT + ks_ * KSB);                    \
    }                                                                                                     \
  } while (0)
#define W8_WAIT_A()                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
               : "+v"(aR[0]), "+v"(aR[1]), "+v"(aR[2]), "+v"(aR[3]), "+v"(aR[4]), "+v"(aR[5]), "+v"(aR[6]), "+v"(aR[7]), \
                 "+v"(aR[8]), "+v"(aR[9]), "+v"(aR[10]), "+v"(aR[11]), "+v"(aR[12]), "+v"(aR[13]), "+v"(aR[14]),        \
                 "+v"(aR[15])::"memory")
#define W8_WAIT_B(SET)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
               : "+v"(bS[SET][0]), "+v"(bS[SET][1]), "+v"(bS[SET][2]), "+v"(bS[SET][3]), "+v"(bS[SET][4]), \
                 "+v"(bS[SET][5]), "+v"(bS[SET][6]), "+v"(bS[SET][7])::"memory")
#define W8_FRAG(lo, hi) __builtin_bit_cast(h16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))
#define W8_MFMAS(I, J, SET)                                                                                \
  do {                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int ks_ = 0; ks_ < NKS; ++ks_) _Pragma("unroll") for (int ti_ = 0; ti_ < NAT; ++ti_) \
    _Pragma("unroll") for (int tj_ = 0; tj_ < NBT; ++tj_) {                                                \
      const h16x8_t fa_ = W8_FRAG(aR[(ti_ * NKS + ks_) * 2], aR[(ti_ * NKS + ks_) * 2 + 1]);               \
      const h16x8_t fb_ = W8_FRAG(bS[SET][(tj_ * NKS + ks_) * 2], bS[SET][(tj_ * NKS + ks_) * 2 + 1]);     \
      if constexpr (MF == 32) acc[I][J][ti_ * NBT + tj_] = rg_mfma_h16_32x32x16(fa_, fb_, acc[I][J][ti_ * NBT + tj_], 0, 0, 0); \
      else acc[I][J][ti_ * NBT + tj_] = rg_mfma_h16_16x16x32(fa_, fb_, acc[I][J][ti_ * NBT + tj_], 0, 0, 0); \
    }                                                                                                      \
    __builtin_amdgcn_s_setprio(0);                                                                         \
  } while (0)
#define W8_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  constexpr int W_ALL = vmcnt_imm(10);              // 5 half-tiles x 2 DMA instructions stay in flight behind the one needed
  // k-tile u in stage S; B0[u] in set SB0, B1[u] -> set 1, B0[u+1] -> set SNX   (schedule: see rg_conv8.hip)
#define W8_TILE(S, SB0, SNX, U)                                                             \
  do {                                                                                      \
    W8_READ_A(S, 0);                                                                        \
    issue_a(ic<1 - (S)>{}, ic<1>{}, (U) + 1);                                               \
    __builtin_amdgcn_s_waitcnt(W_ALL);                                                      \
    W8_SYNC();                                                                              \
    W8_WAIT_A();                                                                            \
    W8_MFMAS(0, 0, SB0);                                                                    \
    W8_SYNC();                                                                              \
    W8_READ_B(S, 1, 1);                                                                     \
    issue_b(ic<(S)>{}, ic<0>{}, (U) + 2);                                                   \
    __builtin_amdgcn_s_waitcnt(W_ALL);                                                      \
    W8_SYNC();                                                                              \
    W8_WAIT_B(1);                                                                           \
    W8_MFMAS(0, 1, 1);                                                                      \
    W8_SYNC();                                                                              \
    W8_READ_A(S, 1);                                                                        \
    issue_a(ic<(S)>{}, ic<0>{}, (U) + 2);                                                   \
    __builtin_amdgcn_s_waitcnt(W_ALL);                                                      \
    W8_SYNC();                                                                              \
    W8_WAIT_A();                                                                            \
    W8_MFMAS(1, 1, 1);                                                                      \
    W8_SYNC();                                                                              \
    W8_READ_B(1 - (S), 0, SNX);                                                             \
    issue_b(ic<(S)>{}, ic<1>{}, (U) + 2);                                                   \
    __builtin_amdgcn_s_waitcnt(W_ALL);                                                      \
    W8_SYNC();                                                                              \
    W8_WAIT_B(SNX);                                                                         \
    W8_MFMAS(1, 0, SB0);                                                                    \
    W8_SYNC();                                                                              \
  } while (0)

  // prologue (issue order B0 A0 B1 A1 per k-tile; B1 reuses the pixel geometry decoded for B0 of the same k-tile)
  issue_b(ic<0>{}, ic<0>{}, 0);
  issue_a(ic<0>{}, ic<0>{}, 0);
  issue_b(ic<0>{}, ic<1>{}, 0);
  issue_a(ic<0>{}, ic<1>{}, 0);
  issue_b(ic<1>{}, ic<0>{}, 1);
  issue_a(ic<1>{}, ic<0>{}, 1);
  issue_b(ic<1>{}, ic<1>{}, 1);
  __builtin_amdgcn_s_waitcnt(W_ALL);
  W8_SYNC();
  W8_READ_B(0, 0, 0);
  W8_WAIT_B(0);
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  for (int u = 0; u < nkt; u += 2) {
    W8_TILE(0, 0, 2, u);
    W8_TILE(1, 2, 0, u + 1);
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#undef W8_TILE
#undef W8_SYNC
#undef W8_MFMAS
#undef W8_FRAG
#undef W8_WAIT_A
#undef W8_WAIT_B
#undef W8_READ_A
#undef W8_READ_B
#undef W8_DSR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: fp32 [256 o][128 cols] per quadrant column through LDS, 16-byte stores (512 contiguous bytes per row)
  float* cs = reinterpret_cast<float*>(lds);
  const int fr = lane & 31, fh2 = lane >> 5;
  const long long ldw = (long long)16 * g.I;
  float* outp = g.out + (g.nsplit > 1 ? (long long)zs * g.O * ldw : 0);
  const int c4 = (t & 31) * 4, rr = t >> 5;         // 32 threads per row (128 floats), 16 rows per pass
#pragma unroll
  for (int ep = 0; ep < 2; ++ep) {
    if (ep) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ti = 0; ti < NAT; ++ti)
#pragma unroll
        for (int tj = 0; tj < NBT; ++tj)
#pragma unroll
          for (int r = 0; r < ACC_R; ++r) {
            // accumulator maps: 32x32 -> row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31; 16x16 -> row 4*(lane>>4) + r, column lane&15
            const int row = MF == 32 ? i * 128 + wm * 64 + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh2
                                     : i * 128 + wm * 64 + ti * 16 + 4 * (lane >> 4) + r;
            const int col = MF == 32 ? wn * 32 + fr : wn * 32 + tj * 16 + (lane & 15);
            cs[row * 128 + col] = acc[i][ep][ti * NBT + tj][r];
          }
    __syncthreads();
    if constexpr (ADAM) {
      // the optimizer step of this 256 x 128 piece: 26 bytes per element (p, m, v read and written, the bf16 image written)
      // instead of the 4 written here + 30 of the streaming Adam.  Four rows per trip with all 12 loads issued first.
      RgAdamHyper hy;
      hy.load(g.hyper);
#pragma unroll 1
      for (int pq = 0; pq < 4; ++pq) {
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        f32x4_nt P[4], M[4], V[4];
        long long off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = rr + 16 * (4 * pq + k);
          off[k] = (long long)(o0 + row) * ldw + c0 + ep * 128 + c4;
          P[k] = *reinterpret_cast<const f32x4_nt*>(g.ap + off[k]);
          M[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(g.am + off[k]));
          V[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(g.av + off[k]));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = rr + 16 * (4 * pq + k);
          const float4 gv = *reinterpret_cast<const float4*>(cs + row * 128 + c4);
          float pe[4] = {P[k].x, P[k].y, P[k].z, P[k].w}, me[4] = {M[k].x, M[k].y, M[k].z, M[k].w};
          float ve[4] = {V[k].x, V[k].y, V[k].z, V[k].w};
          const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) hy.upd(pe[e], ge[e], me[e], ve[e]);
          *reinterpret_cast<f32x4_nt*>(g.ap + off[k]) = f32x4_nt{pe[0], pe[1], pe[2], pe[3]};
          __builtin_nontemporal_store(f32x4_nt{me[0], me[1], me[2], me[3]}, reinterpret_cast<f32x4_nt*>(g.am + off[k]));
          __builtin_nontemporal_store(f32x4_nt{ve[0], ve[1], ve[2], ve[3]}, reinterpret_cast<f32x4_nt*>(g.av + off[k]));
          if (g.ash)
            *reinterpret_cast<uint2*>(g.ash + off[k]) =
                make_uint2((uint32_t)f32_to_h16(pe[0]) | ((uint32_t)f32_to_h16(pe[1]) << 16),
                           (uint32_t)f32_to_h16(pe[2]) | ((uint32_t)f32_to_h16(pe[3]) << 16));
        }
      }
    } else if (g.slab16) {
      w8_store_slab16(cs, reinterpret_cast<uint16_t*>(g.out) + ((long long)zs * g.O + o0) * ldw + c0 + ep * 128, ldw, 256, t);
    } else {
#pragma unroll 4
      for (int p = 0; p < 16; ++p) {
        const int row = rr + 16 * p;
        float4* d = reinterpret_cast<float4*>(outp + (long long)(o0 + row) * ldw + c0 + ep * 128 + c4);
        float4 v = *reinterpret_cast<const float4*>(cs + row * 128 + c4);
        if (g.accumulate) {
          const float4 a = *d;
          v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        *d = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The same pipeline for the layers with 128 low-side channels (O = 128: D.1 and the generator's 128 -> 64 block), which the
// 256 x 256 tile cannot cover: block tile 128 (o) x 512 (tap, i) -- 8 waves of 64 (o) x 4 x 32 columns, FIVE 16 KB half-tiles
// per 64-pixel k-tile (A0 = the 128 channels of `low`, B0..B3 = four 128-column slices of the gathered `high`), two stages =
// all 160 KB of LDS.  105 FLOP per operand byte against 64 of the 128 x 128 tile these layers ran on (wgrad_dma_kernel).
// Phases of k-tile u (stage S), one MFMA group of 8 per phase and wave; every DMA is issued TWO phases after the last read of
// the slot it overwrites (as in wgrad8_kernel: the other wave group is one barrier behind and may still have that read in
// flight one phase later), six phases ahead of its first read:
//   P0: read A0(u)            issue B2(u+1)            MFMA A0 x B0(u)        (B0(u) was read in P3 of k-tile u - 1)
//   P1: read B1(u)            issue B3(u+1), B0(u+2)   MFMA A0 x B1
//   P2: read B2(u)            issue A0(u+2)            MFMA A0 x B2
//   P3: read B3(u), B0(u+1)   issue B1(u+2)            MFMA A0 x B3
// Issue order per k-tile is B0 A0 B1 B2 B3; behind the half-tile the NEXT phase reads there are six younger ones (P1: seven),
// so vmcnt(12) is the wait of every phase (checked by simulation of the issue / read sequence).
template <int NP = 0>      // NP > 0: K-concatenated bf16 planes of fp32 operands, as wgrad8_kernel's NP
__global__ __launch_bounds__(512, 2) void wgrad8n_kernel(W8Args g) {
  constexpr int HT = 64 * 256;                     // bytes per half-tile: 64 pixels x 128 channels bf16
  constexpr int STAGE = 5 * HT;                    // [B0][B1][B2][B3][A0]
  constexpr int OFF_B = 0, OFF_A = 4 * HT;
  constexpr int LDS_BYTES = 2 * STAGE;             // 160 KB
  __shared__ __attribute__((aligned(16))) uint4 lds[LDS_BYTES / 16];

  const int t = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  const int ntiles = g.tiles_o * g.tiles_c;
  const int total = ntiles * g.nsplit;
  int wid = blockIdx.x;
  {
    const int q = total >> 3, r = total & 7, xcd = wid & 7, j = wid >> 3;
    wid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
  }
  const int zs = wid / ntiles;
  const int tid = wid - zs * ntiles;
  const int tile_o = tid / g.tiles_c, tile_c = tid - tile_o * g.tiles_c;
  const int o0 = tile_o * 128, c0 = tile_c * 512;
  const int Ktot = g.Kseg[0] + g.Kseg[1];
  const int nkt_all = ((Ktot + 63) >> 6) * (NP > 0 ? NP : 1);
  const int kt_begin = zs * g.kt_per_split;
  const int nkt = min(nkt_all, kt_begin + g.kt_per_split) - kt_begin;     // host: > 0 and even
  constexpr unsigned OOB = 0x80000000u;

  const __amdgpu_buffer_rsrc_t rsL0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[0], 0, g.low_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH0 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[0], 0, g.high_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsL1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.low[1], 0, g.low_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsH1 = __builtin_amdgcn_make_buffer_rsrc((void*)g.high[1], 0, g.high_bytes, 0x00020000);

  // DMA lane assignment of a half-tile: as in wgrad8_kernel
  const int lrow = wave * 4 + (lane >> 4);
  const int lc = (lane & 15) ^ ((((lane >> 4) & 3) << 2) | (wave & 3));
  const int Wo = 1 << g.lgWo, Ho = 1 << g.lgHo;
  const int a_off = (lrow * g.O + o0 + lc * 8) * 2;
  int b_ci[4], b_kh[4], b_kw[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    const int col = c0 + h * 128 + lc * 8;
    const int tap = col / g.I;
    b_ci[h] = (col - tap * g.I) * 2;
    b_kh[h] = tap >> 2;
    b_kw[h] = tap & 3;
  }
  char* const ldsb = reinterpret_cast<char*>(lds);

  int px_base[2], px_h[2], px_w[2];
  bool px_ok[2];
  int px_plane = 0;
  auto flat_tile = [&](int ktr, int& lo_off, int& hi_off) -> int {
    int q = kt_begin + ktr;
    lo_off = 0; hi_off = 0;
    if constexpr (NP > 0) {
      const int pp = q % NP;
      q = q / NP;
      lo_off = ((0x120100 >> (4 * pp)) & 15) * (int)g.low_plane;
      hi_off = ((0x102010 >> (4 * pp)) & 15) * (int)g.high_plane;
    }
    return q * 64;
  };
  auto decode_pixels = [&](int ktr) {
    int lo_off;
    const int p0 = flat_tile(ktr, lo_off, px_plane);
    const bool seg1 = p0 >= g.Kseg[0];
    const int pb = seg1 ? p0 - g.Kseg[0] : p0;
    const int kend = (ktr < nkt) ? (seg1 ? g.Kseg[1] : g.Kseg[0]) : 0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int p = pb + jj * 32 + lrow;
      px_ok[jj] = p < kend;
      const int wo = p & (Wo - 1), ho = (p >> g.lgWo) & (Ho - 1), n = p >> (g.lgWo + g.lgHo);
      px_h[jj] = 2 * ho - 1;
      px_w[jj] = 2 * wo - 1;
      px_base[jj] = n * g.Hh;
    }
  };
  auto issue_a = [&](auto S, int ktr) {
    constexpr int s = decltype(S)::value;
    int lo_off, hi_off;
    const int p0 = flat_tile(ktr, lo_off, hi_off);
    const bool seg1 = p0 >= g.Kseg[0];
    const int pb = seg1 ? p0 - g.Kseg[0] : p0;
    const int kend = (ktr < nkt) ? (seg1 ? g.Kseg[1] : g.Kseg[0]) : 0;
    const __amdgpu_buffer_rsrc_t rs = seg1 ? rsL1 : rsL0;
    const int so = pb * g.O * 2 + lo_off;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const bool ok = pb + jj * 32 + lrow < kend;
      const unsigned vo = ok ? (unsigned)(a_off + jj * 64 * g.O + so) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(ldsb + s * STAGE + OFF_A + jj * 8192 + wave * 1024), 16, vo, 0, 0, 0);
    }
  };
  // B slice h of k-tile ktr; h == 0 decodes the pixel geometry of that k-tile, slices 1..3 of the SAME k-tile reuse it
  // (B3(u+1) is issued before B0(u+2) in P1)
  auto issue_b = [&](auto S, auto H, int ktr) {
    constexpr int s = decltype(S)::value, h = decltype(H)::value;
    if (h == 0) decode_pixels(ktr);
    int lo_off, hi_off;
    const bool seg1 = flat_tile(ktr, lo_off, hi_off) >= g.Kseg[0];
    const __amdgpu_buffer_rsrc_t rs = seg1 ? rsH1 : rsH0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int hi = px_h[jj] + b_kh[h], wi = px_w[jj] + b_kw[h];
      const bool v = px_ok[jj] && (unsigned)hi < (unsigned)g.Hh && (unsigned)wi < (unsigned)g.Wh;
      const unsigned vo = v ? (unsigned)(((px_base[jj] + hi) * g.Wh + wi) * g.I * 2 + b_ci[h] + px_plane) : OOB;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_vptr_t)(ldsb + s * STAGE + OFF_B + h * HT + jj * 8192 + wave * 1024),
                                               16, vo, 0, 0, 0);
    }
  };

  const int wm = wave >> 2, wn = wave & 3;          // 64-row half of the 128 o rows; 32-column slice of every 128-column B slice
  const int grp = lane >> 4, idx = lane & 15;
  const int q = idx >> 2, p4 = idx & 3, fh = grp >> 1, cb = grp & 1;
  const int f1 = (q << 2) | (2 * fh), f2 = (q << 2) | (2 * fh + 1);
  const unsigned lds_base = (unsigned)(size_t)(lds_vptr_t)lds;
  auto frag_addr = [&](int cb32, int hi) -> unsigned {
    const int prow = 8 * fh + q + 4 * hi;
    const int ca = cb32 * 4 + 2 * cb + (p4 >> 1);
    return (unsigned)(prow * 256 + ((ca ^ (hi ? f2 : f1)) << 4) + (p4 & 1) * 8);
  };
  unsigned aA[2][2][2], bA[2][2];                   // [stage][sub-tile][lo/hi], [stage][lo/hi]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int hl = 0; hl < 2; ++hl) {
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) aA[s][ti][hl] = lds_base + s * STAGE + OFF_A + frag_addr(wm * 2 + ti, hl);
      bA[s][hl] = lds_base + s * STAGE + OFF_B + frag_addr(wn, hl);
    }
  f32x16_t acc[4][2];                               // [B slice j][32-channel sub-tile]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][s][r] = 0.f;
  u32x2_t aR[2][4][2];                              // [sub-tile][k-step][lo/hi]
  u32x2_t bS[3][4][2];                              // set 0: B0, sets 1 / 2 alternate over B1, B2, B3

#define WN_DSR(dst, addr, off) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define WN_READ_A(S)                                                                                      \
  do {                                                                                                    \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) _Pragma("unroll") for (int ti_ = 0; ti_ < 2; ++ti_) { \
      WN_DSR(aR[ti_][ks_][0], aA[S][ti_][0], ks_ * 4096);                                                 \
      WN_DSR(aR[ti_][ks_][1], aA[S][ti_][1], ks_ * 4096);                                                 \
    }                                                                                                     \
  } while (0)
#define WN_READ_B(S, H, SET)                                                                              \
  do {                                                                                                    \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                                                 \
      WN_DSR(bS[SET][ks_][0], bA[S][0], (H) * HT + ks_ * 4096);                                           \
      WN_DSR(bS[SET][ks_][1], bA[S][1], (H) * HT + ks_ * 4096);                                           \
    }                                                                                                     \
  } while (0)
#define WN_WAIT_A()                                                                                        \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
               : "+v"(aR[0][0][0]), "+v"(aR[0][0][1]), "+v"(aR[0][1][0]), "+v"(aR[0][1][1]), "+v"(aR[0][2][0]), \
                 "+v"(aR[0][2][1]), "+v"(aR[0][3][0]), "+v"(aR[0][3][1]), "+v"(aR[1][0][0]), "+v"(aR[1][0][1]), \
                 "+v"(aR[1][1][0]), "+v"(aR[1][1][1]), "+v"(aR[1][2][0]), "+v"(aR[1][2][1]), "+v"(aR[1][3][0]), \
                 "+v"(aR[1][3][1])::"memory")
#define WN_WAIT_B(SET)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
               : "+v"(bS[SET][0][0]), "+v"(bS[SET][0][1]), "+v"(bS[SET][1][0]), "+v"(bS[SET][1][1]),       \
                 "+v"(bS[SET][2][0]), "+v"(bS[SET][2][1]), "+v"(bS[SET][3][0]), "+v"(bS[SET][3][1])::"memory")
#define WN_FRAG(lo, hi) __builtin_bit_cast(h16x8_t, __builtin_shufflevector(lo, hi, 0, 1, 2, 3))
#define WN_MFMAS(J, SET)                                                                                   \
  do {                                                                                                     \
    __builtin_amdgcn_s_setprio(1);                                                                         \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) _Pragma("unroll") for (int ti_ = 0; ti_ < 2; ++ti_) \
        acc[J][ti_] = rg_mfma_h16_32x32x16(WN_FRAG(aR[ti_][ks_][0], aR[ti_][ks_][1]),   \
                                                              WN_FRAG(bS[SET][ks_][0], bS[SET][ks_][1]),   \
                                                              acc[J][ti_], 0, 0, 0);                       \
    __builtin_amdgcn_s_setprio(0);                                                                         \
  } while (0)
#define WN_SYNC() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
  constexpr int WN_ALL = vmcnt_imm(12);
#define WN_TILE(S, U)                                                                       \
  do {                                                                                      \
    WN_READ_A(S);                                                                           \
    issue_b(ic<1 - (S)>{}, ic<2>{}, (U) + 1);                                               \
    __builtin_amdgcn_s_waitcnt(WN_ALL);                                                     \
    WN_SYNC();                                                                              \
    WN_WAIT_A();                                                                            \
    WN_MFMAS(0, 0);                                                                         \
    WN_SYNC();                                                                              \
    WN_READ_B(S, 1, 1);                                                                     \
    issue_b(ic<1 - (S)>{}, ic<3>{}, (U) + 1);                                               \
    issue_b(ic<(S)>{}, ic<0>{}, (U) + 2);                                                   \
    __builtin_amdgcn_s_waitcnt(WN_ALL);                                                     \
    WN_SYNC();                                                                              \
    WN_WAIT_B(1);                                                                           \
    WN_MFMAS(1, 1);                                                                         \
    WN_SYNC();                                                                              \
    WN_READ_B(S, 2, 2);                                                                     \
    issue_a(ic<(S)>{}, (U) + 2);                                                            \
    __builtin_amdgcn_s_waitcnt(WN_ALL);                                                     \
    WN_SYNC();                                                                              \
    WN_WAIT_B(2);                                                                           \
    WN_MFMAS(2, 2);                                                                         \
    WN_SYNC();                                                                              \
    WN_READ_B(S, 3, 1);                                                                     \
    WN_READ_B(1 - (S), 0, 0);                                                               \
    issue_b(ic<(S)>{}, ic<1>{}, (U) + 2);                                                   \
    __builtin_amdgcn_s_waitcnt(WN_ALL);                                                     \
    WN_SYNC();                                                                              \
    WN_WAIT_B(1);                                                                           \
    WN_WAIT_B(0);                                                                           \
    WN_MFMAS(3, 1);                                                                         \
    WN_SYNC();                                                                              \
  } while (0)

  // prologue: k-tile 0 completely, k-tile 1 up to B1 (B2(1) and B3(1) are issued by P0 / P1 of k-tile 0)
  issue_b(ic<0>{}, ic<0>{}, 0);
  issue_a(ic<0>{}, 0);
  issue_b(ic<0>{}, ic<1>{}, 0);
  issue_b(ic<0>{}, ic<2>{}, 0);
  issue_b(ic<0>{}, ic<3>{}, 0);
  issue_b(ic<1>{}, ic<0>{}, 1);
  issue_a(ic<1>{}, 1);
  issue_b(ic<1>{}, ic<1>{}, 1);
  __builtin_amdgcn_s_waitcnt(WN_ALL);               // B0(0), A0(0) have landed
  WN_SYNC();
  WN_READ_B(0, 0, 0);
  WN_WAIT_B(0);
  if (wave >= 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  for (int u = 0; u < nkt; u += 2) {
    WN_TILE(0, u);
    WN_TILE(1, u + 1);
  }
  if (wave < 4) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#undef WN_TILE
#undef WN_SYNC
#undef WN_MFMAS
#undef WN_FRAG
#undef WN_WAIT_A
#undef WN_WAIT_B
#undef WN_READ_A
#undef WN_READ_B
#undef WN_DSR
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: fp32 [128 o][128 cols] per B slice through LDS, 16-byte stores (512 contiguous bytes per row)
  float* cs = reinterpret_cast<float*>(lds);
  const int fr = lane & 31, fh2 = lane >> 5;
  const long long ldw = (long long)16 * g.I;
  float* outp = g.out + (g.nsplit > 1 ? (long long)zs * g.O * ldw : 0);
  const int c4 = (t & 31) * 4, rr = t >> 5;         // 32 threads per row (128 floats), 16 rows per pass
#pragma unroll
  for (int ep = 0; ep < 4; ++ep) {
    if (ep) __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh2;
        cs[row * 128 + wn * 32 + fr] = acc[ep][s][r];
      }
    __syncthreads();
    if (g.slab16) {
      w8_store_slab16(cs, reinterpret_cast<uint16_t*>(g.out) + ((long long)zs * g.O + o0) * ldw + c0 + ep * 128, ldw, 128, t);
      continue;
    }
#pragma unroll 4
    for (int p = 0; p < 8; ++p) {
      const int row = rr + 16 * p;
      float4* d = reinterpret_cast<float4*>(outp + (long long)(o0 + row) * ldw + c0 + ep * 128 + c4);
      float4 v = *reinterpret_cast<const float4*>(cs + row * 128 + c4);
      if (g.accumulate) {
        const float4 a = *d;
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
      }
      *d = v;
    }
  }
}

}  // namespace

#ifndef RG_WGRAD8_KERNEL_ONLY
// Shapes of the ping-pong weight-gradient kernel: O and 16*I multiples of 256, I a multiple of 8 that divides the
// 128-column half-tiles into whole 8-column chunks of one tap (I % 8 == 0), at least 4 k-tiles of 64 pixels per split.
bool rg_wgrad8_supported(int K, int O, int I) {
  return O % 256 == 0 && (16 * I) % 256 == 0 && I % 8 == 0 && K >= 256;
}

// split-K plan: one block per CU when the tiles allow it (256 x 256 fp32 slabs are 256 KB per block and split)
int rg_wgrad8_split(int K, int O, int I, int* kt_per_split) {
  const int tiles = (O / 256) * (16 * I / 256);
  const int nkt = (K + 63) / 64;
  const int target = rg_option("wgrad8_blocks", 256);
  int ns = (target + tiles - 1) / tiles;
  if (ns < 1) ns = 1;
  int per = (nkt + ns - 1) / ns;
  if (per < 4) per = 4;
  per = (per + 1) & ~1;                                   // even number of k-tiles per split (the loop is unrolled by two)
  ns = (nkt + per - 1) / per;
  *kt_per_split = per;
  return ns;
}

// The 128 x 512 form: O a multiple of 128 that the 256-row tile does not divide, 16 * I a multiple of 512.
bool rg_wgrad8n_supported(int K, int O, int I) {
  return O % 128 == 0 && O % 256 != 0 && (16 * I) % 512 == 0 && I % 8 == 0 && K >= 256;
}
int rg_wgrad8n_split(int K, int O, int I, int* kt_per_split) {
  const int tiles = (O / 128) * (16 * I / 512);
  const int nkt = (K + 63) / 64;
  const int target = rg_option("wgrad8_blocks", 256);
  int ns = (target + tiles - 1) / tiles;
  if (ns < 1) ns = 1;
  int per = (nkt + ns - 1) / ns;
  if (per < 4) per = 4;
  per = (per + 1) & ~1;
  ns = (nkt + per - 1) / per;
  *kt_per_split = per;
  return ns;
}
int rg_wgrad8n_launch(const void* low0, const void* high0, const void* low1, const void* high1, float* out, int Kseg,
                      int two, int O, int I, int Ho, int Wo, int nsplit, int kt_per_split, int accumulate, hipStream_t st,
                      int slab16) {
  W8Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_bytes = (unsigned)((size_t)Kseg * O * 2); g.high_bytes = (unsigned)((size_t)Kseg * 4 * I * 2);
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.out = out; g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_o = O / 128; g.tiles_c = 16 * I / 512; g.nsplit = nsplit; g.kt_per_split = kt_per_split;
  g.accumulate = nsplit == 1 ? accumulate : 0;
  g.slab16 = nsplit > 1 ? slab16 : 0;
  hipLaunchKernelGGL(wgrad8n_kernel<0>, dim3((unsigned)(g.tiles_o * g.tiles_c * nsplit)), dim3(512), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad(mfma, ping-pong, 128 x 512)");
  return RG_OK;
}

int rg_wgrad8_launch(const void* low0, const void* high0, const void* low1, const void* high1, float* out, int Kseg,
                     int two, int O, int I, int Ho, int Wo, int nsplit, int kt_per_split, int accumulate,
                     hipStream_t st, int slab16) {
  W8Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_bytes = (unsigned)((size_t)Kseg * O * 2); g.high_bytes = (unsigned)((size_t)Kseg * 4 * I * 2);
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.out = out; g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_o = O / 256; g.tiles_c = 16 * I / 256; g.nsplit = nsplit; g.kt_per_split = kt_per_split;
  g.accumulate = nsplit == 1 ? accumulate : 0;
  g.slab16 = nsplit > 1 ? slab16 : 0;
  if (rg_option("wgrad8_mfma", RG_WGRAD8_MFMA_DEFAULT) == 16)
    hipLaunchKernelGGL((wgrad8_kernel<false, 0, 16>), dim3((unsigned)(g.tiles_o * g.tiles_c * nsplit)), dim3(512), 0, st, g);
  else
    hipLaunchKernelGGL((wgrad8_kernel<false, 0, 32>), dim3((unsigned)(g.tiles_o * g.tiles_c * nsplit)), dim3(512), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad(mfma, ping-pong)");
  return RG_OK;
}

// A plan without split-K writing its tile as bf16 [O][16*I] (the data-parallel wire buffer's slice for this tensor) instead of fp32
int rg_wgrad8_wire_launch(const void* low0, const void* high0, const void* low1, const void* high1, uint16_t* out16, int Kseg,
                          int two, int O, int I, int Ho, int Wo, int kt_per_split, hipStream_t st) {
  W8Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_bytes = (unsigned)((size_t)Kseg * O * 2); g.high_bytes = (unsigned)((size_t)Kseg * 4 * I * 2);
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.out = reinterpret_cast<float*>(out16); g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_o = O / 256; g.tiles_c = 16 * I / 256; g.nsplit = 1; g.kt_per_split = kt_per_split;
  g.slab16 = 1;                                    // zs = 0: the "slab" is the tensor itself
  if (rg_option("wgrad8_mfma", RG_WGRAD8_MFMA_DEFAULT) == 16)
    hipLaunchKernelGGL((wgrad8_kernel<false, 0, 16>), dim3((unsigned)(g.tiles_o * g.tiles_c)), dim3(512), 0, st, g);
  else
    hipLaunchKernelGGL((wgrad8_kernel<false, 0, 32>), dim3((unsigned)(g.tiles_o * g.tiles_c)), dim3(512), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad_wire(mfma, ping-pong)");
  return RG_OK;
}

// One launch = weight gradient + optimizer step of the tensor (a plan without split-K only): see W8Args::ap.
int rg_wgrad8_adam_launch(const void* low0, const void* high0, const void* low1, const void* high1, int Kseg, int two, int O,
                          int I, int Ho, int Wo, int kt_per_split, float* p, float* m, float* v, uint16_t* shadow,
                          const float* hyper, hipStream_t st) {
  W8Args g{};
  g.low[0] = (const uint16_t*)low0; g.high[0] = (const uint16_t*)high0;
  g.low[1] = (const uint16_t*)(two ? low1 : low0); g.high[1] = (const uint16_t*)(two ? high1 : high0);
  g.low_bytes = (unsigned)((size_t)Kseg * O * 2); g.high_bytes = (unsigned)((size_t)Kseg * 4 * I * 2);
  g.Kseg[0] = Kseg; g.Kseg[1] = two ? Kseg : 0;
  g.out = nullptr; g.O = O; g.I = I;
  g.lgWo = rg_ilog2(Wo); g.lgHo = rg_ilog2(Ho); g.Hh = 2 * Ho; g.Wh = 2 * Wo;
  g.tiles_o = O / 256; g.tiles_c = 16 * I / 256; g.nsplit = 1; g.kt_per_split = kt_per_split;
  g.ap = p; g.am = m; g.av = v; g.ash = shadow; g.hyper = hyper;
  if (rg_option("wgrad8_mfma", RG_WGRAD8_MFMA_DEFAULT) == 16)
    hipLaunchKernelGGL((wgrad8_kernel<true, 0, 16>), dim3((unsigned)(g.tiles_o * g.tiles_c)), dim3(512), 0, st, g);
  else
    hipLaunchKernelGGL((wgrad8_kernel<true, 0, 32>), dim3((unsigned)(g.tiles_o * g.tiles_c)), dim3(512), 0, st, g);
  RG_LAUNCH_CHECK("conv_wgrad_adam(mfma, ping-pong)");
  return RG_OK;
}
#endif  // RG_WGRAD8_KERNEL_ONLY
