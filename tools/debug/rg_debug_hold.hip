// tools/debug/rg_debug_hold.hip -- DIAGNOSTIC, not part of librnagan_hip.so / include/rnagan_hip.h.
// Built on demand into tools/debug/librnagan_debug.so (rna_gan_amd.build.build_debug_library) and loaded only when
// RNAGAN_DEBUG_HOG is set (rna_gan_amd/dist.py): a kernel that holds CUs for a given time, the stand-in for an RCCL
// collective's CU footprint on a one-GPU box.
#include <hip/hip_runtime.h>

template <int NR>      // NR live floats per thread: the register weight of the stand-in (RCCL's kernels are register-heavy)
__global__ __launch_bounds__(256) void hold_cus_kernel(long long ticks, float* sink) {
  float r[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) r[i] = (float)(threadIdx.x + i);
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < NR; ++i) r[i] = r[i] * 1.0001f + 1.f;
    __builtin_amdgcn_s_sleep(16);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) s += r[i];
  if (s == 12345.678f) sink[0] = s;
}

// nblocks > 0: light form (~56 VGPRs per wave); nblocks < 0: register-heavy form (~200 VGPRs) with |nblocks| workgroups.
// Returns 0, -1 on bad arguments, or the HIP launch error.
extern "C" int rgdbg_hold_cus(int nblocks, int microseconds, float* sink, void* stream) {
  if (nblocks == 0 || nblocks > 1024 || nblocks < -1024 || microseconds < 0 || !sink) return -1;
  if (nblocks < 0)
    hipLaunchKernelGGL(hold_cus_kernel<192>, dim3((unsigned)(-nblocks)), dim3(256), 0, (hipStream_t)stream,
                       (long long)microseconds * 100, sink);
  else
    hipLaunchKernelGGL(hold_cus_kernel<48>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream,
                       (long long)microseconds * 100, sink);
  return (int)hipGetLastError();
}
