// Which SIMD does wave w of a 768-thread workgroup (12 waves, 150 KB of LDS: one workgroup per CU) run on?  HW_REG_HW_ID bits [5:4] = SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_wave_simd.hip -o /tmp/probe_wave_simd ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(768, 1) void k(unsigned* out)
{
  extern __shared__ unsigned char lds[];
  const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = id;
  if (threadIdx.x == 0) lds[0] = 1;
}
int main()
{
  unsigned* d; const int G = 256;
  hipMalloc(&d, G * 12 * 4);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 150144);
  hipLaunchKernelGGL(k, dim3(G), dim3(768), 150144, 0, d);
  unsigned h[G * 12];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int pattern[12][4] = {};
  for (int b = 0; b < G; ++b) for (int w = 0; w < 12; ++w) pattern[w][(h[b * 12 + w] >> 4) & 3]++;
  for (int w = 0; w < 12; ++w) printf("wave %2d: SIMD0 %3d  SIMD1 %3d  SIMD2 %3d  SIMD3 %3d   (block 0: hw_id %08x simd %u wave slot %u)\n", w, pattern[w][0], pattern[w][1], pattern[w][2], pattern[w][3],
                                     h[w], (h[w] >> 4) & 3, h[w] & 15);
  return 0;
}
