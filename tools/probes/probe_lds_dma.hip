// global_load_lds_dwordx4 (gfx950): where does lane i's 16 bytes land?  Expectation: M0 base + 16 * lane.  Also under a divergent mask (odd lanes only).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/probe_lds_dma.hip -o /tmp/probe_lds_dma && /tmp/probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const uint4* g, uint4* out, int odd_only)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  reinterpret_cast<uint4*>(lds)[threadIdx.x] = make_uint4(0xDEADu, 0, 0, 0);
  __syncthreads();
  unsigned char* wave_base = lds + 1024 * (threadIdx.x >> 6);
  if (!odd_only || (threadIdx.x & 1))
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(g + threadIdx.x), (void __attribute__((address_space(3)))*)wave_base, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  out[threadIdx.x] = reinterpret_cast<const uint4*>(lds)[threadIdx.x];
}
int main()
{
  const int N = 256;
  uint4 h[N], r[N];
  for (int i = 0; i < N; ++i) h[i] = make_uint4(1000 + i, 2 * i, 3 * i, 4 * i);
  uint4 *d, *o;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof h);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int odd = 0; odd < 2; ++odd) {
    hipLaunchKernelGGL(k, dim3(1), dim3(N), N * 16, 0, d, o, odd);
    hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
    int ok = 0, kept = 0, other = 0;
    for (int i = 0; i < N; ++i) {
      if (!memcmp(&r[i], &h[i], 16)) ok++;
      else if (r[i].x == 0xDEADu) kept++;
      else other++;
    }
    printf("odd_only=%d: lane i's data at base + 16 i: %d of %d, untouched slots %d, anything else %d\n", odd, ok, N, kept, other);
  }
  return 0;
}
