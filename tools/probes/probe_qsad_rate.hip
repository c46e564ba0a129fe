// probe_qsad_rate.hip -- issue rate of v_qsad_pk_u16_u8 against v_sad_u8 / v_sad_u16 / v_mad_u32_u16 on gfx950: 8 independent chains per wave,
// N waves per SIMD; prints cycles per instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_qsad_rate.hip -o build/probe_qsad_rate && build/probe_qsad_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int KIND>
__global__ void k(u64* out, unsigned cur_in, int iters)
{
  const unsigned cur = __builtin_amdgcn_readfirstlane((int)cur_in);
  u64 a[8], r = threadIdx.x * 0x0101010101010101ull;
  unsigned b[8];
  for (int i = 0; i < 8; ++i) { a[i] = i; b[i] = i + threadIdx.x; }
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) { u64 d; asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(r), "s"(cur), "v"(a[i])); a[i] = d; }
      if (KIND == 1) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(b[i]) : "v"((unsigned)r), "s"(cur));
      if (KIND == 2) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(b[i]) : "v"((unsigned)r), "s"(cur));
      if (KIND == 3) asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(b[i]) : "v"((unsigned)r), "s"(cur));
      if (KIND == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(b[i]) : "v"((unsigned)r));
    }
  }
  const long long t1 = clock64();
  u64 s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + (u64)(t1 - t0);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (u64)(t1 - t0);
}
int main()
{
  u64* d; hipMalloc(&d, 8 * 1024 * 1024);
  const char* names[5] = { "v_qsad_pk_u16_u8", "v_sad_u8", "v_sad_u16", "v_mad_u32_u16", "v_add_u32" };
  const int iters = 4096;
  for (int kind = 0; kind < 5; ++kind)
    for (int waves = 1; waves <= 4; ++waves) {   // waves per SIMD: one workgroup of 256 * waves threads on one CU
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256 * waves), 0, 0, d, 0x01020304u, iters);
        if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256 * waves), 0, 0, d, 0x01020304u, iters);
        if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256 * waves), 0, 0, d, 0x01020304u, iters);
        if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256 * waves), 0, 0, d, 0x01020304u, iters);
        if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(256 * waves), 0, 0, d, 0x01020304u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // one workgroup per CU (256 CUs): per SIMD `waves` waves x iters x 8 instructions
      const double ns_per = ms * 1e6 / ((double)waves * iters * 8);
      printf("%-18s %d wave(s)/SIMD: %.3f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", names[kind], waves, ns_per, ns_per * 2.4);
    }
  return 0;
}
