// probe_qsad.hip -- what v_qsad_pk_u16_u8 computes on gfx950 (the ISA text leaves the byte windows of the four results to the reader):
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_qsad.hip -o /tmp/probe_qsad && /tmp/probe_qsad
// prints, for reference bytes r[0..7] and original bytes c[0..3], the four 16-bit results next to sum |r[k + i] - c[i]|, k = 0..3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(const unsigned long long* ref, const unsigned* cur, unsigned long long* out)
{
  unsigned long long d;
  const unsigned c = __builtin_amdgcn_readfirstlane((int)cur[0]);
  asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(ref[threadIdx.x]), "s"(c), "v"(0ull));
  out[threadIdx.x] = d;
}
int main()
{
  unsigned long long h_ref[4], *d_ref, *d_out, h_out[4];
  unsigned h_cur = 0, *d_cur;
  unsigned char r[4][8], c[4] = { 10, 200, 33, 97 };
  srand(1);
  for (int t = 0; t < 4; ++t) { h_ref[t] = 0; for (int i = 0; i < 8; ++i) { r[t][i] = (unsigned char)(rand() & 255); h_ref[t] |= (unsigned long long)r[t][i] << (8 * i); } }
  for (int i = 0; i < 4; ++i) h_cur |= (unsigned)c[i] << (8 * i);
  hipMalloc(&d_ref, 32); hipMalloc(&d_out, 32); hipMalloc(&d_cur, 4);
  hipMemcpy(d_ref, h_ref, 32, hipMemcpyHostToDevice); hipMemcpy(d_cur, &h_cur, 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d_ref, d_cur, d_out);
  hipMemcpy(h_out, d_out, 32, hipMemcpyDeviceToHost);
  for (int t = 0; t < 4; ++t) {
    printf("case %d:", t);
    for (int kk = 0; kk < 4; ++kk) {
      int e = 0;
      for (int i = 0; i < 4; ++i) e += abs((int)r[t][kk + i] - (int)c[i]);
      printf("  k=%d got %u expect %d", kk, (unsigned)((h_out[t] >> (16 * kk)) & 0xFFFF), e);
    }
    printf("\n");
  }
  return 0;
}
