// probe: semantics of v_cvt_pk_u8_f32 and of bf16 MFMA accumulation on exact integers (run on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* in, unsigned* out, int n)
{
  int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0, 0);
}
int main()
{
  const float vals[] = { -1000.f, -3.7f, -1.0f, -0.5f, -0.0f, 0.f, 0.25f, 0.5f, 0.75f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 127.49f, 127.5f, 254.5f, 254.99f, 255.f, 255.4f, 255.5f, 256.f, 300.f, 70000.f, 1e9f };
  const int n = sizeof(vals) / sizeof(float);
  float* d; unsigned* o; unsigned h[64];
  hipMalloc(&d, sizeof vals); hipMalloc(&o, 64 * 4);
  hipMemcpy(d, vals, sizeof vals, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n);
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g) = %u\n", vals[i], h[i] & 0xFF);
  return 0;
}
