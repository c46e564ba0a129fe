// probe: does v_cvt_pk_u8_f32 follow MODE.fp_round?  (run on the GPU box)  If it rounds toward -inf under
// fp_round = 2, then cvt_pk_u8 alone is floor + ReLU + clamp + pack for a conv epilogue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* in, unsigned* out, unsigned* out2, float* out3, int n)
{
  int i = threadIdx.x;
  float v = i < n ? in[i] : 0.f;
  // s_setreg_b32 hwreg(HW_REG_MODE, 0, 2): id 1, offset 0, size 2 -> simm16 = (1 << 11) | 1
  unsigned r, r2; float f;
  asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"  // round toward -inf
               "v_cvt_pk_u8_f32 %0, %3, 0, 0\n\t"
               "v_floor_f32 %2, %3\n\t"                                // must be unaffected
               "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"  // back to nearest even
               "v_cvt_pk_u8_f32 %1, %3, 0, 0" : "=&v"(r), "=&v"(r2), "=&v"(f) : "v"(v));
  if (i < n) { out[i] = r; out2[i] = r2; out3[i] = f; }
}
int main()
{
  const float vals[] = { -1000.f, -3.7f, -1.0f, -0.5f, -0.0f, 0.f, 0.25f, 0.5f, 0.75f, 0.99f, 1.0f, 1.5f, 2.5f, 3.5f, 3.99f, 127.49f, 127.5f, 254.5f,
                         254.99f, 255.f, 255.4f, 255.5f, 255.99f, 256.f, 300.f, 70000.f, 1e9f, 0.000061035156f, 17.999939f, 200.00006f };
  const int n = sizeof(vals) / sizeof(float);
  float* d; unsigned *o, *o2; float* o3; unsigned h[64], h2[64]; float h3[64];
  hipMalloc(&d, sizeof vals); hipMalloc(&o, 64 * 4); hipMalloc(&o2, 64 * 4); hipMalloc(&o3, 64 * 4);
  hipMemcpy(d, vals, sizeof vals, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, o2, o3, n);
  hipMemcpy(h, o, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(h3, o3, n * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    float e = floorf(vals[i]); e = e < 0 ? 0 : (e > 255 ? 255 : e);
    printf("v=%-12g  rtn-mode cvt=%3u  rne-mode cvt=%3u  floor+clamp=%3g  v_floor=%g %s\n", vals[i], h[i] & 0xFF, h2[i] & 0xFF, e, h3[i], (h[i] & 0xFF) == (unsigned)e ? "" : "MISMATCH");
    bad += (h[i] & 0xFF) != (unsigned)e;
  }
  printf("%s\n", bad ? "cvt_pk_u8_f32 does NOT follow fp_round" : "cvt_pk_u8_f32 FOLLOWS fp_round (floor + clamp in one instruction)");
  return 0;
}
