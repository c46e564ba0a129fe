// probe: how much VALU work does the SIMD hide under an MFMA chain?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o probe_coissue probe_coissue.hip && ./probe_coissue
// (a) ONE instruction stream: every MFMA (two independent accumulator chains) is followed by K independent VALU instructions,
//     one or two such waves per SIMD;
// (b) TWO streams on a SIMD: an 8-wave workgroup whose waves 0-3 run MFMAs only and waves 4-7 VALU only (wave w sits on SIMD w % 4).
// All times are s_memtime ticks per MFMA (per VALU instruction for the VALU-only rows), averaged over the waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define STR2(x) #x
#define STR(x) STR2(x)
#define VA(i) "v_add_u32 v" STR(i) ", v" STR(i) ", %[one]\n"
#define V0
#define V1 VA(140)
#define V2 V1 VA(141)
#define V3 V2 VA(142)
#define V4 V3 VA(143)
#define V6 V4 VA(144) VA(145)
#define V8 V6 VA(146) VA(147)
#define V10 V8 VA(148) VA(149)
#define V12 V10 VA(150) VA(151)
#define M32(acc, R) "v_mfma_f32_32x32x16_f16 %[" #acc "], %[w0], v[" STR(R) ":" STR(R) "+3], %[" #acc "]\n"
#define M16(acc, R) "v_mfma_f32_16x16x32_f16 %[" #acc "], %[w0], v[" STR(R) ":" STR(R) "+3], %[" #acc "]\n"
#define MI8(acc, R) "v_mfma_i32_32x32x32_i8 %[" #acc "], %[w0], v[" STR(R) ":" STR(R) "+3], %[" #acc "]\n"
#define BODYI8(V) MI8(b0, 100) V MI8(b1, 104) V MI8(b0, 108) V MI8(b1, 112) V MI8(b0, 100) V MI8(b1, 104) V MI8(b0, 108) V MI8(b1, 112) V
#define BODY32(V) M32(b0, 100) V M32(b1, 104) V M32(b0, 108) V M32(b1, 112) V M32(b0, 100) V M32(b1, 104) V M32(b0, 108) V M32(b1, 112) V
#define BODY16(V) M16(a0, 100) V M16(a1, 104) V M16(a0, 108) V M16(a1, 112) V M16(a0, 100) V M16(a1, 104) V M16(a0, 108) V M16(a1, 112) V
#define BODYV V8 V8 V8 V8 V8 V8 V8 V8
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
             "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "s20", "scc", "memory"
#define LOOP(B) "s_mov_b32 s20, %[n]\n1:\n" B "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
#define RUN(B) asm volatile(LOOP(B) : [b0] "+v"(b0), [b1] "+v"(b1), [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [n] "s"(iters), [one] "v"(one) : CLOB)

// MODE 40..45 = v_mfma_i32_32x32x32_i8 with K = 0, 2, 4, 6, 8, 12 (twice the MACs of 32x32x16 f16 per instruction);
// MODE: 0..6 = 32x32x16 with K = 0, 2, 4, 6, 8, 10, 12;  10..15 = 16x16x32 with K = 0, 1, 2, 3, 4, 6;  20 = VALU only;
// 30/31 = role split (waves 0-3: 32x32x16 / 16x16x32 only, waves 4-7: VALU only)
template <int MODE>
__global__ __launch_bounds__(512) void probe(const f16x8* w, float* out, unsigned long long* cyc, int iters)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f16x8 w0 = w[lane];
  const unsigned one = 1;
  f32x16 b0, b1;
  f32x4 a0 = { 0, 0, 0, 0 }, a1 = { 0, 0, 0, 0 };
  for (int i = 0; i < 16; ++i) { b0[i] = 0; b1[i] = 0; }
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (MODE == 0) RUN(BODY32(V0));
  if (MODE == 1) RUN(BODY32(V2));
  if (MODE == 2) RUN(BODY32(V4));
  if (MODE == 3) RUN(BODY32(V6));
  if (MODE == 4) RUN(BODY32(V8));
  if (MODE == 5) RUN(BODY32(V10));
  if (MODE == 6) RUN(BODY32(V12));
  if (MODE == 10) RUN(BODY16(V0));
  if (MODE == 11) RUN(BODY16(V1));
  if (MODE == 12) RUN(BODY16(V2));
  if (MODE == 13) RUN(BODY16(V3));
  if (MODE == 14) RUN(BODY16(V4));
  if (MODE == 15) RUN(BODY16(V6));
  if (MODE == 20) RUN(BODYV);
  if (MODE == 40) RUN(BODYI8(V0));
  if (MODE == 41) RUN(BODYI8(V2));
  if (MODE == 42) RUN(BODYI8(V4));
  if (MODE == 43) RUN(BODYI8(V6));
  if (MODE == 44) RUN(BODYI8(V8));
  if (MODE == 45) RUN(BODYI8(V12));
  if (MODE == 30) { if (wave < 4) RUN(BODY32(V0)); else RUN(BODYV); }
  if (MODE == 31) { if (wave < 4) RUN(BODY16(V0) BODY16(V0)); else RUN(BODYV); }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = b0[0] + b1[1] + a0[0] + a1[1];
}

template <int MODE>
static void run(const char* what, int threads, int blocks_per_cu, double per_trip, const f16x8* dw, float* dout, unsigned long long* dcyc)
{
  const int iters = 1000, grid = 256 * blocks_per_cu, waves = threads / 64;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(threads), 79872, 0, dw, dout, dcyc, iters);  // 78 KiB: at most 2 blocks per CU
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 8);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  if (MODE == 30 || MODE == 31) {
    double sm = 0, sv = 0;
    for (int b = 0; b < grid; ++b)
      for (int wv = 0; wv < 8; ++wv) (wv < 4 ? sm : sv) += (double)h[b * 8 + wv];
    printf("%-64s MFMA waves: %7.2f ticks per MFMA; VALU waves: %6.2f ticks per VALU instruction (64 per trip)\n", what,
           sm / (grid * 4) / (iters * per_trip), sv / (grid * 4) / (iters * 64.0));
    return;
  }
  double s = 0;
  for (int b = 0; b < grid; ++b)
    for (int wv = 0; wv < waves; ++wv) s += (double)h[b * 8 + wv];
  printf("%-64s %d wave(s)/SIMD: %7.2f ticks per %s\n", what, blocks_per_cu * waves / 4, s / (grid * waves) / (iters * per_trip), MODE == 20 ? "VALU instruction" : "MFMA");
}

int main()
{
  f16x8* dw; float* dout; unsigned long long* dcyc;
  hipMalloc(&dw, 128 * 16); hipMalloc(&dout, 512 * 512 * 4); hipMalloc(&dcyc, 512 * 8 * 8);
  std::vector<_Float16> hw(128 * 8, (_Float16)0.001f);
  hipMemcpy(dw, hw.data(), 128 * 16, hipMemcpyHostToDevice);
  for (int b = 1; b <= 2; ++b) {
    run<20>("VALU only (v_add_u32, 8 independent registers)", 256, b, 64, dw, dout, dcyc);
    run<0>("32x32x16 f16 + 0 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<1>("32x32x16 f16 + 2 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<2>("32x32x16 f16 + 4 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<3>("32x32x16 f16 + 6 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<4>("32x32x16 f16 + 8 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<5>("32x32x16 f16 + 10 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<6>("32x32x16 f16 + 12 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<40>("32x32x32 i8 + 0 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<41>("32x32x32 i8 + 2 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<42>("32x32x32 i8 + 4 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<43>("32x32x32 i8 + 6 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<44>("32x32x32 i8 + 8 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<45>("32x32x32 i8 + 12 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<10>("16x16x32 f16 + 0 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<11>("16x16x32 f16 + 1 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<12>("16x16x32 f16 + 2 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<13>("16x16x32 f16 + 3 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<14>("16x16x32 f16 + 4 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
    run<15>("16x16x32 f16 + 6 VALU per MFMA", 256, b, 8, dw, dout, dcyc);
  }
  run<30>("8-wave workgroup: waves 0-3 32x32x16 only, waves 4-7 VALU only", 512, 1, 8, dw, dout, dcyc);
  run<31>("8-wave workgroup: waves 0-3 16x16x32 only, waves 4-7 VALU only", 512, 1, 16, dw, dout, dcyc);
  return 0;
}
