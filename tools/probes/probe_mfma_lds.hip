// probe: how many cycles per v_mfma_f32_16x16x32_f16 does ONE wave per SIMD sustain when every pair of MFMAs takes its B operand
// from LDS through a register ring (the shape of the depth kernel's conv3 chain)?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o probe_mfma_lds probe_mfma_lds.hip && ./probe_mfma_lds
// Variants: no LDS at all; ring of 4 / 8 fragments; 1 or 2 MFMAs per fragment; one or two workgroups (waves per SIMD) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define STR2(x) #x
#define STR(x) STR2(x)
// one fragment group: wait until at most W reads are outstanding, two MFMAs on fragment registers v[R:R+3], refill the slot
#define GROUP_LDS(R, W, OFF)                                                      \
  "s_waitcnt lgkmcnt(" STR(W) ")\n"                                               \
  "v_mfma_f32_16x16x32_f16 %[a0], %[w0], v[" STR(R) ":" STR(R) "+3], %[a0]\n"     \
  "v_mfma_f32_16x16x32_f16 %[a1], %[w1], v[" STR(R) ":" STR(R) "+3], %[a1]\n"     \
  "ds_read_b128 v[" STR(R) ":" STR(R) "+3], %[addr] offset:" STR(OFF) "\n"
#define GROUP_LDS1(R, W, OFF)                                                     \
  "s_waitcnt lgkmcnt(" STR(W) ")\n"                                               \
  "v_mfma_f32_16x16x32_f16 %[a0], %[w0], v[" STR(R) ":" STR(R) "+3], %[a0]\n"     \
  "ds_read_b128 v[" STR(R) ":" STR(R) "+3], %[addr] offset:" STR(OFF) "\n"
#define GROUP_REG(R)                                                              \
  "v_mfma_f32_16x16x32_f16 %[a0], %[w0], v[" STR(R) ":" STR(R) "+3], %[a0]\n"     \
  "v_mfma_f32_16x16x32_f16 %[a1], %[w1], v[" STR(R) ":" STR(R) "+3], %[a1]\n"
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", \
             "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "s20", "scc", "memory"

typedef __attribute__((ext_vector_type(16))) float f32x16;
#define GROUP_REG32(R)                                                            \
  "v_mfma_f32_32x32x16_f16 %[b0], %[w0], v[" STR(R) ":" STR(R) "+3], %[b0]\n"     \
  "v_mfma_f32_32x32x16_f16 %[b1], %[w1], v[" STR(R) ":" STR(R) "+3], %[b1]\n"
// the same question for v_mfma_f32_32x32x16_f16 (conv1 / conv2 of the depth kernel), fragments in registers
__global__ __launch_bounds__(256) void probe32(const f16x8* w, float* out, unsigned long long* cyc, int iters)
{
  const int lane = threadIdx.x & 63;
  f16x8 w0 = w[lane], w1 = w[64 + lane];
  f32x16 b0, b1;
  for (int i = 0; i < 16; ++i) { b0[i] = 0; b1[i] = 0; }
  unsigned long long t0, t1;
  const unsigned addr = 0;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("s_mov_b32 s20, %[n]\n"
               "1:\n" GROUP_REG32(100) GROUP_REG32(104) GROUP_REG32(108) GROUP_REG32(112) GROUP_REG32(100) GROUP_REG32(104) GROUP_REG32(108) GROUP_REG32(112)
               "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
               : [b0] "+v"(b0), [b1] "+v"(b1) : [w0] "v"(w0), [w1] "v"(w1), [n] "s"(iters), [addr] "v"(addr) : CLOB);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = b0[0] + b1[1];
}

template <int VARIANT>
__global__ __launch_bounds__(256) void probe(const f16x8* w, float* out, unsigned long long* cyc, int iters)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u;  // f16 1.0 pairs, 64 KiB
  __syncthreads();
  f16x8 w0 = w[lane], w1 = w[64 + lane];
  f32x4 a0 = { 0, 0, 0, 0 }, a1 = { 0, 0, 0, 0 };
  const unsigned addr = (unsigned)(reinterpret_cast<uintptr_t>(lds)) + lane * 16 + (threadIdx.x >> 6) * 4096;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (VARIANT == 0) {  // MFMAs only, fragments stay in registers: 16 MFMAs per trip
    asm volatile("s_mov_b32 s20, %[n]\n"
                 "1:\n" GROUP_REG(100) GROUP_REG(104) GROUP_REG(108) GROUP_REG(112) GROUP_REG(100) GROUP_REG(104) GROUP_REG(108) GROUP_REG(112)
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
                 : [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [n] "s"(iters), [addr] "v"(addr) : CLOB);
  } else if (VARIANT == 1) {  // ring of 4 fragments, 2 MFMAs per fragment
    asm volatile("ds_read_b128 v[100:103], %[addr] offset:0\n ds_read_b128 v[104:107], %[addr] offset:1024\n"
                 "ds_read_b128 v[108:111], %[addr] offset:2048\n ds_read_b128 v[112:115], %[addr] offset:3072\n"
                 "s_mov_b32 s20, %[n]\n"
                 "1:\n" GROUP_LDS(100, 3, 16384) GROUP_LDS(104, 3, 17408) GROUP_LDS(108, 3, 18432) GROUP_LDS(112, 3, 19456)
                 GROUP_LDS(100, 3, 0) GROUP_LDS(104, 3, 1024) GROUP_LDS(108, 3, 2048) GROUP_LDS(112, 3, 3072)
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
                 : [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [n] "s"(iters), [addr] "v"(addr) : CLOB);
  } else if (VARIANT == 2) {  // ring of 8 fragments
    asm volatile("ds_read_b128 v[100:103], %[addr] offset:0\n ds_read_b128 v[104:107], %[addr] offset:1024\n"
                 "ds_read_b128 v[108:111], %[addr] offset:2048\n ds_read_b128 v[112:115], %[addr] offset:3072\n"
                 "ds_read_b128 v[116:119], %[addr] offset:16384\n ds_read_b128 v[120:123], %[addr] offset:17408\n"
                 "ds_read_b128 v[124:127], %[addr] offset:18432\n ds_read_b128 v[128:131], %[addr] offset:19456\n"
                 "s_mov_b32 s20, %[n]\n"
                 "1:\n" GROUP_LDS(100, 7, 0) GROUP_LDS(104, 7, 1024) GROUP_LDS(108, 7, 2048) GROUP_LDS(112, 7, 3072)
                 GROUP_LDS(116, 7, 16384) GROUP_LDS(120, 7, 17408) GROUP_LDS(124, 7, 18432) GROUP_LDS(128, 7, 19456)
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
                 : [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [n] "s"(iters), [addr] "v"(addr) : CLOB);
  } else {  // ring of 4, ONE MFMA per fragment (8 MFMAs per trip): twice the LDS traffic per MFMA
    asm volatile("ds_read_b128 v[100:103], %[addr] offset:0\n ds_read_b128 v[104:107], %[addr] offset:1024\n"
                 "ds_read_b128 v[108:111], %[addr] offset:2048\n ds_read_b128 v[112:115], %[addr] offset:3072\n"
                 "s_mov_b32 s20, %[n]\n"
                 "1:\n" GROUP_LDS1(100, 3, 16384) GROUP_LDS1(104, 3, 17408) GROUP_LDS1(108, 3, 18432) GROUP_LDS1(112, 3, 19456)
                 GROUP_LDS1(100, 3, 0) GROUP_LDS1(104, 3, 1024) GROUP_LDS1(108, 3, 2048) GROUP_LDS1(112, 3, 3072)
                 "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
                 : [a0] "+v"(a0), [a1] "+v"(a1) : [w0] "v"(w0), [w1] "v"(w1), [n] "s"(iters), [addr] "v"(addr) : CLOB);
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
}

template <int V>
static void run(const char* what, int mfma_per_trip, int blocks_per_cu, const f16x8* dw, float* dout, unsigned long long* dcyc)
{
  const int iters = 2000, grid = 256 * blocks_per_cu;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<V>, dim3(grid), dim3(256), 79872, 0, dw, dout, dcyc, iters);  // 78 KiB: at most 2 blocks per CU
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-58s %d workgroup(s)/CU: %6.2f cycles per MFMA (wave average)\n", what, blocks_per_cu, s / h.size() / ((double)iters * mfma_per_trip));
}

static void run32(int blocks_per_cu, const f16x8* dw, float* dout, unsigned long long* dcyc)
{
  const int iters = 1000, grid = 256 * blocks_per_cu;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe32), hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe32, dim3(grid), dim3(256), 79872, 0, dw, dout, dcyc, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-58s %d workgroup(s)/CU: %6.2f cycles per MFMA (wave average)\n", "MFMA 32x32x16 f16 only, two chains", blocks_per_cu, s / h.size() / ((double)iters * 16));
}

int main()
{
  f16x8* dw; float* dout; unsigned long long* dcyc;
  hipMalloc(&dw, 128 * 16); hipMalloc(&dout, 512 * 256 * 4); hipMalloc(&dcyc, 512 * 4 * 8);
  std::vector<_Float16> hw(128 * 8, (_Float16)0.001f);
  hipMemcpy(dw, hw.data(), 128 * 16, hipMemcpyHostToDevice);
  for (int b = 1; b <= 2; ++b) {
    run<0>("MFMA 16x16x32 f16 only, two chains", 16, b, dw, dout, dcyc);
    run<1>("+ B from LDS, ring of 4, 2 MFMAs per ds_read_b128", 16, b, dw, dout, dcyc);
    run<2>("+ B from LDS, ring of 8, 2 MFMAs per ds_read_b128", 16, b, dw, dout, dcyc);
    run<3>("+ B from LDS, ring of 4, 1 MFMA per ds_read_b128", 8, b, dw, dout, dcyc);
    run32(b, dw, dout, dcyc);
  }
  return 0;
}
