// probe: SIMD-wide issue rate of the VALU instruction kinds the depth kernel is made of, at 1 / 2 / 3 / 4 waves per SIMD  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 -o probe_valu_mix probe_valu_mix.hip && ./probe_valu_mix
// Each body is 64 instructions of ONE kind over 8 destination registers (dependency distance 8), looped 1000 times; the rows print
// shader cycles per instruction SIMD-WIDE (wave average / waves per SIMD) -- what one more instruction of that kind costs the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)
// I(d, a, b): one instruction writing v(140+d), reading v(150+a), v(150+b)
#define R8(I) I(0, 0, 1) I(1, 2, 3) I(2, 4, 5) I(3, 6, 7) I(4, 1, 2) I(5, 3, 4) I(6, 5, 6) I(7, 7, 0)
#define R64(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I) R8(I)
#define D(d) "v" STR(4##d)
#define S(a) "v" STR(5##a)
#define I_ADD(d, a, b)    "v_add_u32 " D(d) ", " S(a) ", " S(b) "\n"
#define I_XOR(d, a, b)    "v_xor_b32 " D(d) ", " S(a) ", " S(b) "\n"
#define I_MAX3F(d, a, b)  "v_max3_f32 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_MAXF(d, a, b)   "v_max_f32 " D(d) ", " S(a) ", " S(b) "\n"
#define I_MAX3I(d, a, b)  "v_max3_i32 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_CVTU8(d, a, b)  "v_cvt_pk_u8_f32 " D(d) ", " S(a) ", 1, " D(d) "\n"
#define I_PERM(d, a, b)   "v_perm_b32 " D(d) ", " S(a) ", " S(b) ", %[sel]\n"
#define I_CVTI16(d, a, b) "v_cvt_pk_i16_i32 " D(d) ", " S(a) ", " S(b) "\n"
#define I_PKASHR(d, a, b) "v_pk_ashrrev_i16 " D(d) ", 7, " S(a) "\n"
#define I_SATSDWA(d, a, b) "v_sat_pk_u8_i16_sdwa " D(d) ", " S(a) " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD\n"
#define I_CVTF16SDWA(d, a, b) "v_cvt_f16_u16_sdwa " D(d) ", " S(a) " dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1\n"
#define I_DPP(d, a, b)    "v_mov_b32_dpp " D(d) ", " S(a) " row_shr:4 row_mask:0xf bank_mask:0xa\n"
#define I_MAXDPP(d, a, b) "v_max_f32_dpp " D(d) ", " S(a) ", " S(b) " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_PKADD(d, a, b)  "v_pk_add_i16 " D(d) ", " S(a) ", " S(b) "\n"
#define I_PKMAX(d, a, b)  "v_pk_max_i16 " D(d) ", " S(a) ", " S(b) "\n"
#define I_DOT4(d, a, b)   "v_dot4_i32_i8 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_ASHR(d, a, b)   "v_ashrrev_i32 " D(d) ", 7, " S(a) "\n"
#define I_CVTUB(d, a, b)  "v_cvt_f32_ubyte2 " D(d) ", " S(a) "\n"
#define I_ADD3(d, a, b)   "v_add3_u32 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_CNDMASK(d, a, b) "v_cndmask_b32 " D(d) ", " S(a) ", " S(b) ", s[22:23]\n"
#define I_SWAP(d, a, b)   "v_permlane16_swap_b32 " D(d) ", " S(a) "\n"
#define I_FMA(d, a, b)    "v_fma_f32 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_MADU16(d, a, b) "v_pk_mad_u16 " D(d) ", " S(a) ", " S(b) ", " D(d) "\n"
#define I_LDSR(d, a, b)   "ds_read_b128 v[60:63], %[la] offset:" STR(d) "*16\n"
#define I_LDSW(d, a, b)   "ds_write_b32 %[la], " S(a) " offset:" STR(d) "*256\n"
#define I_SALU(d, a, b)   "s_add_u32 s21, s21, 1\n"
#define I_MFMA(d, a, b)   "v_mfma_i32_32x32x32_i8 v[64:79], v[50:53], v[54:57], v[64:79]\n"

#define CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", \
             "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", \
             "s20", "s21", "s22", "s23", "scc", "vcc", "memory"
#define LOOP(B) "s_mov_b32 s20, %[n]\n1:\n" B "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n s_waitcnt lgkmcnt(0)\n"
#define RUN(I) asm volatile(LOOP(R64(I)) :: [n] "s"(iters), [sel] "v"(0x06050201u), [la] "v"(la) : CLOB)

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(float* out, unsigned long long* cyc, int iters)
{
  __shared__ unsigned lds[4096];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned la = (unsigned)(threadIdx.x * 16);
  lds[threadIdx.x] = 0;
  asm volatile("v_mov_b32 v50, 3\n v_mov_b32 v51, 5\n v_mov_b32 v52, 7\n v_mov_b32 v53, 9\n v_mov_b32 v54, 11\n v_mov_b32 v55, 13\n v_mov_b32 v56, 15\n v_mov_b32 v57, 17\n"
               "v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n" ::: CLOB);
  unsigned long long t0, t1;
  __syncthreads();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  if (MODE == 0) RUN(I_ADD);
  if (MODE == 1) RUN(I_XOR);
  if (MODE == 2) RUN(I_MAX3F);
  if (MODE == 3) RUN(I_MAXF);
  if (MODE == 4) RUN(I_MAX3I);
  if (MODE == 5) RUN(I_CVTU8);
  if (MODE == 6) RUN(I_PERM);
  if (MODE == 7) RUN(I_CVTI16);
  if (MODE == 8) RUN(I_PKASHR);
  if (MODE == 9) RUN(I_SATSDWA);
  if (MODE == 10) RUN(I_CVTF16SDWA);
  if (MODE == 11) RUN(I_DPP);
  if (MODE == 12) RUN(I_MAXDPP);
  if (MODE == 13) RUN(I_PKADD);
  if (MODE == 14) RUN(I_PKMAX);
  if (MODE == 15) RUN(I_DOT4);
  if (MODE == 16) RUN(I_ASHR);
  if (MODE == 17) RUN(I_CVTUB);
  if (MODE == 18) RUN(I_ADD3);
  if (MODE == 19) RUN(I_CNDMASK);
  if (MODE == 20) RUN(I_SWAP);
  if (MODE == 21) RUN(I_FMA);
  if (MODE == 22) RUN(I_MADU16);
  if (MODE == 23) RUN(I_LDSR);
  if (MODE == 24) RUN(I_LDSW);
  if (MODE == 25) RUN(I_SALU);
  if (MODE == 26) RUN(I_MFMA);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (lds[lane] == 12345u) out[threadIdx.x] = 1.0f;
}

template <int MODE>
static void run(const char* what, float* dout, unsigned long long* dcyc)
{
  const int iters = 1000;
  printf("%-44s", what);
  for (int wps = 1; wps <= 4; ++wps) {
    const int grid = 256 * wps;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(256), 0, 0, dout, dcyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("  %d w/SIMD: %6.2f", wps, s / h.size() / (iters * 64.0) / wps);
  }
  printf("   cycles per instruction, SIMD-wide\n");
}

int main()
{
  float* dout; unsigned long long* dcyc;
  hipMalloc(&dout, 1024 * 4); hipMalloc(&dcyc, 1024 * 4 * 8);
  run<0>("v_add_u32", dout, dcyc);
  run<1>("v_xor_b32", dout, dcyc);
  run<2>("v_max3_f32", dout, dcyc);
  run<3>("v_max_f32", dout, dcyc);
  run<4>("v_max3_i32", dout, dcyc);
  run<5>("v_cvt_pk_u8_f32", dout, dcyc);
  run<6>("v_perm_b32", dout, dcyc);
  run<7>("v_cvt_pk_i16_i32", dout, dcyc);
  run<8>("v_pk_ashrrev_i16", dout, dcyc);
  run<9>("v_sat_pk_u8_i16 sdwa", dout, dcyc);
  run<10>("v_cvt_f16_u16 sdwa", dout, dcyc);
  run<11>("v_mov_b32 dpp row_shr", dout, dcyc);
  run<12>("v_max_f32 dpp quad_perm", dout, dcyc);
  run<13>("v_pk_add_i16", dout, dcyc);
  run<14>("v_pk_max_i16", dout, dcyc);
  run<15>("v_dot4_i32_i8", dout, dcyc);
  run<16>("v_ashrrev_i32", dout, dcyc);
  run<17>("v_cvt_f32_ubyte2", dout, dcyc);
  run<18>("v_add3_u32", dout, dcyc);
  run<19>("v_cndmask_b32", dout, dcyc);
  run<20>("v_permlane16_swap_b32", dout, dcyc);
  run<21>("v_fma_f32", dout, dcyc);
  run<22>("v_pk_mad_u16", dout, dcyc);
  run<23>("ds_read_b128 (conflict-free)", dout, dcyc);
  run<24>("ds_write_b32", dout, dcyc);
  run<25>("s_add_u32", dout, dcyc);
  run<26>("v_mfma_i32_32x32x32_i8 (one chain)", dout, dcyc);
  return 0;
}
