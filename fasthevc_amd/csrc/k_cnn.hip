// k_cnn.hip -- depth classifier for one 64x64 luma CTU per workgroup iteration, gfx950 only.
//
// What it computes (bit-exact twin of oracle/fhevc_oracle.c: fho_cnn_ctu + fho_depth_from_logits):
//   the network the reference specifies in matlab/dataExtraction/Train...Example.m:75-96
//   (conv3x3x16 -> ReLU -> maxpool2 -> conv3x3x32 -> ReLU -> maxpool2 -> conv3x3x64 -> ReLU -> FC(2)),
//   run convolutionally over the CTU, with three FC heads (64-, 32- and 16-level split decisions) and the
//   top-down assembly of the 16x16 depth map that TEncCu::xCompressCU consumes (TEncCu.cpp:496-1058).
//
// How (HISTORY.md sections 5.1 - 5.1c, 5.6).  One template, fhevc_cnn_depth_kernel<STAMPS, HAD, ARITH>, in two arithmetic forms that
// deliver the same integers (every GPU parity test runs both):
//   ARITH = 0, the 16-bit form (round 1): conv1 v_mfma_f32_32x32x16_bf16, conv2 v_mfma_f32_32x32x16_f16, conv3 v_mfma_f32_16x16x32_f16;
//     activations between the convs as f16 in 8-channel LDS planes; weights carry 2^-shift, biases are the C operand, fp32 rounding toward
//     -inf turns v_cvt_pk_u8_f32 into floor + ReLU + clamp + pack; FC heads on v_dot4_i32_i8; two workgroups per CU (79.8 KB of LDS, 254 VGPRs);
//   ARITH = 1 | 2, the i8 form (round 2, the library's default): conv2 and conv3 on v_mfma_i32_32x32x32_i8 with the activations as signed
//     bytes a - 128 in 16-channel planes (half the LDS), int32 accumulators, requant by shifts (2: the short forms where the blob's shifts
//     and accumulator bounds allow), the 16- / 32-level heads as one v_mfma_i32_16x16x64_i8 GEMM, conv2's fragments re-fetched per CTU;
//     three workgroups per CU (51 KB of LDS, 168 VGPRs), s_setprio per phase.
// Common to both: persistent workgroups of 256 threads, grid-stride over CTUs in an XCD-aware order; A = weights (resident in VGPRs), B =
// im2col fragments read from LDS with one ds_read_b128 per K step through a register ring; the 2x2 max-pools in-lane (conv1 folds the pool
// window into the MFMA's M dimension, conv2 gives a lane one pooled position and four accumulators); the next CTU's samples prefetched
// into registers during conv3 and staged into LDS after the heads; four barriers per CTU.
//   HAD = 1: the per-CTU source Hadamard (TEncCu::updateCtuDataISlice) on packed 16-bit VALU from the prefetched samples, at the tail of
//     conv3 (default); HAD = 2: the same on the bf16 MFMA from the staged tile (8-bit content; measured slower, kept for A/B and tests).
// Round 3 adds, after the kernel: fhevc_cnn_depth_pipe_kernel, the i8 form as a two-stage software pipeline over CTUs (opt-in, measured
// slower), and k_cnn_family.inc, the reference's Bayesian-optimisation network family (NetworkDepth 1: 32 / 64 / 128 filters).
#include "fhevc_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // 8 x 16-bit operand slots of an MFMA fragment (bf16 for conv1, f16 for conv2/conv3)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

namespace {

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1 (a 36-step body with nested loops is past what `#pragma unroll`
// unrolls; register arrays indexed by a loop variable that stays a variable go through v_movrel / scratch)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// ---- LDS map (bytes) ---------------------------------------------------------------------------------------
// conv1 output: 32x32 + 1 halo each side = 34 x 34 positions, 16 B per position and plane
constexpr int A1_ROW = 36 * 16;                   // bytes per row of a plane; inside a row the columns are split by parity: odd x
constexpr int A1_EVEN = 19 * 16;                  // at slot x >> 1, even x at slot 19 + (x >> 1), so that the stride-2 column runs
                                                  // conv2 reads (one pooled column per lane) are contiguous; 19 = 3 mod 8 keeps the
                                                  // two 64-byte halves of conv1's 8-lane store groups on disjoint banks
constexpr int R1_OFF = 0;                         // R1: A1; later A3 u8 [256][64]
constexpr int A3_OFF = R1_OFF;
constexpr int A2_PITCH = 18;                      // conv2 output 16x16 + halo
constexpr int A2_PLANE = 18 * 18 * 16 + 192;      // 5376 = 21 * 256: conv3's B operand takes lane group kg of a ds_read_b128 from plane kg,
                                                  // and {positions 0-3, 12-15 of plane 0} + {4-11 of plane 1} only tile a 256-byte bank row
                                                  // when the planes are a multiple of 256 B apart
constexpr int IN_PITCH = 68;                      // dwords per input row PAIR (66 used): lo = row 2j, hi = row 2j+1
// The two arithmetic variants of the kernel differ in the activations between the convs:
//   I8 = false: f16 (16-bit MFMAs): A1 = 2 planes of 8 channels, A2 = 4 planes of 8 channels;
//   I8 = true : signed bytes a - 128 (v_mfma_i32_32x32x32_i8 in conv2 and conv3): A1 = 1 plane of 16 channels (+ one phantom row that
//               only zero weights meet), A2 = 2 planes of 16 channels -- half the LDS, which is what lets three workgroups share a CU
#ifndef FHEVC_MFMA_HEADS_F16
#define FHEVC_MFMA_HEADS_F16 0  // the 16-bit form keeps its v_dot4 heads: with the MFMA heads it measured the same (0.5756 against 0.5767 ms, parity green)
#endif
// Wave priority per phase (s_setprio 0..3), one hex digit each: 0x<heads><conv3><conv2><conv1>.  With three workgroups per CU the i8
// form gains 4 % when its three conv phases outrank the heads / staging / depth phases of the other workgroups' waves on the same SIMD
// (same-box A/B: none 0.4440, conv2+conv3 0.4336, +conv1 0.4285 at level 1 and 0.4262 at level 2, +heads 0.4355 ms); the 16-bit form
// (two workgroups per CU) gains 3.5 % from the same setting (0.5828 -> 0.5624 ms; conv2+conv3 only: 0.5720)
#ifndef FHEVC_I8_PRIO
#define FHEVC_I8_PRIO 0x0222
#endif
#ifndef FHEVC_F16_PRIO
#define FHEVC_F16_PRIO 0x0123  // two workgroups per CU: the earlier phase outranks the later one (conv1 3, conv2 2, conv3 1): 0.5377 ms at 0x0222,
#endif                         // 0.5199 at 0x0122, 0.5093 at 0x0123 (0x0133 0.5124, 0x0022 0.5216, 0x0112 0.5251); the i8 form measures equal across these
// phase: 0 conv1, 1 conv2, 2 conv3, 3 heads
#define FHEVC_PRIO_OF(phase) (((I8 ? FHEVC_I8_PRIO : FHEVC_F16_PRIO) >> (4 * (phase))) & 3)
// Round 4: the conv phases' level also depends on WHICH of the CU's three workgroups the wave belongs to (FHEVC_SLOT_PRIO = 2, the default of the i8
// form: conv phases at 1 + slot = 1 / 2 / 3, everything else at 0).  With one level for all, two waves of a SIMD that are both in a conv phase tie and
// the arbiter falls back to age; with the levels apart the matrix pipe goes to one of them outright and the other's chain runs in the gaps: same-box
// A/B 0.3801 -> 0.3706 ms (-2.5 %, twice: profiles/r04_ab_slot_priority.log).  slot = the workgroup's LDS base / its LDS size (HW_REG_LDS_ALLOC): 0, 1, 2
// whatever order the dispatcher fills the CUs in (tools/probes/probe_wg_slot.hip: under round-robin dispatch it equals blockIdx / 256).  Measured and
// dropped: 1 = static slot level without phase levels (+4.6 %), 3 = conv phases at 1 + slot and the rest at slot (+2.8 %), 4 = tables: conv1 held at
// 1 (-1.0 %) or at 0 (+1.9 %), heads at 1 for slots 1, 2 (-1.0 %), levels by phase only 1 / 2 / 3 (-1.6 %).
#ifndef FHEVC_SLOT_PRIO
#define FHEVC_SLOT_PRIO 2
#endif
#define FHEVC_SETPRIO_DYN(p) { const int p_ = (p); if (p_ == 1) __builtin_amdgcn_s_setprio(1); else if (p_ == 2) __builtin_amdgcn_s_setprio(2); else if (p_ >= 3) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#if FHEVC_SLOT_PRIO == 0
#define FHEVC_PRIO_ON(phase)  if (FHEVC_PRIO_OF(phase)) __builtin_amdgcn_s_setprio(FHEVC_PRIO_OF(phase));
#define FHEVC_PRIO_OFF(phase) if (FHEVC_PRIO_OF(phase)) __builtin_amdgcn_s_setprio(0);
#elif FHEVC_SLOT_PRIO == 1
#define FHEVC_PRIO_ON(phase)
#define FHEVC_PRIO_OFF(phase)
#elif FHEVC_SLOT_PRIO == 2
#define FHEVC_PRIO_ON(phase)  if (FHEVC_PRIO_OF(phase)) { if (I8) FHEVC_SETPRIO_DYN(1 + prio_slot) else __builtin_amdgcn_s_setprio(FHEVC_PRIO_OF(phase)); }
#define FHEVC_PRIO_OFF(phase) if (FHEVC_PRIO_OF(phase)) __builtin_amdgcn_s_setprio(0);
#elif FHEVC_SLOT_PRIO == 3
#define FHEVC_PRIO_ON(phase)  if (FHEVC_PRIO_OF(phase)) FHEVC_SETPRIO_DYN(1 + prio_slot)
#define FHEVC_PRIO_OFF(phase) if (FHEVC_PRIO_OF(phase)) FHEVC_SETPRIO_DYN(prio_slot)
#else  // 4: a table, one 16-bit group 0x<heads><conv3><conv2><conv1> per slot (slot 0 in the low bits)
#ifndef FHEVC_SLOT_TABLE
#define FHEVC_SLOT_TABLE 0x033302220111ULL
#endif
#define FHEVC_PRIO_ON(phase)  FHEVC_SETPRIO_DYN((int)((FHEVC_SLOT_TABLE >> (16 * prio_slot + 4 * (phase))) & 3))
#define FHEVC_PRIO_OFF(phase) __builtin_amdgcn_s_setprio(0);
#endif
// tuning knobs of the i8 form's pipeline descriptions (VALU instructions offered per MFMA group of a chain; fences around the pools).
// conv2 measured best with the chains pinned as (MFMA, DS read) groups only and the epilogue VALU left to the scheduler, without
// fences: 0.4322 ms (fences, 2 / 5 VALU per group) -> 0.4226 (no fences) -> 0.4212 (no fences, no VALU groups), same-box A/B
#ifndef FHEVC_I8_C2_FILL_POOL
#define FHEVC_I8_C2_FILL_POOL 0
#endif
#ifndef FHEVC_I8_C2_FILL_REQUANT
#define FHEVC_I8_C2_FILL_REQUANT 0
#endif
#ifndef FHEVC_I8_C3_FILL
#define FHEVC_I8_C3_FILL 4
#endif
#ifndef FHEVC_I8_C2_FENCE
#define FHEVC_I8_C2_FENCE 0
#endif
#ifndef FHEVC_F16_C2_FILL_POOL
#define FHEVC_F16_C2_FILL_POOL 1
#endif
#ifndef FHEVC_F16_C2_FILL_REQUANT
#define FHEVC_F16_C2_FILL_REQUANT 3
#endif
#ifndef FHEVC_F16_C2_FENCE
#define FHEVC_F16_C2_FENCE 1
#endif
#ifndef FHEVC_I8_C2_SCHED
#define FHEVC_I8_C2_SCHED 1
#endif
#ifndef FHEVC_I8_C3_SCHED
#define FHEVC_I8_C3_SCHED 0  // conv3 of the i8 form runs 0.7 % faster WITHOUT a pipeline description (0.4231 -> 0.4202 ms); conv2 needs its (MFMA, DS read)
#endif                       // pairing: without it 0.4575 ms
#ifndef FHEVC_CONV1_MFMA_FIRST
#define FHEVC_CONV1_MFMA_FIRST 0
#endif
#ifndef FHEVC_F16_C3_SCHED
#define FHEVC_F16_C3_SCHED 1
#endif
#ifndef FHEVC_CONV1_UNROLL
#define FHEVC_CONV1_UNROLL 2
#endif
#ifndef FHEVC_I8_WG_PER_CU
#define FHEVC_I8_WG_PER_CU 3
#endif
template <bool I8>
struct Lds {
  static constexpr int A1_PLANE = (I8 ? 35 : 34) * A1_ROW;
  static constexpr int A1_PLANES = I8 ? 1 : 2, A2_PLANES = I8 ? 2 : 4;
  static constexpr int R1_BYTES = A1_PLANES * A1_PLANE;     // 39168 / 20160
  static constexpr int R2_OFF = R1_OFF + R1_BYTES;          // R2: input CTU bf16 [66][68]; later A2
  static constexpr int R2_BYTES = A2_PLANES * A2_PLANE;     // 21504 / 10752
  static constexpr int BIAS_OFF = R2_OFF + R2_BYTES;        // b1[16] (float) b2[32] b3[64] (float, or int32 in the i8 variant)
  static constexpr int LOGIT_OFF = BIAS_OFF + 112 * 4;      // int logits[21][2] (the 64-level pair is formed by the readers) + at [44..51] the
                                                            // four waves' partial 64-level sums; at [56..63] two sets of the four waves' source-
                                                            // Hadamard sums (the set of the CTU in flight and the set of the next one)
  static constexpr int HEADW_OFF = LOGIT_OFF + 64 * 4;      // int8 head weights: wh64, wh32, wh16 = 18432 B
  static constexpr int HADP_OFF = HEADW_OFF + 18432;        // MFMA form of the source Hadamard: float partial sums [set 2][M tile 2][block 64]
  static constexpr int HEADX_OFF = HADP_OFF + 1024;         // the MFMA heads' scatter table: per lane four 16-bit LDS addresses (its four partial sums' logits, or
                                                            // the lane's own dummy dword at HEADX_OFF + 512 + 4 lane for a sum that joins nothing): 512 + 256 B
  static constexpr int LDS_BYTES = HEADX_OFF + 768;         // 81600 -> two workgroups per CU (159.4 of 160 KiB) / 51840 -> three
  static constexpr unsigned HALO_FILL = I8 ? 0x80808080u : 0u;  // "activation 0" in the halos of A1 and A2
  static_assert(A2_PLANE % 256 == 0, "conv3 reads lane groups of a ds_read_b128 from different planes");
  static_assert(33 * IN_PITCH * 4 + 4 * IN_PITCH * 4 <= R2_BYTES, "input tile (and conv1's one fragment read past it) must fit the A2 region");
  static_assert(HEADW_OFF % 16 == 0, "head weights are read with ds_read_b128");
  static_assert(A3_OFF + 16384 <= R1_OFF + R1_BYTES, "conv3's output must fit R1");
  static_assert(R2_OFF + R2_BYTES <= 65536, "halo offsets are packed into 16 bits");
};
static_assert(2 * Lds<false>::LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(3 * Lds<true>::LDS_BYTES <= 160 * 1024, "three workgroups per CU");
// the 16-bit variant's names, as its epilogues and chains use them
constexpr int A1_PLANE = Lds<false>::A1_PLANE;

constexpr int HEAD64_OFF = 0, HEAD32_OFF = 2 * 4096, HEAD16_OFF = 4 * 4096;  // into whead (int8)
constexpr int HEADM_OFF = 2 * 4096, HEADM_STEP = 10 * 64, HEADM_BYTES = 16 * HEADM_STEP;  // the i8 form's MFMA image of wh32 + wh16

// max(v, value of the horizontally adjacent lane): with old = 0 and bound_ctrl the DPP move folds into ONE
// v_max_f32_dpp quad_perm:[1,0,3,2] (the (v, v, bound_ctrl = 0) form costs v_mov + v_mov_dpp + v_max)
__device__ __forceinline__ float max_with_xor1(float v)
{
  const int o = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true);
  return fmaxf(v, __builtin_bit_cast(float, o));
}
// Activations between the convs travel as f16 (integers 0..255, exact): under the kernel's round-down mode v_cvt_pk_u8_f32 is
// floor + ReLU + clamp + pack of four values into a dword, and v_cvt_f16_u16 with SDWA byte select / word destination turns
// two of its bytes into one packed f16 pair: 2 instructions per output where floor + med3 + half a v_perm needed 2.5
template <int HI>
__device__ __forceinline__ unsigned f16_pair_of_bytes(unsigned d)  // HI = 0: bytes 0, 1; HI = 1: bytes 2, 3
{
  unsigned r;
  if (HI == 0) {
    asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(r) : "v"(d));
    asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(r) : "v"(d));
  } else {
    asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:BYTE_2" : "=v"(r) : "v"(d));
    asm("v_cvt_f16_u16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(r) : "v"(d));
  }
  return r;
}
__device__ __forceinline__ unsigned u8x4_floor_clamp(float a, float b, float c, float d)
{
  unsigned r = __builtin_amdgcn_cvt_pk_u8_f32(a, 0, 0);
  r = __builtin_amdgcn_cvt_pk_u8_f32(b, 1, r);
  r = __builtin_amdgcn_cvt_pk_u8_f32(c, 2, r);
  return __builtin_amdgcn_cvt_pk_u8_f32(d, 3, r);
}
// two fp32 holding integers 0..255 -> two bf16 (exact: the low 16 mantissa bits are zero)
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
__device__ __forceinline__ bf16x8 lds_frag(const unsigned char* p)
{
  return *reinterpret_cast<const bf16x8*>(p);
}
__device__ __forceinline__ unsigned bytemax(unsigned a, unsigned b)
{
  u16x2 al = __builtin_bit_cast(u16x2, a & 0x00FF00FFu), bl = __builtin_bit_cast(u16x2, b & 0x00FF00FFu);
  u16x2 ah = __builtin_bit_cast(u16x2, (a >> 8) & 0x00FF00FFu), bh = __builtin_bit_cast(u16x2, (b >> 8) & 0x00FF00FFu);
  unsigned lo = __builtin_bit_cast(unsigned, __builtin_elementwise_max(al, bl));
  unsigned hi = __builtin_bit_cast(unsigned, __builtin_elementwise_max(ah, bh));
  return lo | (hi << 8);
}
__device__ __forceinline__ int sdot4(unsigned a, unsigned b, int c)  // v_dot4_i32_i8: four signed 8-bit products
{
  return __builtin_amdgcn_sdot4((int)a, (int)b, c, false);
}
// init the 16 accumulator rows of a 32-channel tile from the bias table in LDS: reg i -> channel
// (i&3) + 8*(i>>2) + 4*h (C/D layout of v_mfma_f32_32x32x16_bf16)
__device__ __forceinline__ f32x16 bias_tile(const float* b32, int h)
{
  f32x16 acc;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(b32 + 8 * g + 4 * h);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
  return acc;
}

// one sample as the network sees it: rounded to 8 bits, clamped to 0..255.  The oracle centres it (x - 128); here the LDS
// image keeps x itself and conv1's bias carries -128 * (sum of the filter's weights) (fhevc_api.hip), which is the same
// integer arithmetic: the halo and everything outside the picture hold 128
template <typename T>
__device__ __forceinline__ int load_sample8(const T* p, int shift)
{
  int v = (int)*p;
  if (shift > 0) v = min(255, (v + (1 << (shift - 1))) >> shift);
  return min(255, max(0, v));
}

// ---- fused epilogues (forceinline: everything stays in registers) ------------------------------------------
// 16-lane row sum with DPP only (every lane ends with the row total)
__device__ __forceinline__ int dpp_row_sum(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);  // row_mirror
  return v;
}
// conv1: the two MFMAs of a unit hold, per lane, the four pre-pool outputs of 8 channels for ONE pooled position
// (rows m of A: channel = m[1:0] + 4*m[3] + 8*m[2], pre-pool row = m[4], pre-pool column = m[5]; lane half h = m[2]
// holds channels 8h..8h+7 = plane h).  2x2 max-pool = 3 in-lane max per channel, then requant and ONE 16-byte store.
__device__ __forceinline__ void conv1_store(const f32x16& acc0, const f32x16& acc1, unsigned char* dst)
{
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)  // regs k (py 0) and k+8 (py 1) of both MFMAs (px 0, 1); the accumulators are already scaled by 2^-s (weights, bias)
    m[k] = fmaxf(fmaxf(acc0[k], acc0[k + 8]), fmaxf(acc1[k], acc1[k + 8]));
  const unsigned d0 = u8x4_floor_clamp(m[0], m[1], m[2], m[3]), d1 = u8x4_floor_clamp(m[4], m[5], m[6], m[7]);
  *reinterpret_cast<uint4*>(dst) = make_uint4(f16_pair_of_bytes<0>(d0), f16_pair_of_bytes<1>(d0), f16_pair_of_bytes<0>(d1), f16_pair_of_bytes<1>(d1));
}
// conv2: a lane owns ONE pooled position; its four pre-pool outputs sit in four accumulators (dy, dx), so the 2x2
// max-pool is in-lane.  The accumulators start from the pre-scaled bias tile and conv2's weights carry 2^-s, so they
// already hold (acc + b) * 2^-s (max-pooling commutes with the common bias): pool, floor, clamp, pack.
// First half (dy = 0 done): the horizontal maximum, in place
__device__ __forceinline__ void conv2_pool_h(f32x16& acc0, const f32x16& acc1)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) acc0[k] = fmaxf(acc0[k], acc1[k]);
}
// Second half (dy = 1 done): the vertical maximum into the same registers (this frees the two accumulators for the next
// half-chain before it starts: the phase is register-tight) ...
__device__ __forceinline__ void conv2_pool_v(f32x16& top, const f32x16& acc0, const f32x16& acc1)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) top[k] = fmaxf(fmaxf(top[k], acc0[k]), acc1[k]);
}
// ... then requant and four 8-byte stores under the next half-chain (reg i -> channel (i & 3) + 8 * (i >> 2) + 4h: plane i >> 2,
// bytes 8h + 2 * (i & 3)); dst = this lane's position in plane 0 + 8h
__device__ __forceinline__ void conv2_requant_store(const f32x16& m, unsigned char* dst)
{
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const unsigned d = u8x4_floor_clamp(m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3]);
    *reinterpret_cast<uint2*>(dst + g * A2_PLANE) = make_uint2(f16_pair_of_bytes<0>(d), f16_pair_of_bytes<1>(d));
  }
}
// conv3 on v_mfma_f32_16x16x32_bf16 (the shape that holds the higher clock under the power limit): a K step is ONE tap x all 32
// input channels (lane group kg = lane >> 4 reads activation plane kg), N = the 16 positions of one output row, M = 16 output
// channels; a wave owns two M tiles (its 32 channels) and 8 rows.  D layout: lane (x = lane & 15, rg = lane >> 4), reg i ->
// channel 16 mt + 4 rg + i at position x.
// Requant: the kernel runs with MODE.fp_round = toward -inf (set once at its top), under which v_cvt_pk_u8_f32 rounds DOWN and
// saturates to 0..255 (verified on hardware: tools/probes/probe_cvt_mode): one instruction = floor + ReLU + clamp + pack.  The
// weights carry 2^-s and the accumulators start from the pre-scaled bias, so they already hold (acc + b) * 2^-s exactly.
// dst = the position's 64-byte row + 4 rg; channel 32 tile + 16 mt + 4 rg + i lives in logical 16-B chunk 2 tile + mt, byte
// 4 rg + i; psw = chunk swizzle of this position
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ __forceinline__ void conv3_store(const f32x4& acc0, const f32x4& acc1, unsigned char* dst, int tile, int psw)
{
  unsigned d0 = 0, d1 = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    d0 = __builtin_amdgcn_cvt_pk_u8_f32(acc0[k], k, d0);
    d1 = __builtin_amdgcn_cvt_pk_u8_f32(acc1[k], k, d1);
  }
  // stored as a - 128 (signed bytes) for the heads' v_dot4_i32_i8; their biases carry + 128 * sum of weights (fhevc_api.hip)
  *reinterpret_cast<unsigned*>(dst + (((2 * tile + 0) ^ psw) << 4)) = d0 ^ 0x80808080u;
  *reinterpret_cast<unsigned*>(dst + (((2 * tile + 1) ^ psw) << 4)) = d1 ^ 0x80808080u;
}

// ---- the i8 variant's epilogues: int32 accumulators (bias + 128 * sum of weights as the C operand: the activations travel as
// a - 128), requant = arithmetic shift, clamp to 0..255, back to a - 128.  Four values -> one dword in 9 instructions:
// 4 v_ashrrev_i32, 2 v_cvt_pk_i16_i32 (saturating), 2 v_sat_pk_u8_i16 (SDWA: low / high word of the result), 1 v_xor
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;
// Two shorter forms where the numbers allow (mode, uniform per layer, chosen by the host: build_weight_image):
//   mode 1 (shift <= 7): saturate to int16 FIRST, then one packed 16-bit shift per pair (sat16(x) >> s and x >> s clamp to the same
//           byte: 32767 >> 7 = 255) -- 7 instructions;
//   mode 2 (shift == 8 and every accumulator provably inside 24 bits): bytes 1, 2 of x ARE x >> 8 as an int16, one v_perm_b32 per
//           pair -- 5 instructions
__device__ __forceinline__ unsigned requant4_i8(int a, int b, int c, int d, int shift, int mode)
{
  s16x2_t p0, p1;
  if (mode == 2) {
    p0 = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_perm((unsigned)b, (unsigned)a, 0x06050201u));
    p1 = __builtin_bit_cast(s16x2_t, __builtin_amdgcn_perm((unsigned)d, (unsigned)c, 0x06050201u));
  } else if (mode == 1) {
    const s16x2_t sh = { (short)shift, (short)shift };
    p0 = __builtin_amdgcn_cvt_pk_i16(a, b) >> sh;
    p1 = __builtin_amdgcn_cvt_pk_i16(c, d) >> sh;
  } else {
    p0 = __builtin_amdgcn_cvt_pk_i16(a >> shift, b >> shift);
    p1 = __builtin_amdgcn_cvt_pk_i16(c >> shift, d >> shift);
  }
  unsigned r;
  asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD" : "=v"(r) : "v"(p0));
  asm("v_sat_pk_u8_i16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD" : "+v"(r) : "v"(p1));
  return r ^ 0x80808080u;
}
// conv1 (still on the bf16 MFMA: K = 16 gains nothing from the i8 shape): pool, floor + clamp, a - 128; the lane's 8 channels
// (8h .. 8h+7) are 8 bytes of the position's 16
__device__ __forceinline__ void conv1_store_i8(const f32x16& acc0, const f32x16& acc1, unsigned char* dst)
{
  float m[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) m[k] = fmaxf(fmaxf(acc0[k], acc0[k + 8]), fmaxf(acc1[k], acc1[k + 8]));
  *reinterpret_cast<uint2*>(dst) = make_uint2(u8x4_floor_clamp(m[0], m[1], m[2], m[3]) ^ 0x80808080u, u8x4_floor_clamp(m[4], m[5], m[6], m[7]) ^ 0x80808080u);
}
#ifndef FHEVC_X_NOBIAS
#define FHEVC_X_NOBIAS 0   // (sensitivity experiments, tools/experiments: 1 = no bias-tile reads, WRONG results, timing only)
#endif
#ifndef FHEVC_X_C3_HALF
#define FHEVC_X_C3_HALF 0  // (sensitivity experiments: 1 = conv3 reads half of its fragments, WRONG results, timing only)
#endif
__device__ __forceinline__ i32x16 bias_tile_i8(const int* b32, int h)  // the integer twin of bias_tile
{
  i32x16 acc;
  if (FHEVC_X_NOBIAS) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    return acc;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int4 v = *reinterpret_cast<const int4*>(b32 + 8 * g + 4 * h);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
  return acc;
}
__device__ __forceinline__ void pool_h_i8(i32x16& acc0, const i32x16& acc1)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) acc0[k] = max(acc0[k], acc1[k]);
}
__device__ __forceinline__ void pool_v_i8(i32x16& top, const i32x16& acc0, const i32x16& acc1)
{
#pragma unroll
  for (int k = 0; k < 16; ++k) top[k] = max(max(top[k], acc0[k]), acc1[k]);
}
// reg i -> channel (i & 3) + 8 g + 4 h, g = i >> 2: plane g >> 1, bytes 8 (g & 1) + 4 h ..; dst = the lane's position in plane 0 + 4 h
template <int MODE>
__device__ __forceinline__ void conv2_requant_store_i8m(const i32x16& m, unsigned char* dst, int shift)
{
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<unsigned*>(dst + (g >> 1) * A2_PLANE + 8 * (g & 1)) = requant4_i8(m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3], shift, MODE);
}
// conv3: channel 32 tile + 8 g + 4 h + (i & 3) of a position lives in logical 16-B chunk 2 tile + (g >> 1), bytes 8 (g & 1) + 4 h ..;
// dst = the position's 64-byte row + 4 h; psw = chunk swizzle of the position
template <int MODE>
__device__ __forceinline__ void conv3_store_i8m(const i32x16& acc, unsigned char* dst, int tile, int psw, int shift)
{
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<unsigned*>(dst + (((2 * tile + (g >> 1)) ^ psw) << 4) + 8 * (g & 1)) =
        requant4_i8(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], shift, MODE);
}

// ---- MFMA chains with a register ring of B fragments ---------------------------------------------------------
// Each v_mfma needs one ds_read_b128 (its im2col fragment).  hipcc places the read right in front of its MFMA and
// waits for it, exposing the LDS latency 18 times per chain; instead the reads run RING steps ahead of their MFMA
// and sched_group_barrier pins the (MFMA, DS read, a few VALU of the previous unit's epilogue) interleave.
#ifndef FHEVC_RING2
#define FHEVC_RING2 4
#endif
#ifndef FHEVC_RING3
#define FHEVC_RING3 4
#endif
constexpr int RING = FHEVC_RING2;    // conv2's ring (the register-tightest phase)
#ifndef FHEVC_RING2_I8
#define FHEVC_RING2_I8 2             // the i8 form's conv2: two fragments ahead are enough with three waves per SIMD, and the 8 registers
#endif                               // saved take the kernel's last spills away (0.4227 -> 0.4186 ms)
constexpr int RINGI = FHEVC_RING2_I8;
constexpr int RING3 = FHEVC_RING3;   // conv3's ring
static_assert(12 % RING == 0, "conv2 hands its ring slots from unit to unit unchanged");
template <int VALU_PER_MFMA>
__device__ __forceinline__ void sched_chain18()
{
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    // 1 MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                    // 1 DS read
    if (VALU_PER_MFMA > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_MFMA, 0);  // VALU filler
  }
}
// conv2 MFMA half-chain = pre-pool outputs (dy, dx = 0) and (dy, dx = 1) of 32 pooled positions (two pooled rows x 16
// columns), K = 9 taps x 16 channels.  Output column 2 pc + dx and tap kx read input column 2 pc + c, c = dx + kx in 0..3:
// 12 fragments (3 input rows x 4 columns c), 18 MFMAs (c = 1, 2 feed both accumulators).  base = the lane's pooled column
// in input row (first pre-pool row + dy).  The fragment ring runs ACROSS half-chains: the last RING reads fetch the
// first fragments of the next one, so only the first of a phase exposes the LDS latency.
__device__ __forceinline__ const unsigned char* conv2_frag(const unsigned char* base, int f)
{
  return base + (f / 4) * A1_ROW + (((f % 4) & 1) ? 0 : A1_EVEN) + ((f % 4) >> 1) * 16;
}
template <bool FIRST, bool LAST>
__device__ __forceinline__ void conv2_half(const unsigned char* base, const unsigned char* next, const bf16x8 (&wA2)[9], bf16x8 (&ring)[RING],
                                           const f32x16& binit, f32x16& acc0, f32x16& acc1)
{
  acc0 = binit;  // C operand of the first MFMA of each accumulator: the bias costs nothing
  acc1 = binit;
  if (FIRST) {
#pragma unroll
    for (int f = 0; f < RING; ++f) ring[f] = lds_frag(conv2_frag(base, f));
  }
#pragma unroll
  for (int f = 0; f < 12; ++f) {  // 12 % RING == 0: ring slots line up from half-chain to half-chain
    const int ky = f / 4, c = f % 4;
    const bf16x8 b = ring[f % RING];
    if (c < 3) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wA2[ky * 3 + c]), __builtin_bit_cast(f16x8, b), acc0, 0, 0, 0);
    if (c > 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wA2[ky * 3 + c - 1]), __builtin_bit_cast(f16x8, b), acc1, 0, 0, 0);
    if (f + RING < 12) ring[f % RING] = lds_frag(conv2_frag(base, f + RING));
    else if (!LAST) ring[f % RING] = lds_frag(conv2_frag(next, f + RING - 12));
  }
}
// conv3 MFMA chain of one output row: 9 fragments (taps), 18 MFMAs (two M tiles per fragment).  Rows k = 0..7 of a wave are
// y = y0 + 4 * (k >> 1) + (k & 1); fragment g = 9 k + tap of the phase lives at a compile-time offset from the lane's base
// pointer, and the 4-deep ring runs across the rows.
__device__ __forceinline__ constexpr int conv3_frag_off(int g)
{
  const int k = g / 9, t = g % 9;
  return ((4 * (k >> 1) + (k & 1) + t / 3) * A2_PITCH + t % 3) * 16;
}
template <int K>
__device__ __forceinline__ void conv3_row(const unsigned char* base, const bf16x8 (&wA3)[18], bf16x8 (&ring)[RING3],
                                          const f32x4& b0, const f32x4& b1, f32x4& acc0, f32x4& acc1)
{
  acc0 = b0;  // C operand of the first MFMA of each accumulator: the bias costs nothing
  acc1 = b1;
  if (K == 0) {
#pragma unroll
    for (int g = 0; g < RING3; ++g) ring[g] = lds_frag(base + conv3_frag_off(g));
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int g = 9 * K + t;
    const bf16x8 b = ring[g % RING3];
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wA3[t]), __builtin_bit_cast(f16x8, b), acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wA3[9 + t]), __builtin_bit_cast(f16x8, b), acc1, 0, 0, 0);
    if (g + RING3 < 72) ring[g % RING3] = lds_frag(base + conv3_frag_off(g + RING3));
  }
}
__device__ __forceinline__ void sched_row18()
{
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // 2 MFMAs (the two M tiles of a fragment)
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
  }
}
// ---- the i8 variant's chains (v_mfma_i32_32x32x32_i8: K = 32 bytes, lanes 0-31 hold K 0-15, lanes 32-63 K 16-31) ----------
__device__ __forceinline__ i32x16 mfma_i8(const bf16x8& a, const bf16x8& b, const i32x16& c)
{
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
// conv2 (round 4: 4.5 K steps of 32 instead of 6): rows ky = 0, 1 of the taps pair up along ky -- fragment c = input column c of the lane's pooled position,
// the lane halves read rows one A1_ROW apart (the lane's base pointer carries it) --, row ky = 2 pairs up along kx with the lane halves TWO columns apart (same
// parity plane, next slot): fragment 4 = columns (0, 2), fragment 5 = columns (1, 3) of row 2 (`d2`: the lane's offset to them; conflict-free like the others).
// A fragments: 0-2 = taps (ky = h, kx), 3 = [(2, 0) | (2, 2)], 4 = [(2, 1) | 0], 5 = [0 | (2, 1)]: dx = 0 multiplies fragment 4 by A3 and fragment 5 by A4,
// dx = 1 (one column on) fragment 5 by A3 and fragment 4 by A5.  Per half-chain 6 ds_read_b128 and 10 MFMAs for the two accumulators where the row-pair form (rounds 2-3: K rows 2 and a
// phantom row 3) needed 8 and 12; measured -3.1 % of the launch as a sensitivity build before it was written (profiles/r04_ab_conv2_k_steps.log).
__device__ __forceinline__ int conv2_lane_i8(int h) { return h ? A1_ROW + 16 : 2 * A1_ROW; }
__device__ __forceinline__ const unsigned char* conv2_frag_i8(const unsigned char* base, int d2, int f)
{
  if (f < 4) return base + ((f & 1) ? 0 : A1_EVEN) + (f >> 1) * 16;
  return base + d2 + (f == 4 ? A1_EVEN : 0);
}
constexpr int C2F = 6;  // fragments per half-chain
// R0: the ring slot of this half-chain's fragment 0
template <bool FIRST, bool LAST, int R0>
__device__ __forceinline__ void conv2_half_i8(const unsigned char* base, const unsigned char* next, int d2, const bf16x8 (&wA2)[9], bf16x8 (&ring)[RINGI],
                                              const int* bias, int h, i32x16& acc0, i32x16& acc1)
{
  const i32x16 binit = bias_tile_i8(bias, h);  // read per half-chain, not held across the phase (168 registers)
  acc0 = binit;
  acc1 = binit;
  if (FIRST) {
#pragma unroll
    for (int f = 0; f < RINGI; ++f) ring[(R0 + f) % RINGI] = lds_frag(conv2_frag_i8(base, d2, f));
  }
#pragma unroll
  for (int f = 0; f < C2F; ++f) {
    const bf16x8 b = ring[(R0 + f) % RINGI];
    if (f < 3) acc0 = mfma_i8(wA2[f], b, acc0);                    // rows 0, 1: columns 0 .. 2 for dx = 0
    if (f >= 1 && f < 4) acc1 = mfma_i8(wA2[f - 1], b, acc1);      //            columns 1 .. 3 for dx = 1
    if (f == 4) { acc0 = mfma_i8(wA2[3], b, acc0); acc1 = mfma_i8(wA2[5], b, acc1); }  // row 2, columns (0, 2): taps kx = 0, 2 of dx = 0; tap kx = 1 of dx = 1
    if (f == 5) { acc0 = mfma_i8(wA2[4], b, acc0); acc1 = mfma_i8(wA2[3], b, acc1); }  //        columns (1, 3): tap kx = 1 of dx = 0; taps kx = 0, 2 of dx = 1
    if (f + RINGI < C2F) ring[(R0 + f) % RINGI] = lds_frag(conv2_frag_i8(base, d2, f + RINGI));
    else if (!LAST) ring[(R0 + f) % RINGI] = lds_frag(conv2_frag_i8(next, d2, f + RINGI - C2F));
  }
}
template <int VALU_PER_GROUP>
__device__ __forceinline__ void sched_chain12_i8()
{
#define FHEVC_G(NM)                                                                          \
  __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);                                        \
  __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                         \
  if (VALU_PER_GROUP > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GROUP, 0);
  FHEVC_G(1) FHEVC_G(2) FHEVC_G(2) FHEVC_G(1) FHEVC_G(2) FHEVC_G(2)  // fragments 0 .. 5
#undef FHEVC_G
}
// conv3: one MFMA = one tap x all 32 input channels (lane half = activation plane) x the wave's 32 output channels x TWO output
// rows (B column n: x = n & 15 of row y + 8 (n >> 4): rows EIGHT apart are 8 * 288 B = 9 * 256 B apart, so the 16-lane groups of
// a ds_read_b128 -- {0-3, 12-15 of one row} + {4-11 of the other} -- tile a 256-byte bank row): 9 MFMAs and 9 ds_read_b128 per
// row pair, 4 row pairs per wave (rows y0 + {0, 1, 4, 5} + {0, 8}).  Two row pairs run interleaved (two independent accumulators: a
// lone chain of 9 would wait for each MFMA's result): super-chain S = row pairs 2S, 2S+1, fragment g = 18 S + 2 tap + (pair & 1)
__device__ __forceinline__ constexpr int conv3_pair_row_i8(int k) { return 4 * (k >> 1) + (k & 1); }
__device__ __forceinline__ constexpr int conv3_frag_off_i8(int g)
{
  const int k = 2 * (g / 18) + (g & 1), t = (g % 18) / 2;
  return ((conv3_pair_row_i8(k) + t / 3) * A2_PITCH + t % 3) * 16;
}
template <int S>
__device__ __forceinline__ void conv3_pairs_i8(const unsigned char* base, const bf16x8 (&wA3)[18], bf16x8 (&ring)[RING3], const int* bias, int h,
                                               i32x16& acc0, i32x16& acc1)
{
  const i32x16 binit = bias_tile_i8(bias, h);  // read per super-chain, not held across the phase (168 registers)
  acc0 = binit;
  acc1 = binit;
  if (S == 0) {
#pragma unroll
    for (int g = 0; g < RING3; ++g) ring[g] = lds_frag(base + conv3_frag_off_i8(g));
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int g = 18 * S + 2 * t;
    acc0 = mfma_i8(wA3[t], ring[g % RING3], acc0);
    if (g + RING3 < 36) ring[g % RING3] = lds_frag(base + conv3_frag_off_i8(g + RING3));
    acc1 = mfma_i8(wA3[t], ring[(g + (FHEVC_X_C3_HALF ? 2 : 1)) % RING3], acc1);
    if (!FHEVC_X_C3_HALF && g + 1 + RING3 < 36) ring[(g + 1) % RING3] = lds_frag(base + conv3_frag_off_i8(g + 1 + RING3));
  }
}
template <int VALU_PER_GROUP>
__device__ __forceinline__ void sched_pairs18_i8()
{
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (VALU_PER_GROUP > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GROUP, 0);
  }
}
// conv3 of the 16-bit form on v_mfma_f32_32x32x16_f16 (FHEVC_F16_CONV3_32): one MFMA = one tap x 16 input channels (lane half h = plane
// 2 c2 + h) x the wave's 32 output channels x TWO output rows 8 apart (the mapping of the i8 form: conflict-free ds_read_b128 groups);
// 18 MFMAs and 18 reads per row pair, two row pairs interleaved (two accumulators), fragment g = 36 S + 2 step + (pair & 1)
__device__ __forceinline__ constexpr int conv3_frag_off_f32(int g)
{
  const int k = 2 * (g / 36) + (g & 1), st = (g % 36) / 2, t = st >> 1, c2 = st & 1;
  return 2 * c2 * A2_PLANE + ((conv3_pair_row_i8(k) + t / 3) * A2_PITCH + t % 3) * 16;
}
template <int S>
__device__ __forceinline__ void conv3_pairs_f32(const unsigned char* base, const bf16x8 (&wA3)[18], bf16x8 (&ring)[RING3], const f32x16& binit,
                                                f32x16& acc0, f32x16& acc1)
{
  acc0 = binit;
  acc1 = binit;
  if (S == 0) {
#pragma unroll
    for (int g = 0; g < RING3; ++g) ring[g] = lds_frag(base + conv3_frag_off_f32(g));
  }
#pragma unroll
  for (int st = 0; st < 18; ++st) {
    const int g = 36 * S + 2 * st;
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wA3[st]), __builtin_bit_cast(f16x8, ring[g % RING3]), acc0, 0, 0, 0);
    if (g + RING3 < 72) ring[g % RING3] = lds_frag(base + conv3_frag_off_f32(g + RING3));
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wA3[st]), __builtin_bit_cast(f16x8, ring[(g + 1) % RING3]), acc1, 0, 0, 0);
    if (g + 1 + RING3 < 72) ring[(g + 1) % RING3] = lds_frag(base + conv3_frag_off_f32(g + 1 + RING3));
  }
}
template <int VALU_PER_GROUP>
__device__ __forceinline__ void sched_pairs36_f32()
{
#pragma unroll
  for (int i = 0; i < 36; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (VALU_PER_GROUP > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER_GROUP, 0);
  }
}
// its epilogue: reg i -> channel 32 tile + (i & 3) + 8 g + 4 h of the lane's position: v_cvt_pk_u8_f32 (floor + clamp under the round-down
// mode), a - 128, one dword per g at chunk 2 tile + (g >> 1), bytes 8 (g & 1) + 4 h
__device__ __forceinline__ void conv3_store_f32(const f32x16& acc, unsigned char* dst, int tile, int psw)
{
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<unsigned*>(dst + (((2 * tile + (g >> 1)) ^ psw) << 4) + 8 * (g & 1)) =
        u8x4_floor_clamp(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]) ^ 0x80808080u;
}
// One thread's 16 samples of a CTU (picture row ld_row, columns 16*ld_seg..) fetched ahead of use.  fast = 0:
// picture edge or unaligned plane, P0 falls back to guarded scalar loads.
struct Prefetched { uint4 a, b; int fast; };
// CTU coordinates of a work item: frame, CTU row inside the band, CTU column.  A workgroup walks its items with a
// mixed-radix increment (advance) instead of dividing the work index: integer division costs ~40 VALU instructions.
struct CtuPos { int f, ry, cx; };
__device__ __forceinline__ CtuPos advance(CtuPos c, const CtuPos& step, int band_rows, int ctus_x)
{
  c.cx += step.cx; c.ry += step.ry; c.f += step.f;
  if (c.cx >= ctus_x) { c.cx -= ctus_x; c.ry++; }
  if (c.ry >= band_rows) { c.ry -= band_rows; c.f++; }
  return c;
}
template <bool ZERO>
__device__ __forceinline__ Prefetched prefetch_ctu(const FhevcFrames& F, bool live, CtuPos c, int ld_row, int ld_seg)
{
  Prefetched p;
  p.fast = 0;  // a, b stay undefined unless fast (stage_ctu reads them only then): no zero-fill instructions on the way
  if (ZERO) { p.a = make_uint4(0, 0, 0, 0); p.b = make_uint4(0, 0, 0, 0); }  // the fused source Hadamard reads them either way
  if (!live) return p;
  const int pf = c.f, pcy = F.row_begin + c.ry, pcx = c.cx;
  const int py = pcy * 64 + ld_row, px0 = pcx * 64 + ld_seg * 16;
  if (py >= F.height || px0 + 16 > F.width) return p;
  const long long base = (long long)pf * F.frame_stride + (long long)py * F.stride + px0;
  if (F.sample_bytes == 2) {
    const int16_t* src = reinterpret_cast<const int16_t*>(F.luma) + base;
    if (reinterpret_cast<uintptr_t>(src) & 15) return p;
    p.a = *reinterpret_cast<const uint4*>(src);
    p.b = *reinterpret_cast<const uint4*>(src + 8);
  } else {
    const uint8_t* src = reinterpret_cast<const uint8_t*>(F.luma) + base;
    if (reinterpret_cast<uintptr_t>(src) & 15) return p;
    p.a = *reinterpret_cast<const uint4*>(src);
  }
  p.fast = 1;
  return p;
}

typedef __attribute__((ext_vector_type(2))) short s16x2;
// ---- source Hadamard fused into the CTU load (TEncCu::updateCtuDataISlice, TEncCu.cpp:1230-1343; the stand-alone twin is
// k_hadamard.hip) ----------------------------------------------------------------------------------------------------------
// A thread holds 16 samples of one picture row = one row of two horizontally adjacent 8x8 blocks (registers 0-3: block A,
// 4-7: block B, two samples per register); the 8 rows of a block sit in the 8 lanes lane ^ {4, 8, 16} of the wave.  The
// 2-D Walsh-Hadamard transform runs on packed 16-bit VALU (coefficients below 2^15 through five stages up to 10 bit; DC is
// never formed): horizontal distances 2 and 4 between registers, vertical distances 1, 2, 4 across lanes with a register-
// PAIR exchange (DPP row shifts under bank masks for lane ^ 4 and lane ^ 8, v_permlane16_swap for lane ^ 16): after the
// exchange one register of the pair holds the butterfly of P in the lanes of the upper rows and of Q in the lower ones, so
// every lane does useful work; where a coefficient ends up does not matter for a sum of absolute values.  The last stage
// (horizontal distance 1, inside a register) is folded into the sum: |a + b| + |a - b| = 2 max(|a|, |b|); the block's DC term
// |a + b| of that pair is the plain sum of its samples, subtracted at the end (TEncCu.cpp:1319).
__device__ __forceinline__ unsigned hpk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b))); }
__device__ __forceinline__ unsigned hpk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b))); }
template <int SHR, int SHL, int BANK_HI, int BANK_LO>
__device__ __forceinline__ void had_exchange(unsigned& P, unsigned& Q)
{
  const unsigned p2 = (unsigned)__builtin_amdgcn_update_dpp((int)P, (int)Q, SHR, 0xF, BANK_HI, false);  // lanes of the upper rows take Q from lane - k
  const unsigned q2 = (unsigned)__builtin_amdgcn_update_dpp((int)Q, (int)P, SHL, 0xF, BANK_LO, false);  // lanes of the lower rows take P from lane + k
  P = hpk_add(p2, q2);
  Q = hpk_sub(p2, q2);
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
// -> the sum over this wave's 16 whole 8x8 blocks of (sum |WHT8x8| - |DC| + 2) >> 2, in every lane.  Threads without samples
// (rows or columns outside the picture) pass zeros: their blocks then contribute (0 + 2) >> 2 = 0.
// the thread's 16 samples as eight registers of two 16-bit samples each (uint8 planes: bytes -> 16-bit pairs, selector 0x0C = zero byte)
template <int SB>
__device__ __forceinline__ void hadamard_samples(const Prefetched& pre, unsigned (&r)[8])
{
  if (SB == 2) {
    r[0] = pre.a.x; r[1] = pre.a.y; r[2] = pre.a.z; r[3] = pre.a.w; r[4] = pre.b.x; r[5] = pre.b.y; r[6] = pre.b.z; r[7] = pre.b.w;
  } else {
    r[0] = __builtin_amdgcn_perm(0u, pre.a.x, 0x0C010C00u); r[1] = __builtin_amdgcn_perm(0u, pre.a.x, 0x0C030C02u);
    r[2] = __builtin_amdgcn_perm(0u, pre.a.y, 0x0C010C00u); r[3] = __builtin_amdgcn_perm(0u, pre.a.y, 0x0C030C02u);
    r[4] = __builtin_amdgcn_perm(0u, pre.a.z, 0x0C010C00u); r[5] = __builtin_amdgcn_perm(0u, pre.a.z, 0x0C030C02u);
    r[6] = __builtin_amdgcn_perm(0u, pre.a.w, 0x0C010C00u); r[7] = __builtin_amdgcn_perm(0u, pre.a.w, 0x0C030C02u);
  }
}
__device__ __forceinline__ int wave_src_hadamard_core(unsigned (&r)[8], int lane);
template <int SB>
__device__ __forceinline__ int wave_src_hadamard(const Prefetched& pre, int lane)
{
  unsigned r[8];
  hadamard_samples<SB>(pre, r);
  return wave_src_hadamard_core(r, lane);
}
// the transform itself: straight-line code (the pipelined kernel places it inside conv2's MFMA chain)
__device__ __forceinline__ int wave_src_hadamard_core(unsigned (&r)[8], int lane)
{
  int t[2];  // per block: 2 * (this lane's share of sum |coefficients| / 2) - (this lane's share of the block's sample sum)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    unsigned* q = r + 4 * k;
    const unsigned ss = hpk_add(hpk_add(q[0], q[1]), hpk_add(q[2], q[3]));   // sample sum of the half-row: DC's share
    const int dc = (int)(ss & 0xFFFFu) + (int)(ss >> 16);
    // horizontal distance 2 (registers 0|1, 2|3) and 4 (0|2, 1|3)
    unsigned a = hpk_add(q[0], q[1]), b = hpk_sub(q[0], q[1]), c = hpk_add(q[2], q[3]), d = hpk_sub(q[2], q[3]);
    q[0] = hpk_add(a, c); q[2] = hpk_sub(a, c); q[1] = hpk_add(b, d); q[3] = hpk_sub(b, d);
    // vertical distance 1, 2 (lane ^ 4, lane ^ 8: inside a DPP row of 16 lanes), 4 (lane ^ 16: between DPP rows)
    had_exchange<0x114, 0x104, 0xA, 0x5>(q[0], q[1]);
    had_exchange<0x114, 0x104, 0xA, 0x5>(q[2], q[3]);
    had_exchange<0x118, 0x108, 0xC, 0x3>(q[0], q[2]);
    had_exchange<0x118, 0x108, 0xC, 0x3>(q[1], q[3]);
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const u32x2_t sw = __builtin_amdgcn_permlane16_swap(q[j], q[j + 1], false, false);
      q[j] = hpk_add(sw.x, sw.y);
      q[j + 1] = hpk_sub(sw.x, sw.y);
    }
    unsigned acc = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const s16x2 v = __builtin_bit_cast(s16x2, q[j]);
      const unsigned m = __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, (s16x2)(-v)));
      acc += max(m & 0xFFFFu, m >> 16);
    }
    t[k] = (int)(2 * acc) - dc;
  }
  int hb = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {  // the block's 8 lanes: two row rotations inside the DPP row, one exchange between rows
    int v = t[k];
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);   // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);   // row_ror:8
    v += __shfl_xor(v, 16);
    hb += (v + 2) >> 2;                                              // xCalcHADs8x8_ISlice: (sum + 2) >> 2 per block
  }
  // one lane per block pair ((lane & 0x1C) == 0): lanes 0-3 and 32-35; sum them
  int w = ((lane & 0x1C) == 0) ? hb : 0;
  w += __builtin_amdgcn_update_dpp(0, w, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
  w += __builtin_amdgcn_update_dpp(0, w, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
  return __builtin_amdgcn_readlane(w, 0) + __builtin_amdgcn_readlane(w, 32);
}

// two int16 samples of one dword -> two 8-bit samples (load_sample8 on both halves): max(v, 0) first, so that the
// rounding shift may be a logical one ((0 + rnd) >> s = 0 like every negative sample), then min(255)
__device__ __forceinline__ unsigned sample8_pair(unsigned w, int shift, unsigned rnd2)
{
  const s16x2 zero = { 0, 0 }, top = { 255, 255 };
  s16x2 t = __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), zero);
  const u16x2 sh = { (unsigned short)shift, (unsigned short)shift };
  t = __builtin_bit_cast(s16x2, (u16x2)((__builtin_bit_cast(u16x2, t) + __builtin_bit_cast(u16x2, rnd2)) >> sh));
  t = __builtin_elementwise_min(t, top);
  return __builtin_bit_cast(unsigned, t);
}

// Stage one CTU into LDS (region R2): 8-bit samples as bf16, two picture rows per dword, halo = 128 (the centre).
// halo coordinates: hy = row + 1, hx = col + 1; dword (hy >> 1) * IN_PITCH + hx, half (hy & 1)
template <bool HALO = true>
__device__ __forceinline__ void stage_ctu(unsigned char* lds, int r2_off, const Prefetched& pre, const FhevcFrames& F, CtuPos c,
                                          int tid, int ld_row, int ld_seg, int shift_in, unsigned in_cells)
{
  unsigned short* inh = reinterpret_cast<unsigned short*>(lds + r2_off);
  const int hy = ld_row + 1;
  unsigned short* dst = inh + 2 * ((hy >> 1) * IN_PITCH + ld_seg * 16 + 1) + (hy & 1);
  if (pre.fast) {
    if (F.sample_bytes == 2) {
      const unsigned wds[8] = { pre.a.x, pre.a.y, pre.a.z, pre.a.w, pre.b.x, pre.b.y, pre.b.z, pre.b.w };
      const unsigned rnd2 = shift_in > 0 ? (0x00010001u << (shift_in - 1)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {  // load_sample8 on both halves of a dword with packed 16-bit VALU, v_cvt_f32_ubyte0/2
        const unsigned t = sample8_pair(wds[j], shift_in, rnd2);
        dst[4 * j] = (unsigned short)(__float_as_uint((float)(t & 0xFF)) >> 16);
        dst[4 * j + 2] = (unsigned short)(__float_as_uint((float)((t >> 16) & 0xFF)) >> 16);
      }
    } else {
      const unsigned wds[4] = { pre.a.x, pre.a.y, pre.a.z, pre.a.w };
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned v = (wds[j >> 2] >> (8 * (j & 3))) & 0xFF;
        dst[2 * j] = (unsigned short)(__float_as_uint((float)v) >> 16);
      }
    }
  } else {  // picture edge or unaligned plane: guarded scalar loads
    const int f = c.f, cy = F.row_begin + c.ry, cx = c.cx;
    const int py = cy * 64 + ld_row, px0 = cx * 64 + ld_seg * 16;
    const long long base = (long long)f * F.frame_stride + (long long)py * F.stride + px0;
    const bool row_ok = py < F.height;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
      int v = 128;
      if (row_ok && px0 + j < F.width) {
        if (F.sample_bytes == 2) v = load_sample8(reinterpret_cast<const int16_t*>(F.luma) + base + j, shift_in);
        else v = (int)reinterpret_cast<const uint8_t*>(F.luma)[base + j];
      }
      dst[2 * j] = (unsigned short)(__float_as_uint((float)v) >> 16);
    }
  }
  // input halo: 66*66 - 64*64 = 260 two-byte cells of bf16(128) (the pipelined kernel's tile region is never overwritten: it fills the halo once)
  if (HALO) {
    asm volatile("" : "+v"(in_cells));
    *reinterpret_cast<unsigned short*>(lds + (in_cells & 0xFFFF)) = 0x4300;
    if (tid < 260 - 256) *reinterpret_cast<unsigned short*>(lds + (in_cells >> 16)) = 0x4300;
  }
}
// LDS byte offsets of the halo cells a thread zeroes (computed once per kernel: the index arithmetic with its three-way
// divergence cost ~700 cycles per CTU when it ran inside the phases).  Cell e of: the conv1 output halo (264 x 16 B),
// the conv2 output halo (272 x 16 B), the input tile halo (260 x 2 B).
template <bool I8>
__device__ __forceinline__ int a1_halo_off(int e)
{
  const int pl = e / 132, k0 = e - pl * 132;
  int y, x;
  if (k0 < 34) { y = 0; x = k0; }
  else if (k0 < 68) { y = 33; x = k0 - 34; }
  else { const int k = k0 - 68; y = 1 + (k >> 1); x = (k & 1) ? 33 : 0; }
  return R1_OFF + pl * Lds<I8>::A1_PLANE + y * A1_ROW + ((x & 1) ? 0 : A1_EVEN) + (x >> 1) * 16;
}
template <bool I8>
__device__ __forceinline__ int a2_halo_off(int e)
{
  const int pl = e / 68, k0 = e - pl * 68;
  int y, x;
  if (k0 < 18) { y = 0; x = k0; }
  else if (k0 < 36) { y = 17; x = k0 - 18; }
  else { const int k = k0 - 36; y = 1 + (k >> 1); x = (k & 1) ? 17 : 0; }
  return Lds<I8>::R2_OFF + pl * A2_PLANE + (y * A2_PITCH + x) * 16;
}
template <bool I8>
__device__ __forceinline__ int in_halo_off(int e)
{
  int y, x;
  if (e < 66) { y = 0; x = e; }
  else if (e < 132) { y = 65; x = e - 66; }
  else { const int k = e - 132; y = 1 + (k >> 1); x = (k & 1) ? 65 : 0; }
  return Lds<I8>::R2_OFF + 2 * (2 * ((y >> 1) * IN_PITCH + x) + (y & 1));
}
// a thread's two cells of each halo, packed as (first | second << 16); all offsets are below 64 KiB.  The A1 and A2 halos have
// 132 and 68 cells per plane: with fewer cells than threads (the i8 variant) the surplus threads rewrite the last cell
struct HaloCells { unsigned a1, a2, in; };
template <bool I8>
__device__ __forceinline__ HaloCells halo_cells(int tid)
{
  constexpr int N1 = 132 * Lds<I8>::A1_PLANES, N2 = 68 * Lds<I8>::A2_PLANES;
  HaloCells hc;
  hc.a1 = (unsigned)a1_halo_off<I8>(min(tid, N1 - 1)) | ((unsigned)a1_halo_off<I8>(min(tid + 256, N1 - 1)) << 16);
  hc.a2 = (unsigned)a2_halo_off<I8>(min(tid, N2 - 1)) | ((unsigned)a2_halo_off<I8>(min(tid + 256, N2 - 1)) << 16);
  hc.in = (unsigned)in_halo_off<I8>(tid) | ((unsigned)in_halo_off<I8>(min(tid + 256, 259)) << 16);
  return hc;
}
// "activation 0" into the halo of the conv1 output (region R1, shared with the conv3 output): 132 positions per plane
template <bool I8>
__device__ __forceinline__ void zero_a1_halo(unsigned char* lds, int tid, unsigned cells)
{
  asm volatile("" : "+v"(cells));  // unpack the two offsets here, every time: hoisted out of the CTU loop they are two more kernel-long registers
  constexpr unsigned Z = Lds<I8>::HALO_FILL;
  *reinterpret_cast<uint4*>(lds + (cells & 0xFFFF)) = make_uint4(Z, Z, Z, Z);
  if (tid < 132 * Lds<I8>::A1_PLANES - 256) *reinterpret_cast<uint4*>(lds + (cells >> 16)) = make_uint4(Z, Z, Z, Z);
}

// in-kernel stamp (diagnostic build only): s_memtime with its own lgkmcnt wait, fenced against reordering
__device__ __forceinline__ unsigned long long stamp()
{
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// Per-phase re-derivation of the thread coordinates from an opaque copy of threadIdx.x: everything computed from them
// (LDS addresses, lane classes) then lives only inside its phase instead of in ~20 kernel-long registers, which the
// conv2 phase (weights 108 + accumulators 64 + fragment ring) had pushed into scratch (reloads stall on vmcnt(0)).
#define FHEVC_PHASE_IDS                                   \
  int tid = (int)threadIdx.x;                             \
  asm volatile("" : "+v"(tid));                           \
  if (TRIO) tid &= 255;                                   \
  const int lane = tid & 63, wave = tid >> 6;             \
  const int r = lane & 31, h = lane >> 5;                 \
  (void)r; (void)h; (void)wave;

// STAMPS = true is a separate diagnostic instantiation (fhevc_debug_cnn_phase_cycles): wave 0 of every workgroup
// adds the cycles of each phase (incl. the barrier that ends it) into d_stamps[vbidx * FHEVC_STAMP_SLOTS + phase].
// HAD: also write the per-CTU source Hadamard (d_had), computed from the samples the kernel loads anyway (one pass over the frame):
//      1 = on packed 16-bit VALU from the prefetched samples (up to 10 bit: wave_src_hadamard), 2 = on the bf16 MFMA from the staged
//      tile (8-bit content only: the tile holds the samples rounded to 8 bits; src_hadamard_mfma)
// I8: conv2 and conv3 on v_mfma_i32_32x32x32_i8 with the activations as signed bytes (see Lds); the same integers come out.
// I8 = 2: the same with the short requant forms (requant4_i8: conv2 mode 1, conv3 mode 2), where the host found them valid
// TRIO = 1 (round 4): the CU's three workgroups as ONE workgroup of 768 threads = three groups of four waves, each group a "workgroup" of the form above with its own
// third of the LDS and its own CTUs, but behind COMMON barriers and one barrier interval apart: at any time the three groups are in three different
// intervals of the four (conv1 | conv2 | conv3 | heads + staging), so the intervals that load the matrix pipe never coincide -- three independent workgroups fall
// into step and run them together (profiles/r04_depth_kernel_intervals.md).  Every group executes the same number of barriers: g empty ones before its first
// CTU, 2 - g after its last, and four in an iteration that has no CTU left.
template <bool STAMPS, int HAD, int ARITH, int TRIO = 0>
__global__ __launch_bounds__(TRIO ? 768 : 256, TRIO ? 1 : (ARITH ? FHEVC_I8_WG_PER_CU : 2)) void fhevc_cnn_depth_kernel(FhevcFrames F, FhevcCnnWeights W,
                                                                  uint8_t* __restrict__ d_depth, int32_t* __restrict__ d_had,
                                                                  int32_t* __restrict__ d_logits,
                                                                  uint32_t* __restrict__ d_flags,
                                                                  unsigned long long* __restrict__ d_stamps,
                                                                  uint8_t* __restrict__ d_depth_max, int margin_split, int margin_stop)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
  const int group = TRIO ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;
  unsigned char* const lds = lds_all + (TRIO ? group * Lds<ARITH != 0>::LDS_BYTES : 0);
  // the grid as the CTU walk sees it: three virtual workgroups per real one in the TRIO form
  const int vgrid = TRIO ? 3 * (int)gridDim.x : (int)gridDim.x;
  const int vbidx = TRIO ? group * (int)gridDim.x + (int)blockIdx.x : (int)blockIdx.x;
  // fp32 rounding mode of this wave: toward -inf (MODE.fp_round, hwreg id 1, bits 1:0 <- 2).  Every fp32 operation of
  // the kernel is exact (integers scaled by powers of two), so only v_cvt_pk_u8_f32 notices: it becomes floor + clamp
  __builtin_amdgcn_s_setreg((1 << 11) | 1, 2);
  const int tid = TRIO ? (int)(threadIdx.x & 255) : (int)threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  constexpr bool I8 = ARITH != 0, FASTRQ = ARITH == 2;
  int prio_slot = 0;   // which of the CU's workgroups this is, from where its LDS allocation starts (see FHEVC_SLOT_PRIO)
  if (FHEVC_SLOT_PRIO && I8) {
    const unsigned la = __builtin_amdgcn_s_getreg((31 << 11) | 6);   // HW_REG_LDS_ALLOC: base [11:0], size [20:12], both in 256-byte granules
    const unsigned base = la & 0xFFFu, sz = (la >> 12) & 0x1FFu;
    prio_slot = base >= 2 * sz && sz ? 2 : (base >= sz && sz ? 1 : 0);
    if (TRIO) prio_slot = group;
  }
  (void)prio_slot;
  if (FHEVC_SLOT_PRIO == 1 || FHEVC_SLOT_PRIO == 3) FHEVC_SETPRIO_DYN(prio_slot)
  constexpr bool MFMA_HEADS = FHEVC_MFMA_HEADS_F16 || I8;  // the two smaller FC heads as an i8 MFMA GEMM (P4)
  (void)FASTRQ;
  using L = Lds<I8>;
  const HaloCells hc = halo_cells<I8>(tid);  // three registers for the life of the kernel
  // head bias + QP prior on "split" (class 1): uniform, lives in scalar registers
  const int hb64a = W.bhead[0], hb64b = W.bhead[1] + W.bhead[6 + 0 * 52 + F.qp];
  const int hb32a = W.bhead[2], hb32b = W.bhead[3] + W.bhead[6 + 1 * 52 + F.qp];
  const int hb16a = W.bhead[4], hb16b = W.bhead[5] + W.bhead[6 + 2 * 52 + F.qp];
  // ---- resident weight fragments (A operands) ----
  // conv1's two fragments are only live from the end of a CTU's heads to its conv1: they are re-fetched (L2-hot) under
  // the heads of the previous CTU, which frees 8 registers in the conv2 phase, the tightest one
  const uint4* frag1p = W.frag + FHEVC_FRAG_CONV1 + lane;
  bf16x8 wA1a = __builtin_bit_cast(bf16x8, frag1p[0]);
  bf16x8 wA1b = __builtin_bit_cast(bf16x8, frag1p[64]);
  bf16x8 wA2[9];   // the i8 variant uses 6 of them (2 row pairs x 3 columns of taps) ...
  bf16x8 wA3[18];  // ... and 9 of these (one per tap): 60 weight registers instead of 108
  const int tile3 = wave & 1;
  if (I8) {
#pragma unroll
    for (int s = 0; s < 6; ++s) wA2[s] = __builtin_bit_cast(bf16x8, W.frag_i8[FHEVC_FRAGI8_CONV2 + s * 64 + lane]);
#pragma unroll
    for (int s = 0; s < 9; ++s) wA3[s] = __builtin_bit_cast(bf16x8, W.frag_i8[FHEVC_FRAGI8_CONV3 + (tile3 * 9 + s) * 64 + lane]);
  } else {
#pragma unroll
    for (int s = 0; s < 9; ++s) wA2[s] = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV2 + s * 64 + lane]);
#pragma unroll
    for (int s = 0; s < 18; ++s) wA3[s] = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV3 + (tile3 * 18 + s) * 64 + lane]);
  }
  const int shift2 = W.shift[1], shift3 = W.shift[2];  // (the i8 variant's requant shifts and forms: scalar registers)
  (void)shift2; (void)shift3;

  float* biasL = reinterpret_cast<float*>(lds + L::BIAS_OFF);
  int* logitL = reinterpret_cast<int*>(lds + L::LOGIT_OFF);
  if (tid < 112) {  // all pre-scaled (exact: integer * 2^-s); b1 is the accumulator init of conv1, whose weights carry 2^-s1
    if (I8 && tid >= 16) {  // conv2 / conv3 biases of the i8 variant: int32, + 128 * (sum of the filter's weights)
      reinterpret_cast<int*>(biasL)[tid] = W.bias_i8[tid];
    } else {
      float b = W.bias[tid];
      b *= (tid < 16 ? W.scale[0] : (tid < 48 ? W.scale[1] : W.scale[2]));
      biasL[tid] = b;
    }
  }
  // head weights stay in LDS for the life of the workgroup; the four 16-B chunks of a 64-B row are XOR-swizzled by
  // the row so that the 16 lanes of a ds_read_b128 group (a 4x4 block of positions) hit 16 distinct slots
  for (int i = tid; i < (MFMA_HEADS ? 8192 : 18432) / 16; i += 256) {
    const int row = i >> 2, c = i & 3;
    const int sw = (i < 2 * 4096 * 2 / 16) ? ((row >> 3) & 3) : ((row >> 2) & 3);  // wh64, wh32: rows of 8; wh16: rows of 4
    *reinterpret_cast<uint4*>(lds + L::HEADW_OFF + row * 64 + ((c ^ sw) << 4)) = reinterpret_cast<const uint4*>(W.whead)[i];
  }
  if (MFMA_HEADS) {
    // the 16- and 32-level heads run on v_mfma_i32_16x16x64_i8 (P4; both arithmetic forms: conv3's output is bytes in either): B operand = [K step j = position (py, px) of a 16x16
    // block][column n][64 channels], columns 0, 1 = the 16-level classes, 2 + 2 sub + class = the 32-level weights of a block at
    // sub-position sub = (by & 1, bx & 1) of its quadrant: 16 steps x 10 columns x 64 B = the same 10 240 B, in MFMA order
    for (int i = tid; i < HEADM_BYTES / 16; i += 256) {
      const int j = i / 40, rem = i - j * 40, n = rem >> 2, kg = rem & 3;
      int src;
      if (n < 2) src = 16384 + (n * 16 + j) * 64;
      else {
        const int sub = (n - 2) >> 1, cls = n & 1, py = j >> 2, px = j & 3;
        src = 8192 + (cls * 64 + ((sub >> 1) * 4 + py) * 8 + (sub & 1) * 4 + px) * 64;
      }
      *reinterpret_cast<uint4*>(lds + L::HEADW_OFF + HEADM_OFF + i * 16) = *reinterpret_cast<const uint4*>(W.whead + src + kg * 16);
    }
  }

  if (MFMA_HEADS && tid < 64) {
    // the MFMA heads' scatter table (P4): lane (n = column, rg = row group) holds the partial sums of blocks (by = rg, bx = 0..3) for weight column n.  Columns
    // 0, 1: every block's 16-level logit; column 2 + 2 sub + class: the 32-level logit of the block's quadrant, only from the blocks at sub-position sub
    const int n = tid & 15, rg = tid >> 4, subn = (n - 2) >> 1;
    const bool is16 = n < 2, is32 = n >= 2 && n < 10 && ((rg & 1) == (subn >> 1));
    unsigned a[4];
    for (int i = 0; i < 4; ++i) {
      const bool act = is16 || (is32 && ((i & 1) == (subn & 1)));
      const int idx = is16 ? 2 * (5 + 4 * rg + i) + n : 2 * (1 + (rg >> 1) * 2 + (i >> 1)) + (n & 1);
      a[i] = act ? (unsigned)(L::LOGIT_OFF + 4 * idx) : (unsigned)(L::HEADX_OFF + 512 + 4 * tid);
    }
    *reinterpret_cast<uint2*>(lds + L::HEADX_OFF + tid * 8) = make_uint2(a[0] | (a[1] << 16), a[2] | (a[3] << 16));
  }
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  const int shift_in = F.bit_depth - 8;

  unsigned long long tsum[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tprev = 0;
#define FHEVC_STAMP(k)                                     \
  if (STAMPS) {                                            \
    const unsigned long long tn = stamp();                 \
    tsum[k] += tn - tprev;                                 \
    tprev = tn;                                            \
  }
  // the in-kernel clock (MI355X_MICROARCH.md, 'DVFS give-back' item 6): shader cycles (s_memtime) over the constant 100 MHz counter (s_memrealtime),
  // stamped once around the workgroup's whole CTU loop -> d_stamps[.. + 8], [.. + 9]
  unsigned long long tclk0 = 0, treal0 = 0;
  if (STAMPS) {
    tprev = stamp();
    tclk0 = tprev;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(treal0)::"memory");
  }

  // this thread's 16 samples of a CTU: picture row (tid >> 2), columns 16 * (tid & 3) ..
  const int ld_row = tid >> 2, ld_seg = tid & 3;
  // XCD-aware order (speed only): blockIdx % 8 share an L2, give each XCD a contiguous run of CTUs per sweep so that the
  // 128-byte lines shared by horizontally adjacent CTUs (HM's unaligned margins) are fetched into one L2, not two
  const int vblock = (vgrid & 7) ? vbidx : ((vbidx & 7) * (vgrid >> 3) + (vbidx >> 3));
  CtuPos pos, step;  // this workgroup's current item and its stride (the only divisions of the kernel)
  {
    const int vb = min(vblock, total - 1), g = vgrid;
    pos.f = vb / per_frame; pos.ry = (vb - pos.f * per_frame) / F.ctus_x; pos.cx = (vb - pos.f * per_frame) - pos.ry * F.ctus_x;
    step.f = g / per_frame; step.ry = (g - step.f * per_frame) / F.ctus_x; step.cx = (g - step.f * per_frame) - step.ry * F.ctus_x;
  }
  Prefetched pre = prefetch_ctu<HAD == 1>(F, vblock < total, pos, ld_row, ld_seg);
  // the heads' per-thread LDS rows (see P4): position (y, x) of quadrant `wave`, 16x16 block lane >> 4
  uint2 head_addr;
  {
    const int q = wave, blk = lane >> 4;
    const int y = (q >> 1) * 8 + (blk >> 1) * 4 + ((lane >> 2) & 3), x = (q & 1) * 8 + (blk & 1) * 4 + (lane & 3);
    const unsigned psw = (x >> 2) & 3, sw16 = y & 3, sw32 = y & 3, sw64 = (y >> 1) & 3;  // chunk swizzles: a3 rows (P3), weight rows (prologue above)
    const unsigned a_0 = (unsigned)(y * 16 + x) * 64 + (psw << 4);
    const unsigned w16_0 = (unsigned)((y & 3) * 4 + (x & 3)) * 64 + (sw16 << 4);
    const unsigned w32_0 = (unsigned)((y & 7) * 8 + (x & 7)) * 64 + (sw32 << 4);
    const unsigned w64_0 = (unsigned)((y >> 1) * 8 + (x >> 1)) * 64 + (sw64 << 4);  // 2x2 sum pool: four positions share a weight row
    head_addr = make_uint2(a_0 | (w16_0 << 16), w32_0 | (w64_0 << 16));
  }
  // the i8 form's MFMA heads: lane (m = lane & 15, kg = lane >> 4) reads, as the A operand, channels 16 kg .. of position (py = wave, px = step)
  // of block m = (by, bx) (chunk swizzle of that position = bx), and as the B operand column min(m, 9) of K step 4 wave + px
  unsigned headm_addr = 0;
  if (MFMA_HEADS) {
    const int m = lane & 15, kg = lane >> 4, by = m >> 2, bx = m & 3;
    const unsigned am = (unsigned)((4 * by + wave) * 16 + 4 * bx) * 64 + ((unsigned)(kg ^ bx) << 4);
    const unsigned bm = (unsigned)(HEADM_OFF + 4 * wave * HEADM_STEP + min(m, 9) * 64 + 16 * kg);
    headm_addr = am | (bm << 16);
  }
  (void)headm_addr;
  int had_set = 0;  // which of the two sets of per-wave Hadamard sums belongs to the CTU in flight
  if (vblock < total) {  // prologue: first CTU of this workgroup
    stage_ctu(lds, L::R2_OFF, pre, F, pos, tid, ld_row, ld_seg, shift_in, hc.in);
    zero_a1_halo<I8>(lds, tid, hc.a1);
    if (HAD == 1) {
      const int hs = F.sample_bytes == 2 ? wave_src_hadamard<2>(pre, lane) : wave_src_hadamard<1>(pre, lane);
      if (lane == 0) logitL[56 + wave] = hs;
    }
  }
  __syncthreads();
  FHEVC_STAMP(0)
  // conv1's first LDS reads (bias half-tile, first fragment) are issued one phase early, before the depth phase of the
  // previous CTU: that phase and conv1's start are both latency-bound, their LDS round trips now overlap
  float4 early_b0, early_b1;
  bf16x8 early_bq;
#define FHEVC_CONV1_EARLY_READS                                                                          \
  {                                                                                                      \
    FHEVC_PHASE_IDS                                                                                      \
    early_b0 = *reinterpret_cast<const float4*>(biasL + 8 * h);                                          \
    early_b1 = *reinterpret_cast<const float4*>(biasL + 8 * h + 4);                                      \
    const unsigned char* p = lds + L::R2_OFF + (2 * r + 2 * h) * 4 + wave * (IN_PITCH * 4);              \
    const uint2 lo = *reinterpret_cast<const uint2*>(p);                                                 \
    const uint2 hi = *reinterpret_cast<const uint2*>(p + IN_PITCH * 4);                                  \
    early_bq = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));                           \
  }
  FHEVC_CONV1_EARLY_READS
  // HAD == 2: the source Hadamard's A operands (constant +-1 / 0 fragments) travel with them: five L2-hot loads per CTU, requested
  // one phase before their MFMAs; an opaque lane offset keeps the loads inside the loop (hoisted they would be 20 kernel-long registers)
  bf16x8 wH[5];
#define FHEVC_HAD_FRAG_LOADS                                                                                                              \
  if (HAD == 2) {                                                                                                                         \
    unsigned hoff = (unsigned)(threadIdx.x & 63) * 16u + (unsigned)((threadIdx.x >> 6) & 1) * (5u * 64u * 16u); /* M tile = wave & 1 */  \
    asm volatile("" : "+v"(hoff));                                                                                                        \
    const unsigned char* hf = reinterpret_cast<const unsigned char*>(W.frag + FHEVC_FRAG_HAD) + hoff;                                     \
    _Pragma("unroll") for (int st = 0; st < 5; ++st) wH[st] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(hf + st * 64 * 16)); \
  }
  FHEVC_HAD_FRAG_LOADS

  // Per CTU: P1 conv1 | P2 conv2 | P3 conv3 (next CTU's samples are requested) | P4 heads + next CTU staged into LDS |
  // P5 depth map + conv1 halo re-zeroed -- four barriers; P5 runs into the next P1 without one (disjoint LDS).
  if (TRIO) {
    for (int i = 0; i < group; ++i) __syncthreads();  // one barrier interval behind the group before
  }
  const int trio_iters = TRIO ? (total + vgrid - 1) / vgrid : 0;  // the same for all three groups: the barriers must pair up
  for (int work = vblock, it = 0; TRIO ? it < trio_iters : work < total; work += vgrid, ++it) {
    if (TRIO && work >= total) {  // (only ever the last iteration) no CTU left for this group: keep the others' barriers company
      __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads();
      continue;
    }
    const int f = pos.f, cy = F.row_begin + pos.ry, cx = pos.cx;
    const CtuPos next = advance(pos, step, band_rows, F.ctus_x);

    // (the samples of this CTU were staged into LDS during the previous iteration's P4, or by the prologue)

    // ====== P1: conv1 (1 -> 16): one MFMA per 32 positions x 2 rows, K = 4x3 window, fused maxpool + requant ======
    {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(0)
      // unit = one pooled row of 32 positions (picture rows 2yp, 2yp+1, all 64 columns); lane (n, h): pooled column n,
      // K slots = the 4x4 input window of the 2x2 pre-pool outputs: lanes h=0 hold window columns 0-1, h=1 columns 2-3,
      // each column as two row-pair dwords -> the fragment is two ds_read_b64, all 16 K slots carry data
      const unsigned char* inb = lds + L::R2_OFF + (2 * r + 2 * h) * 4;
      f32x16 bias1;  // reg i -> channel (i&3) + 4*((i>>2)&1) + 8h, the same for both pre-pool rows and both MFMAs
      {
        const float4 b0 = early_b0, b1 = early_b1;  // read from LDS before the previous CTU's depth phase (or by the prologue)
        bias1[0] = b0.x; bias1[1] = b0.y; bias1[2] = b0.z; bias1[3] = b0.w;
        bias1[4] = b1.x; bias1[5] = b1.y; bias1[6] = b1.z; bias1[7] = b1.w;
#pragma unroll
        for (int i = 0; i < 8; ++i) bias1[8 + i] = bias1[i];
      }
      auto frag1 = [](const unsigned char* p) {
        const uint2 lo = *reinterpret_cast<const uint2*>(p);
        const uint2 hi = *reinterpret_cast<const uint2*>(p + IN_PITCH * 4);
        return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
      };
      // this wave's units: pooled rows wave, wave + 4, ...: both addresses advance by constants (no per-unit index arithmetic)
      const unsigned char* fp = inb + wave * (IN_PITCH * 4);
      unsigned char* dp = lds + R1_OFF + (I8 ? 8 * h : h * A1_PLANE) + (wave + 1) * A1_ROW + (((r + 1) & 1) ? 0 : A1_EVEN) + ((r + 1) >> 1) * 16;  // column r (halo +1)
      bf16x8 bq = early_bq;  // = frag1(fp), in flight since before the previous CTU's depth phase
      if (HAD == 2) {
        // Source Hadamard of THIS CTU (TEncCu::updateCtuDataISlice / xCalcHADs8x8_ISlice, TEncCu.cpp:1230-1343) as one small GEMM on the
        // bf16 MFMA, fed from the staged tile (8-bit content: the tile holds the samples themselves, exact in bf16): D[m][n] = sum_k
        // A[m][k] B[k][n], m = one of the 64 coefficients (u, v) of the 2-D Walsh-Hadamard transform (M tile = wave & 1), n = one of
        // the 64 blocks (N tile = wave >> 1: block rows 4 nt .., n = 8 (by & 3) + bx), k = a sample of the block.  The tile stores
        // picture rows 2P - 1 (low half) and 2P (high half) in one dword, so a block's 8 rows span FIVE row pairs P = 4 by + s, the
        // first and the last half used: 5 K steps of 16 slots = (2 rows) x (4 columns per lane half), the slots of rows -1 and 8 and
        // the whole DC row carry zero weights (host: build_weight_image).  Products are +-sample, sums stay below 2^24: exact.
        // Then sum |D| over the lane's 16 rows (abs is a source modifier), add the other lane half (permlane32 swap), and the
        // (M tile, block) partial goes to LDS; wave 1 finishes the CTU's total in the depth phase.  ~25 VALU instead of ~130.
        // (Runs BEFORE the conv1 units: its A operands were requested one phase early, like conv1's first reads, and are dead after the
        // five MFMAs -- at the end of the phase they would be live across the conv1 units and spill.)
        const int mt = wave & 1, nt = wave >> 1;
        const unsigned* hb = reinterpret_cast<const unsigned*>(lds + L::R2_OFF) + (16 * nt + 4 * (r >> 3)) * IN_PITCH + 8 * (r & 7) + 1 + 4 * h;
        f32x16 hacc;
#pragma unroll
        for (int k = 0; k < 16; ++k) hacc[k] = 0.0f;
#pragma unroll
        for (int st = 0; st < 5; ++st) {
          const unsigned* q = hb + st * IN_PITCH;  // 4-byte aligned only (column 8 bx + 1): two ds_read2_b32
          const bf16x8 smp = __builtin_bit_cast(bf16x8, make_uint4(q[0], q[1], q[2], q[3]));
          hacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wH[st], smp, hacc, 0, 0, 0);
        }
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += __builtin_fabsf(hacc[k]);
        const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(t), __float_as_uint(t), false, false);
        t = __uint_as_float(sw.x) + __uint_as_float(sw.y);   // rows of both lane halves: the (M tile, block) partial, in every lane
        if (h == 0) reinterpret_cast<float*>(lds + L::HADP_OFF)[128 * had_set + 64 * mt + 32 * nt + r] = t;
      }
#pragma unroll FHEVC_CONV1_UNROLL
      for (int i = 0; i < 8; ++i) {
        const f32x16 acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1a, bq, bias1, 0, 0, 0);
        const f32x16 acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1b, bq, bias1, 0, 0, 0);
        fp += 4 * IN_PITCH * 4;
        bq = frag1(fp);  // next unit's fragment travels during the epilogue (the last one is redundant: it reads past the tile, inside R2)
        if (I8) conv1_store_i8(acc0, acc1, dp);
        else conv1_store(acc0, acc1, dp);
        dp += 4 * A1_ROW;
        if (FHEVC_CONV1_MFMA_FIRST && (i & 1)) {  // (experiment) both units' MFMAs of an unrolled pair ahead of their epilogues
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 60, 0);
        }
      }
      FHEVC_PRIO_OFF(0)
    }
    __syncthreads();
    FHEVC_STAMP(1)

    // ================= P2: conv2 (16 -> 32), K = 9 taps x 16 ch, fused maxpool + requant =================
    if constexpr (I8) {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(1)
      constexpr unsigned Z = L::HALO_FILL;  // the input tile is dead: "activation 0" into the A2 halo (68 positions x 2 planes)
      unsigned a2cells = hc.a2;
      asm volatile("" : "+v"(a2cells));
      *reinterpret_cast<uint4*>(lds + (a2cells & 0xFFFF)) = make_uint4(Z, Z, Z, Z);
      // the 32- and 16-level logits start from their head biases: P4's MFMA heads add to them (the previous CTU's readers are two barriers back)
      if (tid < 40) logitL[2 + tid] = tid < 8 ? ((tid & 1) ? hb32b : hb32a) : ((tid & 1) ? hb16b : hb16a);
      // lane -> pooled position as in the 16-bit form below; the lane half picks the input row of a tap pair (ky = 2 q + h)
      const int q = r >> 2;
      const int pr = (q ^ (q >> 1) ^ (q >> 2)) & 1, pc = ((r >> 3) << 2) | (r & 3);
      const unsigned char* a1p = lds + R1_OFF + (2 * pr + h) * A1_ROW + pc * 16;
      unsigned char* a2dst = lds + L::R2_OFF + ((pr + 1) * A2_PITCH + pc + 1) * 16 + 4 * h;  // pooled row pr of unit 0, plane 0, bytes 4h ..
      const int u0 = wave, u1 = wave + 4;
      const unsigned char* h00 = a1p + (4 * u0) * A1_ROW;  // half-chains: (u0, dy 0), (u0, dy 1), (u1, dy 0), (u1, dy 1)
      const unsigned char* h10 = a1p + (4 * u1) * A1_ROW;
      const int* b2t = reinterpret_cast<const int*>(biasL) + 16;
      i32x16 t0, t1, a0, a1;
      bf16x8 ring[RINGI];
      const int c2l = conv2_lane_i8(h);
      conv2_half_i8<true, false, 0>(h00, h00 + A1_ROW, c2l, wA2, ring, b2t, h, t0, t1);
      if (FHEVC_I8_C2_SCHED) __builtin_amdgcn_sched_group_barrier(0x100, RINGI + 4, 0);  // bias tile + the ring's first fragments go out together
      if (FHEVC_I8_C2_SCHED) sched_chain12_i8<0>();
      conv2_half_i8<false, false, C2F % RINGI>(h00 + A1_ROW, h10, c2l, wA2, ring, b2t, h, a0, a1);
      pool_h_i8(t0, t1);
      if (FHEVC_I8_C2_SCHED) sched_chain12_i8<FHEVC_I8_C2_FILL_POOL>();
      if (FHEVC_I8_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      pool_v_i8(t0, a0, a1);
      if (FHEVC_I8_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      conv2_half_i8<false, false, (2 * C2F) % RINGI>(h10, h10 + A1_ROW, c2l, wA2, ring, b2t, h, t1, a0);
      conv2_requant_store_i8m<FASTRQ ? 1 : 0>(t0, a2dst + (2 * u0) * A2_PITCH * 16, shift2);
      if (FHEVC_I8_C2_SCHED) sched_chain12_i8<FHEVC_I8_C2_FILL_REQUANT>();
      conv2_half_i8<false, true, (3 * C2F) % RINGI>(h10 + A1_ROW, h10 + A1_ROW, c2l, wA2, ring, b2t, h, a1, t0);
      pool_h_i8(t1, a0);
      if (FHEVC_I8_C2_SCHED) sched_chain12_i8<FHEVC_I8_C2_FILL_POOL>();
      if (FHEVC_I8_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      pool_v_i8(t1, a1, t0);
      conv2_requant_store_i8m<FASTRQ ? 1 : 0>(t1, a2dst + (2 * u1) * A2_PITCH * 16, shift2);
      FHEVC_PRIO_OFF(1)
    } else {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(1)
      // the input tile is dead: zero the A2 halo (68 positions x 4 planes) while conv2 fills the interior
      *reinterpret_cast<uint4*>(lds + (hc.a2 & 0xFFFF)) = make_uint4(0, 0, 0, 0);
      if (tid < 272 - 256) *reinterpret_cast<uint4*>(lds + (hc.a2 >> 16)) = make_uint4(0, 0, 0, 0);
      if (MFMA_HEADS && tid < 40) logitL[2 + tid] = tid < 8 ? ((tid & 1) ? hb32b : hb32a) : ((tid & 1) ? hb16b : hb16a);  // see the i8 form above
      // lane -> pooled position: the 32 columns of a B operand are two pooled rows (pr) x 16 pooled columns (pc), assigned so
      // that each 16-lane group of a ds_read_b128 ({0-3, 12-15, 20-27} and {4-11, 16-19, 28-31}) is one row's 16 columns =
      // 256 contiguous bytes (conflict-free whatever the row pitch)
      // = lanes 4q..4q+3 with an even number of bits set in q: pr = parity of q, pc = 4 * (q >> 1) + (r & 3)
      const int q = r >> 2;
      const int pr = (q ^ (q >> 1) ^ (q >> 2)) & 1, pc = ((r >> 3) << 2) | (r & 3);
      // unit u = pooled rows 2u, 2u+1 = pre-pool rows 4u .. 4u+3 (input halo rows 4u .. 4u+5); this lane: input row 4u + 2pr + dy + ky
      const unsigned char* a1p = lds + R1_OFF + h * A1_PLANE + (2 * pr) * A1_ROW + pc * 16;
      unsigned char* a2dst = lds + L::R2_OFF + ((pr + 1) * A2_PITCH + pc + 1) * 16 + 8 * h;  // pooled row pr of unit 0, plane 0
      const int u0 = wave, u1 = wave + 4;
      const unsigned char* h00 = a1p + (4 * u0) * A1_ROW;  // half-chains: (u0, dy 0), (u0, dy 1), (u1, dy 0), (u1, dy 1)
      const unsigned char* h10 = a1p + (4 * u1) * A1_ROW;
      const f32x16 b2t = bias_tile(biasL + 16, h);  // pre-scaled conv2 biases in the accumulator layout, once per phase
      f32x16 t0, t1, a0, a1;
      bf16x8 ring[RING];
      conv2_half<true, false>(h00, h00 + A1_ROW, wA2, ring, b2t, t0, t1);
      __builtin_amdgcn_sched_group_barrier(0x100, RING + 4, 0);  // bias tile + the ring's first fragments go out together
      sched_chain18<0>();
      conv2_half<false, false>(h00 + A1_ROW, h10, wA2, ring, b2t, a0, a1);
      conv2_pool_h(t0, t1);
      sched_chain18<FHEVC_F16_C2_FILL_POOL>();
      if (FHEVC_F16_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      conv2_pool_v(t0, a0, a1);
      if (FHEVC_F16_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      conv2_half<false, false>(h10, h10 + A1_ROW, wA2, ring, b2t, t1, a0);  // (the registers of t1, a0, a1 are free again)
      conv2_requant_store(t0, a2dst + (2 * u0) * A2_PITCH * 16);
      sched_chain18<FHEVC_F16_C2_FILL_REQUANT>();
      conv2_half<false, true>(h10 + A1_ROW, h10 + A1_ROW, wA2, ring, b2t, a1, t0);
      conv2_pool_h(t1, a0);
      sched_chain18<FHEVC_F16_C2_FILL_POOL>();
      if (FHEVC_F16_C2_FENCE) __builtin_amdgcn_sched_barrier(0);
      conv2_pool_v(t1, a1, t0);
      conv2_requant_store(t1, a2dst + (2 * u1) * A2_PITCH * 16);
      FHEVC_PRIO_OFF(1)
    }
    __syncthreads();
    FHEVC_STAMP(2)

    // ================= P3: conv3 (32 -> 64), K = 9 taps x 32 ch, requant to u8 =================
    pre = prefetch_ctu<HAD == 1>(F, work + vgrid < total, next, ld_row, ld_seg);  // next CTU's samples travel under conv3 and the heads
    if constexpr (I8) {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(2)
      const int x = lane & 15, rs = (lane >> 4) & 1;  // B column n = lane & 31: row 8 rs of the pair, position x; lane half h = activation plane
      const int y0 = 2 * (wave >> 1);                 // this wave's row pairs: y0 + {0, 1, 4, 5} + {0, 8}
      const unsigned char* a2 = lds + L::R2_OFF + h * A2_PLANE + ((y0 + 8 * rs) * A2_PITCH + x) * 16;
      const int psw = (x >> 2) & 3;                   // chunk swizzle of the a3 rows (see the 16-bit form below)
      unsigned char* a3dst = lds + A3_OFF + ((y0 + 8 * rs) * 16 + x) * 64 + 4 * h;
      const int* b3p = reinterpret_cast<const int*>(biasL) + 48 + 32 * tile3;
      bf16x8 ring[RING3];
      i32x16 p0, p1, q0, q1;
      conv3_pairs_i8<0>(a2, wA3, ring, b3p, h, p0, p1);
      if (FHEVC_I8_C3_SCHED) __builtin_amdgcn_sched_group_barrier(0x100, RING3 + 4, 0);
      if (FHEVC_I8_C3_SCHED) sched_pairs18_i8<0>();
      conv3_pairs_i8<1>(a2, wA3, ring, b3p, h, q0, q1);
      if (FHEVC_I8_C3_SCHED) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      conv3_store_i8m<FASTRQ ? 2 : 0>(p0, a3dst + conv3_pair_row_i8(0) * 1024, tile3, psw, shift3);
      conv3_store_i8m<FASTRQ ? 2 : 0>(p1, a3dst + conv3_pair_row_i8(1) * 1024, tile3, psw, shift3);
      if (FHEVC_I8_C3_SCHED) sched_pairs18_i8<FHEVC_I8_C3_FILL>();
      conv3_store_i8m<FASTRQ ? 2 : 0>(q0, a3dst + conv3_pair_row_i8(2) * 1024, tile3, psw, shift3);
      conv3_store_i8m<FASTRQ ? 2 : 0>(q1, a3dst + conv3_pair_row_i8(3) * 1024, tile3, psw, shift3);
      FHEVC_PRIO_OFF(2)
    } else if constexpr (FHEVC_F16_CONV3_32 != 0) {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(2)
      const int x = lane & 15, rs = (lane >> 4) & 1;  // B column n = lane & 31: row 8 rs of the pair, position x; lane half h = plane 2 c2 + h
      const int y0 = 2 * (wave >> 1);                 // this wave's row pairs: y0 + {0, 1, 4, 5} + {0, 8}
      const unsigned char* a2 = lds + L::R2_OFF + h * A2_PLANE + ((y0 + 8 * rs) * A2_PITCH + x) * 16;
      const int psw = (x >> 2) & 3;
      unsigned char* a3dst = lds + A3_OFF + ((y0 + 8 * rs) * 16 + x) * 64 + 4 * h;
      const f32x16 b3t = bias_tile(biasL + 48 + 32 * tile3, h);
      bf16x8 ring[RING3];
      f32x16 p0, p1, q0, q1;
      conv3_pairs_f32<0>(a2, wA3, ring, b3t, p0, p1);
      __builtin_amdgcn_sched_group_barrier(0x100, RING3 + 4, 0);
      sched_pairs36_f32<0>();
      conv3_pairs_f32<1>(a2, wA3, ring, b3t, q0, q1);
      conv3_store_f32(p0, a3dst + conv3_pair_row_i8(0) * 1024, tile3, psw);
      conv3_store_f32(p1, a3dst + conv3_pair_row_i8(1) * 1024, tile3, psw);
      sched_pairs36_f32<1>();
      conv3_store_f32(q0, a3dst + conv3_pair_row_i8(2) * 1024, tile3, psw);
      conv3_store_f32(q1, a3dst + conv3_pair_row_i8(3) * 1024, tile3, psw);
      FHEVC_PRIO_OFF(2)
    } else {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(2)
      const int x = lane & 15, kg = lane >> 4;  // B column = position x of the row, K group = activation plane kg; D rows 4 kg ..
      const int y0 = 2 * (wave >> 1);           // this wave's rows: y0 + {0, 1, 4, 5, 8, 9, 12, 13} (+ {2, 3, ...} for the next wave pair)
      const unsigned char* a2 = lds + L::R2_OFF + kg * A2_PLANE + (y0 * A2_PITCH + x) * 16;
      // a3 row of a position p = 64 B = four 16-B chunks; chunk c is stored at c ^ ((p >> 2) & 3) so that the 16
      // lanes of a ds_read_b128 group in the heads (consecutive positions, same logical chunk) hit 16 distinct slots
      const int psw = (x >> 2) & 3;
      unsigned char* a3dst = lds + A3_OFF + (y0 * 16 + x) * 64 + 4 * kg;
      const f32x4 b30 = *reinterpret_cast<const f32x4*>(biasL + 48 + 32 * tile3 + 4 * kg);       // M tile 0: channels 32 t + 4 kg + i
      const f32x4 b31 = *reinterpret_cast<const f32x4*>(biasL + 48 + 32 * tile3 + 16 + 4 * kg);  // M tile 1
      bf16x8 ring[RING3];
      f32x4 p0, p1, q0, q1;
#define FHEVC_ROW_OFF(k) ((4 * ((k) >> 1) + ((k) & 1)) * 1024)
      conv3_row<0>(a2, wA3, ring, b30, b31, p0, p1);
      if (FHEVC_F16_C3_SCHED) __builtin_amdgcn_sched_group_barrier(0x100, RING3 + 2, 0);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<1>(a2, wA3, ring, b30, b31, q0, q1);
      conv3_store(p0, p1, a3dst + FHEVC_ROW_OFF(0), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<2>(a2, wA3, ring, b30, b31, p0, p1);
      conv3_store(q0, q1, a3dst + FHEVC_ROW_OFF(1), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<3>(a2, wA3, ring, b30, b31, q0, q1);
      conv3_store(p0, p1, a3dst + FHEVC_ROW_OFF(2), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<4>(a2, wA3, ring, b30, b31, p0, p1);
      conv3_store(q0, q1, a3dst + FHEVC_ROW_OFF(3), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<5>(a2, wA3, ring, b30, b31, q0, q1);
      conv3_store(p0, p1, a3dst + FHEVC_ROW_OFF(4), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<6>(a2, wA3, ring, b30, b31, p0, p1);
      conv3_store(q0, q1, a3dst + FHEVC_ROW_OFF(5), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_row<7>(a2, wA3, ring, b30, b31, q0, q1);
      conv3_store(p0, p1, a3dst + FHEVC_ROW_OFF(6), tile3, psw);
      if (FHEVC_F16_C3_SCHED) sched_row18();
      conv3_store(q0, q1, a3dst + FHEVC_ROW_OFF(7), tile3, psw);
      FHEVC_PRIO_OFF(2)
#undef FHEVC_ROW_OFF
    }
    if (HAD == 1) {  // the next CTU's source Hadamard from its samples in flight: VALU work at the tail of the MFMA-bound phase
      FHEVC_PHASE_IDS
      const int hs = F.sample_bytes == 2 ? wave_src_hadamard<2>(pre, lane) : wave_src_hadamard<1>(pre, lane);
      if (lane == 0) logitL[56 + 4 * (had_set ^ 1) + wave] = hs;
    }
    __syncthreads();
    FHEVC_STAMP(3)

    // ================= P4: FC heads on v_dot4_i32_i8 (int8 weights resident in LDS, activations a - 128) =================
    {  // conv1's fragments for the next CTU: issued first so that they have landed before the depth phase, whose spill
       // reloads wait for vmcnt(0)
      unsigned lane_off = (unsigned)(threadIdx.x & 63) * 16u;
      asm volatile("" : "+v"(lane_off));  // opaque 32-bit lane offset on a scalar base: keeps the re-fetch inside the loop, no kernel-long pointer pair
      const unsigned char* fp = reinterpret_cast<const unsigned char*>(W.frag + FHEVC_FRAG_CONV1) + lane_off;
      wA1a = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fp));
      wA1b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(fp + 64 * 16));
      if (I8) {  // the i8 variant runs three workgroups per CU on 168 registers: conv2's fragments are dead during conv3 and come back here too
        const unsigned char* f2 = reinterpret_cast<const unsigned char*>(W.frag_i8 + FHEVC_FRAGI8_CONV2) + lane_off;
#pragma unroll
        for (int s = 0; s < 6; ++s) wA2[s] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(f2 + s * 64 * 16));
      }
    }
    if constexpr (MFMA_HEADS) {
      FHEVC_PHASE_IDS
      FHEVC_PRIO_ON(3)
      // 16- and 32-level heads as ONE GEMM on v_mfma_i32_16x16x64_i8: rows = the 16 blocks of the CTU, K step = one position of a block x
      // 64 channels (this wave: the four positions of block row py = wave), columns = the weight variants (see the prologue).  D: lane
      // (n = lane & 15, rg = lane >> 4), register i = block (by = rg, bx = i), column n: partial sums over this wave's positions
      unsigned hm = headm_addr, ha0 = head_addr.x, ha1 = head_addr.y;
      asm volatile("" : "+v"(hm), "+v"(ha0), "+v"(ha1));
      const unsigned char* hw = lds + L::HEADW_OFF;
      const unsigned char* ap = lds + A3_OFF + (hm & 0xFFFFu);
      const unsigned char* bp = hw + (hm >> 16);
      i32x4 hacc = { 0, 0, 0, 0 };
#pragma unroll
      for (int px = 0; px < 4; ++px)
        hacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(ap + px * 64), *reinterpret_cast<const i32x4*>(bp + px * HEADM_STEP), hacc, 0, 0, 0);
      // 64-level head (unique weights per position: nothing for an MFMA to reuse) on v_dot4_i32_i8, the lane = one position of quadrant `wave`
      const unsigned a_0 = ha0 & 0xFFFFu, w64_0 = ha1 >> 16;
      int s64a = 0, s64b = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint4 a = *reinterpret_cast<const uint4*>(lds + A3_OFF + (a_0 ^ (qq << 4)));
        const unsigned char* w64 = hw + HEAD64_OFF + (w64_0 ^ (qq << 4));
        const uint4 d0 = *reinterpret_cast<const uint4*>(w64), d1 = *reinterpret_cast<const uint4*>(w64 + 4096);
        s64a = sdot4(a.x, d0.x, s64a); s64a = sdot4(a.y, d0.y, s64a); s64a = sdot4(a.z, d0.z, s64a); s64a = sdot4(a.w, d0.w, s64a);
        s64b = sdot4(a.x, d1.x, s64b); s64b = sdot4(a.y, d1.y, s64b); s64b = sdot4(a.z, d1.z, s64b); s64b = sdot4(a.w, d1.w, s64b);
      }
      // the MFMA's partial sums join the logits (initialised to the head biases in P3) by LDS atomic adds: columns 0, 1 every block's
      // 16-level logit, column 2 + 2 sub + class the 32-level logit of the quadrant -- only from the blocks that sit at sub-position sub
      // (where each of the lane's four sums goes was worked out once per kernel: the scatter table in LDS -- four unconditional ds_add_u32 instead of four
      // predicated ones behind ~45 VALU / 30 SALU of lane classification per CTU)
      {
        const uint2 hx = *reinterpret_cast<const uint2*>(lds + L::HEADX_OFF + lane * 8);
        atomicAdd(reinterpret_cast<int*>(lds + (hx.x & 0xFFFFu)), hacc[0]);
        atomicAdd(reinterpret_cast<int*>(lds + (hx.x >> 16)), hacc[1]);
        atomicAdd(reinterpret_cast<int*>(lds + (hx.y & 0xFFFFu)), hacc[2]);
        atomicAdd(reinterpret_cast<int*>(lds + (hx.y >> 16)), hacc[3]);
      }
      const int r64a = dpp_row_sum(s64a), r64b = dpp_row_sum(s64b);
      const int q64a = __builtin_amdgcn_readlane(r64a, 0) + __builtin_amdgcn_readlane(r64a, 16) +
                       __builtin_amdgcn_readlane(r64a, 32) + __builtin_amdgcn_readlane(r64a, 48);
      const int q64b = __builtin_amdgcn_readlane(r64b, 0) + __builtin_amdgcn_readlane(r64b, 16) +
                       __builtin_amdgcn_readlane(r64b, 32) + __builtin_amdgcn_readlane(r64b, 48);
      if (lane == 0) *reinterpret_cast<int2*>(logitL + 44 + 2 * wave) = make_int2(q64a, q64b);  // the readers add the four waves' parts
      FHEVC_PRIO_OFF(3)
    } else
    {
      FHEVC_PHASE_IDS
      // wave = 32x32 quadrant q; 16-lane DPP row = one 16x16 block of it; lane bits [1:0] = x & 3, [3:2] = y & 3 (head_addr)
      const int q = wave, blk = lane >> 4;
      // the thread's a3 row and its three weight rows, each with its chunk swizzle applied: step qq reads chunk (qq ^ swizzle) of a
      // 64-byte row = the row's address XOR (qq << 4); four 16-bit offsets in two kernel-long registers instead of ~40 VALU per CTU
      unsigned ha0 = head_addr.x, ha1 = head_addr.y;
      asm volatile("" : "+v"(ha0), "+v"(ha1));
      const unsigned a_0 = ha0 & 0xFFFFu, w16_0 = ha0 >> 16, w32_0 = ha1 & 0xFFFFu, w64_0 = ha1 >> 16;
      const unsigned char* hw = lds + L::HEADW_OFF;
      int s16a = 0, s16b = 0, s32a = 0, s32b = 0, s64a = 0, s64b = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {  // 16 channels per step keeps this phase's register footprint small
        const uint4 a = *reinterpret_cast<const uint4*>(lds + A3_OFF + (a_0 ^ (qq << 4)));
        const unsigned char *w16 = hw + HEAD16_OFF + (w16_0 ^ (qq << 4)), *w32 = hw + HEAD32_OFF + (w32_0 ^ (qq << 4)), *w64 = hw + HEAD64_OFF + (w64_0 ^ (qq << 4));
        const uint4 b0 = *reinterpret_cast<const uint4*>(w16), b1 = *reinterpret_cast<const uint4*>(w16 + 1024);
        const uint4 c0 = *reinterpret_cast<const uint4*>(w32), c1 = *reinterpret_cast<const uint4*>(w32 + 4096);
        const uint4 d0 = *reinterpret_cast<const uint4*>(w64), d1 = *reinterpret_cast<const uint4*>(w64 + 4096);
        s16a = sdot4(a.x, b0.x, s16a); s16a = sdot4(a.y, b0.y, s16a); s16a = sdot4(a.z, b0.z, s16a); s16a = sdot4(a.w, b0.w, s16a);
        s16b = sdot4(a.x, b1.x, s16b); s16b = sdot4(a.y, b1.y, s16b); s16b = sdot4(a.z, b1.z, s16b); s16b = sdot4(a.w, b1.w, s16b);
        s32a = sdot4(a.x, c0.x, s32a); s32a = sdot4(a.y, c0.y, s32a); s32a = sdot4(a.z, c0.z, s32a); s32a = sdot4(a.w, c0.w, s32a);
        s32b = sdot4(a.x, c1.x, s32b); s32b = sdot4(a.y, c1.y, s32b); s32b = sdot4(a.z, c1.z, s32b); s32b = sdot4(a.w, c1.w, s32b);
        s64a = sdot4(a.x, d0.x, s64a); s64a = sdot4(a.y, d0.y, s64a); s64a = sdot4(a.z, d0.z, s64a); s64a = sdot4(a.w, d0.w, s64a);
        s64b = sdot4(a.x, d1.x, s64b); s64b = sdot4(a.y, d1.y, s64b); s64b = sdot4(a.z, d1.z, s64b); s64b = sdot4(a.w, d1.w, s64b);
      }
      // reductions: DPP inside the 16-lane row (= one 16x16 block), v_readlane across the four rows of the wave
      const int r16a = dpp_row_sum(s16a), r16b = dpp_row_sum(s16b);
      const int r32a = dpp_row_sum(s32a), r32b = dpp_row_sum(s32b);
      const int r64a = dpp_row_sum(s64a), r64b = dpp_row_sum(s64b);
      if ((lane & 15) == 0) {  // one owner per 16x16 block: no atomics
        const int bi = ((q >> 1) * 2 + (blk >> 1)) * 4 + (q & 1) * 2 + (blk & 1);
        *reinterpret_cast<int2*>(logitL + (5 + bi) * 2) = make_int2(r16a + hb16a, r16b + hb16b);  // plain stores: nothing to read back
      }
      const int q32a = __builtin_amdgcn_readlane(r32a, 0) + __builtin_amdgcn_readlane(r32a, 16) +
                       __builtin_amdgcn_readlane(r32a, 32) + __builtin_amdgcn_readlane(r32a, 48);
      const int q32b = __builtin_amdgcn_readlane(r32b, 0) + __builtin_amdgcn_readlane(r32b, 16) +
                       __builtin_amdgcn_readlane(r32b, 32) + __builtin_amdgcn_readlane(r32b, 48);
      const int q64a = __builtin_amdgcn_readlane(r64a, 0) + __builtin_amdgcn_readlane(r64a, 16) +
                       __builtin_amdgcn_readlane(r64a, 32) + __builtin_amdgcn_readlane(r64a, 48);
      const int q64b = __builtin_amdgcn_readlane(r64b, 0) + __builtin_amdgcn_readlane(r64b, 16) +
                       __builtin_amdgcn_readlane(r64b, 32) + __builtin_amdgcn_readlane(r64b, 48);
      if (lane == 0) {
        *reinterpret_cast<int2*>(logitL + (1 + q) * 2) = make_int2(q32a + hb32a, q32b + hb32b);  // one wave per quadrant
        *reinterpret_cast<int2*>(logitL + 44 + 2 * q) = make_int2(q64a, q64b);  // the readers add the four waves' parts
      }
    }
    // the A2/input region (R2) is free since the P3 barrier: stage the next CTU now, its P1 needs no extra barrier
    FHEVC_STAMP(6)  // heads only
    if (work + vgrid < total) {
      FHEVC_PHASE_IDS  // the staging addresses are re-derived per CTU: hoisted, they were spilled to scratch in the 168-register form
      stage_ctu(lds, L::R2_OFF, pre, F, next, tid, tid >> 2, tid & 3, shift_in, hc.in);
    }
    FHEVC_STAMP(7)  // staging of the next CTU; slot 4 below is then the wait at the barrier
    __syncthreads();
    FHEVC_STAMP(4)
    FHEVC_CONV1_EARLY_READS  // the next CTU's tile is staged (harmless reads if there is none)
    FHEVC_HAD_FRAG_LOADS

    // ================= P5: top-down depth map (forced split at the picture edge), branch-free =================
    {
      FHEVC_PHASE_IDS
      const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
      const int ux = tid & 15, uy = tid >> 4;
      const int4 pa = *reinterpret_cast<const int4*>(logitL + 44), pb = *reinterpret_cast<const int4*>(logitL + 48);
      const int2 l64 = make_int2(hb64a + pa.x + pa.z + pb.x + pb.z, hb64b + pa.y + pa.w + pb.y + pb.w);
      const int2 l32 = *reinterpret_cast<const int2*>(logitL + 2 * (1 + (uy >> 3) * 2 + (ux >> 3)));
      const int2 l16 = *reinterpret_cast<const int2*>(logitL + 2 * (5 + (uy >> 2) * 4 + (ux >> 2)));
      // soft decisions: d_depth follows the splits surer than +margin_split, d_depth_max those not rejected by more
      // than -margin_stop (both 0: the plain map in both); CUs crossing the picture edge are split either way
      const bool e64 = (vw < 64) || (vh < 64);  // uniform: the per-unit edge tests run for CTUs on the picture edge only
      bool inside = true, e32 = false, e16 = false;
      if (e64) {
        inside = (ux * 4 < vw) && (uy * 4 < vh);
        e32 = ((ux >> 3) * 32 + 32 > vw) || ((uy >> 3) * 32 + 32 > vh);
        e16 = ((ux >> 2) * 16 + 16 > vw) || ((uy >> 2) * 16 + 16 > vh);
      }
      const int d64 = l64.y - l64.x, d32 = l32.y - l32.x, d16 = l16.y - l16.x;
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      {
        const bool s64 = e64 || d64 > margin_split, s32 = e32 || d32 > margin_split, s16 = e16 || d16 > margin_split;
        const int d = (inside && s64) ? (s32 ? (s16 ? 3 : 2) : 1) : 0;
        d_depth[o * 256 + tid] = (uint8_t)d;
      }
      if (d_depth_max != nullptr) {
        const bool s64 = e64 || d64 > -margin_stop, s32 = e32 || d32 > -margin_stop, s16 = e16 || d16 > -margin_stop;
        const int d = (inside && s64) ? (s32 ? (s16 ? 3 : 2) : 1) : 0;
        d_depth_max[o * 256 + tid] = (uint8_t)d;
      }
      if (d_logits != nullptr && tid < 42) d_logits[o * 42 + tid] = tid == 0 ? l64.x : (tid == 1 ? l64.y : logitL[tid]);
      if (HAD == 1 && tid == 64) {  // (a lane of wave 1: wave 0 also assembles the split-flag word)
        const int4 hp = *reinterpret_cast<const int4*>(logitL + 56 + 4 * had_set);
        d_had[o] = hp.x + hp.y + hp.z + hp.w;
      }
      if (HAD == 2 && wave == 1) {  // MFMA form: lane = 8x8 block; its two M tiles' sums of |coefficients| (DC row zeroed), (s + 2) >> 2, CTU total
        const float* hp = reinterpret_cast<const float*>(lds + L::HADP_OFF) + 128 * had_set + lane;
        const int sb = (int)(hp[0] + hp[64]);                     // exact: integers below 2^24
        int v = dpp_row_sum((sb + 2) >> 2);                      // xCalcHADs8x8_ISlice: (sum + 2) >> 2 per block (TEncCu.cpp:1319-1321)
        v = __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
        if (lane == 0) d_had[o] = v;
      }
      if (d_flags != nullptr && wave == 0) {  // the 21 decisions as one word: lane k < 21 evaluates node k
        const int k = lane;
        const int bi = k - 5, qq = k < 5 ? k - 1 : (bi >> 3) * 2 + ((bi >> 1) & 1);  // own / parent quadrant
        const int qx = (qq & 1) * 32, qy = (qq >> 1) * 32, bxx = (bi & 3) * 16, byy = (bi >> 2) * 16;
        const int2 a64 = l64;
        const bool n64 = (vw < 64) || (vh < 64) || (a64.y - a64.x > margin_split);
        bool bit = n64;
        if (k >= 1 && k < 21) {
          const int2 a32 = *reinterpret_cast<const int2*>(logitL + 2 * (1 + qq));
          const bool n32 = n64 && (qx < vw) && (qy < vh) && ((qx + 32 > vw) || (qy + 32 > vh) || (a32.y - a32.x > margin_split));
          bit = n32;
          if (k >= 5) {
            const int2 a16 = *reinterpret_cast<const int2*>(logitL + 2 * k);
            bit = n32 && (bxx < vw) && (byy < vh) && ((bxx + 16 > vw) || (byy + 16 > vh) || (a16.y - a16.x > margin_split));
          }
        }
        const unsigned long long m = __ballot(bit && k < 21);
        if (lane == 0) d_flags[o] = (uint32_t)(m & 0x1FFFFFu);
      }
      zero_a1_halo<I8>(lds, tid, hc.a1);  // R1 held the conv3 output until the P4 barrier; conv1 of the next CTU needs a zero halo
    }
    FHEVC_STAMP(5)
    had_set ^= 1;
    pos = next;
    // no barrier here: the next P1 reads R2 (staged before the P4 barrier) and writes the A1 interior (R1, last read
    // before the P4 barrier, halo rewritten above by disjoint addresses); the logits are re-initialised in P2.
  }
  if (TRIO) {
    for (int i = group; i < 2; ++i) __syncthreads();
  }
  if (STAMPS) {
    unsigned long long treal1;
    const unsigned long long tclk1 = stamp();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(treal1)::"memory");
    if (tid == 0) {
      for (int k = 0; k < 8; ++k) d_stamps[vbidx * FHEVC_STAMP_SLOTS + k] = tsum[k];
      d_stamps[vbidx * FHEVC_STAMP_SLOTS + 8] = tclk1 - tclk0;
      d_stamps[vbidx * FHEVC_STAMP_SLOTS + 9] = treal1 - treal0;
      d_stamps[vbidx * FHEVC_STAMP_SLOTS + 10] = (unsigned long long)prio_slot;
    }
  }
#undef FHEVC_STAMP
#undef FHEVC_CONV1_EARLY_READS
#undef FHEVC_HAD_FRAG_LOADS
}


// =====================================================================================================================================
// The i8 form as a TWO-STAGE SOFTWARE PIPELINE over the CTUs of a workgroup (round 3; FHEVC_CNN_PIPE).  Same network, same arithmetic, same
// per-wave work split as fhevc_cnn_depth_kernel<.., ARITH = 1 | 2>; what changes is WHEN each piece runs.  The kernel above runs a CTU
// through five phases with four barriers, and in three of them (conv1's pooling / requant, the heads, staging + depth map) the matrix
// pipe has nothing to do while in the other two the vector ALUs idle: the counters show matrix and vector instructions co-executing in
// only a third of the matrix-busy cycles.  Here every barrier interval pairs an MFMA-bound piece of one CTU with VALU-bound pieces of
// its neighbours IN THE SAME WAVE, so the vector work sits in the shadow of the wave's own MFMA chain:
//     X(k):  conv2(k)   ||  heads(k-1)  +  staging(k+1)  +  source Hadamard(k+1)          barrier
//     Y(k):  conv3(k)   ||  conv1(k+1)  +  depth map(k-1)  +  prefetch(k+2)               barrier
// Two barriers per CTU instead of four.  Every activation map has a region of its own (tile, A1, A2, A3 are all live in both
// intervals): 76 800 B of LDS, two workgroups per CU, up to 256 VGPRs (conv3's and conv1's accumulators are live together).  The halos
// are written once per kernel (no region is ever recycled for another map).
//   hazards: conv2(k) reads A1(k) (written in Y(k-1)) and writes A2 (last read by conv3(k-1) in Y(k-1)); staging(k+1) writes the tile
//   (last read by conv1(k) in Y(k-1)); heads(k-1) read A3(k-1) (written in Y(k-1)); conv3(k) writes A3 (last read in X(k)); conv1(k+1)
//   writes A1 (last read in X(k)).  Logits: set k & 1 is initialised in Y(k), summed by heads(k) in X(k+1), read by the depth map in Y(k+1).
struct LdsPipe {
  static constexpr int A1_OFF = 0;                                // conv1 output [35][36][16 B] (a - 128), parity-split columns
  static constexpr int A3_OFF = A1_OFF + 35 * A1_ROW;             // 20160: conv3 output [256][64 B], swizzled chunks
  static constexpr int A2_OFF = A3_OFF + 16384;                   // 36544: conv2 output, 2 planes [18][18][16 B]
  static constexpr int T_OFF = A2_OFF + 2 * A2_PLANE;             // 47296: input tile bf16 [33 row pairs][68 dwords] (+ conv1's one fragment read past it)
  static constexpr int BIAS_OFF = T_OFF + 37 * IN_PITCH * 4;      // 57360: b1 (float) b2 b3 (int32)
  static constexpr int LOGIT_OFF = BIAS_OFF + 112 * 4;            // 57808: two sets of 64 ints (as Lds::LOGIT_OFF [0..51])
  static constexpr int HADS_OFF = LOGIT_OFF + 2 * 64 * 4;         // 58320: four sets of the four waves' source-Hadamard sums
  static constexpr int HEADW_OFF = HADS_OFF + 16 * 4;             // 58384: wh64 (8192 B) + the MFMA image of wh32 / wh16 (10240 B)
  static constexpr int LDS_BYTES = HEADW_OFF + 18432;             // 76816
  static_assert(A3_OFF % 16 == 0 && A2_OFF % 16 == 0 && T_OFF % 16 == 0 && BIAS_OFF % 16 == 0 && LOGIT_OFF % 16 == 0 && HEADW_OFF % 16 == 0, "");
  static_assert(2 * LDS_BYTES <= 160 * 1024, "two workgroups per CU");
};
#ifndef FHEVC_PIPE_RING3
#define FHEVC_PIPE_RING3 4
#endif
#ifndef FHEVC_PIPE_SCHED_X
#define FHEVC_PIPE_SCHED_X 9    // VALU instructions offered per (MFMA, fragment read) group of conv2's chains in interval X: Hadamard, heads, pools, requant
#endif
#ifndef FHEVC_PIPE_FENCE_Y
#define FHEVC_PIPE_FENCE_Y 0    // 1: a hard sched_barrier after every step of interval Y instead of the group description: the interleave then
#endif                          // holds in the ISA (7-8 VALU behind every MFMA) and the kernel gains 1.7 % (0.4066 against 0.4135 ms; default kernel 0.3827)
#ifndef FHEVC_PIPE_SCHED_Y
#define FHEVC_PIPE_SCHED_Y 10   // VALU instructions offered per conv3 MFMA step of interval Y (0: no pipeline description)
#endif

// top-down depth map of one CTU from its logits (the depth phase of the kernel above, as a function: the pipelined kernel needs it twice)
__device__ __forceinline__ void depth_map_of_ctu(const FhevcFrames& F, const int* logitL, const int* hads, int tid, int lane, int wave, int f, int cy, int cx,
                                                 int band_rows, int hb64a, int hb64b, int margin_split, int margin_stop, uint8_t* __restrict__ d_depth,
                                                 uint8_t* __restrict__ d_depth_max, int32_t* __restrict__ d_logits, uint32_t* __restrict__ d_flags,
                                                 int32_t* __restrict__ d_had, bool had)
{
  const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
  const int ux = tid & 15, uy = tid >> 4;
  const int4 pa = *reinterpret_cast<const int4*>(logitL + 44), pb = *reinterpret_cast<const int4*>(logitL + 48);
  const int2 l64 = make_int2(hb64a + pa.x + pa.z + pb.x + pb.z, hb64b + pa.y + pa.w + pb.y + pb.w);
  const int2 l32 = *reinterpret_cast<const int2*>(logitL + 2 * (1 + (uy >> 3) * 2 + (ux >> 3)));
  const int2 l16 = *reinterpret_cast<const int2*>(logitL + 2 * (5 + (uy >> 2) * 4 + (ux >> 2)));
  const bool e64 = (vw < 64) || (vh < 64);
  bool inside = true, e32 = false, e16 = false;
  if (e64) {
    inside = (ux * 4 < vw) && (uy * 4 < vh);
    e32 = ((ux >> 3) * 32 + 32 > vw) || ((uy >> 3) * 32 + 32 > vh);
    e16 = ((ux >> 2) * 16 + 16 > vw) || ((uy >> 2) * 16 + 16 > vh);
  }
  const int d64 = l64.y - l64.x, d32 = l32.y - l32.x, d16 = l16.y - l16.x;
  const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
  {
    const bool s64 = e64 || d64 > margin_split, s32 = e32 || d32 > margin_split, s16 = e16 || d16 > margin_split;
    const int d = (inside && s64) ? (s32 ? (s16 ? 3 : 2) : 1) : 0;
    d_depth[o * 256 + tid] = (uint8_t)d;
  }
  if (d_depth_max != nullptr) {
    const bool s64 = e64 || d64 > -margin_stop, s32 = e32 || d32 > -margin_stop, s16 = e16 || d16 > -margin_stop;
    const int d = (inside && s64) ? (s32 ? (s16 ? 3 : 2) : 1) : 0;
    d_depth_max[o * 256 + tid] = (uint8_t)d;
  }
  if (d_logits != nullptr && tid < 42) d_logits[o * 42 + tid] = tid == 0 ? l64.x : (tid == 1 ? l64.y : logitL[tid]);
  if (had && tid == 64) {
    const int4 hp = *reinterpret_cast<const int4*>(hads);
    d_had[o] = hp.x + hp.y + hp.z + hp.w;
  }
  if (d_flags != nullptr && wave == 0) {  // the 21 decisions as one word: lane k < 21 evaluates node k
    const int k = lane;
    const int bi = k - 5, qq = k < 5 ? k - 1 : (bi >> 3) * 2 + ((bi >> 1) & 1);
    const int qx = (qq & 1) * 32, qy = (qq >> 1) * 32, bxx = (bi & 3) * 16, byy = (bi >> 2) * 16;
    const bool n64 = (vw < 64) || (vh < 64) || (l64.y - l64.x > margin_split);
    bool bit = n64;
    if (k >= 1 && k < 21) {
      const int2 a32 = *reinterpret_cast<const int2*>(logitL + 2 * (1 + qq));
      const bool n32 = n64 && (qx < vw) && (qy < vh) && ((qx + 32 > vw) || (qy + 32 > vh) || (a32.y - a32.x > margin_split));
      bit = n32;
      if (k >= 5) {
        const int2 a16 = *reinterpret_cast<const int2*>(logitL + 2 * k);
        bit = n32 && (bxx < vw) && (byy < vh) && ((bxx + 16 > vw) || (byy + 16 > vh) || (a16.y - a16.x > margin_split));
      }
    }
    const unsigned long long m = __ballot(bit && k < 21);
    if (lane == 0) d_flags[o] = (uint32_t)(m & 0x1FFFFFu);
  }
}

template <int HAD, int ARITH>
__global__ __launch_bounds__(256, 2) void fhevc_cnn_depth_pipe_kernel(FhevcFrames F, FhevcCnnWeights W, uint8_t* __restrict__ d_depth, int32_t* __restrict__ d_had,
                                                                       int32_t* __restrict__ d_logits, uint32_t* __restrict__ d_flags,
                                                                       uint8_t* __restrict__ d_depth_max, int margin_split, int margin_stop)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  __builtin_amdgcn_s_setreg((1 << 11) | 1, 2);  // fp32 rounding toward -inf: v_cvt_pk_u8_f32 = floor + clamp (conv1's requant); see the kernel above
  using P = LdsPipe;
  constexpr bool FASTRQ = ARITH == 2;
  constexpr int RING3P = FHEVC_PIPE_RING3;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int hb64a = W.bhead[0], hb64b = W.bhead[1] + W.bhead[6 + 0 * 52 + F.qp];
  const int hb32a = W.bhead[2], hb32b = W.bhead[3] + W.bhead[6 + 1 * 52 + F.qp];
  const int hb16a = W.bhead[4], hb16b = W.bhead[5] + W.bhead[6 + 2 * 52 + F.qp];
  // ---- resident weight fragments: 8 + 24 + 36 registers ----
  const bf16x8 wA1a = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV1 + lane]);
  const bf16x8 wA1b = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV1 + 64 + lane]);
  bf16x8 wA2[9], wA3[18];
  const int tile3 = wave & 1;
#pragma unroll
  for (int s = 0; s < 6; ++s) wA2[s] = __builtin_bit_cast(bf16x8, W.frag_i8[FHEVC_FRAGI8_CONV2 + s * 64 + lane]);
#pragma unroll
  for (int s = 0; s < 9; ++s) wA3[s] = __builtin_bit_cast(bf16x8, W.frag_i8[FHEVC_FRAGI8_CONV3 + (tile3 * 9 + s) * 64 + lane]);
  const int shift2 = W.shift[1], shift3 = W.shift[2];

  float* biasL = reinterpret_cast<float*>(lds + P::BIAS_OFF);
  int* logit0 = reinterpret_cast<int*>(lds + P::LOGIT_OFF);
  int* hadsL = reinterpret_cast<int*>(lds + P::HADS_OFF);
  if (tid < 112) {
    if (tid >= 16) reinterpret_cast<int*>(biasL)[tid] = W.bias_i8[tid];
    else biasL[tid] = W.bias[tid] * W.scale[0];
  }
  // wh64 with its chunk swizzle, and the MFMA image of wh32 / wh16 (as in the kernel above)
  for (int i = tid; i < 8192 / 16; i += 256) {
    const int row = i >> 2, c = i & 3, sw = (row >> 3) & 3;
    *reinterpret_cast<uint4*>(lds + P::HEADW_OFF + row * 64 + ((c ^ sw) << 4)) = reinterpret_cast<const uint4*>(W.whead)[i];
  }
  for (int i = tid; i < HEADM_BYTES / 16; i += 256) {
    const int j = i / 40, rem = i - j * 40, n = rem >> 2, kg = rem & 3;
    int src;
    if (n < 2) src = 16384 + (n * 16 + j) * 64;
    else {
      const int sub = (n - 2) >> 1, cls = n & 1, py = j >> 2, px = j & 3;
      src = 8192 + (cls * 64 + ((sub >> 1) * 4 + py) * 8 + (sub & 1) * 4 + px) * 64;
    }
    *reinterpret_cast<uint4*>(lds + P::HEADW_OFF + HEADM_OFF + i * 16) = *reinterpret_cast<const uint4*>(W.whead + src + kg * 16);
  }
  // the halos, once: "activation 0" (0x80) everywhere in A1 and A2, bf16(128) everywhere in the tile; the interiors are rewritten per CTU
  for (int i = tid; i < (35 * A1_ROW) / 16; i += 256) *reinterpret_cast<uint4*>(lds + P::A1_OFF + i * 16) = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
  for (int i = tid; i < (2 * A2_PLANE) / 16; i += 256) *reinterpret_cast<uint4*>(lds + P::A2_OFF + i * 16) = make_uint4(0x80808080u, 0x80808080u, 0x80808080u, 0x80808080u);
  for (int i = tid; i < (37 * IN_PITCH * 4) / 16; i += 256) *reinterpret_cast<uint4*>(lds + P::T_OFF + i * 16) = make_uint4(0x43004300u, 0x43004300u, 0x43004300u, 0x43004300u);
  if (tid < 128 + 16) logit0[tid] = 0;
  __syncthreads();  // the fills above and the first staging below write the same tile cells from different threads

  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  const int shift_in = F.bit_depth - 8;
  const int ld_row = tid >> 2, ld_seg = tid & 3;
  const int grid = (int)gridDim.x;
  const int vblock = (grid & 7) ? (int)blockIdx.x : (int)((blockIdx.x & 7) * (grid >> 3) + (blockIdx.x >> 3));
  CtuPos pos, step;
  {
    const int vb = min(vblock, total - 1);
    pos.f = vb / per_frame; pos.ry = (vb - pos.f * per_frame) / F.ctus_x; pos.cx = (vb - pos.f * per_frame) - pos.ry * F.ctus_x;
    step.f = grid / per_frame; step.ry = (grid - step.f * per_frame) / F.ctus_x; step.cx = (grid - step.f * per_frame) - step.ry * F.ctus_x;
  }
  // heads: per-thread LDS rows (as in the kernel above)
  uint2 head_addr;
  {
    const int q = wave, blk = lane >> 4;
    const int y = (q >> 1) * 8 + (blk >> 1) * 4 + ((lane >> 2) & 3), x = (q & 1) * 8 + (blk & 1) * 4 + (lane & 3);
    const unsigned psw = (x >> 2) & 3, sw64 = (y >> 1) & 3;
    const unsigned a_0 = (unsigned)(y * 16 + x) * 64 + (psw << 4);
    const unsigned w64_0 = (unsigned)((y >> 1) * 8 + (x >> 1)) * 64 + (sw64 << 4);
    head_addr = make_uint2(a_0, w64_0);
  }
  unsigned headm_addr;
  {
    const int m = lane & 15, kg = lane >> 4, by = m >> 2, bx = m & 3;
    const unsigned am = (unsigned)((4 * by + wave) * 16 + 4 * bx) * 64 + ((unsigned)(kg ^ bx) << 4);
    const unsigned bm = (unsigned)(HEADM_OFF + 4 * wave * HEADM_STEP + min(m, 9) * 64 + 16 * kg);
    headm_addr = am | (bm << 16);
  }
  // ---- the pieces ----
  // conv1 of the CTU whose samples sit in the tile: T -> A1 (8 units = pooled rows wave, wave + 4, ..: see P1 of the kernel above)
  f32x16 bias1;
  {
    const float4 b0 = *reinterpret_cast<const float4*>(W.bias + 8 * h), b1 = *reinterpret_cast<const float4*>(W.bias + 8 * h + 4);
    const float sc = W.scale[0];
    bias1[0] = b0.x * sc; bias1[1] = b0.y * sc; bias1[2] = b0.z * sc; bias1[3] = b0.w * sc;
    bias1[4] = b1.x * sc; bias1[5] = b1.y * sc; bias1[6] = b1.z * sc; bias1[7] = b1.w * sc;
#pragma unroll
    for (int i = 0; i < 8; ++i) bias1[8 + i] = bias1[i];
  }
  auto frag1 = [](const unsigned char* p) {
    const uint2 lo = *reinterpret_cast<const uint2*>(p);
    const uint2 hi = *reinterpret_cast<const uint2*>(p + IN_PITCH * 4);
    return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
  };
  const unsigned char* c1_in = lds + P::T_OFF + (2 * r + 2 * h) * 4 + wave * (IN_PITCH * 4);
  unsigned char* c1_out = lds + P::A1_OFF + 8 * h + (wave + 1) * A1_ROW + (((r + 1) & 1) ? 0 : A1_EVEN) + ((r + 1) >> 1) * 16;
#define FHEVC_PIPE_CONV1_UNIT(i)                                                                                   \
  {                                                                                                                \
    const bf16x8 bq_ = frag1(c1_in + (i) * (4 * IN_PITCH * 4));                                                    \
    const f32x16 acc0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1a, bq_, bias1, 0, 0, 0);                       \
    const f32x16 acc1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1b, bq_, bias1, 0, 0, 0);                       \
    conv1_store_i8(acc0_, acc1_, c1_out + (i) * (4 * A1_ROW));                                                     \
  }
  // conv2: A1 -> A2 (lane -> pooled position and units as in P2 of the kernel above)
  const int q2 = r >> 2;
  const int pr2 = (q2 ^ (q2 >> 1) ^ (q2 >> 2)) & 1, pc2 = ((r >> 3) << 2) | (r & 3);
  const unsigned char* a1p = lds + P::A1_OFF + (2 * pr2 + h) * A1_ROW + pc2 * 16;
  unsigned char* a2dst = lds + P::A2_OFF + ((pr2 + 1) * A2_PITCH + pc2 + 1) * 16 + 4 * h;
  const int* b2t = reinterpret_cast<const int*>(biasL) + 16;
  // conv3: A2 -> A3 (B column n = lane & 31: row 8 rs of the pair, position x; lane half = activation plane)
  const int x3 = lane & 15, rs3 = (lane >> 4) & 1, y03 = 2 * (wave >> 1);
  const unsigned char* a2 = lds + P::A2_OFF + h * A2_PLANE + ((y03 + 8 * rs3) * A2_PITCH + x3) * 16;
  const int psw3 = (x3 >> 2) & 3;
  unsigned char* a3dst = lds + P::A3_OFF + ((y03 + 8 * rs3) * 16 + x3) * 64 + 4 * h;
  const int* b3p = reinterpret_cast<const int*>(biasL) + 48 + 32 * tile3;

  // ---- prologue: first CTU staged, its conv1 done, the second CTU's samples in flight ----
  CtuPos p_prev = pos, p_cur = pos, p_next = advance(pos, step, band_rows, F.ctus_x);   // c(k-1), c(k), c(k+1)
  Prefetched pre = prefetch_ctu<HAD == 1>(F, vblock < total, p_cur, ld_row, ld_seg);
  if (vblock < total) {
    stage_ctu<false>(lds, P::T_OFF, pre, F, p_cur, tid, ld_row, ld_seg, shift_in, 0u);
    if (HAD == 1) {
      const int hs = F.sample_bytes == 2 ? wave_src_hadamard<2>(pre, lane) : wave_src_hadamard<1>(pre, lane);
      if (lane == 0) hadsL[wave] = hs;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) FHEVC_PIPE_CONV1_UNIT(i)
  pre = prefetch_ctu<HAD == 1>(F, vblock + grid < total, p_next, ld_row, ld_seg);
  __syncthreads();

  int k = 0;
  for (int work = vblock; work < total; work += grid, ++k) {
    const CtuPos p_next2 = advance(p_next, step, band_rows, F.ctus_x);
    const bool have_next = work + grid < total;
    int* logit_prev = logit0 + 64 * ((k + 1) & 1);   // set (k - 1) & 1: heads(k-1) add to it in X(k), the depth map reads it in Y(k)
    int* logit_cur = logit0 + 64 * (k & 1);          // set k & 1: initialised in Y(k)
    // ================= X(k): staging(k+1) + Hadamard(k+1), then conv2(k) with heads(k-1) in its shadow =================
    unsigned hr[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };  // the next CTU's samples for the source Hadamard (zeros: no next CTU, or not through the aligned path)
    if (have_next) {
      stage_ctu<false>(lds, P::T_OFF, pre, F, p_next, tid, ld_row, ld_seg, shift_in, 0u);
      if (HAD == 1) {
        if (F.sample_bytes == 2) hadamard_samples<2>(pre, hr); else hadamard_samples<1>(pre, hr);
      }
    }
    {
      // source Hadamard(k+1): straight-line VALU work in the shadow of conv2's MFMAs (the set's write is harmless without a next CTU)
      if (HAD == 1) {
        const int hs = wave_src_hadamard_core(hr, lane);
        hadsL[4 * ((k + 1) & 3) + wave] = hs;   // every lane holds the wave's sum: 64 lanes, one address, one value
      }
      // heads(k-1): 16- and 32-level heads as ONE GEMM on v_mfma_i32_16x16x64_i8, 64-level head on v_dot4_i32_i8 (P4 of the kernel above)
      const unsigned char* hw = lds + P::HEADW_OFF;
      const unsigned char* ap = lds + P::A3_OFF + (headm_addr & 0xFFFFu);
      const unsigned char* bp = hw + (headm_addr >> 16);
      i32x4 hacc = { 0, 0, 0, 0 };
#pragma unroll
      for (int px = 0; px < 4; ++px)
        hacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(ap + px * 64), *reinterpret_cast<const i32x4*>(bp + px * HEADM_STEP), hacc, 0, 0, 0);
      int s64a = 0, s64b = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint4 a = *reinterpret_cast<const uint4*>(lds + P::A3_OFF + (head_addr.x ^ (qq << 4)));
        const unsigned char* w64 = hw + HEAD64_OFF + (head_addr.y ^ (qq << 4));
        const uint4 d0 = *reinterpret_cast<const uint4*>(w64), d1 = *reinterpret_cast<const uint4*>(w64 + 4096);
        s64a = sdot4(a.x, d0.x, s64a); s64a = sdot4(a.y, d0.y, s64a); s64a = sdot4(a.z, d0.z, s64a); s64a = sdot4(a.w, d0.w, s64a);
        s64b = sdot4(a.x, d1.x, s64b); s64b = sdot4(a.y, d1.y, s64b); s64b = sdot4(a.z, d1.z, s64b); s64b = sdot4(a.w, d1.w, s64b);
      }
      // conv2(k): four half-chains (u0 dy 0, u0 dy 1, u1 dy 0, u1 dy 1), pools and requants under the following chains
      const int u0 = wave, u1 = wave + 4;
      const unsigned char* h00 = a1p + (4 * u0) * A1_ROW;
      const unsigned char* h10 = a1p + (4 * u1) * A1_ROW;
      i32x16 t0, t1, a0, a1;
      bf16x8 ring[RINGI];
      const int c2l = conv2_lane_i8(h);
      conv2_half_i8<true, false, 0>(h00, h00 + A1_ROW, c2l, wA2, ring, b2t, h, t0, t1);
      __builtin_amdgcn_sched_group_barrier(0x100, RINGI + 4, 0);
      sched_chain12_i8<FHEVC_PIPE_SCHED_X>();
      conv2_half_i8<false, false, C2F % RINGI>(h00 + A1_ROW, h10, c2l, wA2, ring, b2t, h, a0, a1);
      pool_h_i8(t0, t1);
      sched_chain12_i8<FHEVC_PIPE_SCHED_X>();
      pool_v_i8(t0, a0, a1);
      conv2_half_i8<false, false, (2 * C2F) % RINGI>(h10, h10 + A1_ROW, c2l, wA2, ring, b2t, h, t1, a0);
      conv2_requant_store_i8m<FASTRQ ? 1 : 0>(t0, a2dst + (2 * u0) * A2_PITCH * 16, shift2);
      sched_chain12_i8<FHEVC_PIPE_SCHED_X>();
      conv2_half_i8<false, true, (3 * C2F) % RINGI>(h10 + A1_ROW, h10 + A1_ROW, c2l, wA2, ring, b2t, h, a1, t0);
      pool_h_i8(t1, a0);
      sched_chain12_i8<FHEVC_PIPE_SCHED_X>();
      pool_v_i8(t1, a1, t0);
      conv2_requant_store_i8m<FASTRQ ? 1 : 0>(t1, a2dst + (2 * u1) * A2_PITCH * 16, shift2);
      // the heads' partial sums join the logits of set (k - 1) & 1
      {
        const int n = lane & 15, rg = lane >> 4, subn = (n - 2) >> 1;
        const bool is16 = n < 2, is32 = n >= 2 && n < 10 && ((rg & 1) == (subn >> 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool act = is16 || (is32 && ((i & 1) == (subn & 1)));
          const int idx = is16 ? 2 * (5 + 4 * rg + i) + n : 2 * (1 + (rg >> 1) * 2 + (i >> 1)) + (n & 1);
          if (act) atomicAdd(logit_prev + idx, hacc[i]);
        }
      }
      const int r64a = dpp_row_sum(s64a), r64b = dpp_row_sum(s64b);
      const int q64a = __builtin_amdgcn_readlane(r64a, 0) + __builtin_amdgcn_readlane(r64a, 16) + __builtin_amdgcn_readlane(r64a, 32) + __builtin_amdgcn_readlane(r64a, 48);
      const int q64b = __builtin_amdgcn_readlane(r64b, 0) + __builtin_amdgcn_readlane(r64b, 16) + __builtin_amdgcn_readlane(r64b, 32) + __builtin_amdgcn_readlane(r64b, 48);
      if (lane == 0) *reinterpret_cast<int2*>(logit_prev + 44 + 2 * wave) = make_int2(q64a, q64b);
    }
    __syncthreads();
    // ================= Y(k): depth map(k-1), prefetch(k+2), then conv3(k) with conv1(k+1) in its shadow =================
    if (k >= 1)
      depth_map_of_ctu(F, logit_prev, hadsL + 4 * ((k - 1) & 3), tid, lane, wave, p_prev.f, F.row_begin + p_prev.ry, p_prev.cx, band_rows, hb64a, hb64b,
                       margin_split, margin_stop, d_depth, d_depth_max, d_logits, d_flags, d_had, HAD == 1);
    if (tid < 40) logit_cur[2 + tid] = tid < 8 ? ((tid & 1) ? hb32b : hb32a) : ((tid & 1) ? hb16b : hb16a);
    pre = prefetch_ctu<HAD == 1>(F, work + 2 * grid < total, p_next2, ld_row, ld_seg);
    {
      // conv3(k)'s 36 MFMAs in program order j = 0..35 (super-chain j / 18: accumulators p0 / p1, then q0 / q1; tap (j % 18) / 2), one
      // fragment read ahead per MFMA.  conv1(k+1)'s eight units are cut into slices and dealt over those steps so that every MFMA has
      // a few VALU instructions behind it: unit u's two MFMAs go out before step 9u / 2, its pooling (2 x 8 v_max3 / v_max) after the
      // next two steps, requant + store after the third.  The requant of p0 / p1 (four 5-instruction groups each) follows their chain
      // in the same way; only q0 / q1's requant is left for the tail of the interval.
      bf16x8 ring[RING3P];
      i32x16 p0, p1, q0, q1;
      f32x16 c1a, c1b;
      float c1m[8];
#pragma unroll
      for (int g = 0; g < RING3P; ++g) ring[g] = lds_frag(a2 + conv3_frag_off_i8(g));
      static_for<0, 36>([&](auto J) {
        constexpr int j = decltype(J)::value;
        constexpr int t = (j % 18) >> 1;
        if constexpr (j == 0) { const i32x16 binit = bias_tile_i8(b3p, h); p0 = binit; p1 = binit; }
        if constexpr (j == 18) { const i32x16 binit = bias_tile_i8(b3p, h); q0 = binit; q1 = binit; }
        static_for<0, 8>([&](auto U) {
          constexpr int u = decltype(U)::value;
          if constexpr (j == (9 * u) / 2) {
            const bf16x8 bq_ = frag1(c1_in + u * (4 * IN_PITCH * 4));
            c1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1a, bq_, bias1, 0, 0, 0);
            c1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1b, bq_, bias1, 0, 0, 0);
          }
        });
        if constexpr (j < 18) { if constexpr (j & 1) p1 = mfma_i8(wA3[t], ring[j % RING3P], p1); else p0 = mfma_i8(wA3[t], ring[j % RING3P], p0); }
        else { if constexpr (j & 1) q1 = mfma_i8(wA3[t], ring[j % RING3P], q1); else q0 = mfma_i8(wA3[t], ring[j % RING3P], q0); }
        if constexpr (j + RING3P < 36) ring[j % RING3P] = lds_frag(a2 + conv3_frag_off_i8(j + RING3P));
        static_for<0, 8>([&](auto U) {
          constexpr int u = decltype(U)::value;
          constexpr int j0 = (9 * u) / 2;
          if constexpr (j == j0 + 1) {
#pragma unroll
            for (int c = 0; c < 4; ++c) c1m[c] = fmaxf(fmaxf(c1a[c], c1a[c + 8]), fmaxf(c1b[c], c1b[c + 8]));
          }
          if constexpr (j == j0 + 2) {
#pragma unroll
            for (int c = 4; c < 8; ++c) c1m[c] = fmaxf(fmaxf(c1a[c], c1a[c + 8]), fmaxf(c1b[c], c1b[c + 8]));
          }
          if constexpr (j == j0 + 3)
            *reinterpret_cast<uint2*>(c1_out + u * (4 * A1_ROW)) = make_uint2(u8x4_floor_clamp(c1m[0], c1m[1], c1m[2], c1m[3]) ^ 0x80808080u,
                                                                               u8x4_floor_clamp(c1m[4], c1m[5], c1m[6], c1m[7]) ^ 0x80808080u);
        });
        // requant of super-chain 0's accumulators, one group of four channels per step: p0 after steps 19..22, p1 after 23..26
        if constexpr (j >= 19 && j < 27) {
          constexpr int g = (j - 19) & 3;
          const i32x16& acc = j < 23 ? p0 : p1;
          unsigned char* dst = a3dst + conv3_pair_row_i8(j < 23 ? 0 : 1) * 1024;
          *reinterpret_cast<unsigned*>(dst + (((2 * tile3 + (g >> 1)) ^ psw3) << 4) + 8 * (g & 1)) =
              requant4_i8(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], shift3, FASTRQ ? 2 : 0);
        }
        // the pipeline description of this step: its MFMA(s) and fragment read(s), then the slices' VALU work and their one LDS store
        if (FHEVC_PIPE_FENCE_Y) __builtin_amdgcn_sched_barrier(0);  // nothing moves across a step boundary: the slices stay where they were dealt
        else if (FHEVC_PIPE_SCHED_Y) {
          constexpr bool issue = (j % 9 == 0) || (j % 9 == 4);  // steps 0, 4, 9, 13, 18, 22, 27, 31: a conv1 unit's two MFMAs and two ds_read_b64 go out first
          __builtin_amdgcn_sched_group_barrier(0x100, issue ? 3 : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, issue ? 3 : 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, FHEVC_PIPE_SCHED_Y, 0);
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
      });
      conv3_store_i8m<FASTRQ ? 2 : 0>(q0, a3dst + conv3_pair_row_i8(2) * 1024, tile3, psw3, shift3);
      conv3_store_i8m<FASTRQ ? 2 : 0>(q1, a3dst + conv3_pair_row_i8(3) * 1024, tile3, psw3, shift3);
    }
    __syncthreads();
    p_prev = p_cur; p_cur = p_next; p_next = p_next2;
  }
  // ================= drain: heads and depth map of the last CTU =================
  if (k >= 1) {
    int* logit_prev = logit0 + 64 * ((k + 1) & 1);
    {
      const unsigned char* hw = lds + P::HEADW_OFF;
      const unsigned char* ap = lds + P::A3_OFF + (headm_addr & 0xFFFFu);
      const unsigned char* bp = hw + (headm_addr >> 16);
      i32x4 hacc = { 0, 0, 0, 0 };
#pragma unroll
      for (int px = 0; px < 4; ++px)
        hacc = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<const i32x4*>(ap + px * 64), *reinterpret_cast<const i32x4*>(bp + px * HEADM_STEP), hacc, 0, 0, 0);
      int s64a = 0, s64b = 0;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const uint4 a = *reinterpret_cast<const uint4*>(lds + P::A3_OFF + (head_addr.x ^ (qq << 4)));
        const unsigned char* w64 = hw + HEAD64_OFF + (head_addr.y ^ (qq << 4));
        const uint4 d0 = *reinterpret_cast<const uint4*>(w64), d1 = *reinterpret_cast<const uint4*>(w64 + 4096);
        s64a = sdot4(a.x, d0.x, s64a); s64a = sdot4(a.y, d0.y, s64a); s64a = sdot4(a.z, d0.z, s64a); s64a = sdot4(a.w, d0.w, s64a);
        s64b = sdot4(a.x, d1.x, s64b); s64b = sdot4(a.y, d1.y, s64b); s64b = sdot4(a.z, d1.z, s64b); s64b = sdot4(a.w, d1.w, s64b);
      }
      {
        const int n = lane & 15, rg = lane >> 4, subn = (n - 2) >> 1;
        const bool is16 = n < 2, is32 = n >= 2 && n < 10 && ((rg & 1) == (subn >> 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool act = is16 || (is32 && ((i & 1) == (subn & 1)));
          const int idx = is16 ? 2 * (5 + 4 * rg + i) + n : 2 * (1 + (rg >> 1) * 2 + (i >> 1)) + (n & 1);
          if (act) atomicAdd(logit_prev + idx, hacc[i]);
        }
      }
      const int r64a = dpp_row_sum(s64a), r64b = dpp_row_sum(s64b);
      const int q64a = __builtin_amdgcn_readlane(r64a, 0) + __builtin_amdgcn_readlane(r64a, 16) + __builtin_amdgcn_readlane(r64a, 32) + __builtin_amdgcn_readlane(r64a, 48);
      const int q64b = __builtin_amdgcn_readlane(r64b, 0) + __builtin_amdgcn_readlane(r64b, 16) + __builtin_amdgcn_readlane(r64b, 32) + __builtin_amdgcn_readlane(r64b, 48);
      if (lane == 0) *reinterpret_cast<int2*>(logit_prev + 44 + 2 * wave) = make_int2(q64a, q64b);
    }
    __syncthreads();
    depth_map_of_ctu(F, logit_prev, hadsL + 4 * ((k - 1) & 3), tid, lane, wave, p_prev.f, F.row_begin + p_prev.ry, p_prev.cx, band_rows, hb64a, hb64b,
                     margin_split, margin_stop, d_depth, d_depth_max, d_logits, d_flags, d_had, HAD == 1);
  }
#undef FHEVC_PIPE_CONV1_UNIT
}

#include "k_cnn_family.inc"
#include "k_cnn_layers.inc"
#include "k_cnn_d2.inc"

// split-flag words -> depth maps (whole pictures, CTU raster order): one thread per row of 16 units = one 16-byte store,
// 16 threads per CTU, 16 CTUs per workgroup and sweep (HBM-write-bound: 256 B per CTU)
__global__ __launch_bounds__(256) void fhevc_expand_flags_kernel(FhevcFrames F, const uint32_t* __restrict__ flags,
                                                                  uint8_t* __restrict__ depth)
{
  const int per_frame = F.ctus_x * F.ctus_y;
  const long long total = (long long)per_frame * F.num_frames;
  const int tid = threadIdx.x, uy = tid & 15, sub = tid >> 4;
  for (long long c = (long long)blockIdx.x * 16 + sub; c < total; c += (long long)gridDim.x * 16) {
    const int rem = (int)(c % per_frame), cy = rem / F.ctus_x, cx = rem - cy * F.ctus_x;
    const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
    const uint32_t w = flags[c];
    uint32_t out[4] = { 0, 0, 0, 0 };
    if (uy * 4 < vh && (w & 1u)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // the four units of a 16x16 block share their depth; the picture edge may cut the group
        const int q = (uy >> 3) * 2 + (j >> 1), bi = (uy >> 2) * 4 + j;
        const uint32_t d = ((w >> (1 + q)) & 1u) ? (((w >> (5 + bi)) & 1u) ? 3u : 2u) : 1u;
        const int nvalid = min(4, max(0, (vw >> 2) - 4 * j));
        out[j] = (d * 0x01010101u) & (nvalid >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nvalid)) - 1u));
      }
    }
    *reinterpret_cast<uint4*>(depth + c * 256 + uy * 16) = make_uint4(out[0], out[1], out[2], out[3]);
  }
}

}  // namespace

hipError_t fhevc_launch_expand_flags(const FhevcFrames& fr, const uint32_t* d_flags, uint8_t* d_depth, hipStream_t stream)
{
  const long long total = (long long)fr.ctus_x * fr.ctus_y * fr.num_frames;
  if (total <= 0) return hipSuccess;
  const long long groups = (total + 15) / 16;
  const int grid = (int)(groups < 8192 ? groups : 8192);
  hipLaunchKernelGGL(fhevc_expand_flags_kernel, dim3(grid), dim3(256), 0, stream, fr, d_flags, d_depth);
  return hipGetLastError();
}

// > 64 KiB of dynamic LDS needs an opt-in per function AND per device: fhevc_create calls this with its device current
hipError_t fhevc_cnn_prepare_device()
{
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<false, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, Lds<false>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<false, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, Lds<false>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<false, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, Lds<false>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_family_kernel<32, 64, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsFam<32, 64, 128>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_family_kernel<16, 32, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsFam<16, 32, 64>::LDS_BYTES);
  // (the i8 variant's 51 072 B need no opt-in; its pipelined form's 76 816 B do)
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_pipe_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsPipe::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_pipe_kernel<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsPipe::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_pipe_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsPipe::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_pipe_kernel<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsPipe::LDS_BYTES);
  // the layer kernels that stage a 32 x 32 map of up to 64 channels (+ halo) in LDS: 34 x 34 x 64 B = 73 984 B
#define FHEVC_LAYER_LDS(KCV) \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_layer_conv_kernel<KCV, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_layer_conv_kernel<KCV, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  FHEVC_LAYER_LDS(1) FHEVC_LAYER_LDS(2) FHEVC_LAYER_LDS(3) FHEVC_LAYER_LDS(4)
#undef FHEVC_LAYER_LDS
#define FHEVC_LAYER_FUSE(KCV) \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_layer_conv_kernel<KCV, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); \
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_layer_conv_kernel<KCV, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  FHEVC_LAYER_FUSE(1) FHEVC_LAYER_FUSE(2)
#undef FHEVC_LAYER_FUSE
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_d2_kernel<1, 2, 3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsD2<1, 2, 3>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_d2_kernel<1, 2, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LdsD2<1, 2, 3>::LDS_BYTES);
  // the i8 depth kernel as one 768-thread workgroup per CU: three groups x 50 048 B
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<false, 0, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * Lds<true>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<false, 1, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * Lds<true>::LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<true, 1, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * Lds<true>::LDS_BYTES);
  return e;
}

// d_had != nullptr: the fused source Hadamard (only where fhevc_cnn_can_fuse_hadamard says so)
bool fhevc_cnn_can_fuse_hadamard(const FhevcFrames& fr)
{
  // every thread's 16 samples must come through the aligned fast path or lie wholly outside the picture, the packed
  // 16-bit butterflies need samples below 2^10, and the height must be a multiple of 8: the prefetch zero-fills rows below the
  // picture, so a bottom 8x8 block that is only partly inside would enter the sum with real rows plus zero rows, where
  // updateCtuDataISlice (TEncCu.cpp:1324-1343) and the stand-alone kernel count WHOLE blocks only
  const size_t sb = (size_t)fr.sample_bytes;
  return fr.bit_depth <= 10 && (fr.width % 16) == 0 && (fr.height % 8) == 0 && (reinterpret_cast<uintptr_t>(fr.luma) % 16) == 0 &&
         ((size_t)fr.stride * sb) % 16 == 0 && ((size_t)fr.frame_stride * sb) % 16 == 0;
}

namespace {
template <bool STAMPS, int HAD, int ARITH>
void launch_depth_kernel(int grid, hipStream_t stream, const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_had, int32_t* d_logits,
                         uint32_t* d_flags, unsigned long long* d_stamps, uint8_t* d_depth_max, int margin_split, int margin_stop)
{
  hipLaunchKernelGGL((fhevc_cnn_depth_kernel<STAMPS, HAD, ARITH>), dim3(grid), dim3(256), Lds<ARITH != 0>::LDS_BYTES, stream, fr, w, d_depth, d_had, d_logits,
                     d_flags, d_stamps, d_depth_max, margin_split, margin_stop);
}
// 0: 16-bit MFMAs; 1: i8, general requant; 2: i8 with the short requant forms
int cnn_arith(const FhevcCnnWeights& w) { return !w.i8 ? 0 : (w.requant_mode[1] == 1 && w.requant_mode[2] == 2) ? 2 : 1; }
}  // namespace

// the switches of FhevcKnobs, read once per context (fhevc_create)
FhevcKnobs fhevc_read_knobs()
{
  FhevcKnobs k;
  auto digit = [](const char* e) { return (e && e[0] >= '1' && e[0] <= '4') ? e[0] - '0' : 0; };
  k.wg_per_cu = digit(std::getenv("FHEVC_CNN_WG_PER_CU"));
  k.debug_wg_per_cu = digit(std::getenv("FHEVC_DEBUG_WG_PER_CU"));
  if (const char* rq = std::getenv("FHEVC_CNN_REQUANT")) k.requant_general = std::strcmp(rq, "general") == 0;
  k.family_layers = std::getenv("FHEVC_FAMILY_LAYERS") != nullptr;
  if (const char* fd = std::getenv("FHEVC_FUSED_D2")) k.fused_d2 = fd[0] != '0';
  k.layers_no_fuse = std::getenv("FHEVC_LAYERS_NO_FUSE") != nullptr;
  k.layers_no_dbuf = std::getenv("FHEVC_LAYERS_NO_DBUF") != nullptr;
  if (const char* kb = std::getenv("FHEVC_LAYERS_LDS_KB")) k.layers_lds_limit = (size_t)std::min(80, std::max(8, std::atoi(kb))) * 1024;   // <= the opt-in of fhevc_cnn_prepare_device
  if (const char* lg = std::getenv("FHEVC_LAYERS_GRID")) k.layers_grid = std::atoi(lg);
  if (const char* tr = std::getenv("FHEVC_CNN_TRIO")) k.trio = tr[0] != '0';
  if (const char* rq = std::getenv("FHEVC_D2_REQUANT")) k.d2_requant_general = std::strcmp(rq, "general") == 0;
  return k;
}

// workgroups per CU of the persistent grid (FHEVC_CNN_WG_PER_CU overrides: a tuning knob)
static int cnn_wg_per_cu(const FhevcCnnWeights& w, const FhevcKnobs& knobs)
{
  if (knobs.wg_per_cu) return knobs.wg_per_cu;
  return w.i8 ? FHEVC_I8_WG_PER_CU : 2;  // what the variants' LDS and register budgets are sized for
}

hipError_t fhevc_launch_cnn(const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_had, int32_t* d_logits,
                            uint32_t* d_flags, uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, const FhevcKnobs& knobs, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  int grid = cnn_wg_per_cu(w, knobs) * num_cus;
  if (total < grid) grid = (int)total;
#define FHEVC_LAUNCH(HAD, ARITH) launch_depth_kernel<false, HAD, ARITH>(grid, stream, fr, w, d_depth, d_had, d_logits, d_flags, nullptr, d_depth_max, margin_split, margin_stop)
  // the fused source Hadamard's form: on the MFMA from the staged tile for 8-bit content (the tile IS the samples), on packed
  // 16-bit VALU from the prefetched samples otherwise (the tile is rounded to 8 bits); w.had_valu forces the latter (A/B, tests)
  const int had = d_had == nullptr ? 0 : (fr.bit_depth == 8 && !w.had_valu) ? 2 : 1;
  if (knobs.trio && !knobs.wg_per_cu && !w.pipe && cnn_arith(w) == 2 && had != 2) {  // three groups behind common barriers, one interval apart: one workgroup per CU
    const int tgrid = (int)std::min<long long>(num_cus, (total + 2) / 3);
#define FHEVC_LAUNCH_TRIO(HAD) hipLaunchKernelGGL((fhevc_cnn_depth_kernel<false, HAD, 2, 1>), dim3(tgrid), dim3(768), 3 * Lds<true>::LDS_BYTES, stream, fr, w, d_depth, d_had, \
                                                  d_logits, d_flags, nullptr, d_depth_max, margin_split, margin_stop)
    if (had) FHEVC_LAUNCH_TRIO(1); else FHEVC_LAUNCH_TRIO(0);
#undef FHEVC_LAUNCH_TRIO
    return hipGetLastError();
  }
  if (w.i8 && w.pipe && had != 2) {  // the software-pipelined form of the i8 kernel: two workgroups per CU
    int pgrid = 2 * num_cus;
    if (knobs.wg_per_cu == 1) pgrid = num_cus;
    if (total < pgrid) pgrid = (int)total;
#define FHEVC_LAUNCH_PIPE(HAD, ARITH) hipLaunchKernelGGL((fhevc_cnn_depth_pipe_kernel<HAD, ARITH>), dim3(pgrid), dim3(256), LdsPipe::LDS_BYTES, stream, fr, w, d_depth, d_had, \
                                                         d_logits, d_flags, d_depth_max, margin_split, margin_stop)
    if (cnn_arith(w) == 2) { if (had) FHEVC_LAUNCH_PIPE(1, 2); else FHEVC_LAUNCH_PIPE(0, 2); }
    else { if (had) FHEVC_LAUNCH_PIPE(1, 1); else FHEVC_LAUNCH_PIPE(0, 1); }
#undef FHEVC_LAUNCH_PIPE
    return hipGetLastError();
  }
#define FHEVC_LAUNCH_HAD(ARITH) do { if (had == 2) FHEVC_LAUNCH(2, ARITH); else if (had == 1) FHEVC_LAUNCH(1, ARITH); else FHEVC_LAUNCH(0, ARITH); } while (0)
  switch (cnn_arith(w)) {
    case 2: FHEVC_LAUNCH_HAD(2); break;
    case 1: FHEVC_LAUNCH_HAD(1); break;
    default: FHEVC_LAUNCH_HAD(0); break;
  }
#undef FHEVC_LAUNCH_HAD
#undef FHEVC_LAUNCH
  return hipGetLastError();
}

// diagnostic build of the same kernel with s_memtime stamps; d_stamps: grid * FHEVC_STAMP_SLOTS (8 phase cycle sums, then the loop's shader cycles and 100 MHz ticks)
hipError_t fhevc_launch_cnn_stamped(const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_had, int num_cus, const FhevcKnobs& knobs,
                                    unsigned long long* d_stamps, int* grid_out, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  // FHEVC_DEBUG_WG_PER_CU=1: one workgroup per CU, i.e. the phase times without a second workgroup on the same SIMDs
  int grid = (knobs.debug_wg_per_cu ? knobs.debug_wg_per_cu : cnn_wg_per_cu(w, knobs)) * num_cus;
  if (total < grid) grid = (int)total;
  *grid_out = grid;
  if (total <= 0) return hipSuccess;
  const int arith = cnn_arith(w);
  if (arith == 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_cnn_depth_kernel<true, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, Lds<false>::LDS_BYTES);
    if (e != hipSuccess) return e;
    launch_depth_kernel<true, 0, 0>(grid, stream, fr, w, d_depth, nullptr, nullptr, nullptr, d_stamps, nullptr, 0, 0);
  } else if (arith == 1) launch_depth_kernel<true, 0, 1>(grid, stream, fr, w, d_depth, nullptr, nullptr, nullptr, d_stamps, nullptr, 0, 0);
  else if (d_had && knobs.trio && !knobs.debug_wg_per_cu && !knobs.wg_per_cu) {
    const int tgrid = (int)std::min<long long>(num_cus, (total + 2) / 3);
    *grid_out = 3 * tgrid;
    hipLaunchKernelGGL((fhevc_cnn_depth_kernel<true, 1, 2, 1>), dim3(tgrid), dim3(768), 3 * Lds<true>::LDS_BYTES, stream, fr, w, d_depth, d_had, nullptr, nullptr, d_stamps, nullptr, 0, 0);
  } else if (d_had) launch_depth_kernel<true, 1, 2>(grid, stream, fr, w, d_depth, d_had, nullptr, nullptr, d_stamps, nullptr, 0, 0);  // as bench.py times it: with the fused source Hadamard
  else launch_depth_kernel<true, 0, 2>(grid, stream, fr, w, d_depth, nullptr, nullptr, nullptr, d_stamps, nullptr, 0, 0);
  return hipGetLastError();
}

// ---- the reference's Bayesian-optimisation family, one convolution per block (k_cnn_family.inc) ----
bool fhevc_cnn_family_supported(int c1, int c2, int c3) { return (c1 == 32 && c2 == 64 && c3 == 128) || (c1 == 16 && c2 == 32 && c3 == 64); }

hipError_t fhevc_launch_cnn_family(const FhevcFrames& fr, const FhevcFamilyWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                                   uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  int grid = 2 * num_cus;
  if (total < grid) grid = (int)total;
  if (w.c[0] == 32 && w.c[1] == 64 && w.c[2] == 128)
    hipLaunchKernelGGL((fhevc_cnn_family_kernel<32, 64, 128>), dim3(grid), dim3(256), (LdsFam<32, 64, 128>::LDS_BYTES), stream, fr, w, d_depth, d_logits, d_flags, d_depth_max, margin_split, margin_stop);
  else if (w.c[0] == 16 && w.c[1] == 32 && w.c[2] == 64)
    hipLaunchKernelGGL((fhevc_cnn_family_kernel<16, 32, 64>), dim3(grid), dim3(256), (LdsFam<16, 32, 64>::LDS_BYTES), stream, fr, w, d_depth, d_logits, d_flags, d_depth_max, margin_split, margin_stop);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ---- the members with two convolutions per block as ONE kernel (k_cnn_d2.inc): padded widths 32 / 64 / 96 = the reference's 23 / 46 / 92 x 2 ----
bool fhevc_cnn_d2_supported(const FhevcLayersWeights& w)
{
  static const int cp[6] = { 32, 32, 64, 64, 96, 96 }, pool[6] = { 0, 1, 0, 1, 0, 0 }, hh[6] = { 64, 64, 32, 32, 16, 16 };
  if (w.num_layers != 6 || w.c3_pad != 96) return false;
  for (int i = 0; i < 6; ++i)
    if (w.l[i].cout_pad != cp[i] || (w.l[i].pool != 0) != (pool[i] != 0) || w.l[i].H != hh[i] || w.l[i].kc != (i == 0 ? 0 : cp[i - 1] / 32)) return false;
  return true;
}

hipError_t fhevc_launch_cnn_d2(const FhevcFrames& fr, const FhevcLayersWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                               uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  if (!fhevc_cnn_d2_supported(w)) return hipErrorInvalidValue;
  int grid = num_cus;   // one 512-thread workgroup per CU (119 KB of LDS)
  if (total < grid) grid = (int)total;
  // every layer qualifies for its short requant form (conv1a: shift <= 7; the others: shift 8, accumulators inside 24 bits): the instantiation built for that
  if (w.d2_short) hipLaunchKernelGGL((fhevc_cnn_d2_kernel<1, 2, 3, true>), dim3(grid), dim3(512), (LdsD2<1, 2, 3>::LDS_BYTES), stream, fr, w, d_depth, d_logits, d_flags, d_depth_max, margin_split, margin_stop);
  else hipLaunchKernelGGL((fhevc_cnn_d2_kernel<1, 2, 3, false>), dim3(grid), dim3(512), (LdsD2<1, 2, 3>::LDS_BYTES), stream, fr, w, d_depth, d_logits, d_flags, d_depth_max, margin_split, margin_stop);
  return hipGetLastError();
}

// ---- any member of the family, layer by layer through HBM (k_cnn_layers.inc) ----
void fhevc_layer_lds_image(int kc, int pool, int H, int* pad, int* mask)
{
  // tools/lds_swizzle_search.py: (pad bytes, mask) per [kc - 1][pool][H == 16, 32, 64]; every entry reads at 4 LDS cycles per ds_read_b128 (conflict-free)
  static const unsigned char T[4][2][3][2] = {
    { { { 16, 0 }, { 0, 1 }, { 0, 1 } }, { { 192, 3 }, { 16, 1 }, { 0, 3 } } },
    { { { 32, 1 }, { 0, 3 }, { 0, 3 } }, { { 16, 1 }, { 32, 3 }, { 0, 7 } } },
    { { { 16, 0 }, { 0, 1 }, { 0, 1 } }, { { 64, 3 }, { 16, 1 }, { 0, 3 } } },
    { { { 64, 3 }, { 0, 7 }, { 0, 7 } }, { { 32, 3 }, { 64, 7 }, { 0, 15 } } } };
  *pad = 0; *mask = 0;
  if (kc < 1 || kc > 4 || (H != 16 && H != 32 && H != 64)) return;
  if (!(kc == 2 || kc == 4 || (kc == 3 && pool))) return;   // two-way conflicted plain images stay plain (the kernel's SWZ)
  const unsigned char* e = T[kc - 1][pool ? 1 : 0][H == 16 ? 0 : (H == 32 ? 1 : 2)];
  *pad = e[0]; *mask = e[1];
}

hipError_t fhevc_launch_cnn_layers(const FhevcFrames& fr, const FhevcLayersWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                                   uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, const FhevcKnobs& knobs, hipStream_t stream)
{
  const int total = (fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  const int grid = 3 * num_cus;   // x 4 waves: a multiple of 12 waves (M tiles of 1 .. 4 divide it)
  for (int first = 0; first < total; first += w.chunk) {
    const int count = total - first < w.chunk ? total - first : w.chunk;
    if (fr.sample_bytes == 2) hipLaunchKernelGGL((fhevc_layers_stage_kernel<int16_t>), dim3(count < 4096 ? count : 4096), dim3(256), 0, stream, fr, first, count, w.in0);
    else hipLaunchKernelGGL((fhevc_layers_stage_kernel<uint8_t>), dim3(count < 4096 ? count : 4096), dim3(256), 0, stream, fr, first, count, w.in0);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return le;
    const int8_t* in = w.in0;
    // members with two or three convolutions per block: the first convolution is computed inside the second one's LDS staging (FUSE0) where the
    // second one's strips fit LDS and its width is at most 64
    const bool no_fuse = knobs.layers_no_fuse;   // (tests: the unfused form of the same member)
    const bool fuse0 = !no_fuse && w.num_layers > 3 && w.l[0].kc == 0 && !w.l[0].pool && w.l[1].kc >= 1 && w.l[1].kc <= 2 && w.l[0].cout_pad == w.l[1].kc * 32;
    for (int i = fuse0 ? 1 : 0; i < w.num_layers; ++i) {
      const FhevcLayer& L = w.l[i];
      const FhevcFirstConv first = { w.in0, w.l[0].frag, w.l[0].bias, w.l[0].shift };
      // the input map (or, at 64 x 64, a strip of 32 rows of it) staged in LDS per workgroup item where it fits 80 KB; the first layer reads HBM directly
      int strip = L.H;
      const size_t lds_limit = knobs.layers_lds_limit;   // (experiments: FHEVC_LAYERS_LDS_KB)
      // maps that come from HBM: two buffers where a strip of at least 16 rows fits twice (the next item streams in behind the current one's MFMAs)
      const bool no_dbuf = knobs.layers_no_dbuf;   // (tests, experiments: the single-buffered staging)
      const bool fused_here = fuse0 && i == 1;
      const size_t in_pitch = (size_t)(L.H + 2) * (L.kc * 32) + L.in_pad, extra = fused_here ? 16 + 36 * 66 : 0;
      auto image = [&](int rows) { return (((size_t)(rows + 2) * in_pitch + 255) & ~(size_t)255); };   // a strip's LDS image: whole 256-byte rows (the XOR stays inside one)
      int strip2 = L.H;
      while (strip2 > 8 && 2 * image(strip2) > lds_limit) strip2 >>= 1;
      const int dbuf = (!no_dbuf && !fused_here && L.kc > 0 && strip2 >= 16 && 2 * image(strip2) <= lds_limit) ? 1 : 0;
      if (dbuf) strip = strip2;
      else while (strip > 8 && image(strip) + extra > lds_limit) strip >>= 1;
      const size_t map_bytes = image(strip);
      const bool use_lds = L.kc > 0 && map_bytes + extra <= lds_limit;
      const int lcap = knobs.layers_grid;   // (experiments: FHEVC_LAYERS_GRID)
      const int litems = count * (L.H / strip), lgrid = litems < lcap ? litems : lcap;
#define FHEVC_LAYER(KCV, POOLV) do { if (use_lds && fuse0 && i == 1) hipLaunchKernelGGL((fhevc_layer_conv_kernel<KCV, POOLV, KCV != 0, KCV == 1 || KCV == 2>), dim3(lgrid), dim3(256), map_bytes + 16 + 36 * 66, stream, in, L.out, L.frag, L.bias, L.shift, L.H, L.cout_pad, count, strip, first, 0, L.in_pad, L.out_pad, L.swz); \
                                      else if (use_lds) hipLaunchKernelGGL((fhevc_layer_conv_kernel<KCV, POOLV, KCV != 0>), dim3(lgrid), dim3(256), map_bytes << dbuf, stream, in, L.out, L.frag, L.bias, L.shift, L.H, L.cout_pad, count, strip, first, dbuf, L.in_pad, L.out_pad, L.swz); \
                                      else hipLaunchKernelGGL((fhevc_layer_conv_kernel<KCV, POOLV, false>), dim3(grid), dim3(256), 0, stream, in, L.out, L.frag, L.bias, L.shift, L.H, L.cout_pad, count, L.H, first, 0, L.in_pad, L.out_pad, 0); } while (0)
      switch (L.kc * 2 + (L.pool ? 1 : 0)) {
        case 0: FHEVC_LAYER(0, false); break; case 1: FHEVC_LAYER(0, true); break;
        case 2: FHEVC_LAYER(1, false); break; case 3: FHEVC_LAYER(1, true); break;
        case 4: FHEVC_LAYER(2, false); break; case 5: FHEVC_LAYER(2, true); break;
        case 6: FHEVC_LAYER(3, false); break; case 7: FHEVC_LAYER(3, true); break;
        case 8: FHEVC_LAYER(4, false); break; case 9: FHEVC_LAYER(4, true); break;
        default: return hipErrorInvalidValue;
      }
#undef FHEVC_LAYER
      if (const hipError_t le = hipGetLastError(); le != hipSuccess) return le;   // a refused launch must not let the later layers run on stale activations
      in = L.out;
    }
#define FHEVC_HEADS(CPV) hipLaunchKernelGGL((fhevc_layers_heads_kernel<CPV>), dim3(count < 2 * num_cus ? count : 2 * num_cus), dim3(256), (size_t)18 * 18 * w.c3_pad, stream, fr, in, w.c3, w.c3_pad, w.whead, w.bhead, first, count, d_depth, d_logits, d_flags, d_depth_max, margin_split, margin_stop)
    switch (w.c3_pad >> 5) { case 1: FHEVC_HEADS(1); break; case 2: FHEVC_HEADS(2); break; case 3: FHEVC_HEADS(3); break; case 4: FHEVC_HEADS(4); break; default: return hipErrorInvalidValue; }
#undef FHEVC_HEADS
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return le;
  }
  return hipSuccess;
}
