// k_cnn.hip -- depth classifier for one 64x64 luma CTU per workgroup iteration, gfx950 only.
//
// What it computes (bit-exact twin of oracle/fhevc_oracle.c: fho_cnn_ctu + fho_depth_from_logits):
//   the network the reference specifies in matlab/dataExtraction/Train...Example.m:75-96
//   (conv3x3x16 -> ReLU -> maxpool2 -> conv3x3x32 -> ReLU -> maxpool2 -> conv3x3x64 -> ReLU -> FC(2)),
//   run convolutionally over the CTU, with three FC heads (64-, 32- and 16-level split decisions) and the
//   top-down assembly of the 16x16 depth map that TEncCu::xCompressCU consumes (TEncCu.cpp:496-1058).
//
// How:
//   * persistent workgroups (256 threads = 4 waves), grid = 2 per CU, grid-stride over CTUs;
//   * every conv layer is an im2col GEMM on v_mfma_f32_32x32x16_bf16 with A = weights (rows = output channels,
//     resident in VGPRs for the whole kernel) and B = im2col (columns = 32 spatial positions) read from LDS
//     with one ds_read_b128 per K-step: activations live in LDS as 8-channel planes [y][x][8] bf16 so the
//     16-lane groups of a ds_read_b128 cover 256 contiguous bytes;
//   * operands are fixed-point integers (|w| <= 127, activations 0..255, |acc| < 2^24) so the fp32
//     accumulation is exact in any order -> the integer depth map is bit-exact against the CPU oracle;
//   * ReLU/requant/max-pool are fused into the MFMA epilogue (in-lane max for the vertical pair,
//     DPP quad_perm for the horizontal pair), FC heads run on v_dot4_u32_u8.
#include "fhevc_internal.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

namespace {

// ---- LDS map (bytes) ---------------------------------------------------------------------------------------
constexpr int A1_PITCH = 34;                      // conv1 output 32x32 + 1 halo each side, positions per row
constexpr int A1_PLANE = 34 * 34 * 16;            // one 8-channel plane: 16 B per position
constexpr int R1_OFF = 0;                         // R1: A1 (2 planes); later A3 u8 [256][64] + pooled [64][64]
constexpr int R1_BYTES = 2 * A1_PLANE;            // 36992
constexpr int A3_OFF = R1_OFF;
constexpr int P3_OFF = R1_OFF + 16384;            // maxpool2x2(a3): [8*8][64] u8
constexpr int A2_PITCH = 18;                      // conv2 output 16x16 + halo
constexpr int A2_PLANE = 18 * 18 * 16;            // 5184
constexpr int R2_OFF = R1_OFF + R1_BYTES;         // R2: input CTU bf16 [66][68]; later A2 (4 planes)
constexpr int R2_BYTES = 4 * A2_PLANE;            // 20736
constexpr int IN_PITCH = 68;                      // bf16 elements per input row (66 used)
constexpr int BIAS_OFF = R2_OFF + R2_BYTES;       // float b1[16] b2[32] b3[64]
constexpr int LOGIT_OFF = BIAS_OFF + 112 * 4;     // int logits[21][2]
constexpr int LDS_BYTES = LOGIT_OFF + 48 * 4;     // 58368 -> two workgroups per CU
static_assert(66 * IN_PITCH * 2 <= R2_BYTES, "input tile must fit the A2 region");
static_assert(P3_OFF + 4096 <= R1_OFF + R1_BYTES, "pooled map must fit R1");

constexpr int HEAD64_OFF = 0, HEAD32_OFF = 2 * 4096, HEAD16_OFF = 4 * 4096;  // into whead (uint8, w+128)

__device__ __forceinline__ float dpp_xor1(float v)  // value of the horizontally adjacent lane (quad_perm [1,0,3,2])
{
  int i = __builtin_bit_cast(int, v);
  i = __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false);
  return __builtin_bit_cast(float, i);
}
// two fp32 holding integers 0..255 -> two bf16 (exact: the low 16 mantissa bits are zero)
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// ReLU + fixed-point requantisation: clamp(floor(acc * 2^-shift), 0, 255); acc*2^-shift is exact in fp32
__device__ __forceinline__ float requant(float acc, float scale)
{
  return __builtin_amdgcn_fmed3f(floorf(acc * scale), 0.0f, 255.0f);
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
__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c)
{
  return __builtin_amdgcn_udot4(a, b, c, false);
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

template <typename T>
__device__ __forceinline__ int load_centered(const T* p, int shift)
{
  int v = (int)*p;
  if (shift > 0) v = min(255, (v + (1 << (shift - 1))) >> shift);
  v = min(255, max(0, v));
  return v - 128;
}

__global__ __launch_bounds__(256, 2) void fhevc_cnn_depth_kernel(FhevcFrames F, FhevcCnnWeights W,
                                                                  uint8_t* __restrict__ d_depth,
                                                                  int32_t* __restrict__ d_logits)
{
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;

  // ---- resident weight fragments (A operands) ----
  const bf16x8 wA1 = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV1 + lane]);
  bf16x8 wA2[9];
#pragma unroll
  for (int s = 0; s < 9; ++s) wA2[s] = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV2 + s * 64 + lane]);
  bf16x8 wA3[18];
  const int tile3 = wave & 1;
#pragma unroll
  for (int s = 0; s < 18; ++s) wA3[s] = __builtin_bit_cast(bf16x8, W.frag[FHEVC_FRAG_CONV3 + (tile3 * 18 + s) * 64 + lane]);

  float* biasL = reinterpret_cast<float*>(lds + BIAS_OFF);
  int* logitL = reinterpret_cast<int*>(lds + LOGIT_OFF);
  if (tid < 112) biasL[tid] = W.bias[tid];

  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  const int shift_in = F.bit_depth - 8;

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = work / per_frame;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + rem / F.ctus_x;
    const int cx = rem % F.ctus_x;

    // ================= P0: CTU -> LDS (centred 8-bit as bf16, halo 0); zero the A1 halo =================
    {
      unsigned short* in = reinterpret_cast<unsigned short*>(lds + R2_OFF);
      const int row = tid >> 2, seg = tid & 3;
      const int py = cy * 64 + row, px0 = cx * 64 + seg * 16;
      const long long base = (long long)f * F.frame_stride + (long long)py * F.stride + px0;
      unsigned short* dst = in + (row + 1) * IN_PITCH + seg * 16 + 1;
      const bool row_ok = py < F.height;
      if (F.sample_bytes == 2) {
        const int16_t* src = reinterpret_cast<const int16_t*>(F.luma) + base;
        if (row_ok && px0 + 16 <= F.width && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
          const uint4 q0 = *reinterpret_cast<const uint4*>(src), q1 = *reinterpret_cast<const uint4*>(src + 8);
          const unsigned wds[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            short s0 = (short)(wds[j] & 0xFFFF), s1 = (short)(wds[j] >> 16);
            dst[2 * j] = (unsigned short)(__float_as_uint((float)load_centered(&s0, shift_in)) >> 16);
            dst[2 * j + 1] = (unsigned short)(__float_as_uint((float)load_centered(&s1, shift_in)) >> 16);
          }
        } else {
#pragma unroll 4
          for (int j = 0; j < 16; ++j) {
            int v = 0;
            if (row_ok && px0 + j < F.width) v = load_centered(src + j, shift_in);
            dst[j] = (unsigned short)(__float_as_uint((float)v) >> 16);
          }
        }
      } else {
        const uint8_t* src = reinterpret_cast<const uint8_t*>(F.luma) + base;
        if (row_ok && px0 + 16 <= F.width && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
          const uint4 q = *reinterpret_cast<const uint4*>(src);
          const unsigned wds[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            int v = (int)((wds[j >> 2] >> (8 * (j & 3))) & 0xFF) - 128;
            dst[j] = (unsigned short)(__float_as_uint((float)v) >> 16);
          }
        } else {
#pragma unroll 4
          for (int j = 0; j < 16; ++j) {
            int v = 0;
            if (row_ok && px0 + j < F.width) v = (int)src[j] - 128;
            dst[j] = (unsigned short)(__float_as_uint((float)v) >> 16);
          }
        }
      }
      // input halo: 66*66 - 64*64 = 260 positions
      for (int e = tid; e < 260; e += 256) {
        int y, x;
        if (e < 66) { y = 0; x = e; }
        else if (e < 132) { y = 65; x = e - 66; }
        else { const int k = e - 132; y = 1 + (k >> 1); x = (k & 1) ? 65 : 0; }
        in[y * IN_PITCH + x] = 0;
      }
      // A1 halo: 132 positions x 2 planes
      for (int e = tid; e < 264; e += 256) {
        const int pl = e / 132, k0 = e - pl * 132;
        int y, x;
        if (k0 < 34) { y = 0; x = k0; }
        else if (k0 < 68) { y = 33; x = k0 - 34; }
        else { const int k = k0 - 68; y = 1 + (k >> 1); x = (k & 1) ? 33 : 0; }
        *reinterpret_cast<uint4*>(lds + R1_OFF + pl * A1_PLANE + (y * A1_PITCH + x) * 16) = make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();

    // ================= P1: conv1 (1 -> 16), K = 9 taps padded to 16, fused maxpool + requant =================
    {
      if (tid < 42) {  // logits start from the head biases
        const int k = tid >> 1, cls = tid & 1;
        logitL[tid] = W.bhead[(k == 0 ? 0 : (k < 5 ? 2 : 4)) + cls];
      }
      const unsigned short* in = reinterpret_cast<const unsigned short*>(lds + R2_OFF);
      int toff[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) toff[j] = h ? (j == 0 ? 2 * IN_PITCH + 2 : 0) : ((j / 3) * IN_PITCH + (j % 3));
      const unsigned hmask = h ? 0u : 0xFFFFFFFFu;
      f32x16 bias1;
#pragma unroll
      for (int i = 0; i < 16; ++i) bias1[i] = 0.0f;
      {
        const float4 b0 = *reinterpret_cast<const float4*>(biasL + 4 * h);
        const float4 b1 = *reinterpret_cast<const float4*>(biasL + 8 + 4 * h);
        bias1[0] = b0.x; bias1[1] = b0.y; bias1[2] = b0.z; bias1[3] = b0.w;
        bias1[4] = b1.x; bias1[5] = b1.y; bias1[6] = b1.z; bias1[7] = b1.w;
      }
#pragma unroll 1
      for (int i = 0; i < 16; ++i) {
        const int u = wave + 4 * i;
        const int yp = u >> 1, xh = u & 1;
        const int x = 32 * xh + r;
        f32x16 acc[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          const int base = (2 * yp + rr) * IN_PITCH + x;
          unsigned e[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) e[j] = in[base + toff[j]];
          uint4 q;
          q.x = e[0] | ((e[1] & hmask) << 16);
          q.y = (e[2] | (e[3] << 16)) & hmask;
          q.z = (e[4] | (e[5] << 16)) & hmask;
          q.w = (e[6] | (e[7] << 16)) & hmask;
          acc[rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA1, __builtin_bit_cast(bf16x8, q), bias1, 0, 0, 0);
        }
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float m = fmaxf(acc[0][k], acc[1][k]);
          m = fmaxf(m, dpp_xor1(m));
          v[k] = requant(m, W.scale[0]);
        }
        // channels: regs 0-3 -> 4h+k (plane 0), regs 4-7 -> 8+4h+k (plane 1); even lane stores plane 0, odd plane 1
        const unsigned p0a = pack_bf16(v[0], v[1]), p0b = pack_bf16(v[2], v[3]);
        const unsigned p1a = pack_bf16(v[4], v[5]), p1b = pack_bf16(v[6], v[7]);
        const int pl = r & 1;
        uint2 o;
        o.x = pl ? p1a : p0a;
        o.y = pl ? p1b : p0b;
        const int pcol = 16 * xh + (r >> 1) + 1;
        *reinterpret_cast<uint2*>(lds + R1_OFF + pl * A1_PLANE + ((yp + 1) * A1_PITCH + pcol) * 16 + h * 8) = o;
      }
    }
    __syncthreads();

    // ================= P2: conv2 (16 -> 32), K = 9 taps x 16 ch, fused maxpool + requant =================
    {
      // the input tile is dead: zero the A2 halo (68 positions x 4 planes) while conv2 fills the interior
      for (int e = tid; e < 272; e += 256) {
        const int pl = e / 68, k0 = e - pl * 68;
        int y, x;
        if (k0 < 18) { y = 0; x = k0; }
        else if (k0 < 36) { y = 17; x = k0 - 18; }
        else { const int k = k0 - 36; y = 1 + (k >> 1); x = (k & 1) ? 17 : 0; }
        *reinterpret_cast<uint4*>(lds + R2_OFF + pl * A2_PLANE + (y * A2_PITCH + x) * 16) = make_uint4(0, 0, 0, 0);
      }
      const unsigned char* a1 = lds + R1_OFF + h * A1_PLANE + r * 16;
      const f32x16 bias2 = bias_tile(biasL + 16, h);
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int yp = wave + 4 * i;
        f32x16 acc0 = bias2, acc1 = bias2;
#pragma unroll
        for (int ir = 0; ir < 4; ++ir) {  // input rows 2yp-1 .. 2yp+2 (halo coordinates 2yp .. 2yp+3)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bf16x8 b = lds_frag(a1 + ((2 * yp + ir) * A1_PITCH + kx) * 16);
            if (ir < 3) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA2[ir * 3 + kx], b, acc0, 0, 0, 0);
            if (ir > 0) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA2[(ir - 1) * 3 + kx], b, acc1, 0, 0, 0);
          }
        }
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          float m = fmaxf(acc0[k], acc1[k]);
          m = fmaxf(m, dpp_xor1(m));
          v[k] = requant(m, W.scale[1]);
        }
        if ((r & 1) == 0) {  // regs 4g..4g+3 -> channels 8g+4h.. of plane g
          unsigned char* dst = lds + R2_OFF + ((yp + 1) * A2_PITCH + (r >> 1) + 1) * 16 + h * 8;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(dst + g * A2_PLANE) =
                make_uint2(pack_bf16(v[4 * g], v[4 * g + 1]), pack_bf16(v[4 * g + 2], v[4 * g + 3]));
        }
      }
    }
    __syncthreads();

    // ================= P3: conv3 (32 -> 64), K = 9 taps x 32 ch, requant to u8 =================
    {
      const int yy = r >> 4, x = r & 15;
      const f32x16 bias3 = bias_tile(biasL + 48 + 32 * tile3, h);
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int yp = (wave >> 1) + 2 * i;
        const unsigned char* a2 = lds + R2_OFF + h * A2_PLANE + ((2 * yp + yy) * A2_PITCH + x) * 16;
        f32x16 acc = bias3;
#pragma unroll
        for (int s = 0; s < 18; ++s) {
          const int tap = s >> 1, cb = s & 1;
          const bf16x8 b = lds_frag(a2 + 2 * cb * A2_PLANE + ((tap / 3) * A2_PITCH + (tap % 3)) * 16);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA3[s], b, acc, 0, 0, 0);
        }
        unsigned char* dst = lds + A3_OFF + ((2 * yp + yy) * 16 + x) * 64 + 32 * tile3 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          unsigned d = 0;
#pragma unroll
          for (int k = 0; k < 4; ++k) d = __builtin_amdgcn_cvt_pk_u8_f32(requant(acc[4 * g + k], W.scale[2]), k, d);
          *reinterpret_cast<unsigned*>(dst + 8 * g) = d;
        }
      }
    }
    __syncthreads();

    // ================= P4: FC heads on v_dot4_u32_u8 (weights stored as w+128) =================
    {
      const int p = tid, y = p >> 4, x = p & 15;
      const unsigned char* arow = lds + A3_OFF + p * 64;
      const unsigned char* w16 = W.whead + HEAD16_OFF + ((y & 3) * 4 + (x & 3)) * 64;
      const unsigned char* w32 = W.whead + HEAD32_OFF + ((y & 7) * 8 + (x & 7)) * 64;
      unsigned sa = 0, s16a = 0, s16b = 0, s32a = 0, s32b = 0;
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {  // 16 channels per step: keeps the register footprint of this phase small
        const uint4 a = *reinterpret_cast<const uint4*>(arow + q * 16);
        const uint4 b0 = *reinterpret_cast<const uint4*>(w16 + q * 16), b1 = *reinterpret_cast<const uint4*>(w16 + 1024 + q * 16);
        const uint4 c0 = *reinterpret_cast<const uint4*>(w32 + q * 16), c1 = *reinterpret_cast<const uint4*>(w32 + 4096 + q * 16);
        sa = udot4(a.x, 0x01010101u, sa); sa = udot4(a.y, 0x01010101u, sa); sa = udot4(a.z, 0x01010101u, sa); sa = udot4(a.w, 0x01010101u, sa);
        s16a = udot4(a.x, b0.x, s16a); s16a = udot4(a.y, b0.y, s16a); s16a = udot4(a.z, b0.z, s16a); s16a = udot4(a.w, b0.w, s16a);
        s16b = udot4(a.x, b1.x, s16b); s16b = udot4(a.y, b1.y, s16b); s16b = udot4(a.z, b1.z, s16b); s16b = udot4(a.w, b1.w, s16b);
        s32a = udot4(a.x, c0.x, s32a); s32a = udot4(a.y, c0.y, s32a); s32a = udot4(a.z, c0.z, s32a); s32a = udot4(a.w, c0.w, s32a);
        s32b = udot4(a.x, c1.x, s32b); s32b = udot4(a.y, c1.y, s32b); s32b = udot4(a.z, c1.z, s32b); s32b = udot4(a.w, c1.w, s32b);
      }
      const int part16[2] = { (int)s16a - 128 * (int)sa, (int)s16b - 128 * (int)sa };
      const int part32[2] = { (int)s32a - 128 * (int)sa, (int)s32b - 128 * (int)sa };
      // 64-level head: features = maxpool2x2(a3); the thread at an even (y, x) owns pooled position (y/2, x/2)
      int part64[2] = { 0, 0 };
      if (((y | x) & 1) == 0) {
        unsigned sp = 0, s0 = 0, s1 = 0;
        const unsigned char* w64 = W.whead + HEAD64_OFF + ((y >> 1) * 8 + (x >> 1)) * 64;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
          const uint4 a = *reinterpret_cast<const uint4*>(arow + q * 16);
          const uint4 b = *reinterpret_cast<const uint4*>(arow + 64 + q * 16);
          const uint4 c = *reinterpret_cast<const uint4*>(arow + 16 * 64 + q * 16);
          const uint4 d = *reinterpret_cast<const uint4*>(arow + 17 * 64 + q * 16);
          uint4 m;
          m.x = bytemax(bytemax(a.x, b.x), bytemax(c.x, d.x));
          m.y = bytemax(bytemax(a.y, b.y), bytemax(c.y, d.y));
          m.z = bytemax(bytemax(a.z, b.z), bytemax(c.z, d.z));
          m.w = bytemax(bytemax(a.w, b.w), bytemax(c.w, d.w));
          const uint4 u0 = *reinterpret_cast<const uint4*>(w64 + q * 16), u1 = *reinterpret_cast<const uint4*>(w64 + 4096 + q * 16);
          sp = udot4(m.x, 0x01010101u, sp); sp = udot4(m.y, 0x01010101u, sp);
          sp = udot4(m.z, 0x01010101u, sp); sp = udot4(m.w, 0x01010101u, sp);
          s0 = udot4(m.x, u0.x, s0); s0 = udot4(m.y, u0.y, s0); s0 = udot4(m.z, u0.z, s0); s0 = udot4(m.w, u0.w, s0);
          s1 = udot4(m.x, u1.x, s1); s1 = udot4(m.y, u1.y, s1); s1 = udot4(m.z, u1.z, s1); s1 = udot4(m.w, u1.w, s1);
        }
        part64[0] = (int)s0 - 128 * (int)sp;
        part64[1] = (int)s1 - 128 * (int)sp;
      }
      // wave = 4 rows of 16 positions: lane bits 0-3 = x, bits 4-5 = y & 3
#pragma unroll
      for (int cls = 0; cls < 2; ++cls) {
        int s16 = part16[cls], s32 = part32[cls], s64 = part64[cls];
        s16 += __shfl_xor(s16, 1); s16 += __shfl_xor(s16, 2); s16 += __shfl_xor(s16, 16); s16 += __shfl_xor(s16, 32);
        s32 += __shfl_xor(s32, 1); s32 += __shfl_xor(s32, 2); s32 += __shfl_xor(s32, 4); s32 += __shfl_xor(s32, 16); s32 += __shfl_xor(s32, 32);
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) s64 += __shfl_xor(s64, m);
        if ((lane & 0x33) == 0) atomicAdd(&logitL[(5 + wave * 4 + (lane >> 2)) * 2 + cls], s16);  // one owner per 16x16 block
        if ((lane & 0x37) == 0) atomicAdd(&logitL[(1 + (wave >> 1) * 2 + (lane >> 3)) * 2 + cls], s32);
        if (lane == 0) atomicAdd(&logitL[cls], s64);
      }
    }
    __syncthreads();

    // ================= P5: top-down depth map (forced split at the picture edge) =================
    {
      const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
      const int x = (tid & 15) * 4, y = (tid >> 4) * 4;
      int d = 0;
      if (x < vw && y < vh) {
        const bool s64 = (vw < 64 || vh < 64) || (logitL[1] > logitL[0]);
        if (s64) {
          const int q = (y >> 5) * 2 + (x >> 5);
          const bool cross32 = ((x >> 5) * 32 + 32 > vw) || ((y >> 5) * 32 + 32 > vh);
          if (cross32 || logitL[(1 + q) * 2 + 1] > logitL[(1 + q) * 2]) {
            const int bi = (y >> 4) * 4 + (x >> 4);
            const bool cross16 = ((x >> 4) * 16 + 16 > vw) || ((y >> 4) * 16 + 16 > vh);
            d = (cross16 || logitL[(5 + bi) * 2 + 1] > logitL[(5 + bi) * 2]) ? 3 : 2;
          } else {
            d = 1;
          }
        }
      }
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      d_depth[o * 256 + tid] = (uint8_t)d;
      if (d_logits != nullptr && tid < 42) d_logits[o * 42 + tid] = logitL[tid];
    }
    // no barrier needed here: the next iteration's first LDS writes (P0) touch R1/R2, last read before the
    // P4 barrier, and the logits are re-initialised only after the P0 barrier.
  }
}

}  // namespace

hipError_t fhevc_launch_cnn(const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_logits,
                            int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  int grid = 2 * num_cus;
  if (total < grid) grid = (int)total;
  hipLaunchKernelGGL(fhevc_cnn_depth_kernel, dim3(grid), dim3(256), 0, stream, fr, w, d_depth, d_logits);
  return hipGetLastError();
}
