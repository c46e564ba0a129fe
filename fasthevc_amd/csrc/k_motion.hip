// k_motion.hip -- source-only integer motion search per CU node (config 4: P slices), gfx950 only.
//
// Bit-exact twin of oracle/fhevc_oracle.c: fho_motion_ctu_dist.  For every CU node (64x64, 32x32, 16x16, 8x8: 85 per CTU) of a
// picture: full search over [-R, R]^2 integer vectors in the PREVIOUS ORIGINAL picture, cost = distortion + vector cost, raster
// order over the window and strict "<" as TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848), vector cost as
// TComRdCost::getCostOfVectorWithPredictor (TComRdCost.h:166-174; the host tabulates it with HM's double arithmetic),
// reference samples outside the picture replicated from the border (TComPicYuv::extendPicBorder, TComPicYuv.cpp:229-270).
// Two distortions (template argument SAD):
//   SAD  -- what HM's integer search uses: xPatternSearch's setDistParam selects DF_SAD (TComRdCost.cpp:205-236, xGetSAD* :518-..).
//           In this mode the kernel reproduces the reference's own xPatternSearch bit for bit (vector, SAD, cost): the oracle is pinned
//           to tests/golden/ref_pattern_search.npz, the kernel to the oracle.  One v_sad_u16 per pair of samples.
//   SATD -- TComRdCost::xGetHADs (TComRdCost.cpp:1753-1824) at integer positions.  HM applies Hadamard to the FRACTIONAL refinement
//           only (HadamardME, TEncSearch.cpp:836; cfg/encoder_lowdelay_P_main.cfg:37): at integer positions it is this build's own
//           choice (the P-picture rule's features are fitted on it), the default of fhevc_motion_search.
// Either distortion of a node is the sum of its 8x8 tiles' (both shift the block's sum once), so ONE pass over the 64 tiles of a
// CTU per vector serves all four levels.
//
// Mapping: workgroup (4 waves) = one CTU at a time, grid-stride; lane = one 8x8 tile (its 64 original samples stay in
// registers as 32 packed pairs); the four waves split the vectors of the window; the reference window ((64 + 2R)^2
// samples, border replicated) is staged in LDS once per CTU: each HBM sample is read once per CTU (+ the 2R halo).
// Per vector a lane reads its displaced 8x8 block from LDS (dword reads + v_alignbit for odd offsets), subtracts, runs the
// packed-16 Hadamard of k_hadamard.hip (|coefficients| stay below 2^15 through five stages up to 10 bit; the sixth, inside a
// packed pair, is folded into the absolute sum: |a+b| + |a-b| = 2 max(|a|,|b|)) and the level sums meet through lane shuffles.
#include "fhevc_internal.h"

namespace {

// MR = the largest search range an instantiation is laid out for: FHEVC_MOTION_MAX_RANGE (8: the window in 15 KB of static LDS, the vector costs in the
// kernel arguments) or FHEVC_MOTION_WIDE_MAX_RANGE (64, round 4: HM's own SearchRange for content ABOVE 8 bit, where k_motion_wide.hip's byte SADs do
// not apply -- the window in 76.8 KB of dynamic LDS, the (2 R + 1)^2 vector costs in HBM).  Same code, same raster order, same first-found minimum.
template <int MR> struct MotionGeom {
  static constexpr int RP = 64 + 2 * MR + 8;  // LDS row pitch of the reference window in samples (multiple of 8: 16-byte row starts)
  static constexpr int WIN_ROWS = 64 + 2 * MR;
  static constexpr int REF_SAMPLES = WIN_ROWS * RP + 8;
};

typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk_abs(unsigned a)
{
  const i16x2 v = __builtin_bit_cast(i16x2, a);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, (i16x2)(-v)));
}
// sum of |WHT8x8(d)| of a block held as 8 rows x 4 packed pairs (low half = even column), |samples| < 2^10
__device__ __forceinline__ unsigned had8x8_packed(unsigned (&d)[32])
{
#pragma unroll
  for (int hs = 1; hs < 8; hs <<= 1)  // vertical: rows y, y + hs
#pragma unroll
    for (int i = 0; i < 8; i += hs << 1)
#pragma unroll
      for (int y = i; y < i + hs; ++y)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned a = d[y * 4 + j], b = d[(y + hs) * 4 + j];
          d[y * 4 + j] = pk_add(a, b); d[(y + hs) * 4 + j] = pk_sub(a, b);
        }
#pragma unroll
  for (int hs = 1; hs < 4; hs <<= 1)  // horizontal distance 2 and 4: pairs j, j + hs
#pragma unroll
    for (int y = 0; y < 8; ++y)
#pragma unroll
      for (int i = 0; i < 4; i += hs << 1)
#pragma unroll
        for (int j = i; j < i + hs; ++j) {
          const unsigned a = d[y * 4 + j], b = d[y * 4 + j + hs];
          d[y * 4 + j] = pk_add(a, b); d[y * 4 + j + hs] = pk_sub(a, b);
        }
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {  // horizontal distance 1, inside a pair: |a + b| + |a - b| = 2 max(|a|, |b|)
    const unsigned a = pk_abs(d[i]);
    acc += max(a & 0xFFFFu, a >> 16);
  }
  return 2 * acc;
}
// 32-bit twin (12-bit content): d[64] row-major
__device__ __forceinline__ unsigned had8x8_wide(int (&v)[64])
{
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int hs = 1; hs < 8; hs <<= 1)
#pragma unroll
      for (int i = 0; i < 8; i += hs << 1)
#pragma unroll
        for (int j = i; j < i + hs; ++j) {
          const int a = v[8 * y + j], b = v[8 * y + j + hs];
          v[8 * y + j] = a + b; v[8 * y + j + hs] = a - b;
        }
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int hs = 1; hs < 8; hs <<= 1)
#pragma unroll
      for (int i = 0; i < 8; i += hs << 1)
#pragma unroll
        for (int j = i; j < i + hs; ++j) {
          const int a = v[8 * j + x], b = v[8 * (j + hs) + x];
          v[8 * j + x] = a + b; v[8 * (j + hs) + x] = a - b;
        }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += (unsigned)abs(v[i]);
  return s;
}

template <typename T>
__device__ __forceinline__ int sample_at(const T* plane, long long off) { return (int)plane[off]; }

// T = int16_t (HM Pel planes) or uint8_t; PACKED = bit depth <= 10; SAD: see the header
template <typename T, bool PACKED, bool SAD, int MR>
__global__ __launch_bounds__(256) void fhevc_motion_kernel(FhevcFrames F, int range, FhevcMvCost mvc, const uint32_t* __restrict__ mvtab, FhevcMotionNode* __restrict__ out)
{
  constexpr int RP = MotionGeom<MR>::RP;
  constexpr bool BIG = MR > FHEVC_MOTION_MAX_RANGE;
  extern __shared__ __attribute__((aligned(16))) short s_dyn[];
  __shared__ __attribute__((aligned(16))) short s_small[BIG ? 8 : MotionGeom<MR>::REF_SAMPLES];
  short* const s_ref = BIG ? s_dyn : s_small;
  __shared__ unsigned s_cost[4][4][64], s_satd[4][4][64], s_idx[4][4][64], s_zero[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * (F.num_frames - 1);  // frame f >= 1 is searched in frame f - 1
  const int side = 2 * range + 1, nmv = side * side, centre = (nmv - 1) >> 1;
  const int win = 64 + 2 * range;
  const int shift = F.bit_depth - 8;
  const int delta = (8 - (range & 7)) & 7;  // the window starts at column 64 cx - range: delta samples after a multiple of 8
  const T* plane = reinterpret_cast<const T*>(F.luma);

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = 1 + work / per_frame;
    const int rem = work % per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const long long cur_base = (long long)f * F.frame_stride, ref_base = (long long)(f - 1) * F.frame_stride;
    // ---- stage the reference window: rows cy*64 - R .. + win, columns cx*64 - R .. + win, coordinates clamped to the picture ----
    __syncthreads();  // the previous CTU's readers are done
    {
      // chunks of 8 samples starting at a column that is a multiple of 8 (delta = what the window's first column lacks to one): a chunk
      // inside the picture is ONE 16-byte (uint8 planes: 8-byte) load where the plane allows it, and one 16-byte LDS store
      const int chunks = (win + delta + 7) >> 3;
      for (int it = tid; it < win * chunks; it += 256) {
        const int wr = it / chunks, wc = (it - wr * chunks) * 8;
        const int py = min(max(cy * 64 - range + wr, 0), F.height - 1);
        const int px0 = cx * 64 - range - delta + wc;
        short v[8];
        const long long row = ref_base + (long long)py * F.stride;
        const T* src = plane + row + px0;
        if (px0 >= 0 && px0 + 8 <= F.width && (reinterpret_cast<uintptr_t>(src) & (8 * sizeof(T) - 1)) == 0) {
          if (sizeof(T) == 2) {
            const uint4 q = *reinterpret_cast<const uint4*>(src);
            *reinterpret_cast<uint4*>(s_ref + wr * RP + wc) = q;
            continue;
          } else {
            const uint2 q = *reinterpret_cast<const uint2*>(src);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = (short)((q.x >> (8 * k)) & 0xFF); v[4 + k] = (short)((q.y >> (8 * k)) & 0xFF); }
          }
        } else if (px0 >= 0 && px0 + 8 <= F.width) {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (short)sample_at(plane, row + px0 + k);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (short)sample_at(plane, row + min(max(px0 + k, 0), F.width - 1));
        }
        short* dst = s_ref + wr * RP + wc;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = v[k];
      }
    }
    // ---- this lane's original 8x8 tile (all four waves hold the same 64 tiles) ----
    const int px = cx * 64 + tx * 8, py = cy * 64 + ty * 8;
    const bool inside = (px + 8 <= F.width) && (py + 8 <= F.height);
    unsigned O[32];
    if (inside) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const long long row = cur_base + (long long)(py + j) * F.stride + px;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          O[4 * j + k] = ((unsigned)sample_at(plane, row + 2 * k) & 0xFFFFu) | ((unsigned)sample_at(plane, row + 2 * k + 1) << 16);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 32; ++i) O[i] = 0;
    }
    __syncthreads();

    unsigned bc[4], bs[4], bi[4], zero8 = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) { bc[l] = 0xFFFFFFFFu; bs[l] = 0; bi[l] = 0; }
    unsigned z[4] = { 0, 0, 0, 0 };
    (void)zero8;
    for (int m = wave; m < nmv; m += 4) {  // raster order inside a wave; the waves interleave and are merged by (cost, index)
      const int dy = m / side - range, dx = m % side - range;
      const int col = tx * 8 + range + dx + delta, row0 = ty * 8 + range + dy;
      const unsigned sh = (unsigned)(col & 1) * 16u;  // uniform: R + dx
      unsigned t8;
      if (PACKED && SAD) {  // sum |org - ref| on pairs of unsigned 16-bit samples: v_sad_u16
        t8 = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned* q = reinterpret_cast<const unsigned*>(s_ref) + (((row0 + j) * RP + col) >> 1);
          const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
          t8 = __builtin_amdgcn_sad_u16(O[4 * j + 0], __builtin_amdgcn_alignbit(d1, d0, sh), t8);
          t8 = __builtin_amdgcn_sad_u16(O[4 * j + 1], __builtin_amdgcn_alignbit(d2, d1, sh), t8);
          t8 = __builtin_amdgcn_sad_u16(O[4 * j + 2], __builtin_amdgcn_alignbit(d3, d2, sh), t8);
          t8 = __builtin_amdgcn_sad_u16(O[4 * j + 3], __builtin_amdgcn_alignbit(d4, d3, sh), t8);
        }
      } else if (PACKED) {
        unsigned D[32];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const unsigned* q = reinterpret_cast<const unsigned*>(s_ref) + (((row0 + j) * RP + col) >> 1);
          const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
          D[4 * j + 0] = pk_sub(O[4 * j + 0], __builtin_amdgcn_alignbit(d1, d0, sh));
          D[4 * j + 1] = pk_sub(O[4 * j + 1], __builtin_amdgcn_alignbit(d2, d1, sh));
          D[4 * j + 2] = pk_sub(O[4 * j + 2], __builtin_amdgcn_alignbit(d3, d2, sh));
          D[4 * j + 3] = pk_sub(O[4 * j + 3], __builtin_amdgcn_alignbit(d4, d3, sh));
        }
        t8 = had8x8_packed(D);
      } else {
        int v[64];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned o = O[4 * j + (k >> 1)];
            const int os = (k & 1) ? (int)(short)(o >> 16) : (int)(short)(o & 0xFFFFu);
            v[8 * j + k] = os - (int)s_ref[(row0 + j) * RP + col + k];
          }
        if (SAD) {
          t8 = 0;
#pragma unroll
          for (int i = 0; i < 64; ++i) t8 += (unsigned)abs(v[i]);
        } else t8 = had8x8_wide(v);
      }
      if (SAD) t8 = inside ? t8 : 0u;
      else t8 = inside ? ((t8 + 2) >> 2) : 0u;  // xCalcHADs8x8: (sum + 2) >> 2 (TComRdCost.cpp:1747)
      // node sums: 16x16 = tiles (tx ^ 1, ty ^ 1), 32x32 = + bits 1, 64x64 = + bits 2
      unsigned s[4];
      s[3] = t8;
      unsigned a = t8 + __shfl_xor(t8, 1);
      s[2] = a + __shfl_xor(a, 8);
      a = s[2] + __shfl_xor(s[2], 2);
      s[1] = a + __shfl_xor(a, 16);
      a = s[1] + __shfl_xor(s[1], 4);
      s[0] = a + __shfl_xor(a, 32);
      const unsigned vc = BIG ? mvtab[m] : mvc.c[m];
#pragma unroll
      for (int l = 0; l < 4; ++l) {
        const unsigned sd = s[l] >> shift;  // DISTORTION_PRECISION_ADJUSTMENT on the block's sum (TComRdCost.cpp:1823)
        const unsigned c = sd + vc;
        if (m == centre) z[l] = sd;
        if (c < bc[l]) { bc[l] = c; bs[l] = sd; bi[l] = (unsigned)m; }
      }
    }
#pragma unroll
    for (int l = 0; l < 4; ++l) {
      s_cost[wave][l][lane] = bc[l]; s_satd[wave][l][lane] = bs[l]; s_idx[wave][l][lane] = bi[l];
      if ((centre & 3) == wave) s_zero[l][lane] = z[l];
    }
    __syncthreads();
    if (tid < FHEVC_NODES) {
      int l, ni;
      if (tid == 0) { l = 0; ni = 0; } else if (tid < 5) { l = 1; ni = tid - 1; } else if (tid < 21) { l = 2; ni = tid - 5; } else { l = 3; ni = tid - 21; }
      const int n = 64 >> l, cnt = 1 << l, tn = n >> 3;
      const int bx = ni % cnt, by = ni / cnt;
      const int rep = (by * tn) * 8 + bx * tn;  // a lane of the node (all of them hold the node's sums)
      FhevcMotionNode o;
      if (cx * 64 + bx * n + n > F.width || cy * 64 + by * n + n > F.height) {
        o.satd_zero = o.satd_best = o.cost_best = 0xFFFFFFFFu; o.mvx = 0; o.mvy = 0;
      } else {
        unsigned c = s_cost[0][l][rep], sd = s_satd[0][l][rep], ix = s_idx[0][l][rep];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const unsigned c2 = s_cost[w][l][rep], i2 = s_idx[w][l][rep];
          if (c2 < c || (c2 == c && i2 < ix)) { c = c2; ix = i2; sd = s_satd[w][l][rep]; }
        }
        o.satd_zero = s_zero[l][rep]; o.satd_best = sd; o.cost_best = c;
        o.mvx = (short)((int)(ix % side) - range); o.mvy = (short)((int)(ix / side) - range);
      }
      const long long oc = (long long)((f - 1) * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      out[oc * FHEVC_NODES + tid] = o;
    }
  }
}

}  // namespace

hipError_t fhevc_launch_motion(const FhevcFrames& fr, int range, const FhevcMvCost& mvc, FhevcMotionNode* d_out, int num_cus, bool sad, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * (fr.num_frames - 1);
  if (total <= 0) return hipSuccess;
  const int grid = (int)(total < 4LL * num_cus ? total : 4LL * num_cus);
#define FHEVC_MOTION(T, P) do { if (sad) hipLaunchKernelGGL((fhevc_motion_kernel<T, P, true, FHEVC_MOTION_MAX_RANGE>), dim3(grid), dim3(256), 0, stream, fr, range, mvc, nullptr, d_out); \
                                else hipLaunchKernelGGL((fhevc_motion_kernel<T, P, false, FHEVC_MOTION_MAX_RANGE>), dim3(grid), dim3(256), 0, stream, fr, range, mvc, nullptr, d_out); } while (0)
  if (fr.sample_bytes == 2 && fr.bit_depth <= 10) FHEVC_MOTION(int16_t, true);
  else if (fr.sample_bytes == 2) FHEVC_MOTION(int16_t, false);
  else FHEVC_MOTION(uint8_t, true);
#undef FHEVC_MOTION
  return hipGetLastError();
}

// ranges 9 .. 64 on 16-bit planes above 8 bit (SAD, HM's integer-search distortion): the same kernel laid out for a window of up to 192 x 192 samples
hipError_t fhevc_launch_motion_big(const FhevcFrames& fr, int range, const uint32_t* d_mvtab, FhevcMotionNode* d_out, int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * (fr.num_frames - 1);
  if (total <= 0) return hipSuccess;
  if (fr.sample_bytes != 2 || range > FHEVC_MOTION_WIDE_MAX_RANGE) return hipErrorInvalidValue;
  constexpr int MRB = FHEVC_MOTION_WIDE_MAX_RANGE;
  const size_t lds = (size_t)MotionGeom<MRB>::REF_SAMPLES * sizeof(short);   // 76 816 B: two workgroups per CU
  const int grid = (int)(total < 2LL * num_cus ? total : 2LL * num_cus);
  const FhevcMvCost none = {};
  hipError_t e;
  if (fr.bit_depth <= 10) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_motion_kernel<int16_t, true, true, MRB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fhevc_motion_kernel<int16_t, true, true, MRB>), dim3(grid), dim3(256), lds, stream, fr, range, none, d_mvtab, d_out);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fhevc_motion_kernel<int16_t, false, true, MRB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fhevc_motion_kernel<int16_t, false, true, MRB>), dim3(grid), dim3(256), lds, stream, fr, range, none, d_mvtab, d_out);
  }
  return hipGetLastError();
}
