// k_hadamard.hip -- integer Hadamard kernels, gfx950 only.
//
//  * fhevc_src_hadamard_kernel: bit-exact twin of TEncCu::updateCtuDataISlice / xCalcHADs8x8_ISlice
//    (TEncCu.cpp:1230-1343): per CTU, sum over whole 8x8 blocks of (sum|WHT(src)| - |DC| + 2) >> 2.
//    One pass over the planar-Y frame: HBM-bound (each sample read once, 4 B written per CTU).
//    One lane owns one 8x8 block (8 loads of 16 B, or 8 B for uint8), one wave one CTU: both butterfly passes run
//    in registers, 8 consecutive lanes read one contiguous 128-byte run of a picture row.  Up to 10 bits the butterflies
//    run on packed 16-bit VALU (two samples per lane-op, wrapping adds): every coefficient except DC is bounded by
//    32 * 1023 and DC is left out of the sum anyway, so arithmetic modulo 2^16 is exact; the stage inside a packed pair
//    is folded into the absolute sum with |a+b| + |a-b| = 2 max(|a|,|b|).  With 32-bit butterflies the kernel is VALU-bound.
//  * fhevc_satd_kernel: twin of TComRdCost::calcHAD / xGetHADs (TComRdCost.cpp:297-334, 1527-1824) for one
//    block pair; parity entry point, not a throughput path.
#include "fhevc_internal.h"

namespace {

__device__ __forceinline__ void wht8_inlane(int v[8])
{
#pragma unroll
  for (int hstep = 1; hstep < 8; hstep <<= 1)
#pragma unroll
    for (int i = 0; i < 8; i += hstep << 1)
#pragma unroll
      for (int j = i; j < i + hstep; ++j) {
        const int a = v[j], b = v[j + hstep];
        v[j] = a + b;
        v[j + hstep] = a - b;
      }
}

// One lane = one 8x8 block, one wave = one CTU (lane = 8*block_row + block_col), one workgroup = 4 consecutive CTUs.
// For load j (row j of every lane's block) 8 consecutive lanes read one contiguous 128-byte run (int16) of a picture
// row, so every fetched line is used whole; both transform passes are in-lane: no LDS, no cross-lane traffic until
// the final 64-lane sum.
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
// sum of |WHT8x8| without DC, of a block held as 8 rows x 4 packed pairs (low half = even column), samples < 2^10
__device__ __forceinline__ int src_had_packed(unsigned (&d)[32])
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
  // register 0 holds (a, b) with DC = a + b: only |a - b| counts; both are sums of 32 samples, no wrap
  const int a0 = (int)(short)(d[0] & 0xFFFF), b0 = (int)(short)(d[0] >> 16);
  unsigned acc = 0;
#pragma unroll
  for (int i = 1; i < 32; ++i) {
    const unsigned a = pk_abs(d[i]);
    acc += max(a & 0xFFFFu, a >> 16);
  }
  return (int)(2 * acc) + abs(a0 - b0);
}

template <typename T, bool PACKED>
__global__ __launch_bounds__(256) void fhevc_src_hadamard_kernel(FhevcFrames F, int32_t* __restrict__ out)
{
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = lane & 7, by = lane >> 3;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  // XCD-aware order: blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 share an L2), so give every XCD a
  // contiguous run of CTUs: horizontally adjacent CTUs share 128-byte lines when the plane is not 128-byte aligned
  // (HM's 80-sample margin), and then the shared line is fetched into one L2 instead of two.  Speed only.
  const int vblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // grid is a multiple of 8
  for (int work = vblock * 4 + wave; work < total; work += gridDim.x * 4) {
    const int f = work / per_frame;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
    const bool ok = (bx * 8 + 8 <= vw) && (by * 8 + 8 <= vh);  // only WHOLE 8x8 blocks count (TEncCu.cpp:1334-1336)
    int s = 0;
    if (PACKED) {
      unsigned d[32];
      if (ok) {
        const T* p = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride +
                     (long long)(cy * 64 + by * 8) * F.stride + cx * 64 + bx * 8;
        const bool al = (reinterpret_cast<uintptr_t>(p) & (8 * sizeof(T) - 1)) == 0 && ((F.stride * sizeof(T)) & (8 * sizeof(T) - 1)) == 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const T* rp = p + (long long)j * F.stride;
          if (al && sizeof(T) == 2) {
            const uint4 q = *reinterpret_cast<const uint4*>(rp);
            d[4 * j] = q.x; d[4 * j + 1] = q.y; d[4 * j + 2] = q.z; d[4 * j + 3] = q.w;
          } else if (al) {
            const uint2 q = *reinterpret_cast<const uint2*>(rp);  // bytes -> 16-bit pairs (0x0C selects a zero byte)
            d[4 * j] = __builtin_amdgcn_perm(0u, q.x, 0x0C010C00u); d[4 * j + 1] = __builtin_amdgcn_perm(0u, q.x, 0x0C030C02u);
            d[4 * j + 2] = __builtin_amdgcn_perm(0u, q.y, 0x0C010C00u); d[4 * j + 3] = __builtin_amdgcn_perm(0u, q.y, 0x0C030C02u);
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) d[4 * j + k] = ((unsigned)rp[2 * k] & 0xFFFFu) | ((unsigned)rp[2 * k + 1] << 16);
          }
        }
        s = src_had_packed(d);
      }
    } else {
    int v[64];
    if (ok) {
      const T* p = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride +
                   (long long)(cy * 64 + by * 8) * F.stride + cx * 64 + bx * 8;
      const bool al = (reinterpret_cast<uintptr_t>(p) & (8 * sizeof(T) - 1)) == 0 && ((F.stride * sizeof(T)) & (8 * sizeof(T) - 1)) == 0;
      if (al) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (sizeof(T) == 2) {
            const uint4 q = *reinterpret_cast<const uint4*>(p + (long long)j * F.stride);
            v[8 * j + 0] = (short)(q.x & 0xFFFF); v[8 * j + 1] = (short)(q.x >> 16);
            v[8 * j + 2] = (short)(q.y & 0xFFFF); v[8 * j + 3] = (short)(q.y >> 16);
            v[8 * j + 4] = (short)(q.z & 0xFFFF); v[8 * j + 5] = (short)(q.z >> 16);
            v[8 * j + 6] = (short)(q.w & 0xFFFF); v[8 * j + 7] = (short)(q.w >> 16);
          } else {
            const uint2 q = *reinterpret_cast<const uint2*>(p + (long long)j * F.stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[8 * j + k] = (q.x >> (8 * k)) & 0xFF; v[8 * j + 4 + k] = (q.y >> (8 * k)) & 0xFF; }
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) v[8 * j + k] = (int)p[(long long)j * F.stride + k];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 64; ++i) v[i] = 0;
    }
    // rows then columns: un-normalised Walsh-Hadamard butterflies, all in registers
#pragma unroll
    for (int j = 0; j < 8; ++j) wht8_inlane(v + 8 * j);
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int hstep = 1; hstep < 8; hstep <<= 1)
#pragma unroll
        for (int i = 0; i < 8; i += hstep << 1)
#pragma unroll
          for (int j = i; j < i + hstep; ++j) {
            const int a = v[8 * j + k], b = v[8 * (j + hstep) + k];
            v[8 * j + k] = a + b;
            v[8 * (j + hstep) + k] = a - b;
          }
#pragma unroll
    for (int i = 1; i < 64; ++i) s += abs(v[i]);  // v[0] is the DC coefficient: left out (TEncCu.cpp:1319)
    }
    s = ok ? ((s + 2) >> 2) : 0;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) s += __shfl_xor(s, m);
    if (lane == 0) {
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      out[o] = s;
    }
  }
}

// ---- generic SATD of one block pair (w, h <= 64) --------------------------------------------------------
__device__ int had_tile(const int16_t* org, int so, const int16_t* cur, int sc, int n)
{
  int d[64];
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) d[y * n + x] = (int)org[y * so + x] - (int)cur[y * sc + x];
  for (int pass = 0; pass < 2; ++pass) {
    const int inner = pass == 0 ? 1 : n, outer = pass == 0 ? n : 1;
    for (int l = 0; l < n; ++l)
      for (int hs = 1; hs < n; hs <<= 1)
        for (int i = 0; i < n; i += hs << 1)
          for (int j = i; j < i + hs; ++j) {
            const int ia = l * outer + j * inner, ib = l * outer + (j + hs) * inner;
            const int a = d[ia], b = d[ib];
            d[ia] = a + b;
            d[ib] = a - b;
          }
  }
  int s = 0;
  for (int i = 0; i < n * n; ++i) s += abs(d[i]);
  return s;
}

__global__ __launch_bounds__(64) void fhevc_satd_kernel(const int16_t* org, int so, const int16_t* cur, int sc,
                                                         int w, int h, int bit_depth, uint32_t* out)
{
  const int t = ((w % 8) == 0 && (h % 8) == 0) ? 8 : (((w % 4) == 0 && (h % 4) == 0) ? 4 : 2);
  const int tx = w / t, ty = h / t;
  unsigned sum = 0;
  for (int i = threadIdx.x; i < tx * ty; i += 64) {
    const int x = (i % tx) * t, y = (i / tx) * t;
    const int s = had_tile(org + y * so + x, so, cur + y * sc + x, sc, t);
    sum += (t == 8) ? (unsigned)((s + 2) >> 2) : (t == 4) ? (unsigned)((s + 1) >> 1) : (unsigned)s;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
  if (threadIdx.x == 0) *out = sum >> (bit_depth - 8);
}

}  // namespace

hipError_t fhevc_launch_src_hadamard(const FhevcFrames& fr, int32_t* d_out, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  long long groups = (total + 3) / 4;
  const int grid = (int)(((groups < 4096 ? groups : 4096) + 7) & ~7LL);  // multiple of 8: see the XCD remap in the kernel
  if (fr.sample_bytes == 2 && fr.bit_depth <= 10)
    hipLaunchKernelGGL((fhevc_src_hadamard_kernel<int16_t, true>), dim3(grid), dim3(256), 0, stream, fr, d_out);
  else if (fr.sample_bytes == 2)
    hipLaunchKernelGGL((fhevc_src_hadamard_kernel<int16_t, false>), dim3(grid), dim3(256), 0, stream, fr, d_out);
  else
    hipLaunchKernelGGL((fhevc_src_hadamard_kernel<uint8_t, true>), dim3(grid), dim3(256), 0, stream, fr, d_out);
  return hipGetLastError();
}

hipError_t fhevc_launch_satd(const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                             int w, int h, int bit_depth, uint32_t* d_out, hipStream_t stream)
{
  hipLaunchKernelGGL(fhevc_satd_kernel, dim3(1), dim3(64), 0, stream, d_org, org_stride, d_cur, cur_stride, w, h,
                     bit_depth, d_out);
  return hipGetLastError();
}
