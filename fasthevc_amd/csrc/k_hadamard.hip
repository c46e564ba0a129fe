// k_hadamard.hip -- integer Hadamard kernels, gfx950 only.
//
//  * fhevc_src_hadamard_kernel: bit-exact twin of TEncCu::updateCtuDataISlice / xCalcHADs8x8_ISlice
//    (TEncCu.cpp:1230-1343): per CTU, sum over whole 8x8 blocks of (sum|WHT(src)| - |DC| + 2) >> 2.
//    One pass over the planar-Y frame: HBM-bound (each sample read once, 4 B written per CTU).
//    One wave handles 8 picture rows x 64 columns: lane = 8*row + block, each lane loads its 8 samples with
//    one 16-byte (int16) or 8-byte (uint8) load, so 8 consecutive lanes cover one contiguous 128-byte run.
//    Horizontal butterflies run in-lane, vertical ones across lanes (xor 8/16/32).
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

template <typename T>
__global__ __launch_bounds__(256) void fhevc_src_hadamard_kernel(FhevcFrames F, int32_t* __restrict__ out)
{
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = lane & 7, row = lane >> 3;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = work / per_frame;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
    const T* base = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride +
                    (long long)(cy * 64) * F.stride + cx * 64;
    int acc = 0;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int by = wave + 4 * it;  // block row 0..7
      const bool ok = (bx * 8 + 8 <= vw) && (by * 8 + 8 <= vh);  // only WHOLE 8x8 blocks count (TEncCu.cpp:1334-1336)
      int v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = 0;
      if (ok) {
        const T* p = base + (long long)(by * 8 + row) * F.stride + bx * 8;
        if (sizeof(T) == 2) {
          if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
            const uint4 q = *reinterpret_cast<const uint4*>(p);
            v[0] = (short)(q.x & 0xFFFF); v[1] = (short)(q.x >> 16); v[2] = (short)(q.y & 0xFFFF); v[3] = (short)(q.y >> 16);
            v[4] = (short)(q.z & 0xFFFF); v[5] = (short)(q.z >> 16); v[6] = (short)(q.w & 0xFFFF); v[7] = (short)(q.w >> 16);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (int)p[j];
          }
        } else {
          if ((reinterpret_cast<uintptr_t>(p) & 7) == 0) {
            const uint2 q = *reinterpret_cast<const uint2*>(p);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = (q.x >> (8 * j)) & 0xFF; v[4 + j] = (q.y >> (8 * j)) & 0xFF; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (int)p[j];
          }
        }
      }
      wht8_inlane(v);
      // vertical butterflies: partner rows are 8, 16 and 32 lanes away; the lower lane keeps a+b, the upper a-b
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) {
        const bool upper = (lane & m) != 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o = __shfl_xor(v[j], m);
          v[j] = upper ? (o - v[j]) : (v[j] + o);
        }
      }
      int s = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += abs(v[j]);
      if (row == 0) s -= abs(v[0]);  // DC coefficient lives in (row 0, column 0)
      // per-block sum over its 8 rows, then (s+2)>>2 per block, then over the 8 blocks of the wave
      s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
      s = (s + 2) >> 2;
      if (!ok) s = 0;
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
      acc += s;
    }
    if (lane == 0) wsum[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      out[o] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    __syncthreads();
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
  const int grid = (int)(total < 8192 ? total : 8192);
  if (fr.sample_bytes == 2)
    hipLaunchKernelGGL(fhevc_src_hadamard_kernel<int16_t>, dim3(grid), dim3(256), 0, stream, fr, d_out);
  else
    hipLaunchKernelGGL(fhevc_src_hadamard_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream, fr, d_out);
  return hipGetLastError();
}

hipError_t fhevc_launch_satd(const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                             int w, int h, int bit_depth, uint32_t* d_out, hipStream_t stream)
{
  hipLaunchKernelGGL(fhevc_satd_kernel, dim3(1), dim3(64), 0, stream, d_org, org_stride, d_cur, cur_stride, w, h,
                     bit_depth, d_out);
  return hipGetLastError();
}
