// k_firstpass.hip -- 35-mode intra first pass for every CU node of a CTU, gfx950 only.
//
// Source-only twin of the first pass of TEncSearch::estIntraPredLumaQT (TEncSearch.cpp:2233-2295):
//   reference samples (TComPattern.cpp:115-539) are taken from the ORIGINAL plane with HM's coding-order
//   availability rule, smoothed as TComPattern.cpp:196-295, the 35 predictors of TComPrediction.cpp:183-473,
//   731-818 are evaluated and compared with TComRdCost::xGetHADs (TComRdCost.cpp:1753-1824).
//   cost = satd + modeBits * sqrt(lambda), strict '<' keeps the lower mode on ties (TEncSearch.cpp:2288,
//   5385-5408).  Bit-exact against oracle/fhevc_oracle.c: fho_first_pass_ctu.
//
// One workgroup (256 threads) per CTU.  Integer VALU/LDS bound: no HBM re-reads (the CTU and the 85 reference
// lines are staged in LDS once), work item = (level, mode, 8x8 tile): 4 levels x 35 modes x 64 tiles = 8960
// items = exactly 35 per thread, the mode is wave-uniform, partial SATDs meet in LDS with integer atomics.
#include "fhevc_internal.h"

namespace {

constexpr int kLineTotal = 257 + 4 * 129 + 16 * 65 + 64 * 33;  // 3925 samples: all 85 lines of 4n+1
// first sample of a level's lines / first node index of a level (levels: 64, 32, 16, 8)
__device__ __forceinline__ int line_off(int level) { return level == 0 ? 0 : (level == 1 ? 257 : (level == 2 ? 773 : 1813)); }
__device__ __forceinline__ int node_off(int level) { return level == 0 ? 0 : (level == 1 ? 1 : (level == 2 ? 5 : 21)); }

__constant__ int c_angTable[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int c_invAngTable[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
__constant__ int c_filterThr[5] = { 10, 7, 1, 0, 10 };  // TComPrediction.cpp:50-58 (4,8,16,32,64)
// raster 16x16 -> z-order (Morton) of the 4x4 units of a CTU (TComRom.cpp:290-323)
__device__ __forceinline__ int zorder_of(int ux, int uy)
{
  int z = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) z |= (((ux >> b) & 1) << (2 * b)) | (((uy >> b) & 1) << (2 * b + 1));
  return z;
}

__device__ __forceinline__ bool unit_available(int ux, int uy, int x0, int y0, int width, int height, int ctus_x)
{
  if (ux < 0 || uy < 0 || ux >= width || uy >= height) return false;
  const int ca = (uy >> 6) * ctus_x + (ux >> 6), cb = (y0 >> 6) * ctus_x + (x0 >> 6);
  if (ca != cb) return ca < cb;
  return zorder_of((ux & 63) >> 2, (uy & 63) >> 2) < zorder_of((x0 & 63) >> 2, (y0 & 63) >> 2);
}

template <typename T>
__device__ __forceinline__ int sample_at(const T* frame, int stride, int x, int y)
{
  return (int)frame[(long long)y * stride + x];
}

__device__ __forceinline__ void wht8x8(int d[64])
{
#pragma unroll
  for (int y = 0; y < 8; ++y)
#pragma unroll
    for (int hs = 1; hs < 8; hs <<= 1)
#pragma unroll
      for (int i = 0; i < 8; i += hs << 1)
#pragma unroll
        for (int j = i; j < i + hs; ++j) {
          const int a = d[y * 8 + j], b = d[y * 8 + j + hs];
          d[y * 8 + j] = a + b; d[y * 8 + j + hs] = a - b;
        }
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int hs = 1; hs < 8; hs <<= 1)
#pragma unroll
      for (int i = 0; i < 8; i += hs << 1)
#pragma unroll
        for (int j = i; j < i + hs; ++j) {
          const int a = d[j * 8 + x], b = d[(j + hs) * 8 + x];
          d[j * 8 + x] = a + b; d[(j + hs) * 8 + x] = a - b;
        }
}

template <typename T>
__global__ __launch_bounds__(256) void fhevc_first_pass_kernel(FhevcFrames F, double sqrt_lambda,
                                                                FhevcNodeCost* __restrict__ out)
{
  __shared__ short s_org[64 * 64];
  __shared__ short s_ref[kLineTotal];    // unfiltered lines, ref[2n] = TL, +i above, -j left
  __shared__ short s_flt[kLineTotal];    // smoothed lines
  __shared__ int s_satd[85 * 35];
  __shared__ int s_dc[85];
  __shared__ unsigned char s_valid[85];

  const int tid = threadIdx.x;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  const int bd = F.bit_depth;
  const int maxval = (1 << bd) - 1;

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = work / per_frame;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const T* frame = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride;
    const int ox = cx * 64, oy = cy * 64;

    // ---- A: stage the CTU, clear the SATD table, fill the 85 unfiltered reference lines ----
    for (int i = tid; i < 64 * 64; i += 256) {
      const int x = ox + (i & 63), y = oy + (i >> 6);
      s_org[i] = (x < F.width && y < F.height) ? (short)sample_at(frame, F.stride, x, y) : (short)0;
    }
    for (int i = tid; i < 85 * 35; i += 256) s_satd[i] = 0;
    if (tid < 85) {
      const int level = tid < 1 ? 0 : (tid < 5 ? 1 : (tid < 21 ? 2 : 3));
      const int n = 64 >> level, cnt = 1 << level, ni = tid - node_off(level);
      const int x0 = ox + (ni % cnt) * n, y0 = oy + (ni / cnt) * n;
      const bool valid = (x0 + n <= F.width) && (y0 + n <= F.height);
      s_valid[tid] = valid ? 1 : 0;
      if (valid) {
        short* ref = s_ref + line_off(level) + ni * (4 * n + 1);
        // walk the line in HM's order: bottom-left unit first ... above-right unit last (4-sample units,
        // TL is one unit of its own); unavailable units copy the previous sample, leading unavailable units
        // copy the first available one (TComPattern.cpp:461-524); nothing available -> 1 << (bd-1) (:343-354)
        const int L = n / 2, totalUnits = 2 * L + 1;
        int firstAvail = -1;
        for (int u = 0; u < totalUnits && firstAvail < 0; ++u) {
          int ux, uy;
          if (u < L) { ux = x0 - 4; uy = y0 + 4 * (L - 1 - u); }
          else if (u == L) { ux = x0 - 4; uy = y0 - 4; }
          else { ux = x0 + 4 * (u - L - 1); uy = y0 - 4; }
          if (unit_available(ux, uy, x0, y0, F.width, F.height, F.ctus_x)) firstAvail = u;
        }
        if (firstAvail < 0) {
          for (int i = 0; i < 4 * n + 1; ++i) ref[i] = (short)(1 << (bd - 1));
        } else {
          int prev = 0;
          for (int pass = 0; pass < 2; ++pass) {
            // pass 0: find the value the leading unavailable units take; pass 1: fill
            if (pass == 0) {
              const int u = firstAvail;
              if (u < L) prev = sample_at(frame, F.stride, x0 - 1, y0 + 4 * (L - 1 - u) + 3);        // bottom-most sample of the unit
              else if (u == L) prev = sample_at(frame, F.stride, x0 - 1, y0 - 1);
              else prev = sample_at(frame, F.stride, x0 + 4 * (u - L - 1), y0 - 1);
              continue;
            }
            for (int u = 0; u < totalUnits; ++u) {
              int ux, uy;
              if (u < L) { ux = x0 - 4; uy = y0 + 4 * (L - 1 - u); }
              else if (u == L) { ux = x0 - 4; uy = y0 - 4; }
              else { ux = x0 + 4 * (u - L - 1); uy = y0 - 4; }
              const bool av = unit_available(ux, uy, x0, y0, F.width, F.height, F.ctus_x);
              if (u < L) {
                // samples of this unit in line order (bottom -> top): ref[4u + i] = left sample (2n-1 - (4u+i))
                for (int i = 0; i < 4; ++i) {
                  const int j = 2 * n - 1 - (4 * u + i);  // left sample index counted downwards from the top
                  if (av) prev = sample_at(frame, F.stride, x0 - 1, y0 + j);
                  ref[4 * u + i] = (short)prev;
                  if (!av) { /* keeps prev */ }
                }
              } else if (u == L) {
                if (av) prev = sample_at(frame, F.stride, x0 - 1, y0 - 1);
                ref[2 * n] = (short)prev;
              } else {
                for (int i = 0; i < 4; ++i) {
                  const int k = 4 * (u - L - 1) + i;
                  if (av) prev = sample_at(frame, F.stride, x0 + k, y0 - 1);
                  ref[2 * n + 1 + k] = (short)prev;
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- B: smoothed lines (one thread per sample) and DC values (one thread per node) ----
    for (int i = tid; i < kLineTotal; i += 256) {
      const int level = i < 257 ? 0 : (i < 773 ? 1 : (i < 1813 ? 2 : 3));
      const int n = 64 >> level, len = 4 * n + 1;
      const int ni = (i - line_off(level)) / len, k = (i - line_off(level)) - ni * len;
      if (!s_valid[node_off(level) + ni]) continue;
      const short* ref = s_ref + line_off(level) + ni * len;
      int v;
      if (k == 0 || k == 4 * n) v = ref[k];
      else {
        bool strong = false;
        if (n >= 32) {  // strong intra smoothing is on in the reference's configs (sps.getUseStrongIntraSmoothing)
          const int thr = 1 << (bd - 5);
          const int bl = ref[0], tl = ref[2 * n], tr = ref[4 * n];
          strong = (abs(bl + tl - 2 * ref[n]) < thr) && (abs(tl + tr - 2 * ref[3 * n]) < thr);
          if (strong) {
            const int lg = (n == 32) ? 6 : 7;
            if (k < 2 * n) v = ((2 * n - k) * bl + k * tl + n) >> lg;
            else if (k == 2 * n) v = tl;
            else v = ((2 * n - (k - 2 * n)) * tl + (k - 2 * n) * tr + n) >> lg;
          }
        }
        if (!strong) v = (ref[k - 1] + 2 * ref[k] + ref[k + 1] + 2) >> 2;
      }
      s_flt[i] = (short)v;
    }
    if (tid < 85 && s_valid[tid]) {
      const int level = tid < 1 ? 0 : (tid < 5 ? 1 : (tid < 21 ? 2 : 3));
      const int n = 64 >> level, ni = tid - node_off(level);
      const short* ref = s_ref + line_off(level) + ni * (4 * n + 1);  // DC never uses the smoothed line
      int sum = 0;
      for (int i = 0; i < n; ++i) sum += ref[2 * n + 1 + i] + ref[2 * n - 1 - i];
      s_dc[tid] = (sum + n) / (2 * n);
    }
    __syncthreads();

    // ---- C: 8960 (level, mode, tile) items ----
    for (int it = 0; it < 35; ++it) {
      const int q = it * 256 + tid;
      const int level = q / 2240, rq = q - level * 2240;
      const int mode = rq >> 6, tile = rq & 63;
      const int n = 64 >> level, lg = 6 - level, cnt = 1 << level;
      const int tx = (tile & 7) * 8, ty = (tile >> 3) * 8;          // tile origin inside the CTU
      const int ni = (ty / n) * cnt + (tx / n);
      const int node = node_off(level) + ni;
      if (!s_valid[node]) continue;
      const int bx = tx & (n - 1), by = ty & (n - 1);                // tile origin inside the node
      const int idx = 4 - level;  // size index of m_aucIntraFilter: 64->4, 32->3, 16->2, 8->1
      bool use_flt = false;
      if (mode != 1) use_flt = min(abs(mode - 10), abs(mode - 26)) > c_filterThr[idx];
      const short* ref = (use_flt ? s_flt : s_ref) + line_off(level) + ni * (4 * n + 1) + 2 * n;  // ref[0] = TL
      int d[64];
      if (mode == 0) {
        const int topRight = ref[n + 1], bottomLeft = ref[-(n + 1)];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          const int left = ref[-(by + y + 1)];
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const int top = ref[bx + x + 1];
            const int hor = (left << lg) + n + (bx + x + 1) * (topRight - left);
            const int ver = (top << lg) + (by + y + 1) * (bottomLeft - top);
            d[y * 8 + x] = (hor + ver) >> (lg + 1);
          }
        }
      } else if (mode == 1) {
        const int dc = s_dc[node];
#pragma unroll
        for (int i = 0; i < 64; ++i) d[i] = dc;
        if (n <= 16) {
          if (by == 0) {
#pragma unroll
            for (int x = 0; x < 8; ++x) d[x] = (ref[bx + x + 1] + 3 * dc + 2) >> 2;
          }
          if (bx == 0) {
#pragma unroll
            for (int y = 0; y < 8; ++y) d[y * 8] = (ref[-(by + y + 1)] + 3 * dc + 2) >> 2;
          }
          if (bx == 0 && by == 0) d[0] = (ref[1] + ref[-1] + 2 * dc + 2) >> 2;
        }
      } else {
        const bool is_ver = mode >= 18;
        const int ang_mode = is_ver ? mode - 26 : -(mode - 10);
        const int abs_mode = abs(ang_mode);
        const int angle = (ang_mode < 0 ? -1 : 1) * c_angTable[abs_mode];
        const int inv_angle = c_invAngTable[abs_mode];
        const int sgn = is_ver ? 1 : -1;  // main(i) = ref[sgn*i], side(i) = ref[-sgn*i]
        // coordinates in the "vertical" frame: (xx, yy) = is_ver ? (x, y) : (y, x)
        const int bxx = is_ver ? bx : by, byy = is_ver ? by : bx;
#pragma unroll
        for (int yy = 0; yy < 8; ++yy) {
          const int delta = (byy + yy + 1) * angle;
          const int di = delta >> 5, df = delta & 31;
#pragma unroll
          for (int xx = 0; xx < 8; ++xx) {
            const int i0 = bxx + xx + di + 1;
            int a, b = 0;
            a = (i0 >= 0) ? ref[sgn * i0] : ref[-sgn * ((128 - i0 * inv_angle) >> 8)];
            int v = a;
            if (df) {
              const int i1 = i0 + 1;
              b = (i1 >= 0) ? ref[sgn * i1] : ref[-sgn * ((128 - i1 * inv_angle) >> 8)];
              v = ((32 - df) * a + df * b + 16) >> 5;
            }
            if (angle == 0 && n <= 16 && (bxx + xx) == 0)
              v = min(maxval, max(0, v + ((ref[-sgn * (byy + yy + 1)] - ref[0]) >> 1)));
            d[is_ver ? (yy * 8 + xx) : (xx * 8 + yy)] = v;
          }
        }
      }
      const short* org = s_org + ty * 64 + tx;
#pragma unroll
      for (int y = 0; y < 8; ++y)
#pragma unroll
        for (int x = 0; x < 8; ++x) d[y * 8 + x] = (int)org[y * 64 + x] - d[y * 8 + x];
      wht8x8(d);
      int s = 0;
#pragma unroll
      for (int i = 0; i < 64; ++i) s += abs(d[i]);
      atomicAdd(&s_satd[node * 35 + mode], (s + 2) >> 2);
    }
    __syncthreads();

    // ---- D: per node, pick the cheapest mode ----
    if (tid < 85) {
      FhevcNodeCost r;
      if (!s_valid[tid]) { r.satd = 0xFFFFFFFFu; r.mode = 255; r.cost = -1.0; }
      else {
        r.cost = 1e300; r.mode = 0; r.satd = 0;
        for (int m = 0; m < 35; ++m) {
          const unsigned sd = (unsigned)s_satd[tid * 35 + m] >> (bd - 8);
          const int bits = (m == 0) ? 2 : ((m == 1 || m == 26) ? 3 : 6);
          const double c = __dadd_rn((double)sd, __dmul_rn((double)bits, sqrt_lambda));
          if (c < r.cost) { r.cost = c; r.mode = (unsigned)m; r.satd = sd; }
        }
      }
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      out[o * 85 + tid] = r;
    }
    __syncthreads();
  }
}

}  // namespace

hipError_t fhevc_launch_first_pass(const FhevcFrames& fr, double sqrt_lambda, FhevcNodeCost* d_out, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  const int grid = (int)(total < 4096 ? total : 4096);
  if (fr.sample_bytes == 2)
    hipLaunchKernelGGL(fhevc_first_pass_kernel<int16_t>, dim3(grid), dim3(256), 0, stream, fr, sqrt_lambda, d_out);
  else
    hipLaunchKernelGGL(fhevc_first_pass_kernel<uint8_t>, dim3(grid), dim3(256), 0, stream, fr, sqrt_lambda, d_out);
  return hipGetLastError();
}
