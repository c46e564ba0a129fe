// k_firstpass.hip -- 35-mode intra first pass for every CU node of a CTU, gfx950 only.
//
// Source-only twin of the first pass of TEncSearch::estIntraPredLumaQT (TEncSearch.cpp:2233-2295):
//   reference samples (TComPattern.cpp:115-539) are taken from the ORIGINAL plane with HM's coding-order
//   availability rule, smoothed as TComPattern.cpp:196-295, the 35 predictors of TComPrediction.cpp:183-473,
//   731-818 are evaluated and compared with TComRdCost::xGetHADs (TComRdCost.cpp:1753-1824).
//   cost = satd + modeBits * sqrt(lambda), strict '<' keeps the lower mode on ties (TEncSearch.cpp:2288,
//   5385-5408).  Bit-exact against oracle/fhevc_oracle.c: fho_first_pass_ctu.
//
// One workgroup (256 threads) per CTU, persistent over CTUs.  Integer VALU/LDS bound, no HBM re-reads:
//   * the CTU, the row above it and the column left of it are staged in LDS once; the 85 reference lines are
//     built from LDS (one thread per 4-sample unit for availability + copy, one thread per node for HM's
//     substitution walk, one thread per sample for the smoothing);
//   * lane = one 8x8 tile of the CTU for the whole CTU: the tile's original samples stay in registers as packed
//     16-bit pairs; their transpose lives in LDS for the horizontal modes (sum|WHT(X^T)| = sum|WHT(X)|, so those are
//     evaluated in the transposed frame without transposing the prediction back);
//   * wave = one (mode, level) pair at a time, 140 pairs = 35 per wave: everything that depends on the mode is
//     wave-uniform;
//   * angular prediction, residual and Hadamard run on packed 16-bit VALU (two samples per lane-op) when the
//     bit depth is <= 10 (the same boundary at which HM uses its 16-bit SIMD Hadamard, TComRdCost.cpp:1783):
//     five butterfly stages between registers, the sixth (inside a packed pair) folded into the absolute sum with
//     |a+b| + |a-b| = 2 max(|a|,|b|); 12-bit content takes the 32-bit path;
//   * negative-angle modes read a per-wave projected main reference (HM's refMain extension) built once per
//     (mode, level); tile SATDs meet per node through lane shuffles, no atomics.
#include "fhevc_internal.h"

namespace {

typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(2))) short i16x2;

constexpr int kLineTotal = 257 + 4 * 129 + 16 * 65 + 64 * 33;  // 3925 samples: all 85 lines of 4n+1
constexpr int kLinePad = 8;                                    // the unconditional second tap may touch one sample past a line
constexpr int kUnits = 65 + 4 * 33 + 16 * 17 + 64 * 9;         // 1045 four-sample units (n+1 per node: TL is a unit)
constexpr int kMainPerWave = 1088;                             // max over levels of nodes * (2n+1) projected samples
// first sample of a level's lines / first node / first unit of a level (levels: 64, 32, 16, 8)
__device__ __forceinline__ int line_off(int level) { return level == 0 ? 0 : (level == 1 ? 257 : (level == 2 ? 773 : 1813)); }
__device__ __forceinline__ int node_off(int level) { return level == 0 ? 0 : (level == 1 ? 1 : (level == 2 ? 5 : 21)); }
__device__ __forceinline__ int unit_off(int level) { return level == 0 ? 0 : (level == 1 ? 65 : (level == 2 ? 197 : 469)); }

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

// position (ux, uy) of 4-sample unit u of a node at (x0, y0) of size n, in HM's walk order: bottom-left unit first,
// then up the left column, the top-left corner (one sample), then the row above left to right
__device__ __forceinline__ void unit_pos(int u, int n, int x0, int y0, int& ux, int& uy)
{
  const int L = n >> 1;
  if (u < L) { ux = x0 - 4; uy = y0 + 4 * (L - 1 - u); }
  else if (u == L) { ux = x0 - 4; uy = y0 - 4; }
  else { ux = x0 + 4 * (u - L - 1); uy = y0 - 4; }
}

// a sample that an AVAILABLE unit covers: inside this CTU, in the row above it, or in the column left of it
__device__ __forceinline__ short staged(const short* s_org, const short* s_above, const short* s_left, int px, int py, int ox, int oy)
{
  if (py >= oy && px >= ox) return s_org[(py - oy) * 64 + (px - ox)];
  if (py < oy) return s_above[px - ox + 1];
  return s_left[py - oy];
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

__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) - __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, (i16x2)(__builtin_bit_cast(i16x2, a) + __builtin_bit_cast(i16x2, b)));
}
__device__ __forceinline__ unsigned pk_abs(unsigned a)
{
  const i16x2 v = __builtin_bit_cast(i16x2, a);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, (i16x2)(-v)));
}

// (sum|WHT8x8(o - p)| + 2) >> 2 of one tile, rows packed as 4 dwords of two 16-bit samples.  |o - p| <= 1023, so the
// five register-to-register stages stay below 2^15; the stage inside a packed pair is |a+b| + |a-b| = 2 max(|a|,|b|).
__device__ __forceinline__ int satd8x8_packed(const unsigned (&o)[32], const unsigned (&p)[32])
{
  unsigned d[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) d[i] = pk_sub(o[i], p[i]);
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
  for (int i = 0; i < 32; ++i) {
    const unsigned a = pk_abs(d[i]);
    acc += max(a & 0xFFFFu, a >> 16);
  }
  return (int)((acc + 1) >> 1);  // (2 * acc + 2) >> 2
}

// general path (12-bit content): 32-bit butterflies
__device__ __forceinline__ int satd8x8_wide(const unsigned (&o)[32], const unsigned (&p)[32])
{
  int d[64];
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    d[2 * i] = (int)(short)(o[i] & 0xFFFF) - (int)(p[i] & 0xFFFF);
    d[2 * i + 1] = (int)(short)(o[i] >> 16) - (int)(p[i] >> 16);
  }
  wht8x8(d);
  int s = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += abs(d[i]);
  return (s + 2) >> 2;
}

// eight rows of an angular predictor in the "vertical" frame: row yy reads nine consecutive main-reference samples
// R[i0 .. i0+8] (STEP = +1 ascending in memory, -1 descending) and blends neighbours with weights (32 - df, df)
template <int STEP>
__device__ __forceinline__ void angular_rows(const short* mainp, int bxx, int byy, int angle, unsigned (&p)[32])
{
#pragma unroll
  for (int yy = 0; yy < 8; ++yy) {
    const int delta = (byy + yy + 1) * angle;
    const int di = delta >> 5, df = delta & 31;
    const short* rp = mainp + STEP * (bxx + di + 1);
    unsigned short r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = (unsigned short)rp[STEP * k];
    const u16x2 w0 = { (unsigned short)(32 - df), (unsigned short)(32 - df) }, w1 = { (unsigned short)df, (unsigned short)df };
    const u16x2 c16 = { 16, 16 };
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u16x2 a = { r[2 * j], r[2 * j + 1] }, b = { r[2 * j + 1], r[2 * j + 2] };
      const u16x2 t = a * w0 + c16;  // two v_pk_mad_u16 and a shift; <= 32*1023 + 16: no 16-bit overflow up to 10 bits
      const u16x2 v = (u16x2)((b * w1 + t) >> (u16x2){ 5, 5 });
      p[yy * 4 + j] = __builtin_bit_cast(unsigned, v);
    }
  }
}
// the same with 32-bit arithmetic (12-bit content)
template <int STEP>
__device__ __forceinline__ void angular_rows_wide(const short* mainp, int bxx, int byy, int angle, unsigned (&p)[32])
{
#pragma unroll
  for (int yy = 0; yy < 8; ++yy) {
    const int delta = (byy + yy + 1) * angle;
    const int di = delta >> 5, df = delta & 31;
    const short* rp = mainp + STEP * (bxx + di + 1);
    int r[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = (int)rp[STEP * k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v0 = ((32 - df) * r[2 * j] + df * r[2 * j + 1] + 16) >> 5;
      const int v1 = ((32 - df) * r[2 * j + 1] + df * r[2 * j + 2] + 16) >> 5;
      p[yy * 4 + j] = (unsigned)v0 | ((unsigned)v1 << 16);
    }
  }
}

template <bool PACKED>
__device__ __forceinline__ int satd8x8(const unsigned (&o)[32], const unsigned (&p)[32])
{
  return PACKED ? satd8x8_packed(o, p) : satd8x8_wide(o, p);
}

template <typename T, bool PACKED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PACKED ? 3 : 1, PACKED ? 3 : 2))) void fhevc_first_pass_kernel(FhevcFrames F, double sqrt_lambda,
                                                                FhevcNodeCost* __restrict__ out, FhevcNodeCost* __restrict__ out_all)
{
  // s_org holds the CTU while the lines are built; afterwards the same bytes are the four per-wave projected references
  __shared__ __attribute__((aligned(16))) short s_org[4 * kMainPerWave];
  __shared__ short s_refbuf[kLinePad + kLineTotal + kLinePad + 1];  // unfiltered lines, line[2n] = TL, +i above, -j left
  __shared__ short s_fltbuf[kLinePad + kLineTotal + kLinePad + 1];  // smoothed lines
  __shared__ __attribute__((aligned(16))) short s_orgT[64 * 64];     // the CTU transposed: row x, column y
  __shared__ short s_above[132], s_left[64];
  __shared__ int s_satd[85 * 35];
  __shared__ int s_dc[85];
  __shared__ unsigned char s_av[kUnits + 3];
  __shared__ unsigned char s_valid[85];
  static_assert(4 * kMainPerWave >= 64 * 64, "the CTU must fit the shared region");
  short* const s_ref = s_refbuf + kLinePad;
  short* const s_flt = s_fltbuf + kLinePad;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  const int bd = F.bit_depth;
  const int maxval = (1 << bd) - 1;
  const int tx = (lane & 7) * 8, ty = (lane >> 3) * 8;  // this lane's tile inside the CTU

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = work / per_frame;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const T* frame = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride;
    const int ox = cx * 64, oy = cy * 64;

    // ---- A1: stage the CTU (16 samples per thread), the row above (129 samples) and the left column (64) ----
    {
      const int row = tid >> 2, seg = (tid & 3) * 16;
      const int y = oy + row;
      const T* src = frame + (long long)y * F.stride + ox + seg;
      short* dst = &s_org[row * 64 + seg];
      const bool whole = y < F.height && ox + seg + 16 <= F.width;
      if (whole && sizeof(T) == 2 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        reinterpret_cast<uint4*>(dst)[0] = reinterpret_cast<const uint4*>(src)[0];
        reinterpret_cast<uint4*>(dst)[1] = reinterpret_cast<const uint4*>(src)[1];
      } else if (whole && sizeof(T) == 1 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4*>(src);
        const unsigned w[4] = { q.x, q.y, q.z, q.w };
        unsigned o8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = ((w[k >> 1] >> (16 * (k & 1))) & 0xFF) | (((w[k >> 1] >> (16 * (k & 1) + 8)) & 0xFF) << 16);
        reinterpret_cast<uint4*>(dst)[0] = make_uint4(o8[0], o8[1], o8[2], o8[3]);
        reinterpret_cast<uint4*>(dst)[1] = make_uint4(o8[4], o8[5], o8[6], o8[7]);
      } else {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) dst[k] = (y < F.height && ox + seg + k < F.width) ? (short)src[k] : (short)0;
      }
      if (tid < 129) {
        const int x = ox - 1 + tid;
        s_above[tid] = (oy > 0 && x >= 0 && x < F.width) ? (short)frame[(long long)(oy - 1) * F.stride + x] : (short)0;
      } else if (tid >= 192) {
        const int yl = oy + tid - 192;
        s_left[tid - 192] = (ox > 0 && yl < F.height) ? (short)frame[(long long)yl * F.stride + ox - 1] : (short)0;
      }
      if (tid < 85) {
        const int level = tid < 1 ? 0 : (tid < 5 ? 1 : (tid < 21 ? 2 : 3));
        const int n = 64 >> level, cnt = 1 << level, ni = tid - node_off(level);
        s_valid[tid] = ((ox + (ni % cnt) * n + n <= F.width) && (oy + (ni / cnt) * n + n <= F.height)) ? 1 : 0;
      }
    }
    __syncthreads();

    // ---- A2: one thread per 4-sample unit: availability (coding order) and, if available, its samples ----
    for (int ug = tid; ug < kUnits; ug += 256) {
      const int level = ug < 65 ? 0 : (ug < 197 ? 1 : (ug < 469 ? 2 : 3));
      const int n = 64 >> level, cnt = 1 << level, upn = n + 1;
      const int ni = (ug - unit_off(level)) / upn, u = (ug - unit_off(level)) - ni * upn;
      const int x0 = ox + (ni % cnt) * n, y0 = oy + (ni / cnt) * n;
      int ux, uy;
      unit_pos(u, n, x0, y0, ux, uy);
      const bool av = s_valid[node_off(level) + ni] && unit_available(ux, uy, x0, y0, F.width, F.height, F.ctus_x);
      s_av[ug] = av ? 1 : 0;
      if (av) {
        short* line = s_ref + line_off(level) + ni * (4 * n + 1);
        const int L = n >> 1;
        if (u < L) {  // line[4u + i] = left sample 2n-1 - (4u+i), counted downwards from the top
#pragma unroll
          for (int i = 0; i < 4; ++i) line[4 * u + i] = staged(s_org, s_above, s_left, x0 - 1, y0 + 2 * n - 1 - (4 * u + i), ox, oy);
        } else if (u == L) {
          line[2 * n] = staged(s_org, s_above, s_left, x0 - 1, y0 - 1, ox, oy);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) line[4 * u - 3 + i] = staged(s_org, s_above, s_left, x0 + 4 * (u - L - 1) + i, y0 - 1, ox, oy);
        }
      }
    }
    __syncthreads();

    // ---- A3: HM's substitution walk (TComPattern.cpp:461-524), one thread per node: unavailable units copy the last
    // sample before them, leading unavailable units the first available sample; nothing available -> 1 << (bd-1) ----
    if (tid < 85 && s_valid[tid]) {
      const int level = tid < 1 ? 0 : (tid < 5 ? 1 : (tid < 21 ? 2 : 3));
      const int n = 64 >> level, ni = tid - node_off(level), upn = n + 1, L = n >> 1;
      short* line = s_ref + line_off(level) + ni * (4 * n + 1);
      const unsigned char* av = s_av + unit_off(level) + ni * upn;
      int first = -1;
      for (int u = 0; u < upn && first < 0; ++u) if (av[u]) first = u;
      if (first < 0) {
        for (int i = 0; i < 4 * n + 1; ++i) line[i] = (short)(1 << (bd - 1));
      } else {
        short prev = line[first <= L ? 4 * first : 4 * first - 3];
        for (int u = 0; u < upn; ++u) {
          const int s = u <= L ? 4 * u : 4 * u - 3, c = (u == L) ? 1 : 4;
          if (av[u]) prev = line[s + c - 1];
          else for (int i = 0; i < c; ++i) line[s + i] = prev;
        }
      }
    }
    __syncthreads();

    // ---- B: smoothed lines (one thread per sample) and DC values (one thread per node) ----
#pragma unroll
    for (int level = 0; level < 4; ++level) {  // unrolled: sizes are compile-time constants, no runtime division
      const int n = 64 >> level, len = 4 * n + 1, cnt = (1 << (2 * level)) * len;
      for (int i = tid; i < cnt; i += 256) {
        const int ni = i / len, k = i - ni * len;
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
        s_flt[line_off(level) + i] = (short)v;
      }
    }
    if (tid < 85 && s_valid[tid]) {
      const int level = tid < 1 ? 0 : (tid < 5 ? 1 : (tid < 21 ? 2 : 3));
      const int n = 64 >> level, ni = tid - node_off(level);
      const short* ref = s_ref + line_off(level) + ni * (4 * n + 1);  // DC never uses the smoothed line
      int sum = 0;
      for (int i = 0; i < n; ++i) sum += ref[2 * n + 1 + i] + ref[2 * n - 1 - i];
      s_dc[tid] = (sum + n) / (2 * n);
    }
    // this lane's tile stays in registers as packed pairs (low half = even column); its transpose goes to LDS
    unsigned O[32];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const uint4 q = *reinterpret_cast<const uint4*>(&s_org[(ty + y) * 64 + tx]);
      O[4 * y] = q.x; O[4 * y + 1] = q.y; O[4 * y + 2] = q.z; O[4 * y + 3] = q.w;
    }
#pragma unroll
    for (int xx = 0; xx < 8; ++xx) {
      unsigned t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        t[j] = __builtin_amdgcn_perm(O[4 * (2 * j + 1) + (xx >> 1)], O[4 * (2 * j) + (xx >> 1)], (xx & 1) ? 0x07060302u : 0x05040100u);
      *reinterpret_cast<uint4*>(&s_orgT[(tx + xx) * 64 + ty]) = make_uint4(t[0], t[1], t[2], t[3]);
    }
    __syncthreads();  // from here on s_org is the per-wave projected-reference scratch

    // ---- C: 140 (mode, level) pairs, 35 per wave; lane = tile ----
    short* const wmain = s_org + wave * kMainPerWave;
#pragma unroll 1
    for (int it = 0; it < 35; ++it) {
      const int pr = __builtin_amdgcn_readfirstlane(wave + 4 * it);
      const int mode = pr % 35, level = pr / 35;
      const int n = 64 >> level, lg = 6 - level;
      const int ni = ((ty >> lg) << level) + (tx >> lg);
      const int node = node_off(level) + ni;
      const int bx = tx & (n - 1), by = ty & (n - 1);  // tile origin inside the node
      const int idx = 4 - level;                       // size index of m_aucIntraFilter: 64->4, 32->3, 16->2, 8->1
      const bool use_flt = (mode != 1) && (min(abs(mode - 10), abs(mode - 26)) > c_filterThr[idx]);
      const short* lines = (use_flt ? s_flt : s_ref) + line_off(level);
      const short* ref = lines + ni * (4 * n + 1) + 2 * n;  // ref[0] = TL
      // every branch ends in its own copy of the Hadamard: merging the branches first costs ~64 register moves per item
      unsigned P[32];
      int s;
      if (mode == 0) {
        const int topRight = ref[n + 1], bottomLeft = ref[-(n + 1)];
#pragma unroll
        for (int y = 0; y < 8; ++y) {
          const int left = ref[-(by + y + 1)];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            unsigned v2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int x = 2 * j + e;
              const int top = ref[bx + x + 1];
              const int hor = (left << lg) + n + (bx + x + 1) * (topRight - left);
              const int ver = (top << lg) + (by + y + 1) * (bottomLeft - top);
              v2[e] = (unsigned)((hor + ver) >> (lg + 1));
            }
            P[4 * y + j] = v2[0] | (v2[1] << 16);
          }
        }
        s = satd8x8<PACKED>(O, P);
      } else if (mode == 1) {
        const unsigned dc = (unsigned)s_dc[node];
#pragma unroll
        for (int i = 0; i < 32; ++i) P[i] = dc | (dc << 16);
        if (n <= 16) {
          if (by == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned a = (unsigned)((ref[bx + 2 * j + 1] + 3 * (int)dc + 2) >> 2), b = (unsigned)((ref[bx + 2 * j + 2] + 3 * (int)dc + 2) >> 2);
              P[j] = a | (b << 16);
            }
          }
          if (bx == 0) {
#pragma unroll
            for (int y = 0; y < 8; ++y) P[4 * y] = (P[4 * y] & 0xFFFF0000u) | (unsigned)((ref[-(by + y + 1)] + 3 * (int)dc + 2) >> 2);
          }
          if (bx == 0 && by == 0) P[0] = (P[0] & 0xFFFF0000u) | (unsigned)((ref[1] + ref[-1] + 2 * (int)dc + 2) >> 2);
        }
        s = satd8x8<PACKED>(O, P);
      } else {
        const bool is_ver = mode >= 18;
        const int ang_mode = is_ver ? mode - 26 : -(mode - 10);
        const int abs_mode = abs(ang_mode);
        const int angle = (ang_mode < 0 ? -1 : 1) * c_angTable[abs_mode];
        const int inv_angle = c_invAngTable[abs_mode];
        const int sgn = is_ver ? 1 : -1;  // main(i) = ref[sgn*i], side(i) = ref[-sgn*i]
        // (xx, yy) = is_ver ? (x, y) : (y, x): horizontal modes are evaluated against the transposed tile
        const int bxx = is_ver ? bx : by, byy = is_ver ? by : bx;
        if (angle < 0) {
          // HM's refMain with its projected extension (TComPrediction.cpp:278-300), once per node of this level:
          // entries k = (n*angle)>>5 .. n of node v at wmain[v*(2n+1) + n + k]
          const int kmin = (n * angle) >> 5, span = 2 * n + 1;
          const int v = lane & ((1 << (2 * level)) - 1), g = lane >> (2 * level), groups = 64 >> (2 * level);
          const short* l = lines + v * (4 * n + 1) + 2 * n;
          __builtin_amdgcn_wave_barrier();
          for (int k = kmin + g; k <= n; k += groups)  // lane = (node v, every groups-th entry): no division
            wmain[v * span + n + k] = (k >= 0) ? l[sgn * k] : l[-sgn * ((128 - k * inv_angle) >> 8)];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const short* mp = wmain + ni * span + n;
          if (PACKED) angular_rows<1>(mp, bxx, byy, angle, P); else angular_rows_wide<1>(mp, bxx, byy, angle, P);
        } else if (is_ver) {
          if (PACKED) angular_rows<1>(ref, bxx, byy, angle, P); else angular_rows_wide<1>(ref, bxx, byy, angle, P);
        } else {
          if (PACKED) angular_rows<-1>(ref, bxx, byy, angle, P); else angular_rows_wide<-1>(ref, bxx, byy, angle, P);
        }
        if (angle == 0 && n <= 16 && bxx == 0) {  // edge filter of the pure vertical / horizontal modes (first column)
          const int tl = ref[0];
#pragma unroll
          for (int yy = 0; yy < 8; ++yy) {
            const int v = (int)(P[4 * yy] & 0xFFFF) + ((ref[-sgn * (byy + yy + 1)] - tl) >> 1);
            P[4 * yy] = (P[4 * yy] & 0xFFFF0000u) | (unsigned)min(maxval, max(0, v));
          }
        }
        if (is_ver) s = satd8x8<PACKED>(O, P);
        else {
          unsigned OT[32];
#pragma unroll
          for (int xx = 0; xx < 8; ++xx) {
            const uint4 q = *reinterpret_cast<const uint4*>(&s_orgT[(tx + xx) * 64 + ty]);
            OT[4 * xx] = q.x; OT[4 * xx + 1] = q.y; OT[4 * xx + 2] = q.z; OT[4 * xx + 3] = q.w;
          }
          s = satd8x8<PACKED>(OT, P);
        }
      }
      // tiles of one node meet through lane shuffles (lane = 8 * tile_row + tile_col)
      if (level <= 2) { s += __shfl_xor(s, 1); s += __shfl_xor(s, 8); }
      if (level <= 1) { s += __shfl_xor(s, 2); s += __shfl_xor(s, 16); }
      if (level == 0) { s += __shfl_xor(s, 4); s += __shfl_xor(s, 32); }
      const int own = level == 3 ? 0 : (level == 2 ? 9 : (level == 1 ? 27 : 63));
      if ((lane & own) == 0 && s_valid[node]) s_satd[node * 35 + mode] = s;
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
    if (out_all != nullptr) {  // parity output: every (node, mode) pair, not only the winner
      const long long o = (long long)(f * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      for (int i = tid; i < 85 * 35; i += 256) {
        const int node = i / 35, m = i - node * 35;
        FhevcNodeCost r;
        if (!s_valid[node]) { r.satd = 0xFFFFFFFFu; r.mode = 255; r.cost = -1.0; }
        else {
          r.satd = (unsigned)s_satd[i] >> (bd - 8);
          r.mode = (unsigned)m;
          const int bits = (m == 0) ? 2 : ((m == 1 || m == 26) ? 3 : 6);
          r.cost = __dadd_rn((double)r.satd, __dmul_rn((double)bits, sqrt_lambda));
        }
        out_all[o * (85 * 35) + i] = r;
      }
    }
    __syncthreads();
  }
}

// candidate lists (fhevc_intra_first_pass_candidates): one thread per node picks the K modes of smallest cost out of its 35 (node, mode) entries,
// best first, an earlier mode ahead of a later one of equal cost (TEncSearch::xUpdateCandList, TEncSearch.cpp:5385-5408); 255 for edge nodes
__global__ __launch_bounds__(256) void fhevc_first_pass_topk_kernel(const FhevcNodeCost* __restrict__ all, long long nodes, int k, uint8_t* __restrict__ modes)
{
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nodes) return;
  const FhevcNodeCost* a = all + n * 35;
  double cost[8];
  uint8_t mode[8];
  for (int i = 0; i < 8; ++i) { cost[i] = 1e300; mode[i] = 255; }
  const bool edge = a[0].satd == 0xFFFFFFFFu;
  for (int m = 0; m < 35 && !edge; ++m) {
    const double c = a[m].cost;
    // insert behind every entry of smaller or EQUAL cost: the earlier mode keeps its place
    int pos = 8;
#pragma unroll
    for (int i = 7; i >= 0; --i) if (c < cost[i]) pos = i;
#pragma unroll
    for (int i = 7; i > 0; --i) if (i > pos) { cost[i] = cost[i - 1]; mode[i] = mode[i - 1]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i == pos) { cost[i] = c; mode[i] = (uint8_t)m; }
  }
  for (int i = 0; i < k; ++i) modes[n * k + i] = mode[i];
}

}  // namespace

hipError_t fhevc_launch_first_pass_topk(const FhevcNodeCost* d_all, long long nodes, int k, uint8_t* d_modes, hipStream_t stream)
{
  if (nodes <= 0) return hipSuccess;
  hipLaunchKernelGGL(fhevc_first_pass_topk_kernel, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, stream, d_all, nodes, k, d_modes);
  return hipGetLastError();
}

hipError_t fhevc_launch_first_pass(const FhevcFrames& fr, double sqrt_lambda, FhevcNodeCost* d_out, FhevcNodeCost* d_all, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  const int grid = (int)(total < 2048 ? total : 2048);
  const bool packed = fr.bit_depth <= 10;
  if (fr.sample_bytes == 2) {
    if (packed) hipLaunchKernelGGL((fhevc_first_pass_kernel<int16_t, true>), dim3(grid), dim3(256), 0, stream, fr, sqrt_lambda, d_out, d_all);
    else hipLaunchKernelGGL((fhevc_first_pass_kernel<int16_t, false>), dim3(grid), dim3(256), 0, stream, fr, sqrt_lambda, d_out, d_all);
  } else {
    hipLaunchKernelGGL((fhevc_first_pass_kernel<uint8_t, true>), dim3(grid), dim3(256), 0, stream, fr, sqrt_lambda, d_out, d_all);
  }
  return hipGetLastError();
}
