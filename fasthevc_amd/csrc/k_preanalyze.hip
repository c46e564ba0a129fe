// k_preanalyze.hip -- adaptive-QP pre-analysis, gfx950 only.
//
// Bit-exact twin of TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64-152): for every AQ layer d (parts of
// 64 >> d samples, cropped at the picture edge) and every part, the minimum over the four quadrants of the sample
// variance, activity = 1 + minVar.  Sums are 64-bit integers, the variance is formed in double with the reference's
// operation order (this file is built with -ffp-contract=off), so the doubles match bit for bit.  The per-layer
// average (a sequential double sum in raster order in the reference) is formed by the caller in that same order.
//
// One pass over the planar-Y picture for ALL layers: HBM-bound (each sample read once).  One lane owns one 8x8
// block (one wave = one CTU) and reduces it to four 4x4-cell (sum, sum of squares) pairs.  Whole CTUs never touch
// LDS: the quadrants of the 8/16/32/64 parts are the 4x4 / 8x8 / 16x16 / 32x32 sums, which meet through lane
// shuffles (lane = 8 * block_row + block_col).  CTUs cropped by the picture edge take the general path: because
// picture sizes are multiples of the minimum CU size (8), every quadrant of every cropped part is a rectangle of
// whole 4x4 cells, summed from a per-wave LDS image.  No workgroup barriers: waves are independent.
#include "fhevc_internal.h"

namespace {

struct QuadSum { unsigned long long s, q; };

__device__ __forceinline__ QuadSum rect_sum(const unsigned* cs, const unsigned* cq, int x0, int x1, int y0, int y1)
{
  QuadSum r{ 0, 0 };
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      r.s += cs[y * 16 + x];
      r.q += cq[y * 16 + x];
    }
  return r;
}

__device__ __forceinline__ double quad_var(QuadSum a, unsigned npix)
{
  const double avg = __ddiv_rn((double)a.s, (double)npix);
  return __dsub_rn(__ddiv_rn((double)a.q, (double)npix), __dmul_rn(avg, avg));
}

// npix a power of two: x / npix == x * (1 / npix) bit for bit (scaling by a power of two is exact), no f64 division
__device__ __forceinline__ double quad_var_pow2(QuadSum a, double inv_npix)
{
  const double avg = __dmul_rn((double)a.s, inv_npix);
  return __dsub_rn(__dmul_rn((double)a.q, inv_npix), __dmul_rn(avg, avg));
}

template <typename T>
__global__ __launch_bounds__(256) void fhevc_preanalyze_kernel(FhevcFrames F, int layers, long long parts_per_frame,
                                                                double* __restrict__ out)
{
  __shared__ unsigned cell_s[4][256], cell_q[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bx = lane & 7, by = lane >> 3;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * F.num_frames;
  unsigned* cs = cell_s[wave];
  unsigned* cq = cell_q[wave];
  const int vblock = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous runs, as in k_hadamard
  for (int base = vblock * 4; base < total; base += gridDim.x * 4) {
    const int work = base + wave;
    const bool live = work < total;
    const int f = live ? work / per_frame : 0;
    const int rem = work - f * per_frame;
    const int cy = F.row_begin + (live ? rem / F.ctus_x : 0), cx = live ? rem % F.ctus_x : 0;
    const int vw = min(64, F.width - cx * 64), vh = min(64, F.height - cy * 64);
    const bool ok = live && (bx * 8 < vw) && (by * 8 < vh);
    unsigned s4[4] = { 0, 0, 0, 0 }, q4[4] = { 0, 0, 0, 0 };
    if (ok) {
      const T* p = reinterpret_cast<const T*>(F.luma) + (long long)f * F.frame_stride +
                   (long long)(cy * 64 + by * 8) * F.stride + cx * 64 + bx * 8;
      const bool al = (reinterpret_cast<uintptr_t>(p) & (8 * sizeof(T) - 1)) == 0 && ((F.stride * sizeof(T)) & (8 * sizeof(T) - 1)) == 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int v[8];
        if (al) {
          if (sizeof(T) == 2) {
            const uint4 w = *reinterpret_cast<const uint4*>(p + (long long)j * F.stride);
            v[0] = (short)(w.x & 0xFFFF); v[1] = (short)(w.x >> 16); v[2] = (short)(w.y & 0xFFFF); v[3] = (short)(w.y >> 16);
            v[4] = (short)(w.z & 0xFFFF); v[5] = (short)(w.z >> 16); v[6] = (short)(w.w & 0xFFFF); v[7] = (short)(w.w >> 16);
          } else {
            const uint2 w = *reinterpret_cast<const uint2*>(p + (long long)j * F.stride);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = (w.x >> (8 * k)) & 0xFF; v[4 + k] = (w.y >> (8 * k)) & 0xFF; }
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = (int)p[(long long)j * F.stride + k];
        }
        const int c = (j >> 2) * 2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s4[c + (k >> 2)] += (unsigned)v[k];
          q4[c + (k >> 2)] += (unsigned)(v[k] * v[k]);
        }
      }
    }
    double* fout = out + (long long)f * parts_per_frame;
    long long loff[4];  // layer offsets of the caller's layout: layer d holds ceil(H/P) x ceil(W/P) doubles, P = 64 >> d
    loff[0] = 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) loff[d + 1] = loff[d] + (long long)((F.width + (64 >> d) - 1) >> (6 - d)) * ((F.height + (64 >> d) - 1) >> (6 - d));
    if (layers > 3 && ok) {  // 8x8 parts: the lane's own four cells (same on both paths)
      double mv = 1.7976931348623157e308;
#pragma unroll
      for (int c = 0; c < 4; ++c) mv = fmin(mv, quad_var_pow2(QuadSum{ s4[c], q4[c] }, 1.0 / 16.0));
      fout[loff[3] + (long long)(cy * 8 + by) * ((F.width + 7) >> 3) + cx * 8 + bx] = __dadd_rn(1.0, mv);
    }
    const bool whole = live && vw == 64 && vh == 64;  // wave-uniform
    if (whole) {
      // quadrant sums by doubling: 8x8 (this lane), 16x16 (lanes ^1, ^8), 32x32 (lanes ^2, ^16); minima over the four
      // quadrants of a part with the next pair of lane bits
      unsigned long long S = (unsigned long long)s4[0] + s4[1] + s4[2] + s4[3];
      unsigned long long Q = (unsigned long long)q4[0] + q4[1] + q4[2] + q4[3];
#pragma unroll
      for (int d = 2; d >= 0; --d) {
        const int lo = 1 << (2 - d), hi = 8 << (2 - d);       // lane bits that enumerate the quadrants of a layer-d part
        const double inv_npix = d == 2 ? 1.0 / 64.0 : (d == 1 ? 1.0 / 256.0 : 1.0 / 1024.0);  // quadrants of 8x8, 16x16, 32x32
        if (d < layers) {
          double var = quad_var_pow2(QuadSum{ S, Q }, inv_npix);
          var = fmin(var, __shfl_xor(var, lo));
          var = fmin(var, __shfl_xor(var, hi));
          if ((bx & (2 * lo - 1)) == 0 && (by & (2 * lo - 1)) == 0) {  // one lane per part writes
            const int nw = (F.width + (64 >> d) - 1) >> (6 - d);
            const int gx = (cx << d) + (bx >> (3 - d)), gy = (cy << d) + (by >> (3 - d));
            fout[loff[d] + (long long)gy * nw + gx] = __dadd_rn(1.0, var);
          }
        }
        if (d > 0) {  // merge the four quadrants into the next size up
          S += __shfl_xor(S, lo); S += __shfl_xor(S, hi);
          Q += __shfl_xor(Q, lo); Q += __shfl_xor(Q, hi);
        }
      }
    } else if (live) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int ci = (2 * by + (c >> 1)) * 16 + 2 * bx + (c & 1);
        cs[ci] = s4[c];
        cq[ci] = q4[c];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // layers 0..2: 1 + 4 + 16 parts x 4 quadrants = 84 (part, quadrant) items, two passes over the wave
#pragma unroll 1
      for (int pass = 0; pass < 2; ++pass) {
        const int it = pass * 64 + lane;
        const int d = it < 4 ? 0 : (it < 20 ? 1 : 2);
        const int first = d == 0 ? 0 : (d == 1 ? 4 : 20);
        const int part = (it - first) >> 2, quad = it & 3;
        const int pc = 16 >> d;                     // part size in cells
        const int ox = (part & ((1 << d) - 1)) * pc, oy = (part >> d) * pc;
        const int cw = min(pc, (vw >> 2) - ox), ch = min(pc, (vh >> 2) - oy);  // cropped part, in cells
        const bool valid = it < 84 && d < layers && cw > 0 && ch > 0;
        double var = 0.0;
        if (valid) {
          const int hw = cw >> 1, hh = ch >> 1;
          const int x0 = ox + ((quad & 1) ? hw : 0), x1 = ox + ((quad & 1) ? cw : hw);
          const int y0 = oy + ((quad & 2) ? hh : 0), y1 = oy + ((quad & 2) ? ch : hh);
          var = quad_var(rect_sum(cs, cq, x0, x1, y0, y1), (unsigned)(hw * 4) * (unsigned)(hh * 4));
        }
        var = fmin(var, __shfl_xor(var, 1));
        var = fmin(var, __shfl_xor(var, 2));
        if (valid && quad == 0) {
          const int nw = (F.width + (64 >> d) - 1) >> (6 - d);
          const int gx = cx * (1 << d) + (part & ((1 << d) - 1)), gy = cy * (1 << d) + (part >> d);
          fout[loff[d] + (long long)gy * nw + gx] = __dadd_rn(1.0, var);
        }
      }
      __builtin_amdgcn_wave_barrier();  // the next CTU of this wave overwrites the cell image
    }
  }
}

}  // namespace

hipError_t fhevc_launch_preanalyze(const FhevcFrames& fr, int layers, long long parts_per_frame, double* d_activity,
                                   int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * fr.num_frames;
  if (total <= 0) return hipSuccess;
  long long grid = (total + 3) / 4;
  const long long cap = (long long)num_cus * 8;
  if (grid > cap) grid = cap;
  grid = (grid + 7) & ~7LL;
  if (fr.sample_bytes == 2)
    hipLaunchKernelGGL(fhevc_preanalyze_kernel<int16_t>, dim3((unsigned)grid), dim3(256), 0, stream, fr, layers, parts_per_frame, d_activity);
  else
    hipLaunchKernelGGL(fhevc_preanalyze_kernel<uint8_t>, dim3((unsigned)grid), dim3(256), 0, stream, fr, layers, parts_per_frame, d_activity);
  return hipGetLastError();
}
