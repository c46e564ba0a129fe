// k_motion_wide.hip -- the integer full search of config 4 at HM's own SearchRange (up to +-64), SAD distortion, 8-bit content; gfx950 only.
//
// Same definition as k_motion.hip in its SAD mode, i.e. the twin of TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848: raster order,
// strict "<", DF_SAD by setDistParam TComRdCost.cpp:205-236, cost = SAD + getCostOfVectorWithPredictor TComRdCost.h:166-174 with a zero
// predictor, border samples replicated as TComPicYuv::extendPicBorder) for all 85 CU nodes of a CTU at once -- but laid out for a window
// of (2R + 1)^2 = 16 641 vectors instead of 81:
//
//  * lane = a block of DB = 6 dy x 4 dx VECTORS (not a tile): v_qsad_pk_u16_u8 takes 8 reference bytes and 4 original bytes and adds the four
//    SADs of the 4-byte group at byte offsets 0..3 to four packed 16-bit accumulators -- 16 sample differences per lane and instruction,
//    eight times v_sad_u16.  The original bytes are the same for every lane: they sit in SGPRs (the instruction's scalar operand).
//  * the DB + 7 = 13 reference rows a tile needs for DB consecutive dy are read once (three dwords per row and lane: dx0 is a multiple of 4, so the
//    reads are aligned) and each row feeds the up to DB (dy, tile row) pairs it belongs to: 96 qsad and 26 LDS reads per lane and tile.  (DB = 6:
//    22 x 33 = 726 blocks of vectors fill three rounds of 256 lanes to 95 %; DB = 8 gives 561 = 73 %.)  v_qsad_pk_u16_u8 is a quarter-rate
//    instruction (tools/probes/probe_qsad_rate.hip: 4.4 x the issue time of v_sad_u8 for 4 x its work): it still beats v_sad_u8 +
//    v_alignbyte by 1.6 x, and it is what bounds the kernel.
//  * a node's SAD is the sum of its 8x8 tiles' SADs: tiles are visited in z-order, the 16x16 sums stay packed (<= 65 280), the 32x32 and 64x64
//    sums are 32-bit; per vector block that is 16 + 32 + 32 registers.
//  * minima: key = (SAD + vector cost) << 15 | raster index of the vector (cost < 2^17 for the 8x8 and 16x16 nodes, 64-bit keys above):
//    one v_mad_u32_u16 per vector builds it (packed half x 32768 + (cost << 15 | index)), v_min3_u32 trees and a DPP row minimum reduce it, one
//    LDS atomic minimum per wave and node merges the waves: the smallest key IS HM's first-found minimum in raster order.
//  * vectors past +R in the last dx group carry multiplier 0 and addend 0xFFFFFFFF (never the minimum); the last dy block is moved up to
//    end at +R (its first rows repeat vectors of the block before: a minimum does not care).
//
// Workgroup (4 waves) = one CTU at a time, grid-stride; window (64 + 2R + 3 columns, bytes) and the CTU's own bytes staged in LDS once.
#include "fhevc_internal.h"

namespace {

constexpr int WR = FHEVC_MOTION_WIDE_MAX_RANGE;   // 64
constexpr int WP = 64 + 2 * WR + 4;               // window pitch in bytes: 196 = 49 dwords (odd: rows 8 apart land 8 banks apart)
constexpr int WROWS = 64 + 2 * WR;
constexpr int DB = 6;                             // dy per block of vectors
constexpr int NV = 4 * DB;                        // vectors per lane and round

typedef unsigned long long u64;

// (the destination registers must not overlap ANY source, the accumulator included: the hardware writes the low dword before it has read the
//  sources for the high one -- with an overlapping allocation results 2 and 3 come out wrong, tools/probes/probe_qsad.hip; hence "=&v")
__device__ __forceinline__ u64 qsad(u64 ref8, unsigned cur4, u64 acc)
{
  u64 d;
  asm("v_qsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(ref8), "s"(cur4), "v"(acc));
  return d;
}
// (lo / hi half of a) * m + c
__device__ __forceinline__ unsigned mad_lo16(unsigned a, unsigned m, unsigned c)
{
  unsigned d;
  asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "v"(c));
  return d;
}
__device__ __forceinline__ unsigned mad_hi16(unsigned a, unsigned m, unsigned c)
{
  unsigned d;
  asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(d) : "v"(a), "v"(m), "v"(c));
  return d;
}
// minimum over the wave (every lane of the wave ends with it; uniform)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
  const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
  const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
  return min(min(a, b), min(c, d));
}

// T = int16_t (HM Pel planes holding 8-bit content) or uint8_t
template <typename T>
__global__ __launch_bounds__(256, 2) void fhevc_motion_wide_kernel(FhevcFrames F, int range, const uint32_t* __restrict__ mvtab,
                                                                 FhevcMotionNode* __restrict__ out)
{
  __shared__ __attribute__((aligned(16))) unsigned char s_ref[(WROWS + 1) * WP];
  __shared__ __attribute__((aligned(16))) unsigned char s_cur[64 * 64];
  __shared__ unsigned s_key32[FHEVC_NODES], s_zero[FHEVC_NODES];
  __shared__ u64 s_key64[5];
  const int tid = threadIdx.x, lane = tid & 63;
  const int band_rows = F.row_end - F.row_begin;
  const int per_frame = band_rows * F.ctus_x;
  const int total = per_frame * (F.num_frames - 1);
  const int side = 2 * range + 1;
  const int win_rows = 64 + 2 * range, win_cols = 64 + 2 * range + 3;   // + the columns the last dx group reads past +R
  const int ng = (side + 3) >> 2, ndb = (side + DB - 1) / DB, items = ng * ndb;
  const int rounds = (items + 255) >> 8;
  const T* plane = reinterpret_cast<const T*>(F.luma);

  for (int work = blockIdx.x; work < total; work += gridDim.x) {
    const int f = 1 + work / per_frame;
    const int rem = work % per_frame;
    const int cy = F.row_begin + rem / F.ctus_x, cx = rem % F.ctus_x;
    const long long cur_base = (long long)f * F.frame_stride, ref_base = (long long)(f - 1) * F.frame_stride;
    __syncthreads();  // the previous CTU's readers are done
    // ---- stage: reference window (coordinates clamped to the picture = replicated border), the CTU's own samples, the node slots ----
    {
      // chunks of 8 columns starting at a multiple of 8 picture columns (delta = what the window's first column lacks to one): a chunk inside
      // the picture is one 16-byte (uint8 planes: 8-byte) load; its bytes land at window columns wc - delta .. (two dword LDS stores when
      // delta is a multiple of 4, bytes otherwise)
      const int delta = (8 - (range & 7)) & 7;
      const int chunks = (win_cols + delta + 7) >> 3;
      for (int it = tid; it < win_rows * chunks; it += 256) {
        const int wr = it / chunks, wc = (it - wr * chunks) * 8 - delta;   // window column of the chunk's first sample (may be < 0)
        const int py = min(max(cy * 64 - range + wr, 0), F.height - 1);
        const long long row = ref_base + (long long)py * F.stride;
        const int px0 = cx * 64 - range + wc;
        unsigned lo = 0, hi = 0;
        const T* src = plane + row + px0;
        if (px0 >= 0 && px0 + 8 <= F.width && (reinterpret_cast<uintptr_t>(src) & (8 * sizeof(T) - 1)) == 0) {
          if (sizeof(T) == 2) {
            const uint4 q = *reinterpret_cast<const uint4*>(src);
            lo = __builtin_amdgcn_perm(q.y, q.x, 0x06040200u); hi = __builtin_amdgcn_perm(q.w, q.z, 0x06040200u);   // low bytes of the 16-bit samples
          } else {
            const uint2 q = *reinterpret_cast<const uint2*>(src);
            lo = q.x; hi = q.y;
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lo |= ((unsigned)plane[row + min(max(px0 + k, 0), F.width - 1)] & 0xFFu) << (8 * k);
            hi |= ((unsigned)plane[row + min(max(px0 + 4 + k, 0), F.width - 1)] & 0xFFu) << (8 * k);
          }
        }
        unsigned char* dst = s_ref + wr * WP + wc;
        if ((delta & 3) == 0) {
          if (wc >= 0 && wc + 4 <= WP) *reinterpret_cast<unsigned*>(dst) = lo;
          if (wc + 4 >= 0 && wc + 8 <= WP) *reinterpret_cast<unsigned*>(dst + 4) = hi;
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (wc + k >= 0 && wc + k < WP) dst[k] = (unsigned char)(((k < 4 ? lo : hi) >> (8 * (k & 3))) & 0xFFu);
        }
      }
      for (int it = tid; it < 64 * 16; it += 256) {
        const int y = it >> 4, x = (it & 15) * 4;
        const int py = min(cy * 64 + y, F.height - 1);
        const long long row = cur_base + (long long)py * F.stride;
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) v |= ((unsigned)plane[row + min(cx * 64 + x + k, F.width - 1)] & 0xFFu) << (8 * k);
        *reinterpret_cast<unsigned*>(s_cur + y * 64 + x) = v;
      }
      if (tid < FHEVC_NODES) s_key32[tid] = 0xFFFFFFFFu;
      if (tid < 5) s_key64[tid] = ~0ull;
    }
    __syncthreads();
    // ---- the SAD at vector (0, 0) of every node: wave 0, lane = tile, node sums through lane exchanges (as k_motion.hip) ----
    if (tid < 64) {
      const int tx = lane & 7, ty = lane >> 3;
      const bool inside = (cx * 64 + tx * 8 + 8 <= F.width) && (cy * 64 + ty * 8 + 8 <= F.height);
      unsigned t8 = 0;
      for (int j = 0; j < 8; ++j) {
        const unsigned char* c = s_cur + (ty * 8 + j) * 64 + tx * 8;
        const unsigned char* r = s_ref + (ty * 8 + j + range) * WP + tx * 8 + range;
#pragma unroll
        for (int i = 0; i < 8; ++i) t8 += (unsigned)abs((int)c[i] - (int)r[i]);
      }
      t8 = inside ? t8 : 0u;
      unsigned a = t8 + __shfl_xor(t8, 1);
      const unsigned s2 = a + __shfl_xor(a, 8);
      a = s2 + __shfl_xor(s2, 2);
      const unsigned s1 = a + __shfl_xor(a, 16);
      a = s1 + __shfl_xor(s1, 4);
      const unsigned s0 = a + __shfl_xor(a, 32);
      s_zero[21 + lane] = t8;
      if (((tx | ty) & 1) == 0) s_zero[5 + (ty >> 1) * 4 + (tx >> 1)] = s2;
      if (((tx | ty) & 3) == 0) s_zero[1 + (ty >> 2) * 2 + (tx >> 2)] = s1;
      if (lane == 0) s_zero[0] = s0;
    }
    // ---- the search ----
    for (int round = 0; round < rounds; ++round) {
      const int item = min(tid + 256 * round, items - 1);   // spare lanes of the last round repeat the last item
      const int db = item / ng, gx = item - db * ng;
      const int dyb = min(-range + DB * db, range - (DB - 1)), dx0 = -range + 4 * gx;
      // per vector (d, k): addend = cost << 15 | raster index; multiplier per k: 32768, or 0 past +R (then the addend is all ones)
      unsigned addend[NV], mul[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) mul[k] = (dx0 + k <= range) ? 32768u : 0u;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ras = (dyb + d + range) * side + min(dx0 + k, range) + range;
          addend[4 * d + k] = (dx0 + k <= range) ? ((mvtab[ras] << 15) | (unsigned)ras) : 0xFFFFFFFFu;
        }
      const unsigned char* lane_ref = s_ref + (dyb + range) * WP + 4 * gx;
      unsigned s16[2 * DB], s32[NV], s64[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) { s32[i] = 0; s64[i] = 0; }
      for (int b = 0; b < 16; ++b) {   // 16x16 blocks in z-order
        const int q = b >> 2, s = b & 3;
        const int by = 2 * (q >> 1) + (s >> 1), bx = 2 * (q & 1) + (s & 1);
        const unsigned char* blk_ref = lane_ref + (by * 16) * WP + bx * 16;
#pragma unroll
        for (int i = 0; i < 2 * DB; ++i) s16[i] = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {   // its four 8x8 tiles
          const int ty = 2 * by + (t >> 1), tx = 2 * bx + (t & 1);
          if ((cx * 64 + tx * 8 + 8 > F.width) || (cy * 64 + ty * 8 + 8 > F.height)) continue;   // uniform: tile outside the picture adds 0
          // the tile's original bytes -> SGPRs
          unsigned clo[8], chi[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const uint2 c = *reinterpret_cast<const uint2*>(s_cur + (ty * 8 + r) * 64 + tx * 8);
            clo[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)c.x);
            chi[r] = (unsigned)__builtin_amdgcn_readfirstlane((int)c.y);
          }
          const unsigned char* tr = blk_ref + ((t >> 1) * 8) * WP + (t & 1) * 8;
          u64 acc[DB];
#pragma unroll
          for (int d = 0; d < DB; ++d) acc[d] = 0;
#pragma unroll
          for (int j = 0; j < DB + 7; ++j) {   // reference row j of the block of DB dy: tile row r = j - d for vector row d
            const unsigned* p = reinterpret_cast<const unsigned*>(tr + j * WP);
            const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
            const u64 w01 = ((u64)d1 << 32) | d0, w12 = ((u64)d2 << 32) | d1;
#pragma unroll
            for (int d = 0; d < DB; ++d) {
              const int r = j - d;
              if (r >= 0 && r < 8) {
                acc[d] = qsad(w01, clo[r], acc[d]);
                acc[d] = qsad(w12, chi[r], acc[d]);
              }
            }
          }
          // 8x8 node: keys of the lane's vectors, lane minimum, wave minimum, one atomic per wave
          unsigned m = 0xFFFFFFFFu;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const unsigned lo = (unsigned)acc[d], hi = (unsigned)(acc[d] >> 32);
            const unsigned k0 = mad_lo16(lo, mul[0], addend[4 * d + 0]), k1 = mad_hi16(lo, mul[1], addend[4 * d + 1]);
            const unsigned k2 = mad_lo16(hi, mul[2], addend[4 * d + 2]), k3 = mad_hi16(hi, mul[3], addend[4 * d + 3]);
            m = min(min(m, k0), min(min(k1, k2), k3));
            s16[2 * d] += lo;       // packed halves <= 4 * 16 320: no carry between them
            s16[2 * d + 1] += hi;
          }
          m = wave_min_u32(m);
          if (lane == 0) atomicMin(&s_key32[21 + ty * 8 + tx], m);
        }
        // 16x16 node
        {
          unsigned m = 0xFFFFFFFFu;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const unsigned lo = s16[2 * d], hi = s16[2 * d + 1];
            const unsigned k0 = mad_lo16(lo, mul[0], addend[4 * d + 0]), k1 = mad_hi16(lo, mul[1], addend[4 * d + 1]);
            const unsigned k2 = mad_lo16(hi, mul[2], addend[4 * d + 2]), k3 = mad_hi16(hi, mul[3], addend[4 * d + 3]);
            m = min(min(m, k0), min(min(k1, k2), k3));
            s32[4 * d + 0] += lo & 0xFFFFu; s32[4 * d + 1] += lo >> 16;
            s32[4 * d + 2] += hi & 0xFFFFu; s32[4 * d + 3] += hi >> 16;
          }
          m = wave_min_u32(m);
          if (lane == 0) atomicMin(&s_key32[5 + by * 4 + bx], m);
        }
        if (s == 3) {   // the quadrant is complete: 32x32 node q, 64-bit keys (cost << 32 | raster index); first-found inside the lane
          unsigned bh = 0xFFFFFFFFu, bl = 0;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const unsigned h = (mul[v & 3] != 0) ? s32[v] + (addend[v] >> 15) : 0xFFFFFFFFu;
            if (h < bh) { bh = h; bl = addend[v] & 0x7FFFu; }
            s64[v] += s32[v];
            s32[v] = 0;
          }
          atomicMin(&s_key64[1 + q], ((u64)bh << 32) | bl);
        }
      }
      {   // 64x64 node
        unsigned bh = 0xFFFFFFFFu, bl = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const unsigned h = (mul[v & 3] != 0) ? s64[v] + (addend[v] >> 15) : 0xFFFFFFFFu;
          if (h < bh) { bh = h; bl = addend[v] & 0x7FFFu; }
        }
        atomicMin(&s_key64[0], ((u64)bh << 32) | bl);
      }
    }
    __syncthreads();
    if (tid < FHEVC_NODES) {
      int l, ni;
      if (tid == 0) { l = 0; ni = 0; } else if (tid < 5) { l = 1; ni = tid - 1; } else if (tid < 21) { l = 2; ni = tid - 5; } else { l = 3; ni = tid - 21; }
      const int n = 64 >> l, cnt = 1 << l;
      const int bx = ni % cnt, by = ni / cnt;
      FhevcMotionNode o;
      if (cx * 64 + bx * n + n > F.width || cy * 64 + by * n + n > F.height) {
        o.satd_zero = o.satd_best = o.cost_best = 0xFFFFFFFFu; o.mvx = 0; o.mvy = 0;
      } else {
        unsigned cost, ras;
        if (l < 2) { const u64 k = s_key64[tid]; cost = (unsigned)(k >> 32); ras = (unsigned)k; }
        else { const unsigned k = s_key32[tid]; cost = k >> 15; ras = k & 0x7FFFu; }
        o.satd_zero = s_zero[tid]; o.cost_best = cost; o.satd_best = cost - mvtab[ras];
        o.mvx = (short)((int)(ras % side) - range); o.mvy = (short)((int)(ras / side) - range);
      }
      const long long oc = (long long)((f - 1) * band_rows + (cy - F.row_begin)) * F.ctus_x + cx;
      out[oc * FHEVC_NODES + tid] = o;
    }
  }
}

}  // namespace

hipError_t fhevc_launch_motion_wide(const FhevcFrames& fr, int range, const uint32_t* d_mvtab, FhevcMotionNode* d_out, int num_cus, hipStream_t stream)
{
  const long long total = (long long)(fr.row_end - fr.row_begin) * fr.ctus_x * (fr.num_frames - 1);
  if (total <= 0) return hipSuccess;
  const int grid = (int)(total < 2LL * num_cus ? total : 2LL * num_cus);
  if (fr.sample_bytes == 2) hipLaunchKernelGGL((fhevc_motion_wide_kernel<int16_t>), dim3(grid), dim3(256), 0, stream, fr, range, d_mvtab, d_out);
  else hipLaunchKernelGGL((fhevc_motion_wide_kernel<uint8_t>), dim3(grid), dim3(256), 0, stream, fr, range, d_mvtab, d_out);
  return hipGetLastError();
}
