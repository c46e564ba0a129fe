// fhevc_api.hip -- the C ABI of include/fasthevc.h: context, weight image, staging buffers, launches.
// Host-side C++ only; all device work is in k_cnn.hip / k_hadamard.hip / k_firstpass.hip.
#include "../../include/fasthevc.h"
#include "fhevc_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

struct TimedLaunch { hipEvent_t start, stop; int which; };

}  // namespace

struct fhevc_ctx {
  fhevc_cfg cfg{};
  int device = 0, num_cus = 256;
  hipStream_t stream = nullptr;
  int ctus_x = 0, ctus_y = 0, num_ctus = 0;
  int dev_stride = 0;  // samples, staging plane
  // weight image
  bool have_weights = false;
  uint4* d_frag = nullptr; float* d_bias = nullptr; uint8_t* d_whead = nullptr; int32_t* d_bhead = nullptr;
  uint4* d_frag_i8 = nullptr; int32_t* d_bias_i8 = nullptr;  // the i8 variant of conv2 / conv3 (k_cnn.hip)
  // a member of the reference's Bayesian-optimisation network family (FHW3 blob; k_cnn_family.inc): set instead of the arrays above
  bool family = false;
  bool fam_layers = false;            // ... run layer by layer through HBM (k_cnn_layers.inc): every member the fused kernels do not cover
  bool fam_d2 = false;                // ... of those, the members k_cnn_d2.inc runs as one LDS-resident kernel (the layer images are the same; no HBM scratch)
  FhevcLayersWeights lw = {};
  std::vector<void*> lw_bufs;         // everything lw points to (freed with the context / the next blob)
  // the layer path's activation tensors are ONE set per context: a launch on another stream than the previous one waits for that one's last kernel
  // (the host batch alternates two streams; callers may pass any stream per call)
  hipEvent_t lw_done = nullptr; hipStream_t lw_last_stream = nullptr; bool lw_in_flight = false;
  int fam_c[3] = { 0, 0, 0 };
  uint4* f_frag1 = nullptr; float* f_bias1 = nullptr; uint4* f_frag2 = nullptr; uint4* f_frag3 = nullptr; int32_t* f_bias_i8 = nullptr;
  uint8_t* f_whead = nullptr; uint8_t* f_headm = nullptr; int32_t* f_bhead = nullptr;
  int shift[3] = { 0, 0, 0 };
  int requant_mode[3] = { 0, 0, 0 };
  bool cnn_i8 = true;                                         // fhevc_set_cnn_arith / FHEVC_CNN_ARITH at fhevc_create
  float scale[3] = { 1, 1, 1 };
  // staging for the host-buffer entry points
  int16_t* d_luma = nullptr; uint8_t* d_depth = nullptr; int32_t* d_had = nullptr; FhevcNodeCost* d_nodes = nullptr;
  int16_t* d_satd = nullptr; uint32_t* d_satd_out = nullptr;
  hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
  // kernel timing
  bool fuse_hadamard = true;  // FHEVC_FUSE_HADAMARD=0 keeps the stand-alone Hadamard launch (A/B measurements)
  bool motion_sad = false;    // fhevc_set_motion_distortion: SAD (HM's integer-search distortion) instead of Hadamard SATD
  bool cnn_pipe = false;      // FHEVC_CNN_PIPE=1: the i8 form as the two-stage software pipeline over CTUs (k_cnn.hip: fhevc_cnn_depth_pipe_kernel)
  bool had_valu = true;       // FHEVC_HADAMARD_FORM=mfma: the fused Hadamard of 8-bit content on the bf16 MFMA from the staged tile instead of packed
                              // 16-bit VALU (parity-green, and measured 7 % SLOWER in round 3: profiles/r03_ab_hadamard_forms.log) -- kept for A/B and tests
  FhevcKnobs knobs;           // the environment's tuning / test switches, read once in fhevc_create
  bool timing = false;
  std::vector<TimedLaunch> pending;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  double sum_ms[5] = { 0, 0, 0, 0, 0 };
  uint64_t launches[5] = { 0, 0, 0, 0, 0 };
  double* d_act = nullptr;
  int16_t* d_pair = nullptr;          // two staging planes (reference, current) of fhevc_motion_search
  FhevcMotionNode* d_motion = nullptr;
  FhevcNodeCost* d_cand_all = nullptr; uint8_t* d_cand = nullptr;   // fhevc_intra_first_pass_candidates: every (node, mode) cost, the lists
  uint32_t* d_mvtab = nullptr;        // vector costs of the wide search (k_motion_wide.hip), rebuilt when (qp, range) changes
  int mvtab_qp = -1, mvtab_range = -1;
  hipEvent_t mvtab_used = nullptr;    // recorded behind every launch that reads d_mvtab, on whatever stream the caller passed: the rebuild waits for it
  std::vector<uint32_t> mvtab_host;
  // host-batch ring (fhevc_predict_frames): two slots, each with its own stream, device buffers and pinned staging
  struct Slot {
    hipStream_t st = nullptr;
    uint8_t* d_in = nullptr; uint8_t* d_depth = nullptr; int32_t* d_had = nullptr;
    uint8_t* h_in = nullptr; uint8_t* h_depth = nullptr; int32_t* h_had = nullptr;  // pinned staging (pageable callers)
    size_t in_cap = 0, frames_cap = 0, h_in_cap = 0;
    // what is in flight on this slot: where its outputs go once the stream has drained
    int frames = 0; uint8_t* out_depth = nullptr; int32_t* out_had = nullptr; bool staged_out = false;
  } slot[2];
  uint8_t* d_depth_max = nullptr;
  fhevc_stats stats{};
  std::string err;
  // cfg.num_devices > 1: this context is the PRIMARY (device_ids[0]); the other devices are full single-device contexts of their own.
  // The host-buffer entry points shard over them (CTU-row bands of a picture, runs of pictures of a batch); a device that fails is
  // dropped for the rest of the context's life and its share is redone on a device that works (devices_failed counts them)
  std::vector<fhevc_ctx*> peers;
  int fail_peer_for_test = -1;   // FHEVC_TEST_FAIL_DEVICE=<index >= 1>: that device reports a failure on its next share (tests)
};

namespace {

int fail(fhevc_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
  if (c) {
    c->err = what;
    if (e != hipSuccess) { c->err += ": "; c->err += hipGetErrorString(e); }
  }
  return code;
}

#define HIP_TRY(c, call)                                                   \
  do {                                                                     \
    hipError_t e_ = (call);                                                \
    if (e_ != hipSuccess) return fail((c), FHEVC_E_HIP, #call, e_);        \
  } while (0)


// FHW1 blob layout (fasthevc_amd/weights.py)
struct BlobView {
  const int32_t* shift;
  const int8_t* w1; const int32_t* b1;
  const int8_t* w2; const int32_t* b2;
  const int8_t* w3; const int32_t* b3;
  const int8_t* wh64; const int32_t* bh64;
  const int8_t* wh32; const int32_t* bh32;
  const int8_t* wh16; const int32_t* bh16;
  const int32_t* qp_bias;
};
constexpr size_t kBlobBytes = 8 + 12 + 144 + 64 + 4608 + 128 + 18432 + 256 + 8192 + 8 + 8192 + 8 + 2048 + 8 + 3 * 52 * 4;

bool parse_blob(const uint8_t* p, size_t n, BlobView& v, std::vector<uint8_t>& aligned)
{
  if (n != kBlobBytes || std::memcmp(p, "FHW1", 4) != 0) return false;
  uint32_t ver;
  std::memcpy(&ver, p + 4, 4);
  if (ver != 2) return false;
  // copy the int32 sections out to aligned storage: the blob packs int8 and int32 arrays back to back
  aligned.assign(p, p + n);
  size_t off = 8;
  auto take = [&](size_t bytes) { const uint8_t* q = aligned.data() + off; off += bytes; return q; };
  v.shift = reinterpret_cast<const int32_t*>(take(12));
  v.w1 = reinterpret_cast<const int8_t*>(take(144));   v.b1 = reinterpret_cast<const int32_t*>(take(64));
  v.w2 = reinterpret_cast<const int8_t*>(take(4608));  v.b2 = reinterpret_cast<const int32_t*>(take(128));
  v.w3 = reinterpret_cast<const int8_t*>(take(18432)); v.b3 = reinterpret_cast<const int32_t*>(take(256));
  v.wh64 = reinterpret_cast<const int8_t*>(take(8192)); v.bh64 = reinterpret_cast<const int32_t*>(take(8));
  v.wh32 = reinterpret_cast<const int8_t*>(take(8192)); v.bh32 = reinterpret_cast<const int32_t*>(take(8));
  v.wh16 = reinterpret_cast<const int8_t*>(take(2048)); v.bh16 = reinterpret_cast<const int32_t*>(take(8));
  v.qp_bias = reinterpret_cast<const int32_t*>(take(3 * 52 * 4));
  return off == n;
}

inline int32_t rd32(const int32_t* p, int i)  // unaligned-safe read
{
  int32_t v;
  std::memcpy(&v, reinterpret_cast<const uint8_t*>(p) + 4 * (size_t)i, 4);
  return v;
}

// Build the device weight image: MFMA A-operand fragments in lane order (k_cnn.hip header comment).
int build_weight_image(fhevc_ctx* c, const BlobView& b)
{
  int new_shift[3], new_mode[3] = { 0, 0, 0 };
  for (int l = 0; l < 3; ++l) {
    const int s = rd32(b.shift, l);
    if (s < 0 || s > 14) return fail(c, FHEVC_E_WEIGHTS, "shift out of range (0..14)");
    new_shift[l] = s;
  }
  std::vector<uint16_t> frag((size_t)FHEVC_FRAG_TOTAL * 8, 0);
  // all conv weights carry their layer's 2^-shift: w * 2^-s is still exact in bf16 (a power-of-two scaling of an 8-bit integer)
  // and every partial sum is a multiple of 2^-s below 2^24 * 2^-s, so the fp32 accumulation stays exact and the MFMA
  // delivers (acc + b) * 2^-s directly: one multiply per output less in the epilogue
  // conv1 takes bf16 operands (its input tile is bf16), conv2 and conv3 f16 ones (their inputs are written as f16 by the
  // epilogues before them): |w| * 2^-s >= 2^-14 is a normal f16 number and 7 significant bits fit its 11
  auto put_scaled = [&](int layer, int frag_idx, int lane, int j, int v) {
    float f = std::ldexp((float)v, -rd32(b.shift, layer));
    uint16_t bits;
    if (layer == 0) {
      uint32_t u;
      std::memcpy(&u, &f, 4);
      bits = (uint16_t)(u >> 16);
    } else {
      const _Float16 h = (_Float16)f;
      std::memcpy(&bits, &h, 2);
    }
    frag[((size_t)frag_idx + lane) * 8 + j] = bits;
  };
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * h + j;
      // conv1: two MFMAs (jm = pre-pool column px).  Row m = r + 32*jm: channel = m[1:0] + 4*m[3] + 8*m[2],
      // pre-pool row py = m[4].  K slot k = 8h + j addresses the 4x4 input window: column wc = 2h + ((j >> 1) & 1),
      // row wr = 2*(j >> 2) + (j & 1) (two row-pair dwords per column).  Tap (ky, kx) = (wr - py, wc - px).
      for (int jm = 0; jm < 2; ++jm) {
        const int ch = (r & 3) + 4 * ((r >> 3) & 1) + 8 * ((r >> 2) & 1), py = (r >> 4) & 1, px = jm;
        const int wc = 2 * h + ((j >> 1) & 1), wr = 2 * (j >> 2) + (j & 1);
        const int ky = wr - py, kx = wc - px;
        if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) put_scaled(0, FHEVC_FRAG_CONV1 + 64 * jm, lane, j, b.w1[ch * 9 + ky * 3 + kx]);
      }
      // conv2: K-step s = tap, k = input channel
      for (int s = 0; s < 9; ++s) put_scaled(1, FHEVC_FRAG_CONV2 + s * 64, lane, j, b.w2[((r * 16 + k) * 9) + s]);
    }
  }
  // conv3 runs on v_mfma_f32_16x16x32_bf16: lane (m = lane & 15, kg = lane >> 4) holds A[m][8 kg + j]; tile t = the wave's 32
  // output channels, fragment s = 9 mt + tap: M tile mt (16 channels), K = the tap's 32 input channels
  for (int lane = 0; lane < 64; ++lane) {
    const int m = lane & 15, kg = lane >> 4;
    for (int j = 0; j < 8; ++j)
      for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 18; ++s) {
#if FHEVC_F16_CONV3_32
          // v_mfma_f32_32x32x16_f16: lane (row = lane & 31, h = lane >> 5) holds A[row][8 h + j]; fragment s = 2 tap + c2: K = the tap's
          // channels 16 c2 .. 16 c2 + 15, rows = the tile's 32 output channels
          (void)m; (void)kg;
          const int oc = 32 * t + (lane & 31), ic = 16 * (s & 1) + 8 * (lane >> 5) + j, tap = s >> 1;
#else
          const int oc = 32 * t + 16 * (s / 9) + m, ic = 8 * kg + j, tap = s % 9;
#endif
          put_scaled(2, FHEVC_FRAG_CONV3 + (t * 18 + s) * 64, lane, j, b.w3[(oc * 32 + ic) * 9 + tap]);
        }
  }
  // the source Hadamard's constant A operands (k_cnn.hip, HAD == 2; v_mfma_f32_32x32x16_bf16): row m of M tile mt = coefficient
  // c = 32 mt + m = (u = c >> 3, v = c & 7) of the 2-D Walsh-Hadamard transform of an 8x8 block, K slot 8 h + j of step st = the sample
  // at column 4 h + (j >> 1), row 2 st - 1 + (j & 1) of the block (the staged tile keeps picture rows 2P - 1 and 2P in one dword):
  // +-1 by the parity of popcount(u & row) + popcount(v & column); rows -1 and 8 belong to the neighbouring blocks and the DC
  // coefficient is not part of the sum (TEncCu.cpp:1319): zero
  for (int mt = 0; mt < 2; ++mt)
    for (int st = 0; st < 5; ++st)
      for (int lane = 0; lane < 64; ++lane) {
        const int c = 32 * mt + (lane & 31), u = c >> 3, v = c & 7, h = lane >> 5;
        for (int j = 0; j < 8; ++j) {
          const int col = 4 * h + (j >> 1), row = 2 * st - 1 + (j & 1);
          uint16_t bits = 0;
          if (c != 0 && row >= 0 && row <= 7) bits = ((__builtin_popcount(u & row) + __builtin_popcount(v & col)) & 1) ? 0xBF80 : 0x3F80;
          frag[((size_t)FHEVC_FRAG_HAD + (mt * 5 + st) * 64 + lane) * 8 + j] = bits;
        }
      }
  // the i8 variant (v_mfma_i32_32x32x32_i8: a lane holds 16 signed bytes of K; lanes 0-31 K 0-15, lanes 32-63 K 16-31):
  //   conv2 fragments (k_cnn.hip, conv2_half_i8): row = output channel lane & 31, K byte j of lane half h = input channel j at tap
  //                           0-2: (ky = h, kx = 0..2); 3: (2, kx = 2 h); 4: (2, 1) for h = 0, zero for h = 1; 5: zero | (2, 1);
  //   conv3 fragment (tile, tap): row = output channel 32 tile + (lane & 31), K byte j of lane half h = input channel 16 h + j
  std::vector<int8_t> frag8((size_t)FHEVC_FRAGI8_TOTAL * 16, 0);
  for (int lane = 0; lane < 64; ++lane) {
    const int r = lane & 31, h = lane >> 5;
    for (int j = 0; j < 16; ++j) {
      auto w2 = [&](int ky, int kx) { return b.w2[(r * 16 + j) * 9 + ky * 3 + kx]; };
      int8_t* f2 = &frag8[((size_t)FHEVC_FRAGI8_CONV2 + lane) * 16 + j];   // fragment s at f2[s * 64 * 16]
      for (int kx = 0; kx < 3; ++kx) f2[(size_t)kx * 1024] = w2(h, kx);
      f2[3 * 1024] = w2(2, 2 * h);              // [(2, 0) | (2, 2)]
      f2[4 * 1024] = h == 0 ? w2(2, 1) : 0;     // [(2, 1) | 0]
      f2[5 * 1024] = h == 1 ? w2(2, 1) : 0;     // [0 | (2, 1)]
      for (int t = 0; t < 2; ++t)
        for (int tap = 0; tap < 9; ++tap)
          frag8[((size_t)FHEVC_FRAGI8_CONV3 + (t * 9 + tap) * 64 + lane) * 16 + j] = b.w3[((32 * t + r) * 32 + 16 * h + j) * 9 + tap];
    }
  }
  // its biases: the activations travel as a - 128, so sum w a = sum w (a - 128) + 128 sum w (over ALL taps: the halo holds
  // a - 128 = -128, "activation 0", and meets the same correction)
  std::vector<int32_t> bias8(112, 0);
  long long bound[3] = { 0, 0, 0 };  // largest |accumulator| any input can produce: |b'| + 128 * sum |w|
  for (int oc = 0; oc < 32; ++oc) {
    int sw = 0, sa = 0;
    for (int i = 0; i < 16 * 9; ++i) { sw += b.w2[oc * 144 + i]; sa += std::abs((int)b.w2[oc * 144 + i]); }
    bias8[16 + oc] = rd32(b.b2, oc) + 128 * sw;
    bound[1] = std::max(bound[1], (long long)std::abs(bias8[16 + oc]) + 128LL * sa);
  }
  for (int oc = 0; oc < 64; ++oc) {
    int sw = 0, sa = 0;
    for (int i = 0; i < 32 * 9; ++i) { sw += b.w3[oc * 288 + i]; sa += std::abs((int)b.w3[oc * 288 + i]); }
    bias8[48 + oc] = rd32(b.b3, oc) + 128 * sw;
    bound[2] = std::max(bound[2], (long long)std::abs(bias8[48 + oc]) + 128LL * sa);
  }
  // the requant's form per layer (k_cnn.hip: requant4_i8); FHEVC_CNN_REQUANT=general keeps the general one (A/B, tests)
  for (int l = 1; l < 3; ++l) {
    const int sh = rd32(b.shift, l);
    new_mode[l] = c->knobs.requant_general ? 0 : (sh == 8 && bound[l] < (1LL << 23)) ? 2 : (sh <= 7 ? 1 : 0);
  }
  std::vector<float> bias(112);
  for (int i = 0; i < 16; ++i) {  // the kernel feeds conv1 the samples x, not x - 128: sum w (x - 128) + b = sum w x + (b - 128 sum w)
    int sw = 0;
    for (int t = 0; t < 9; ++t) sw += b.w1[i * 9 + t];
    if (std::abs(rd32(b.b1, i)) > 4194304) return fail(c, FHEVC_E_WEIGHTS, "|bias| > 2^22");
    bias[i] = (float)(rd32(b.b1, i) - 128 * sw);  // |.| < 2^22 + 128 * 9 * 127 < 2^23: exact, and conv1's sums stay below 2^24
  }
  for (int i = 0; i < 32; ++i) bias[16 + i] = (float)rd32(b.b2, i);
  for (int i = 0; i < 64; ++i) bias[48 + i] = (float)rd32(b.b3, i);
  for (int i = 16; i < 112; ++i) if (std::fabs(bias[i]) > 4194304.0f) return fail(c, FHEVC_E_WEIGHTS, "|bias| > 2^22");
  // FC heads on v_dot4_i32_i8: conv3's output is kept as a - 128 (signed bytes), so sum w a = sum w (a - 128) + 128 sum w and
  // the second term moves into the head biases (the 64-level weights act on the 2x2 sum pool: four positions each)
  std::vector<uint8_t> whead(4 * 4096 + 2 * 1024);
  for (int i = 0; i < 8192; ++i) whead[i] = (uint8_t)b.wh64[i];
  for (int i = 0; i < 8192; ++i) whead[8192 + i] = (uint8_t)b.wh32[i];
  for (int i = 0; i < 2048; ++i) whead[16384 + i] = (uint8_t)b.wh16[i];
  int32_t bhead[6 + 3 * 52] = { rd32(b.bh64, 0), rd32(b.bh64, 1), rd32(b.bh32, 0), rd32(b.bh32, 1), rd32(b.bh16, 0), rd32(b.bh16, 1) };
  for (int cls = 0; cls < 2; ++cls) {
    int s64 = 0, s32 = 0, s16 = 0;
    for (int i = 0; i < 4096; ++i) { s64 += b.wh64[cls * 4096 + i]; s32 += b.wh32[cls * 4096 + i]; }
    for (int i = 0; i < 1024; ++i) s16 += b.wh16[cls * 1024 + i];
    bhead[0 + cls] += 128 * 4 * s64;
    bhead[2 + cls] += 128 * s32;
    bhead[4 + cls] += 128 * s16;
  }
  for (int i = 0; i < 3 * 52; ++i) bhead[6 + i] = rd32(b.qp_bias, i);

  // everything above validated the blob without touching the context.  From here the base image is overwritten in place: if it is the one in
  // use, a HIP failure below leaves the context WITHOUT weights (predict then fails with FHEVC_E_STATE) rather than with a torn image; if a family
  // member is in use it stays in use until the last copy has succeeded
  if (c->have_weights && !c->family) c->have_weights = false;
  if (!c->d_frag) {
    HIP_TRY(c, hipMalloc(&c->d_frag, frag.size() * 2));
    HIP_TRY(c, hipMalloc(&c->d_bias, bias.size() * 4));
    HIP_TRY(c, hipMalloc(&c->d_whead, whead.size()));
    HIP_TRY(c, hipMalloc(&c->d_bhead, sizeof bhead));
    HIP_TRY(c, hipMalloc(&c->d_frag_i8, frag8.size()));
    HIP_TRY(c, hipMalloc(&c->d_bias_i8, bias8.size() * 4));
  }
  HIP_TRY(c, hipMemcpy(c->d_frag_i8, frag8.data(), frag8.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_bias_i8, bias8.data(), bias8.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_frag, frag.data(), frag.size() * 2, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_whead, whead.data(), whead.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_bhead, bhead, sizeof bhead, hipMemcpyHostToDevice));
  for (int l = 0; l < 3; ++l) { c->shift[l] = new_shift[l]; c->scale[l] = std::ldexp(1.0f, -new_shift[l]); c->requant_mode[l] = new_mode[l]; }
  c->family = false; c->fam_layers = false;   // the dispatch flips only now, with the image complete
  c->have_weights = true;
  return FHEVC_OK;
}

// FHW3 member without a fused kernel: one image per convolution for k_cnn_layers.inc + the activation tensors of a chunk of CTUs in HBM
int build_layers_image(fhevc_ctx* c, const uint8_t* blob, size_t bytes, int C1, int C2, int C3, int depth)
{
  const int C[3] = { C1, C2, C3 };
  for (int b = 0; b < 3; ++b) if (C[b] < 1 || C[b] > 128) return fail(c, FHEVC_E_WEIGHTS, "family widths must be 1..128");
  if (depth < 1 || depth > 3 || (C3 & 3)) return fail(c, FHEVC_E_WEIGHTS, "family members: 1..3 convolutions per block, last width a multiple of 4");
  size_t need = 24 + 36, ci = 1;
  for (int b = 0; b < 3; ++b) for (int j = 0; j < depth; ++j) { need += (size_t)C[b] * ci * 9 + 4 * (size_t)C[b]; ci = (size_t)C[b]; }
  need += (size_t)(2 * 64 + 2 * 64 + 2 * 16) * C3 + 24 + 3 * 52 * 4;
  if (bytes != need) return fail(c, FHEVC_E_WEIGHTS, "FHW3 blob has the wrong size");
  size_t off = 24;
  auto take = [&](size_t n) { const uint8_t* q = blob + off; off += n; return q; };
  auto i32at = [](const uint8_t* p, int i) { int32_t v; std::memcpy(&v, p + 4 * (size_t)i, 4); return v; };
  int32_t shift33[9];
  std::memcpy(shift33, take(36), 36);
  {  // validate the whole blob before the image in use is touched: a rejected blob leaves the context exactly as it was
    size_t o = off, cin_v = 1;
    for (int b = 0; b < 3; ++b)
      for (int j = 0; j < depth; ++j) {
        const size_t nw = (size_t)C[b] * cin_v * 9;
        for (size_t i = 0; i < nw; ++i) if (static_cast<int8_t>(blob[o + i]) == -128) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
        o += nw + 4 * (size_t)C[b];
        cin_v = (size_t)C[b];
        if (shift33[b * 3 + j] < 0 || shift33[b * 3 + j] > 14) return fail(c, FHEVC_E_WEIGHTS, "shift out of range (0..14)");
      }
    const size_t head_bytes[3] = { (size_t)2 * 64 * C3, (size_t)2 * 64 * C3, (size_t)2 * 16 * C3 };
    for (int hd = 0; hd < 3; ++hd) {
      for (size_t i = 0; i < head_bytes[hd]; ++i) if (blob[o + i] == 0x80) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
      o += head_bytes[hd] + 8;
    }
  }
  for (void* q : c->lw_bufs) (void)hipFree(q);
  c->lw_bufs.clear();
  if (c->family && c->fam_layers) c->have_weights = false;   // the layered image in use is gone: a HIP failure below leaves NO weights (never a torn image)
  FhevcLayersWeights lw = {};
  // a chunk of CTUs whose activations live in HBM at once: up to 16 pictures of 1080p (3.2 GB for 23/46/92 x 2), at least one picture row
  lw.chunk = std::min(c->num_ctus * std::max(1, c->cfg.max_frames), 8192);
  if (lw.chunk < 64) lw.chunk = 64;
  auto dev = [&](size_t n, int fill) -> void* { void* q = nullptr; if (hipMalloc(&q, n) != hipSuccess) return nullptr; c->lw_bufs.push_back(q); (void)hipMemset(q, fill, n); return q; };
  // two convolutions per block at padded widths 32 / 64 / 96 (the reference's 23 / 46 / 92 x 2): k_cnn_d2.inc keeps a CTU's activations in LDS -- the same weight
  // images, no activation tensors in HBM (FHEVC_FUSED_D2=0 / FHEVC_FAMILY_LAYERS keep the layer-by-layer path: tests, A/B)
  auto pad32 = [](int v) { return 32 * ((v + 31) / 32); };
  const bool d2 = depth == 2 && pad32(C1) == 32 && pad32(C2) == 64 && pad32(C3) == 96 && c->knobs.fused_d2 && !c->knobs.family_layers;
  if (!d2 && !(lw.in0 = static_cast<int8_t*>(dev((size_t)lw.chunk * 66 * 66, 0)))) return fail(c, FHEVC_E_HIP, "layer buffers");
  int H = 64, cin = 1, li = 0;
  for (int b = 0; b < 3; ++b)
    for (int j = 0; j < depth; ++j, ++li) {
      const int co = C[b], first = (b == 0 && j == 0);
      const int8_t* w = reinterpret_cast<const int8_t*>(take((size_t)co * cin * 9));
      const uint8_t* bp = take(4 * (size_t)co);
      for (size_t i = 0; i < (size_t)co * cin * 9; ++i) if (w[i] == -128) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
      const int sh = shift33[b * 3 + j];
      if (sh < 0 || sh > 14) return fail(c, FHEVC_E_WEIGHTS, "shift out of range (0..14)");
      const int kc = first ? 0 : (cin + 31) / 32, cout_pad = 32 * ((co + 31) / 32), MT = cout_pad / 32, NF = first ? 1 : kc * 9;
      // A fragments: lane (m, h) of (M tile, fragment): row m carries channel 32 mt + 16 m[2] + 4 m[4:3] + m[1:0] (so that a lane's 16 accumulators are 16
      // consecutive channels); byte jj of lane half h = input channel 32 kc + 16 h + jj at the fragment's tap (first layer: tap jj of the one channel, h = 0)
      std::vector<int8_t> frag((size_t)MT * NF * 64 * 16, 0);
      for (int mt = 0; mt < MT; ++mt)
        for (int f = 0; f < NF; ++f)
          for (int lane = 0; lane < 64; ++lane) {
            const int m = lane & 31, h = lane >> 5;
            const int oc = 32 * mt + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
            if (oc >= co) continue;
            for (int jj = 0; jj < 16; ++jj) {
              int8_t v = 0;
              if (first) { if (h == 0 && jj < 9) v = w[(size_t)oc * 9 + jj]; }
              else { const int ic = 32 * (f / 9) + 16 * h + jj; if (ic < cin) v = w[((size_t)oc * cin + ic) * 9 + f % 9]; }
              frag[(((size_t)mt * NF + f) * 64 + lane) * 16 + jj] = v;
            }
          }
      std::vector<int32_t> bias((size_t)cout_pad, 0);
      long long bound = 0;   // largest |accumulator| any input can produce: |b'| + 128 * sum |w| (inputs are a - 128 / centred samples: |x| <= 128)
      for (int oc = 0; oc < co; ++oc) {
        int sw = 0, sa = 0;
        for (int i = 0; i < cin * 9; ++i) { sw += w[(size_t)oc * cin * 9 + i]; sa += std::abs((int)w[(size_t)oc * cin * 9 + i]); }
        bias[(size_t)oc] = i32at(bp, oc) + (first ? 0 : 128 * sw);   // activations travel as a - 128; the first layer's input IS centred
        bound = std::max(bound, std::llabs((long long)bias[(size_t)oc]) + 128LL * sa);
      }
      const int pool = (j == depth - 1) && b < 2, Ho = pool ? H / 2 : H;
      FhevcLayer& L = lw.l[li];
      void* dfrag = dev(frag.size(), 0); void* dbias = dev(bias.size() * 4, 0);
      // the tensor's row pitch carries the padding its consumer's LDS image wants (the last map goes to the heads kernel: none)
      const int ni = li + 1, nb = ni / depth, nj = ni % depth;
      int in_pad = 0, swz = 0, out_pad = 0, unused = 0;
      if (!first) fhevc_layer_lds_image(kc, pool, H, &in_pad, &swz);
      if (ni < 3 * depth) fhevc_layer_lds_image(cout_pad / 32, (nj == depth - 1) && nb < 2, Ho, &out_pad, &unused);
      L.in_pad = in_pad; L.out_pad = out_pad; L.swz = swz;
      L.out = d2 ? nullptr : static_cast<int8_t*>(dev((size_t)lw.chunk * (Ho + 2) * ((size_t)(Ho + 2) * cout_pad + out_pad), 0x80));   // halo = "activation 0", written here once
      if (!dfrag || !dbias || (!d2 && !L.out)) return fail(c, FHEVC_E_HIP, "layer buffers");
      HIP_TRY(c, hipMemcpy(dfrag, frag.data(), frag.size(), hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(dbias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
      L.frag = static_cast<const uint4*>(dfrag); L.bias = static_cast<const int32_t*>(dbias);
      L.shift = sh; L.kc = kc; L.cout_pad = cout_pad; L.H = H; L.pool = pool;
      // requant4_i8's short forms: 1 packs to i16 with saturation BEFORE the shift (exact for shifts up to 7: a saturated value still clamps to 255 / 0 behind
      // it); 2 takes bytes 1-2 of the accumulator (shift 8, exact while the accumulator fits 24 bits)
      L.rq = sh <= 7 ? 1 : (sh == 8 && bound < (1LL << 23)) ? 2 : 0;
      H = Ho; cin = co;
    }
  lw.num_layers = li; lw.c3 = C3; lw.c3_pad = 32 * ((C3 + 31) / 32);
  const int8_t* wh64 = reinterpret_cast<const int8_t*>(take((size_t)2 * 64 * C3));
  const uint8_t* bh64p = take(8);
  const int8_t* wh32 = reinterpret_cast<const int8_t*>(take((size_t)2 * 64 * C3));
  const uint8_t* bh32p = take(8);
  const int8_t* wh16 = reinterpret_cast<const int8_t*>(take((size_t)2 * 16 * C3));
  const uint8_t* bh16p = take(8);
  const uint8_t* qpb = take(3 * 52 * 4);
  // rows of c3_pad bytes (zeros behind the C3 weights): the heads kernel reads activations and weights 16 bytes at a time
  const size_t cp = lw.c3_pad;
  std::vector<uint8_t> whead((size_t)(2 * 64 + 2 * 64 + 2 * 16) * cp, 0);
  for (int r = 0; r < 2 * 64; ++r) std::memcpy(whead.data() + (size_t)r * cp, wh64 + (size_t)r * C3, (size_t)C3);
  for (int r = 0; r < 2 * 64; ++r) std::memcpy(whead.data() + (size_t)(2 * 64 + r) * cp, wh32 + (size_t)r * C3, (size_t)C3);
  for (int r = 0; r < 2 * 16; ++r) std::memcpy(whead.data() + (size_t)(4 * 64 + r) * cp, wh16 + (size_t)r * C3, (size_t)C3);
  for (uint8_t v : whead) if (v == 0x80) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
  int32_t bhead[6 + 3 * 52] = { i32at(bh64p, 0), i32at(bh64p, 1), i32at(bh32p, 0), i32at(bh32p, 1), i32at(bh16p, 0), i32at(bh16p, 1) };
  for (int cls = 0; cls < 2; ++cls) {
    int s64 = 0, s32 = 0, s16 = 0;
    for (int i = 0; i < 64 * C3; ++i) { s64 += wh64[(size_t)cls * 64 * C3 + i]; s32 += wh32[(size_t)cls * 64 * C3 + i]; }
    for (int i = 0; i < 16 * C3; ++i) s16 += wh16[(size_t)cls * 16 * C3 + i];
    bhead[0 + cls] += 128 * 4 * s64; bhead[2 + cls] += 128 * s32; bhead[4 + cls] += 128 * s16;   // the last map travels as a - 128
  }
  for (int i = 0; i < 3 * 52; ++i) bhead[6 + i] = i32at(qpb, i);
  void* dwh = dev(whead.size(), 0); void* dbh = dev(sizeof bhead, 0);
  if (!dwh || !dbh) return fail(c, FHEVC_E_HIP, "layer buffers");
  HIP_TRY(c, hipMemcpy(dwh, whead.data(), whead.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(dbh, bhead, sizeof bhead, hipMemcpyHostToDevice));
  lw.whead = static_cast<const uint8_t*>(dwh); lw.bhead = static_cast<const int32_t*>(dbh);
  if (d2 && !fhevc_cnn_d2_supported(lw)) return fail(c, FHEVC_E_STATE, "layer images do not match the fused two-convolution kernel");
  lw.d2_short = d2 && !c->knobs.d2_requant_general && lw.l[0].rq == 1;
  for (int i = 1; i < 6 && lw.d2_short; ++i) lw.d2_short = lw.l[i].rq == 2;
  c->lw = lw;
  c->fam_c[0] = C1; c->fam_c[1] = C2; c->fam_c[2] = C3;
  c->family = true; c->fam_layers = true; c->fam_d2 = d2; c->have_weights = true;
  return FHEVC_OK;
}

// FHW3: a member of the reference's Bayesian-optimisation network family (fasthevc_amd/weights.py: family_fields); this round the
// kernel runs the members with one convolution per block whose widths it is instantiated for (fhevc_cnn_family_supported)
int build_family_image(fhevc_ctx* c, const uint8_t* blob, size_t bytes)
{
  if (bytes < 24) return fail(c, FHEVC_E_WEIGHTS, "FHW3 blob too short");
  uint32_t ver;
  int32_t hdr[4];
  std::memcpy(&ver, blob + 4, 4);
  std::memcpy(hdr, blob + 8, 16);
  const int C1 = hdr[0], C2 = hdr[1], C3 = hdr[2], depth = hdr[3];
  if (ver != 1) return fail(c, FHEVC_E_WEIGHTS, "unsupported FHW3 version");
  if (depth != 1 || !fhevc_cnn_family_supported(C1, C2, C3) || c->knobs.family_layers)   // (the knob: the generic path for a fused member too)
    return build_layers_image(c, blob, bytes, C1, C2, C3, depth);
  const size_t need = 24 + 36 + (size_t)C1 * 9 + 4 * (size_t)C1 + (size_t)C2 * C1 * 9 + 4 * (size_t)C2 + (size_t)C3 * C2 * 9 + 4 * (size_t)C3 +
                      (size_t)(2 * 64 + 2 * 64 + 2 * 16) * C3 + 24 + 3 * 52 * 4;
  if (bytes != need) return fail(c, FHEVC_E_WEIGHTS, "FHW3 blob has the wrong size");
  size_t off = 24;
  auto take = [&](size_t n) { const uint8_t* q = blob + off; off += n; return q; };
  int32_t shift33[9];
  std::memcpy(shift33, take(36), 36);
  const int sh[3] = { shift33[0], shift33[3], shift33[6] };
  for (int l = 0; l < 3; ++l) if (sh[l] < 0 || sh[l] > 14) return fail(c, FHEVC_E_WEIGHTS, "shift out of range (0..14)");
  const int8_t* w1 = reinterpret_cast<const int8_t*>(take((size_t)C1 * 9));
  const uint8_t* b1p = take(4 * (size_t)C1);
  const int8_t* w2 = reinterpret_cast<const int8_t*>(take((size_t)C2 * C1 * 9));
  const uint8_t* b2p = take(4 * (size_t)C2);
  const int8_t* w3 = reinterpret_cast<const int8_t*>(take((size_t)C3 * C2 * 9));
  const uint8_t* b3p = take(4 * (size_t)C3);
  const int8_t* wh64 = reinterpret_cast<const int8_t*>(take((size_t)2 * 64 * C3));
  const uint8_t* bh64p = take(8);
  const int8_t* wh32 = reinterpret_cast<const int8_t*>(take((size_t)2 * 64 * C3));
  const uint8_t* bh32p = take(8);
  const int8_t* wh16 = reinterpret_cast<const int8_t*>(take((size_t)2 * 16 * C3));
  const uint8_t* bh16p = take(8);
  const uint8_t* qpb = take(3 * 52 * 4);
  auto i32at = [](const uint8_t* p, int i) { int32_t v; std::memcpy(&v, p + 4 * (size_t)i, 4); return v; };
  for (const int8_t* p = w1; p < reinterpret_cast<const int8_t*>(bh16p); ++p) (void)p;
  const int G1 = C1 / 16, K1 = (C1 + 31) / 32, M2 = C2 / 32, K2 = C2 / 32, M3 = C3 / 32;
  // conv1: per group of 16 filters the two fragments of the base network (rows = filter x 2x2 pre-pool position, K = 4x4 window)
  std::vector<uint16_t> frag1((size_t)G1 * 2 * 64 * 8, 0);
  std::vector<float> bias1((size_t)C1);
  for (int g = 0; g < G1; ++g)
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      for (int j = 0; j < 8; ++j)
        for (int jm = 0; jm < 2; ++jm) {
          const int ch = 16 * g + (r & 3) + 4 * ((r >> 3) & 1) + 8 * ((r >> 2) & 1), py = (r >> 4) & 1, px = jm;
          const int wc = 2 * h + ((j >> 1) & 1), wr = 2 * (j >> 2) + (j & 1);
          const int ky = wr - py, kx = wc - px;
          if (ky >= 0 && ky <= 2 && kx >= 0 && kx <= 2) {
            if (w1[ch * 9 + ky * 3 + kx] == -128) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
            const float f = std::ldexp((float)w1[ch * 9 + ky * 3 + kx], -sh[0]);
            uint32_t u;
            std::memcpy(&u, &f, 4);
            frag1[(((size_t)(2 * g + jm) * 64) + lane) * 8 + j] = (uint16_t)(u >> 16);
          }
        }
    }
  for (int i = 0; i < C1; ++i) {
    int sw = 0;
    for (int t = 0; t < 9; ++t) sw += w1[i * 9 + t];
    if (std::abs(i32at(b1p, i)) > 4194304) return fail(c, FHEVC_E_WEIGHTS, "|bias| > 2^22");
    bias1[(size_t)i] = std::ldexp((float)(i32at(b1p, i) - 128 * sw), -sh[0]);
  }
  // conv2 / conv3: fragment (M tile, K chunk, tap): row = output channel 32 mt + (lane & 31), byte j of lane half h = input channel 32 k + 16 h + j
  auto build = [&](const int8_t* w, int CI, int M, int K, std::vector<int8_t>& frag) {
    frag.assign((size_t)M * K * 9 * 64 * 16, 0);
    for (int mt = 0; mt < M; ++mt)
      for (int k = 0; k < K; ++k)
        for (int tap = 0; tap < 9; ++tap)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 16; ++j) {
              const int oc = 32 * mt + (lane & 31), ic = 32 * k + 16 * (lane >> 5) + j;
              if (ic < CI) frag[((((size_t)mt * K + k) * 9 + tap) * 64 + lane) * 16 + j] = w[((size_t)oc * CI + ic) * 9 + tap];
            }
  };
  std::vector<int8_t> frag2, frag3;
  build(w2, C1, M2, K1, frag2);
  build(w3, C2, M3, K2, frag3);
  std::vector<int32_t> bias8((size_t)C2 + C3);
  for (int oc = 0; oc < C2; ++oc) {
    int sw = 0;
    for (int i = 0; i < C1 * 9; ++i) sw += w2[(size_t)oc * C1 * 9 + i];
    bias8[(size_t)oc] = i32at(b2p, oc) + 128 * sw;
  }
  for (int oc = 0; oc < C3; ++oc) {
    int sw = 0;
    for (int i = 0; i < C2 * 9; ++i) sw += w3[(size_t)oc * C2 * 9 + i];
    bias8[(size_t)C2 + oc] = i32at(b3p, oc) + 128 * sw;
  }
  std::vector<uint8_t> whead((size_t)(2 * 64 + 2 * 64 + 2 * 16) * C3);
  std::memcpy(whead.data(), wh64, (size_t)2 * 64 * C3);
  std::memcpy(whead.data() + (size_t)2 * 64 * C3, wh32, (size_t)2 * 64 * C3);
  std::memcpy(whead.data() + (size_t)4 * 64 * C3, wh16, (size_t)2 * 16 * C3);
  // the MFMA image of the two smaller heads: [position j = (py, px) of a 16x16 block][chunk of 64 channels][column][64 B]
  const int KC = C3 / 64;
  std::vector<uint8_t> headm((size_t)16 * KC * 16 * 64, 0);
  for (int j = 0; j < 16; ++j)
    for (int kc = 0; kc < KC; ++kc)
      for (int n = 0; n < 10; ++n) {
        const int8_t* src;
        if (n < 2) src = wh16 + ((size_t)n * 16 + j) * C3 + 64 * kc;
        else {
          const int sub = (n - 2) >> 1, cls = n & 1, py = j >> 2, px = j & 3;
          src = wh32 + ((size_t)cls * 64 + ((sub >> 1) * 4 + py) * 8 + (sub & 1) * 4 + px) * C3 + 64 * kc;
        }
        std::memcpy(headm.data() + (((size_t)j * KC + kc) * 16 + n) * 64, src, 64);
      }
  int32_t bhead[6 + 3 * 52] = { i32at(bh64p, 0), i32at(bh64p, 1), i32at(bh32p, 0), i32at(bh32p, 1), i32at(bh16p, 0), i32at(bh16p, 1) };
  for (int cls = 0; cls < 2; ++cls) {
    int s64 = 0, s32 = 0, s16 = 0;
    for (int i = 0; i < 64 * C3; ++i) { s64 += wh64[(size_t)cls * 64 * C3 + i]; s32 += wh32[(size_t)cls * 64 * C3 + i]; }
    for (int i = 0; i < 16 * C3; ++i) s16 += wh16[(size_t)cls * 16 * C3 + i];
    bhead[0 + cls] += 128 * 4 * s64;   // conv3's map travels as a - 128; the 64-level weights act on four positions each
    bhead[2 + cls] += 128 * s32;
    bhead[4 + cls] += 128 * s16;
  }
  for (int i = 0; i < 3 * 52; ++i) bhead[6 + i] = i32at(qpb, i);
  // validated; the fused member's image is replaced now: if it is the one in use, a HIP failure below leaves NO weights
  if (c->family && !c->fam_layers) c->have_weights = false;
  (void)hipFree(c->f_frag1); (void)hipFree(c->f_bias1); (void)hipFree(c->f_frag2); (void)hipFree(c->f_frag3); (void)hipFree(c->f_bias_i8); (void)hipFree(c->f_whead); (void)hipFree(c->f_headm); (void)hipFree(c->f_bhead);
  c->f_frag1 = nullptr; c->f_headm = nullptr; c->f_bias1 = nullptr; c->f_frag2 = nullptr; c->f_frag3 = nullptr; c->f_bias_i8 = nullptr; c->f_whead = nullptr; c->f_bhead = nullptr;
  HIP_TRY(c, hipMalloc(&c->f_frag1, frag1.size() * 2)); HIP_TRY(c, hipMalloc(&c->f_bias1, bias1.size() * 4));
  HIP_TRY(c, hipMalloc(&c->f_frag2, frag2.size())); HIP_TRY(c, hipMalloc(&c->f_frag3, frag3.size()));
  HIP_TRY(c, hipMalloc(&c->f_bias_i8, bias8.size() * 4)); HIP_TRY(c, hipMalloc(&c->f_whead, whead.size())); HIP_TRY(c, hipMalloc(&c->f_headm, headm.size())); HIP_TRY(c, hipMalloc(&c->f_bhead, sizeof bhead));
  HIP_TRY(c, hipMemcpy(c->f_frag1, frag1.data(), frag1.size() * 2, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_bias1, bias1.data(), bias1.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_frag2, frag2.data(), frag2.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_frag3, frag3.data(), frag3.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_bias_i8, bias8.data(), bias8.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_whead, whead.data(), whead.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_headm, headm.data(), headm.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->f_bhead, bhead, sizeof bhead, hipMemcpyHostToDevice));
  c->fam_c[0] = C1; c->fam_c[1] = C2; c->fam_c[2] = C3;
  c->shift[0] = sh[0]; c->shift[1] = sh[1]; c->shift[2] = sh[2];
  c->family = true; c->fam_layers = false;
  c->have_weights = true;
  return FHEVC_OK;
}

FhevcFamilyWeights family_weights(const fhevc_ctx* c)
{
  FhevcFamilyWeights w;
  w.c[0] = c->fam_c[0]; w.c[1] = c->fam_c[1]; w.c[2] = c->fam_c[2];
  w.frag1 = c->f_frag1; w.bias1 = c->f_bias1; w.frag2 = c->f_frag2; w.frag3 = c->f_frag3; w.bias_i8 = c->f_bias_i8; w.whead = c->f_whead; w.headm = c->f_headm; w.bhead = c->f_bhead;
  w.shift[0] = c->shift[0]; w.shift[1] = c->shift[1]; w.shift[2] = c->shift[2];
  return w;
}

FhevcCnnWeights cnn_weights(const fhevc_ctx* c)
{
  FhevcCnnWeights w;
  w.frag = c->d_frag; w.bias = c->d_bias; w.whead = c->d_whead; w.bhead = c->d_bhead;
  w.scale[0] = c->scale[0]; w.scale[1] = c->scale[1]; w.scale[2] = c->scale[2];
  w.frag_i8 = c->d_frag_i8; w.bias_i8 = c->d_bias_i8;
  w.shift[0] = c->shift[0]; w.shift[1] = c->shift[1]; w.shift[2] = c->shift[2];
  w.requant_mode[0] = 0; w.requant_mode[1] = c->requant_mode[1]; w.requant_mode[2] = c->requant_mode[2];
  w.i8 = c->cnn_i8 ? 1 : 0;
  w.had_valu = c->had_valu ? 1 : 0;
  w.pipe = c->cnn_pipe ? 1 : 0;
  return w;
}

void time_begin(fhevc_ctx* c, hipStream_t s, int which)
{
  if (!c->timing) return;
  TimedLaunch t;
  if (!c->pool.empty()) { t.start = c->pool.back().first; t.stop = c->pool.back().second; c->pool.pop_back(); }
  else { (void)hipEventCreate(&t.start); (void)hipEventCreate(&t.stop); }
  t.which = which;
  (void)hipEventRecord(t.start, s);
  c->pending.push_back(t);
}
void time_end(fhevc_ctx* c, hipStream_t s)
{
  if (!c->timing) return;
  (void)hipEventRecord(c->pending.back().stop, s);
}
void time_resolve(fhevc_ctx* c)
{
  for (auto& t : c->pending) {
    float ms = 0;
    if (hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
      c->sum_ms[t.which] += ms;
      c->launches[t.which] += 1;
      if (t.which == 0) c->stats.last_cnn_ms = ms;
      if (t.which == 1) c->stats.last_hadamard_ms = ms;
      if (t.which == 2) c->stats.last_first_pass_ms = ms;
    }
    c->pool.emplace_back(t.start, t.stop);
  }
  c->pending.clear();
}

FhevcFrames frames_of(const fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride, long long frame_stride,
                      int num_frames, int row_begin, int row_end, int qp = 32)
{
  FhevcFrames f;
  f.luma = d_luma; f.sample_bytes = sample_bytes; f.stride = stride; f.frame_stride = frame_stride;
  f.width = c->cfg.width; f.height = c->cfg.height; f.bit_depth = c->cfg.bit_depth;
  f.ctus_x = c->ctus_x; f.ctus_y = c->ctus_y; f.num_frames = num_frames; f.row_begin = row_begin; f.row_end = row_end;
  f.qp = qp < 0 ? 0 : (qp > 51 ? 51 : qp);
  return f;
}

}  // namespace

extern "C" {

const char* fhevc_version(void) { return "fasthevc_amd 0.1.0 (gfx950)"; }

const char* fhevc_last_error(fhevc_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int fhevc_band(int ctu_rows, int rank, int world, int* begin, int* end)
{
  if (ctu_rows < 0 || world <= 0 || rank < 0 || rank >= world || !begin || !end) return FHEVC_E_INVALID;
  *begin = (int)(((long long)rank * ctu_rows) / world);
  *end = (int)(((long long)(rank + 1) * ctu_rows) / world);
  return FHEVC_OK;
}

int fhevc_create(fhevc_ctx** out, const fhevc_cfg* cfg)
{
  if (!out || !cfg) return FHEVC_E_INVALID;
  *out = nullptr;
  if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 16384 || cfg->height > 16384) return FHEVC_E_INVALID;
  if (cfg->ctu_size != FHEVC_CTU || cfg->max_depth != 3) return FHEVC_E_INVALID;
  if (cfg->bit_depth < 8 || cfg->bit_depth > 12) return FHEVC_E_INVALID;
  if (cfg->backend != FHEVC_BACKEND_HIP) return FHEVC_E_INVALID;  // there is no CPU backend
  if (cfg->num_devices > 16 || (cfg->num_devices > 1 && !cfg->device_ids)) return FHEVC_E_INVALID;
  fhevc_ctx* c = new (std::nothrow) fhevc_ctx();
  if (!c) return FHEVC_E_NOMEM;
  c->cfg = *cfg;
  c->cfg.weights_path = nullptr;
  c->cfg.device_ids = nullptr;
  if (c->cfg.max_frames < 1) c->cfg.max_frames = 1;
  c->device = (cfg->device_ids && cfg->num_devices >= 1) ? cfg->device_ids[0] : 0;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || c->device >= ndev) { delete c; return FHEVC_E_NO_DEVICE; }
  hipDeviceProp_t prop;
  if (hipSetDevice(c->device) != hipSuccess || hipGetDeviceProperties(&prop, c->device) != hipSuccess) { delete c; return FHEVC_E_NO_DEVICE; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete c; return FHEVC_E_NO_DEVICE; }  // code object is gfx950-only
  c->num_cus = prop.multiProcessorCount;
  c->knobs = fhevc_read_knobs();   // every environment switch of the library is read here, once per context
  if (const char* fz = std::getenv("FHEVC_FUSE_HADAMARD")) c->fuse_hadamard = fz[0] != '0';
  if (const char* pp = std::getenv("FHEVC_CNN_PIPE")) c->cnn_pipe = pp[0] == '1';
  if (const char* hf = std::getenv("FHEVC_HADAMARD_FORM")) c->had_valu = std::strcmp(hf, "mfma") != 0;
  // arithmetic of conv2 / conv3 in the depth kernel: "f16" (16-bit MFMAs) or "i8" (v_mfma_i32_32x32x32_i8); both are exact
  if (const char* ar = std::getenv("FHEVC_CNN_ARITH")) c->cnn_i8 = std::strcmp(ar, "f16") != 0;
  c->ctus_x = (cfg->width + 63) / 64;
  c->ctus_y = (cfg->height + 63) / 64;
  c->num_ctus = c->ctus_x * c->ctus_y;
  c->dev_stride = c->ctus_x * 64;
  // A BLOCKING stream (hipStreamDefault): it is ordered after everything issued earlier on the legacy default stream and before
  // everything issued later on it, so a caller that works on the default stream (stream 0, torch's default) and passes
  // stream = NULL needs no extra synchronisation; callers on their own non-blocking streams pass that stream explicitly
  if (hipStreamCreateWithFlags(&c->stream, hipStreamDefault) != hipSuccess) { delete c; return FHEVC_E_NO_DEVICE; }
  if (fhevc_cnn_prepare_device() != hipSuccess) { (void)hipStreamDestroy(c->stream); delete c; return FHEVC_E_NO_DEVICE; }  // per device, not per process
  bool ok = true;
  ok &= hipMalloc(&c->d_luma, (size_t)c->dev_stride * c->ctus_y * 64 * sizeof(int16_t)) == hipSuccess;
  ok &= hipMalloc(&c->d_depth, (size_t)c->num_ctus * 256) == hipSuccess;
  ok &= hipMalloc(&c->d_had, (size_t)c->num_ctus * 4) == hipSuccess;
  ok &= hipMalloc(&c->d_nodes, (size_t)c->num_ctus * FHEVC_NODES_PER_CTU * sizeof(FhevcNodeCost)) == hipSuccess;
  ok &= hipMalloc(&c->d_satd, 2 * 64 * 64 * sizeof(int16_t)) == hipSuccess;
  ok &= hipMalloc(&c->d_satd_out, 4) == hipSuccess;
  for (auto& e : c->ev) ok &= hipEventCreate(&e) == hipSuccess;
  if (!ok) { fhevc_destroy(c); return FHEVC_E_NOMEM; }
  if (hipMemset(c->d_luma, 0, (size_t)c->dev_stride * c->ctus_y * 64 * sizeof(int16_t)) != hipSuccess) { fhevc_destroy(c); return FHEVC_E_HIP; }
  if (cfg->weights_path) {
    FILE* fp = std::fopen(cfg->weights_path, "rb");
    if (!fp) { fhevc_destroy(c); return FHEVC_E_WEIGHTS; }
    std::vector<uint8_t> buf((size_t)4 << 20);   // FHW1 is 42 KB, the largest family member this build runs (FHW3, 32 / 64 / 128) 133 KB
    const size_t n = std::fread(buf.data(), 1, buf.size(), fp);
    std::fclose(fp);
    const int rc = fhevc_set_weights(c, buf.data(), n);
    if (rc != FHEVC_OK) { fhevc_destroy(c); return rc; }
  }
  // the other devices of a multi-device context: one single-device context each (the same picture geometry and weights).  A device
  // that cannot be brought up is left out -- the context works with the ones that can -- and counted in stats.devices_failed
  c->stats.devices = 1;
  for (int d = 1; d < cfg->num_devices; ++d) {
    fhevc_cfg sub = *cfg;
    int id = cfg->device_ids[d];
    sub.num_devices = 1;
    sub.device_ids = &id;
    fhevc_ctx* peer = nullptr;
    if (fhevc_create(&peer, &sub) == FHEVC_OK) { c->peers.push_back(peer); c->stats.devices++; }
    else c->stats.devices_failed++;
  }
  if (const char* ft = std::getenv("FHEVC_TEST_FAIL_DEVICE")) c->fail_peer_for_test = std::atoi(ft);
  (void)hipSetDevice(c->device);
  *out = c;
  return FHEVC_OK;
}

void fhevc_destroy(fhevc_ctx* c)
{
  if (!c) return;
  for (fhevc_ctx* peer : c->peers) fhevc_destroy(peer);
  c->peers.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  time_resolve(c);
  for (auto& p : c->pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
  if (c->lw_done) (void)hipEventDestroy(c->lw_done);
  if (c->mvtab_used) (void)hipEventDestroy(c->mvtab_used);
  (void)hipFree(c->d_frag); (void)hipFree(c->d_bias); (void)hipFree(c->d_whead); (void)hipFree(c->d_bhead);
  (void)hipFree(c->d_frag_i8); (void)hipFree(c->d_bias_i8);
  (void)hipFree(c->f_frag1); (void)hipFree(c->f_bias1); (void)hipFree(c->f_frag2); (void)hipFree(c->f_frag3); (void)hipFree(c->f_bias_i8); (void)hipFree(c->f_whead); (void)hipFree(c->f_headm); (void)hipFree(c->f_bhead);
  (void)hipFree(c->d_luma); (void)hipFree(c->d_depth); (void)hipFree(c->d_had); (void)hipFree(c->d_nodes); (void)hipFree(c->d_satd); (void)hipFree(c->d_satd_out); (void)hipFree(c->d_act); (void)hipFree(c->d_depth_max); (void)hipFree(c->d_pair); (void)hipFree(c->d_motion); (void)hipFree(c->d_mvtab); (void)hipFree(c->d_cand_all); (void)hipFree(c->d_cand);
  for (void* q : c->lw_bufs) (void)hipFree(q);
  for (auto& sl : c->slot) {  // the host-batch ring of fhevc_predict_frames: stream, device buffers, pinned staging
    if (sl.st) { (void)hipStreamSynchronize(sl.st); (void)hipStreamDestroy(sl.st); }
    (void)hipFree(sl.d_in); (void)hipFree(sl.d_depth); (void)hipFree(sl.d_had);
    if (sl.h_in) (void)hipHostFree(sl.h_in);
    if (sl.h_depth) (void)hipHostFree(sl.h_depth);
    if (sl.h_had) (void)hipHostFree(sl.h_had);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int fhevc_set_weights(fhevc_ctx* c, const void* blob, size_t bytes)
{
  if (!c || !blob) return FHEVC_E_INVALID;
  // The primary validates and builds first (a rejected blob changes nothing anywhere); only then do the peers get the blob.  A peer that fails after
  // the primary succeeded would leave devices with different weights: the whole context then reports "weights not set" until a blob is accepted by all
  auto push_to_peers = [&]() {
    for (fhevc_ctx* peer : c->peers) {
      const int rc = fhevc_set_weights(peer, blob, bytes);
      if (rc != FHEVC_OK) {
        c->have_weights = false;
        for (fhevc_ctx* p2 : c->peers) p2->have_weights = false;
        return fail(c, rc, "weights rejected by a peer device");
      }
    }
    (void)hipSetDevice(c->device);
    return (int)FHEVC_OK;
  };
  (void)hipSetDevice(c->device);
  if (bytes >= 4 && std::memcmp(blob, "FHW3", 4) == 0) {  // a member of the reference's Bayesian-optimisation network family
    const int rc = build_family_image(c, static_cast<const uint8_t*>(blob), bytes);
    return rc != FHEVC_OK ? rc : push_to_peers();
  }
  BlobView v;
  std::vector<uint8_t> copy;
  if (!parse_blob(static_cast<const uint8_t*>(blob), bytes, v, copy)) return fail(c, FHEVC_E_WEIGHTS, "not an FHW1 blob");
  const struct { const int8_t* p; size_t n; } i8s[6] = { { v.w1, 144 }, { v.w2, 4608 }, { v.w3, 18432 }, { v.wh64, 8192 }, { v.wh32, 8192 }, { v.wh16, 2048 } };
  for (const auto& a : i8s)
    for (size_t i = 0; i < a.n; ++i) if (a.p[i] == -128) return fail(c, FHEVC_E_WEIGHTS, "weight -128 not allowed");
  const int rc = build_weight_image(c, v);
  return rc != FHEVC_OK ? rc : push_to_peers();
}

int fhevc_enable_kernel_timing(fhevc_ctx* c, int on)
{
  if (!c) return FHEVC_E_INVALID;
  if (!on) time_resolve(c);
  c->timing = on != 0;
  return FHEVC_OK;
}

int fhevc_kernel_timing(fhevc_ctx* c, int which, int reset, double* avg_ms, uint64_t* launches)
{
  if (!c || which < 0 || which > 4) return FHEVC_E_INVALID;
  time_resolve(c);
  if (avg_ms) *avg_ms = c->launches[which] ? c->sum_ms[which] / (double)c->launches[which] : 0.0;
  if (launches) *launches = c->launches[which];
  if (reset) { c->sum_ms[which] = 0; c->launches[which] = 0; }
  return FHEVC_OK;
}

int fhevc_predict_frames_device(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples,
                                long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end, int qp,
                                uint8_t* d_depth_map, int32_t* d_hadamard, int32_t* d_logits, uint32_t* d_flags, void* stream)
{
  return fhevc_predict_frames_device_range(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, ctu_row_begin,
                                           ctu_row_end, qp, 0, 0, d_depth_map, nullptr, d_hadamard, d_logits, d_flags, stream);
}

int fhevc_predict_frames_device_range(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples,
                                      long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end, int qp,
                                      int margin_split, int margin_stop, uint8_t* d_depth_map, uint8_t* d_depth_max, int32_t* d_hadamard, int32_t* d_logits,
                                      uint32_t* d_flags, void* stream)
{
  if (!c || !d_luma || !d_depth_map) return FHEVC_E_INVALID;
  if (!c->have_weights) return fail(c, FHEVC_E_STATE, "weights not set");
  if ((sample_bytes != 1 && sample_bytes != 2) || stride_samples < c->cfg.width || num_frames < 1) return fail(c, FHEVC_E_INVALID, "bad frame layout");
  if (sample_bytes == 1 && c->cfg.bit_depth != 8) return fail(c, FHEVC_E_INVALID, "uint8 samples need bit_depth 8");
  if (ctu_row_begin < 0 || ctu_row_end > c->ctus_y || ctu_row_begin > ctu_row_end) return fail(c, FHEVC_E_INVALID, "bad CTU-row band");
  if (margin_split < 0 || margin_split > (1 << 30) || margin_stop < 0 || margin_stop > (1 << 30)) return fail(c, FHEVC_E_INVALID, "bad decision margin");
  if (num_frames > 1 && frame_stride_samples < (long long)stride_samples * (c->cfg.height - 1) + c->cfg.width) return fail(c, FHEVC_E_INVALID, "frames overlap");
  if (ctu_row_begin == ctu_row_end) return FHEVC_OK;
  (void)hipSetDevice(c->device);
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
  const FhevcFrames fr = frames_of(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, ctu_row_begin, ctu_row_end, qp);
  // the source Hadamard rides on the depth kernel's own pass over the frame wherever the layout allows the fused form
  // (aligned planes, widths that are multiples of 16, up to 10 bit); otherwise it is its own HBM-bound launch
  const bool fuse = d_hadamard && c->fuse_hadamard && !c->family && fhevc_cnn_can_fuse_hadamard(fr);
  if (d_hadamard && !fuse) {
    time_begin(c, s, 1);
    HIP_TRY(c, fhevc_launch_src_hadamard(fr, d_hadamard, s));
    time_end(c, s);
    c->stats.kernels_launched++;
  }
  time_begin(c, s, 0);
  if (c->family && c->fam_layers && c->fam_d2) {
    HIP_TRY(c, fhevc_launch_cnn_d2(fr, c->lw, d_depth_map, d_logits, d_flags, d_depth_max, margin_split, margin_stop, c->num_cus, s));
  }
  else if (c->family && c->fam_layers) {
    if (!c->lw_done) HIP_TRY(c, hipEventCreateWithFlags(&c->lw_done, hipEventDisableTiming));
    if (c->lw_in_flight && c->lw_last_stream != s) HIP_TRY(c, hipStreamWaitEvent(s, c->lw_done, 0));   // the scratch tensors are still being read there
    const hipError_t le = fhevc_launch_cnn_layers(fr, c->lw, d_depth_map, d_logits, d_flags, d_depth_max, margin_split, margin_stop, c->num_cus, c->knobs, s);
    (void)hipEventRecord(c->lw_done, s);   // also after a failed launch: whatever did get queued on s still owns the scratch
    c->lw_last_stream = s; c->lw_in_flight = true;
    if (le != hipSuccess) return fail(c, FHEVC_E_HIP, "fhevc_launch_cnn_layers", le);
  }
  else if (c->family) HIP_TRY(c, fhevc_launch_cnn_family(fr, family_weights(c), d_depth_map, d_logits, d_flags, d_depth_max, margin_split, margin_stop, c->num_cus, s));
  else HIP_TRY(c, fhevc_launch_cnn(fr, cnn_weights(c), d_depth_map, fuse ? d_hadamard : nullptr, d_logits, d_flags, d_depth_max, margin_split, margin_stop, c->num_cus, c->knobs, s));
  time_end(c, s);
  c->stats.kernels_launched++;
  c->stats.frames += (uint64_t)num_frames;
  c->stats.ctus += (uint64_t)num_frames * (uint64_t)(ctu_row_end - ctu_row_begin) * (uint64_t)c->ctus_x;
  return FHEVC_OK;
}

int fhevc_expand_depth_flags_device(fhevc_ctx* c, const uint32_t* d_flags, int num_frames, uint8_t* d_depth_map, void* stream)
{
  if (!c || !d_flags || !d_depth_map || num_frames < 1) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
  const FhevcFrames fr = frames_of(c, nullptr, 1, c->cfg.width, 0, num_frames, 0, c->ctus_y);
  HIP_TRY(c, fhevc_launch_expand_flags(fr, d_flags, d_depth_map, s));
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

// upload one host picture (Pel plane with stride) into the context's staging plane
static int upload_frame(fhevc_ctx* c, const int16_t* luma, int stride_samples)
{
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, hipMemcpy2DAsync(c->d_luma, (size_t)c->dev_stride * 2, luma, (size_t)stride_samples * 2,
                              (size_t)c->cfg.width * 2, (size_t)c->cfg.height, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  c->stats.bytes_h2d += (uint64_t)c->cfg.width * c->cfg.height * 2;
  return FHEVC_OK;
}

// One picture's CTU rows [rb, re) on ONE device: upload those rows, run the depth kernel over the band, bring the band's maps back
// into the caller's whole-picture buffers at the band's place.  depth_max (with its margins) is optional.  Synchronous.
static int predict_band_host(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, int rb, int re, int margin_split, int margin_stop,
                             uint8_t* depth_min, uint8_t* depth_max, int32_t* ctu_src_hadamard)
{
  if (rb >= re) return FHEVC_OK;
  if (!c->have_weights) return fail(c, FHEVC_E_STATE, "weights not set");
  (void)hipSetDevice(c->device);
  if (depth_max && !c->d_depth_max) HIP_TRY(c, hipMalloc(&c->d_depth_max, (size_t)c->num_ctus * 256));
  const int y0 = rb * 64, y1 = std::min(c->cfg.height, re * 64);
  const size_t band_ctus = (size_t)(re - rb) * c->ctus_x, first = (size_t)rb * c->ctus_x;
  HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
  HIP_TRY(c, hipMemcpy2DAsync(c->d_luma + (size_t)y0 * c->dev_stride, (size_t)c->dev_stride * 2, luma + (size_t)y0 * stride_samples, (size_t)stride_samples * 2,
                              (size_t)c->cfg.width * 2, (size_t)(y1 - y0), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
  c->stats.bytes_h2d += (uint64_t)c->cfg.width * (y1 - y0) * 2;
  // the kernel writes a band's results compactly from the start of its output buffers
  int rc = fhevc_predict_frames_device_range(c, c->d_luma, 2, c->dev_stride, 0, 1, rb, re, qp, margin_split, margin_stop, c->d_depth, depth_max ? c->d_depth_max : nullptr,
                                             ctu_src_hadamard ? c->d_had : nullptr, nullptr, nullptr, c->stream);
  if (rc != FHEVC_OK) return rc;
  HIP_TRY(c, hipEventRecord(c->ev[2], c->stream));
  HIP_TRY(c, hipMemcpyAsync(depth_min + first * 256, c->d_depth, band_ctus * 256, hipMemcpyDeviceToHost, c->stream));
  if (depth_max) HIP_TRY(c, hipMemcpyAsync(depth_max + first * 256, c->d_depth_max, band_ctus * 256, hipMemcpyDeviceToHost, c->stream));
  if (ctu_src_hadamard) HIP_TRY(c, hipMemcpyAsync(ctu_src_hadamard + first, c->d_had, band_ctus * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipEventRecord(c->ev[3], c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) c->stats.ms_h2d += ms;
  if (hipEventElapsedTime(&ms, c->ev[1], c->ev[2]) == hipSuccess) c->stats.ms_kernels += ms;
  if (hipEventElapsedTime(&ms, c->ev[2], c->ev[3]) == hipSuccess) c->stats.ms_d2h += ms;
  c->stats.bytes_d2h += (uint64_t)band_ctus * ((depth_max ? 512 : 256) + (ctu_src_hadamard ? 4 : 0));
  return FHEVC_OK;
}

// The devices of a context that still work: the primary first.  (A failed peer is destroyed and forgotten: drop_peer.)
static std::vector<fhevc_ctx*> live_devices(fhevc_ctx* c)
{
  std::vector<fhevc_ctx*> v{ c };
  v.insert(v.end(), c->peers.begin(), c->peers.end());
  return v;
}
static void drop_peer(fhevc_ctx* c, fhevc_ctx* peer, const char* what)
{
  for (size_t i = 0; i < c->peers.size(); ++i)
    if (c->peers[i] == peer) {
      c->err = std::string("device ") + std::to_string(peer->device) + " dropped (" + what + "): " + peer->err;
      fhevc_destroy(peer);
      c->peers.erase(c->peers.begin() + (long)i);
      c->stats.devices_failed++;
      c->stats.devices = 1 + c->peers.size();
      return;
    }
}
// run share(i, device_i) for every live device concurrently (one host thread per device beyond the caller's own); a share that fails
// on a PEER is redone on the primary and the peer is dropped: the call fails only if the primary itself fails
static int run_sharded(fhevc_ctx* c, const std::function<int(int, int, fhevc_ctx*)>& share)
{
  const std::vector<fhevc_ctx*> dev = live_devices(c);
  const int n = (int)dev.size();
  std::vector<int> rc((size_t)n, FHEVC_OK);
  std::vector<std::thread> th;
  for (int i = 1; i < n; ++i)
    th.emplace_back([&, i] {
      rc[(size_t)i] = (c->fail_peer_for_test == i) ? fail(dev[(size_t)i], FHEVC_E_HIP, "failure injected by FHEVC_TEST_FAIL_DEVICE") : share(i, n, dev[(size_t)i]);
    });
  rc[0] = share(0, n, c);
  for (auto& t : th) t.join();
  if (c->fail_peer_for_test >= 1) c->fail_peer_for_test = -1;  // once
  (void)hipSetDevice(c->device);
  if (rc[0] != FHEVC_OK) return rc[0];
  for (int i = 1; i < n; ++i) {
    // statistics of a multi-device context are the sums over its devices
    c->stats.frames += dev[(size_t)i]->stats.frames; c->stats.ctus += dev[(size_t)i]->stats.ctus;
    c->stats.bytes_h2d += dev[(size_t)i]->stats.bytes_h2d; c->stats.bytes_d2h += dev[(size_t)i]->stats.bytes_d2h;
    c->stats.kernels_launched += dev[(size_t)i]->stats.kernels_launched;
    dev[(size_t)i]->stats = fhevc_stats{};
    if (rc[(size_t)i] != FHEVC_OK) {
      const int redo = share(i, n, c);   // the failed device's share, on the primary
      drop_peer(c, dev[(size_t)i], "its share was redone on the primary device");
      if (redo != FHEVC_OK) return redo;
    }
  }
  return FHEVC_OK;
}

int fhevc_predict_frame(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, int slice_type,
                        uint8_t* depth_map, int32_t* ctu_src_hadamard)
{
  (void)slice_type;
  if (!c || !luma || !depth_map || stride_samples < c->cfg.width) return FHEVC_E_INVALID;
  if (!c->have_weights) return fail(c, FHEVC_E_STATE, "weights not set");
  if (c->peers.empty()) {
    const int rc = predict_band_host(c, luma, stride_samples, qp, 0, c->ctus_y, 0, 0, depth_map, nullptr, ctu_src_hadamard);
    // one picture = one frame in the statistics whatever the number of bands
    return rc;
  }
  // CTU-row bands over the devices (fhevc_band): rows [i * rows / n, (i + 1) * rows / n) on device i
  const uint64_t frames0 = c->stats.frames;
  const int rc = run_sharded(c, [&](int i, int n, fhevc_ctx* d) {
    int rb = 0, re = 0;
    (void)fhevc_band(c->ctus_y, i, n, &rb, &re);
    return predict_band_host(d, luma, stride_samples, qp, rb, re, 0, 0, depth_map, nullptr, ctu_src_hadamard);
  });
  c->stats.frames = frames0 + 1;
  return rc;
}

int fhevc_read_yuv_luma(const char* path, int file_width, int file_height, int file_bit_depth, int chroma_format, long long first_frame,
                        int num_frames, int dst_width, int dst_height, int internal_bit_depth, int dst_sample_bytes, void* dst,
                        long long dst_stride_samples, long long dst_frame_stride_samples)
{
  if (!path || !dst || file_width < 1 || file_height < 1 || num_frames < 0 || first_frame < 0) return FHEVC_E_INVALID;
  if (file_bit_depth < 8 || file_bit_depth > 16 || internal_bit_depth < file_bit_depth || internal_bit_depth > 12) return FHEVC_E_INVALID;
  if (dst_width < file_width || dst_height < file_height || dst_stride_samples < dst_width) return FHEVC_E_INVALID;
  if (dst_sample_bytes != 1 && dst_sample_bytes != 2) return FHEVC_E_INVALID;
  if (dst_sample_bytes == 1 && (file_bit_depth != 8 || internal_bit_depth != 8)) return FHEVC_E_INVALID;
  if (num_frames > 1 && dst_frame_stride_samples < dst_stride_samples * (dst_height - 1) + dst_width) return FHEVC_E_INVALID;
  const long long bps = file_bit_depth > 8 ? 2 : 1;
  long long chroma_samples;  // both chroma planes of the FILE's format
  const long long cw = (file_width + 1) / 2, chh = (file_height + 1) / 2;
  switch (chroma_format) {
    case 400: chroma_samples = 0; break;
    case 420: chroma_samples = 2 * cw * chh; break;
    case 422: chroma_samples = 2 * cw * file_height; break;
    case 444: chroma_samples = 2LL * file_width * file_height; break;
    default: return FHEVC_E_INVALID;
  }
  const long long luma_bytes = (long long)file_width * file_height * bps, frame_bytes = luma_bytes + chroma_samples * bps;
  FILE* fp = std::fopen(path, "rb");
  if (!fp) return FHEVC_E_STATE;
  const int shift = internal_bit_depth - file_bit_depth;
  std::vector<uint8_t> row8;
  int done = 0;
  for (; done < num_frames; ++done) {
    if (fseeko(fp, (off_t)((first_frame + done) * frame_bytes), SEEK_SET) != 0) break;
    bool ok = true;
    if (dst_sample_bytes == 1) {
      uint8_t* plane = static_cast<uint8_t*>(dst) + (size_t)done * (size_t)dst_frame_stride_samples;
      if (dst_width == file_width && dst_stride_samples == file_width) ok = std::fread(plane, 1, (size_t)luma_bytes, fp) == (size_t)luma_bytes;  // one read, file -> destination
      else
        for (int y = 0; y < file_height && ok; ++y) ok = std::fread(plane + (size_t)y * dst_stride_samples, 1, (size_t)file_width, fp) == (size_t)file_width;
      if (!ok) break;
      for (int y = 0; y < file_height; ++y) {
        uint8_t* r = plane + (size_t)y * dst_stride_samples;
        for (int x = file_width; x < dst_width; ++x) r[x] = r[file_width - 1];
      }
      for (int y = file_height; y < dst_height; ++y) std::memcpy(plane + (size_t)y * dst_stride_samples, plane + (size_t)(file_height - 1) * dst_stride_samples, (size_t)dst_width);
    } else {
      int16_t* plane = static_cast<int16_t*>(dst) + (size_t)done * (size_t)dst_frame_stride_samples;
      if (bps == 1) row8.resize((size_t)file_width);
      for (int y = 0; y < file_height && ok; ++y) {
        int16_t* r = plane + (size_t)y * dst_stride_samples;
        if (bps == 2) {  // two little-endian bytes per sample: the host is little-endian (x86-64), read them in place
          ok = std::fread(r, 2, (size_t)file_width, fp) == (size_t)file_width;
          if (shift) for (int x = 0; x < file_width; ++x) r[x] = (int16_t)(r[x] << shift);
        } else {
          ok = std::fread(row8.data(), 1, (size_t)file_width, fp) == (size_t)file_width;
          for (int x = 0; x < file_width; ++x) r[x] = (int16_t)((int)row8[(size_t)x] << shift);
        }
        for (int x = file_width; x < dst_width; ++x) r[x] = r[file_width - 1];
      }
      if (!ok) break;
      for (int y = file_height; y < dst_height; ++y) std::memcpy(plane + (size_t)y * dst_stride_samples, plane + (size_t)(file_height - 1) * dst_stride_samples, (size_t)dst_width * 2);
    }
  }
  std::fclose(fp);
  if (done == 0 && num_frames > 0) return FHEVC_E_STATE;
  return done;
}

void* fhevc_alloc_host(fhevc_ctx* c, size_t bytes)
{
  if (!c || bytes == 0) return nullptr;
  (void)hipSetDevice(c->device);
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void fhevc_free_host(fhevc_ctx* c, void* p)
{
  if (!c || !p) return;
  (void)hipSetDevice(c->device);
  (void)hipHostFree(p);
}

static bool is_pinned(const void* p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // pageable memory: "invalid value"
  return a.type == hipMemoryTypeHost;
}

// wait for the slot's stream and hand its outputs to the caller (pageable destinations were staged in pinned memory)
static int drain_slot(fhevc_ctx* c, fhevc_ctx::Slot& sl)
{
  if (sl.frames == 0) return FHEVC_OK;
  HIP_TRY(c, hipStreamSynchronize(sl.st));
  if (sl.staged_out) {
    std::memcpy(sl.out_depth, sl.h_depth, (size_t)sl.frames * c->num_ctus * 256);
    if (sl.out_had) std::memcpy(sl.out_had, sl.h_had, (size_t)sl.frames * c->num_ctus * 4);
  }
  sl.frames = 0;
  return FHEVC_OK;
}

static int predict_frames_one_device(fhevc_ctx* c, const void* luma, int sample_bytes, int stride_samples, long long frame_stride_samples, int num_frames,
                                     int qp, uint8_t* depth_map, int32_t* ctu_src_hadamard);

int fhevc_predict_frames(fhevc_ctx* c, const void* luma, int sample_bytes, int stride_samples, long long frame_stride_samples, int num_frames,
                         int qp, uint8_t* depth_map, int32_t* ctu_src_hadamard)
{
  if (!c || !luma || !depth_map) return FHEVC_E_INVALID;
  if (c->peers.empty() || num_frames < 2) return predict_frames_one_device(c, luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, qp, depth_map, ctu_src_hadamard);
  // a batch over several devices: contiguous runs of pictures, frames [i * F / n, (i + 1) * F / n) on device i ("frames can
  // instead be dealt round-robin -- strictly simpler"); every device runs its own two-stream host batch, all at the same time
  return run_sharded(c, [&](int i, int n, fhevc_ctx* d) {
    int fb = 0, fe = 0;
    (void)fhevc_band(num_frames, i, n, &fb, &fe);
    if (fb >= fe) return (int)FHEVC_OK;
    return predict_frames_one_device(d, static_cast<const uint8_t*>(luma) + (size_t)fb * (size_t)frame_stride_samples * sample_bytes, sample_bytes, stride_samples,
                                     frame_stride_samples, fe - fb, qp, depth_map + (size_t)fb * c->num_ctus * 256,
                                     ctu_src_hadamard ? ctu_src_hadamard + (size_t)fb * c->num_ctus : nullptr);
  });
}

static int predict_frames_one_device(fhevc_ctx* c, const void* luma, int sample_bytes, int stride_samples, long long frame_stride_samples, int num_frames,
                                     int qp, uint8_t* depth_map, int32_t* ctu_src_hadamard)
{
  if (!c || !luma || !depth_map) return FHEVC_E_INVALID;
  if (!c->have_weights) return fail(c, FHEVC_E_STATE, "weights not set");
  if ((sample_bytes != 1 && sample_bytes != 2) || stride_samples < c->cfg.width || num_frames < 1) return fail(c, FHEVC_E_INVALID, "bad frame layout");
  if (sample_bytes == 1 && c->cfg.bit_depth != 8) return fail(c, FHEVC_E_INVALID, "uint8 samples need bit_depth 8");
  const long long frame_extent = (long long)stride_samples * (c->cfg.height - 1) + c->cfg.width;  // samples of one frame, first to last
  if (num_frames > 1 && frame_stride_samples < frame_extent) return fail(c, FHEVC_E_INVALID, "frames overlap");
  (void)hipSetDevice(c->device);
  const int chunk = c->cfg.max_frames;
  const size_t fs_bytes = (size_t)(num_frames > 1 ? frame_stride_samples : frame_extent) * sample_bytes;
  const size_t chunk_in = (size_t)(chunk - 1) * fs_bytes + (size_t)frame_extent * sample_bytes;  // bytes from the first sample of a chunk to its last
  const bool in_pinned = is_pinned(luma), out_pinned = is_pinned(depth_map) && (!ctu_src_hadamard || is_pinned(ctu_src_hadamard));
  for (auto& sl : c->slot) {
    if (!sl.st) HIP_TRY(c, hipStreamCreateWithFlags(&sl.st, hipStreamNonBlocking));
    if (sl.in_cap < chunk_in) {
      (void)hipFree(sl.d_in);
      sl.d_in = nullptr; sl.in_cap = 0;
      HIP_TRY(c, hipMalloc(&sl.d_in, chunk_in + 64));
      sl.in_cap = chunk_in;
    }
    if (sl.frames_cap < (size_t)chunk) {
      (void)hipFree(sl.d_depth); (void)hipFree(sl.d_had);
      if (sl.h_depth) (void)hipHostFree(sl.h_depth);
      if (sl.h_had) (void)hipHostFree(sl.h_had);
      sl.d_depth = nullptr; sl.d_had = nullptr; sl.h_depth = nullptr; sl.h_had = nullptr; sl.frames_cap = 0;
      HIP_TRY(c, hipMalloc(&sl.d_depth, (size_t)chunk * c->num_ctus * 256));
      HIP_TRY(c, hipMalloc(&sl.d_had, (size_t)chunk * c->num_ctus * 4));
      HIP_TRY(c, hipHostMalloc(&sl.h_depth, (size_t)chunk * c->num_ctus * 256, hipHostMallocDefault));
      HIP_TRY(c, hipHostMalloc(&sl.h_had, (size_t)chunk * c->num_ctus * 4, hipHostMallocDefault));
      sl.frames_cap = (size_t)chunk;
    }
    if (!in_pinned && sl.h_in_cap < chunk_in) {
      if (sl.h_in) (void)hipHostFree(sl.h_in);
      sl.h_in = nullptr; sl.h_in_cap = 0;
      HIP_TRY(c, hipHostMalloc(&sl.h_in, chunk_in, hipHostMallocDefault));
      sl.h_in_cap = chunk_in;
    }
  }
  int rc = FHEVC_OK;
  auto hip_rc = [&](hipError_t e, const char* what) { return e == hipSuccess ? FHEVC_OK : fail(c, FHEVC_E_HIP, what, e); };
  for (int f0 = 0, k = 0; f0 < num_frames && rc == FHEVC_OK; f0 += chunk, ++k) {
    fhevc_ctx::Slot& sl = c->slot[k & 1];
    rc = drain_slot(c, sl);  // chunk k-2: its maps reach the caller while chunk k-1 computes
    if (rc != FHEVC_OK) break;
    const int nf = std::min(chunk, num_frames - f0);
    const uint8_t* src = static_cast<const uint8_t*>(luma) + (size_t)f0 * fs_bytes;
    const size_t bytes = (size_t)(nf - 1) * fs_bytes + (size_t)frame_extent * sample_bytes;
    if (!in_pinned) { std::memcpy(sl.h_in, src, bytes); src = sl.h_in; }
    rc = hip_rc(hipMemcpyAsync(sl.d_in, src, bytes, hipMemcpyHostToDevice, sl.st), "upload of a chunk");
    if (rc != FHEVC_OK) break;
    rc = fhevc_predict_frames_device(c, sl.d_in, sample_bytes, stride_samples, (long long)(fs_bytes / sample_bytes), nf, 0, c->ctus_y, qp, sl.d_depth,
                                     ctu_src_hadamard ? sl.d_had : nullptr, nullptr, nullptr, sl.st);
    if (rc != FHEVC_OK) break;
    rc = hip_rc(hipMemcpyAsync(out_pinned ? depth_map + (size_t)f0 * c->num_ctus * 256 : sl.h_depth, sl.d_depth, (size_t)nf * c->num_ctus * 256,
                               hipMemcpyDeviceToHost, sl.st), "download of a chunk's maps");
    if (rc == FHEVC_OK && ctu_src_hadamard)
      rc = hip_rc(hipMemcpyAsync(out_pinned ? (void*)(ctu_src_hadamard + (size_t)f0 * c->num_ctus) : (void*)sl.h_had, sl.d_had, (size_t)nf * c->num_ctus * 4,
                                 hipMemcpyDeviceToHost, sl.st), "download of a chunk's Hadamard sums");
    if (rc != FHEVC_OK) break;
    // only now is the slot "in flight": a failure above leaves nothing for a later drain to copy into this caller's buffers
    sl.frames = nf;
    sl.out_depth = depth_map + (size_t)f0 * c->num_ctus * 256;
    sl.out_had = ctu_src_hadamard ? ctu_src_hadamard + (size_t)f0 * c->num_ctus : nullptr;
    sl.staged_out = !out_pinned;
    c->stats.bytes_h2d += bytes;
    c->stats.bytes_d2h += (uint64_t)nf * c->num_ctus * (256 + (ctu_src_hadamard ? 4 : 0));
  }
  if (rc != FHEVC_OK) {
    // error path: let both streams finish whatever they hold, hand nothing more to the caller, and forget the slots' destinations
    // (the caller's buffers may be gone by the next call)
    for (auto& sl : c->slot) {
      if (sl.st) (void)hipStreamSynchronize(sl.st);
      sl.frames = 0; sl.out_depth = nullptr; sl.out_had = nullptr;
    }
    return rc;
  }
  for (auto& sl : c->slot) {
    const int r2 = drain_slot(c, sl);
    if (rc == FHEVC_OK) rc = r2;
  }
  if (rc != FHEVC_OK)
    for (auto& sl : c->slot) { sl.frames = 0; sl.out_depth = nullptr; sl.out_had = nullptr; }
  return rc;
}

int fhevc_predict_frame_range(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, int slice_type, int margin_split, int margin_stop,
                              uint8_t* depth_min, uint8_t* depth_max, int32_t* ctu_src_hadamard)
{
  (void)slice_type;
  if (!c || !luma || !depth_min || !depth_max || stride_samples < c->cfg.width) return FHEVC_E_INVALID;
  if (!c->have_weights) return fail(c, FHEVC_E_STATE, "weights not set");
  if (c->peers.empty()) return predict_band_host(c, luma, stride_samples, qp, 0, c->ctus_y, margin_split, margin_stop, depth_min, depth_max, ctu_src_hadamard);
  const uint64_t frames0 = c->stats.frames;
  const int rc = run_sharded(c, [&](int i, int n, fhevc_ctx* d) {
    int rb = 0, re = 0;
    (void)fhevc_band(c->ctus_y, i, n, &rb, &re);
    return predict_band_host(d, luma, stride_samples, qp, rb, re, margin_split, margin_stop, depth_min, depth_max, ctu_src_hadamard);
  });
  c->stats.frames = frames0 + 1;
  return rc;
}

int fhevc_satd(fhevc_ctx* c, const int16_t* org, int org_stride, const int16_t* cur, int cur_stride,
               int w, int h, int bit_depth, uint32_t* out)
{
  if (!c || !org || !cur || !out) return FHEVC_E_INVALID;
  if (w < 2 || h < 2 || w > 64 || h > 64 || (w & 1) || (h & 1) || bit_depth < 8 || bit_depth > 12) return fail(c, FHEVC_E_INVALID, "bad SATD block");
  if (org_stride < w || cur_stride < w) return fail(c, FHEVC_E_INVALID, "bad SATD stride");
  (void)hipSetDevice(c->device);
  HIP_TRY(c, hipMemcpy2DAsync(c->d_satd, 64 * 2, org, (size_t)org_stride * 2, (size_t)w * 2, (size_t)h, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpy2DAsync(c->d_satd + 64 * 64, 64 * 2, cur, (size_t)cur_stride * 2, (size_t)w * 2, (size_t)h, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, fhevc_launch_satd(c->d_satd, 64, c->d_satd + 64 * 64, 64, w, h, bit_depth, c->d_satd_out, c->stream));
  HIP_TRY(c, hipMemcpyAsync(out, c->d_satd_out, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

int fhevc_intra_first_pass(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, fhevc_node_cost* out)
{
  if (!c || !luma || !out || stride_samples < c->cfg.width || qp < 0 || qp > 51) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  int rc = upload_frame(c, luma, stride_samples);
  if (rc != FHEVC_OK) return rc;
  // lambda = 0.57 * 2^((qp-12)/3): TEncSlice::calculateLambda, all-intra path (TEncSlice.cpp:433-527)
  const double sqrt_lambda = std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
  const FhevcFrames fr = frames_of(c, c->d_luma, 2, c->dev_stride, 0, 1, 0, c->ctus_y);
  time_begin(c, c->stream, 2);
  HIP_TRY(c, fhevc_launch_first_pass(fr, sqrt_lambda, c->d_nodes, nullptr, c->stream));
  time_end(c, c->stream);
  static_assert(sizeof(fhevc_node_cost) == sizeof(FhevcNodeCost), "node cost layout");
  HIP_TRY(c, hipMemcpyAsync(out, c->d_nodes, (size_t)c->num_ctus * FHEVC_NODES_PER_CTU * sizeof(FhevcNodeCost), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

int fhevc_intra_first_pass_all(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, fhevc_node_cost* best, fhevc_node_cost* all)
{
  if (!c || !luma || !all || stride_samples < c->cfg.width || qp < 0 || qp > 51) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  const size_t n_all = (size_t)c->num_ctus * FHEVC_NODES_PER_CTU * 35;
  FhevcNodeCost* d_all = nullptr;  // a parity entry point: allocated per call
  HIP_TRY(c, hipMalloc(&d_all, n_all * sizeof(FhevcNodeCost)));
  int rc = upload_frame(c, luma, stride_samples);
  if (rc == FHEVC_OK) {
    const double sqrt_lambda = std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
    const FhevcFrames fr = frames_of(c, c->d_luma, 2, c->dev_stride, 0, 1, 0, c->ctus_y);
    hipError_t e = fhevc_launch_first_pass(fr, sqrt_lambda, c->d_nodes, d_all, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(all, d_all, n_all * sizeof(FhevcNodeCost), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && best) e = hipMemcpyAsync(best, c->d_nodes, (size_t)c->num_ctus * FHEVC_NODES_PER_CTU * sizeof(FhevcNodeCost), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, FHEVC_E_HIP, "first pass (all modes)", e);
  }
  (void)hipFree(d_all);
  c->stats.kernels_launched++;
  return rc;
}

// The consumer of the first pass: HM prunes the 35 modes of a PU to numModesForFullRD candidates with exactly these costs
// (TEncSearch::estIntraPredLumaQT, TEncSearch.cpp:2271-2320, xUpdateCandList :5385-5408: the N smallest costs, an earlier mode ahead of a later
// one of the same cost).  modes: numCtus * 85 * num_candidates, best first; 255 in every slot of a node that crosses the picture edge.
int fhevc_intra_first_pass_candidates(fhevc_ctx* c, const int16_t* luma, int stride_samples, int qp, int num_candidates, uint8_t* modes)
{
  if (!c || !luma || !modes || num_candidates < 1 || num_candidates > 8 || stride_samples < c->cfg.width || qp < 0 || qp > 51) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  const size_t nodes = (size_t)c->num_ctus * FHEVC_NODES_PER_CTU;
  if (!c->d_cand_all) HIP_TRY(c, hipMalloc(&c->d_cand_all, nodes * 35 * sizeof(FhevcNodeCost)));
  if (!c->d_cand) HIP_TRY(c, hipMalloc(&c->d_cand, nodes * 8));
  int rc = upload_frame(c, luma, stride_samples);
  if (rc != FHEVC_OK) return rc;
  const double sqrt_lambda = std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
  const FhevcFrames fr = frames_of(c, c->d_luma, 2, c->dev_stride, 0, 1, 0, c->ctus_y);
  time_begin(c, c->stream, 2);
  hipError_t e = fhevc_launch_first_pass(fr, sqrt_lambda, c->d_nodes, c->d_cand_all, c->stream);
  time_end(c, c->stream);
  // the sort runs on the device: 85 x K bytes per CTU come back instead of 85 x 35 x 16
  if (e == hipSuccess) e = fhevc_launch_first_pass_topk(c->d_cand_all, (long long)nodes, num_candidates, c->d_cand, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(modes, c->d_cand, nodes * num_candidates, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, FHEVC_E_HIP, "first-pass candidates", e);
  c->stats.kernels_launched += 2;
  c->stats.bytes_d2h += nodes * num_candidates;
  return FHEVC_OK;
}

int fhevc_aq_parts(int width, int height, int max_aq_depth, long long* layer_offsets)
{
  if (width <= 0 || height <= 0 || max_aq_depth < 1 || max_aq_depth > 4) return FHEVC_E_INVALID;
  long long off = 0;
  for (int d = 0; d < max_aq_depth; ++d) {
    if (layer_offsets) layer_offsets[d] = off;
    const int p = 64 >> d;
    off += (long long)((width + p - 1) / p) * ((height + p - 1) / p);
  }
  if (layer_offsets) layer_offsets[max_aq_depth] = off;
  return (int)off;
}

int fhevc_aq_qp(const double* activity, const double* avg_activity, int width, int height, int max_aq_depth,
                int qp_adaptation_range, int base_qp, int qp_bd_offset, int8_t* qp)
{
  long long off[5];
  if (!activity || !avg_activity || !qp || fhevc_aq_parts(width, height, max_aq_depth, off) < 0) return FHEVC_E_INVALID;
  if (base_qp < -qp_bd_offset || base_qp > 51 || qp_bd_offset < 0 || qp_bd_offset > 48) return FHEVC_E_INVALID;
  const double max_q_scale = std::pow(2.0, qp_adaptation_range / 6.0);
  for (int d = 0; d < max_aq_depth; ++d) {
    const double avg = avg_activity[d];
    for (long long i = off[d]; i < off[d + 1]; ++i) {
      const double act = activity[i];
      const double norm = (max_q_scale * act + avg) / (act + max_q_scale * avg);
      const double qoff = std::log(norm) / std::log(2.0) * 6.0;
      const int v = base_qp + (int)std::floor(qoff + 0.49999);
      qp[i] = (int8_t)std::min(51, std::max(-qp_bd_offset, v));
    }
  }
  return FHEVC_OK;
}

int fhevc_preanalyze_frames_device(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples,
                                   long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                   int max_aq_depth, double* d_activity, void* stream)
{
  if (!c || !d_luma || !d_activity) return FHEVC_E_INVALID;
  if ((sample_bytes != 1 && sample_bytes != 2) || stride_samples < c->cfg.width || num_frames < 1 ||
      ctu_row_begin < 0 || ctu_row_end > c->ctus_y || ctu_row_begin > ctu_row_end || max_aq_depth < 1 || max_aq_depth > 4)
    return fail(c, FHEVC_E_INVALID, "bad pre-analysis arguments");
  if ((c->cfg.width & 7) || (c->cfg.height & 7)) return fail(c, FHEVC_E_INVALID, "pre-analysis needs picture sizes that are multiples of 8");
  if (sample_bytes == 1 && c->cfg.bit_depth != 8) return fail(c, FHEVC_E_INVALID, "uint8 samples need bit depth 8");
  (void)hipSetDevice(c->device);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const FhevcFrames fr = frames_of(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, ctu_row_begin, ctu_row_end);
  const long long per_frame = fhevc_aq_parts(c->cfg.width, c->cfg.height, max_aq_depth, nullptr);
  time_begin(c, st, 3);
  HIP_TRY(c, fhevc_launch_preanalyze(fr, max_aq_depth, per_frame, d_activity, c->num_cus, st));
  time_end(c, st);
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

int fhevc_preanalyze(fhevc_ctx* c, const int16_t* luma, int stride_samples, int max_aq_depth, double* activity, double* avg_activity)
{
  if (!c || !luma || !activity || !avg_activity || stride_samples < c->cfg.width) return FHEVC_E_INVALID;
  long long off[5];
  const int total = fhevc_aq_parts(c->cfg.width, c->cfg.height, max_aq_depth, off);
  if (total < 0) return fail(c, FHEVC_E_INVALID, "bad max_aq_depth");
  (void)hipSetDevice(c->device);
  if (!c->d_act) HIP_TRY(c, hipMalloc(&c->d_act, (size_t)fhevc_aq_parts(c->cfg.width, c->cfg.height, 4, nullptr) * sizeof(double)));
  int rc = upload_frame(c, luma, stride_samples);
  if (rc != FHEVC_OK) return rc;
  rc = fhevc_preanalyze_frames_device(c, c->d_luma, 2, c->dev_stride, 0, 1, 0, c->ctus_y, max_aq_depth, c->d_act, c->stream);
  if (rc != FHEVC_OK) return rc;
  HIP_TRY(c, hipMemcpyAsync(activity, c->d_act, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  // TEncPreanalyzer.cpp:147-150: dSumAct accumulates part by part in raster order; keep that order
  for (int d = 0; d < max_aq_depth; ++d) {
    double sum = 0.0;
    for (long long i = off[d]; i < off[d + 1]; ++i) sum += activity[i];
    avg_activity[d] = sum / (double)(off[d + 1] - off[d]);
  }
  c->stats.bytes_d2h += (uint64_t)total * sizeof(double);
  return FHEVC_OK;
}

int fhevc_intra_first_pass_device(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples,
                                  long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                  int qp, fhevc_node_cost* d_out, void* stream)
{
  if (!c || !d_luma || !d_out) return FHEVC_E_INVALID;
  if ((sample_bytes != 1 && sample_bytes != 2) || stride_samples < c->cfg.width || num_frames < 1 || qp < 0 || qp > 51 ||
      ctu_row_begin < 0 || ctu_row_end > c->ctus_y || ctu_row_begin > ctu_row_end)
    return fail(c, FHEVC_E_INVALID, "bad first-pass arguments");
  if (sample_bytes == 1 && c->cfg.bit_depth != 8) return fail(c, FHEVC_E_INVALID, "uint8 samples need bit depth 8");
  (void)hipSetDevice(c->device);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const double sqrt_lambda = std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
  const FhevcFrames fr = frames_of(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, ctu_row_begin, ctu_row_end);
  time_begin(c, st, 2);
  HIP_TRY(c, fhevc_launch_first_pass(fr, sqrt_lambda, reinterpret_cast<FhevcNodeCost*>(d_out), nullptr, st));
  time_end(c, st);
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

// the vector-cost table of the window, with HM's own arithmetic (TComRdCost.h:166-174, TComRdCost.cpp:109-114, 177-190)
static FhevcMvCost mv_cost_table(int qp, int range)
{
  FhevcMvCost t;
  const double sqrt_lambda = std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
  const double motion_lambda = 65536.0 * sqrt_lambda;
  auto eg = [](int v) { unsigned len = 1, u = (v <= 0) ? (((unsigned)(-v)) << 1) + 1 : ((unsigned)v) << 1; while (u != 1) { u >>= 1; len += 2; } return len; };
  const int side = 2 * range + 1;
  for (int m = 0; m < side * side; ++m) {
    const unsigned bits = eg(((m % side) - range) << 2) + eg(((m / side) - range) << 2);
    t.c[m] = (uint32_t)((motion_lambda * bits) / 65536.0);
  }
  return t;
}

int fhevc_motion_search_device(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples, long long frame_stride_samples,
                               int num_frames, int ctu_row_begin, int ctu_row_end, int qp, int search_range, fhevc_motion_node* d_out, void* stream)
{
  if (!c || !d_luma || !d_out) return FHEVC_E_INVALID;
  if ((sample_bytes != 1 && sample_bytes != 2) || stride_samples < c->cfg.width || num_frames < 2 || qp < 0 || qp > 51 ||
      ctu_row_begin < 0 || ctu_row_end > c->ctus_y || ctu_row_begin > ctu_row_end || search_range < 1 || search_range > FHEVC_MOTION_WIDE_MAX_RANGE)
    return fail(c, FHEVC_E_INVALID, "bad motion-search arguments");
  const bool wide = search_range > FHEVC_MOTION_MAX_RANGE;
  if (wide && !c->motion_sad) return fail(c, FHEVC_E_INVALID, "search ranges above 8 need the SAD distortion (fhevc_set_motion_distortion)");
  if (wide && c->cfg.bit_depth > 8 && sample_bytes != 2) return fail(c, FHEVC_E_INVALID, "bad motion-search arguments");
  const bool big = wide && c->cfg.bit_depth > 8;   // above 8 bit: the generic kernel laid out for the wide window (round 4); 8 bit: the byte-SAD kernel
  if (sample_bytes == 1 && c->cfg.bit_depth != 8) return fail(c, FHEVC_E_INVALID, "uint8 samples need bit depth 8");
  if (frame_stride_samples < (long long)stride_samples * (c->cfg.height - 1) + c->cfg.width) return fail(c, FHEVC_E_INVALID, "frames overlap");
  if (ctu_row_begin == ctu_row_end) return FHEVC_OK;
  (void)hipSetDevice(c->device);
  hipStream_t st = stream ? (hipStream_t)stream : c->stream;
  const FhevcFrames fr = frames_of(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, ctu_row_begin, ctu_row_end, qp);
  static_assert(sizeof(fhevc_motion_node) == sizeof(FhevcMotionNode), "motion node layout");
  if (wide && (c->mvtab_qp != qp || c->mvtab_range != search_range)) {
    // the vector costs of the window in raster order, HM's arithmetic as mv_cost_table
    const double motion_lambda = 65536.0 * std::sqrt(0.57 * std::pow(2.0, ((double)qp - 12.0) / 3.0));
    auto eg = [](int v) { unsigned len = 1, u = (v <= 0) ? (((unsigned)(-v)) << 1) + 1 : ((unsigned)v) << 1; while (u != 1) { u >>= 1; len += 2; } return len; };
    const int side = 2 * search_range + 1;
    if (!c->d_mvtab) HIP_TRY(c, hipMalloc(&c->d_mvtab, sizeof(uint32_t) * (2 * FHEVC_MOTION_WIDE_MAX_RANGE + 1) * (2 * FHEVC_MOTION_WIDE_MAX_RANGE + 1)));
    if (c->mvtab_used) HIP_TRY(c, hipEventSynchronize(c->mvtab_used));  // the last launch that read the old table, on ANY stream (the per-call stream may differ)
    c->mvtab_host.resize((size_t)side * side);
    for (int m = 0; m < side * side; ++m)
      c->mvtab_host[m] = (uint32_t)((motion_lambda * (eg(((m % side) - search_range) << 2) + eg(((m / side) - search_range) << 2))) / 65536.0);
    HIP_TRY(c, hipMemcpy(c->d_mvtab, c->mvtab_host.data(), c->mvtab_host.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->mvtab_qp = qp; c->mvtab_range = search_range;
  }
  time_begin(c, st, 4);
  if (wide) {
    if (big) HIP_TRY(c, fhevc_launch_motion_big(fr, search_range, c->d_mvtab, reinterpret_cast<FhevcMotionNode*>(d_out), c->num_cus, st));
    else HIP_TRY(c, fhevc_launch_motion_wide(fr, search_range, c->d_mvtab, reinterpret_cast<FhevcMotionNode*>(d_out), c->num_cus, st));
    if (!c->mvtab_used) HIP_TRY(c, hipEventCreateWithFlags(&c->mvtab_used, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->mvtab_used, st));
  }
  else HIP_TRY(c, fhevc_launch_motion(fr, search_range, mv_cost_table(qp, search_range), reinterpret_cast<FhevcMotionNode*>(d_out), c->num_cus, c->motion_sad, st));
  time_end(c, st);
  c->stats.kernels_launched++;
  return FHEVC_OK;
}

int fhevc_motion_search(fhevc_ctx* c, const int16_t* cur_luma, const int16_t* ref_luma, int stride_samples, int qp, int search_range,
                        fhevc_motion_node* out)
{
  if (!c || !cur_luma || !ref_luma || !out || stride_samples < c->cfg.width) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  const size_t plane = (size_t)c->dev_stride * c->ctus_y * 64;
  if (!c->d_pair) HIP_TRY(c, hipMalloc(&c->d_pair, 2 * plane * sizeof(int16_t)));
  if (!c->d_motion) HIP_TRY(c, hipMalloc(&c->d_motion, (size_t)c->num_ctus * FHEVC_NODES * sizeof(FhevcMotionNode)));
  HIP_TRY(c, hipMemcpy2DAsync(c->d_pair, (size_t)c->dev_stride * 2, ref_luma, (size_t)stride_samples * 2, (size_t)c->cfg.width * 2,
                              (size_t)c->cfg.height, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpy2DAsync(c->d_pair + plane, (size_t)c->dev_stride * 2, cur_luma, (size_t)stride_samples * 2, (size_t)c->cfg.width * 2,
                              (size_t)c->cfg.height, hipMemcpyHostToDevice, c->stream));
  c->stats.bytes_h2d += (uint64_t)c->cfg.width * c->cfg.height * 4;
  const int rc = fhevc_motion_search_device(c, c->d_pair, 2, c->dev_stride, (long long)plane, 2, 0, c->ctus_y, qp, search_range,
                                            reinterpret_cast<fhevc_motion_node*>(c->d_motion), c->stream);
  if (rc != FHEVC_OK) return rc;
  HIP_TRY(c, hipMemcpyAsync(out, c->d_motion, (size_t)c->num_ctus * FHEVC_NODES * sizeof(FhevcMotionNode), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stats.bytes_d2h += (uint64_t)c->num_ctus * FHEVC_NODES * sizeof(FhevcMotionNode);
  return FHEVC_OK;
}

// ---- P-picture depth range from motion nodes + co-located depths (host-side integer rule; spec in include/fasthevc.h) ----
namespace {
inline int32_t ilog2_q8(uint32_t x)  // floor(256 log2 x) by integer squaring
{
  const int msb = 31 - __builtin_clz(x | 1u);
  uint64_t y = ((uint64_t)x << 31) >> msb;
  int32_t r = msb << 8;
  for (int b = 7; b >= 0; --b) {
    y = (y * y) >> 31;
    if (y >> 32) { r |= 1 << b; y >>= 1; }
  }
  return r;
}
struct PNodeRef { int first, per_row; };
constexpr PNodeRef kLevel[4] = { { 0, 1 }, { 1, 2 }, { 5, 4 }, { 21, 8 } };
int64_t p_split_score(const fhevc_motion_node* nodes, const uint8_t* prev, int lvl, int nx, int ny, int qp, const fhevc_p_rule& r)
{
  const fhevc_motion_node& n = nodes[kLevel[lvl].first + ny * kLevel[lvl].per_row + nx];
  int64_t child_cost = 0, child_satd = 0;
  int moved = 0;
  for (int k = 0; k < 4; ++k) {
    const fhevc_motion_node& c = nodes[kLevel[lvl + 1].first + (2 * ny + (k >> 1)) * kLevel[lvl + 1].per_row + 2 * nx + (k & 1)];
    child_cost += c.cost_best; child_satd += c.satd_best;
    moved += (c.mvx != n.mvx) || (c.mvy != n.mvy);
  }
  const int units = 16 >> lvl;
  int deepest = 0, shallowest = 3;
  for (int y = 0; y < units; ++y)
    for (int x = 0; x < units; ++x) {
      const int d = prev[(ny * units + y) * 16 + nx * units + x];
      deepest = std::max(deepest, d); shallowest = std::min(shallowest, d);
    }
  const int64_t gain = std::max<int64_t>(0, (int64_t)n.cost_best - child_cost);
  const int norm = 512 * (6 - lvl) + (qp * 256) / 6;
  const int64_t f[9] = { ilog2_q8(n.satd_best + 1u) - norm, ilog2_q8((uint32_t)gain + 1u) - norm, ilog2_q8((uint32_t)child_satd + 1u) - norm,
                         ilog2_q8(n.satd_zero + 1u) - ilog2_q8(n.satd_best + 1u), deepest > lvl ? 256 : 0, shallowest > lvl ? 256 : 0,
                         deepest > lvl + 1 ? 256 : 0, 64 * moved, 8 * qp };
  int64_t s = r.w[lvl][9];
  for (int i = 0; i < 9; ++i) s += (int64_t)r.w[lvl][i] * f[i];
  return s;
}
}  // namespace

void fhevc_p_rule_default(fhevc_p_rule* rule)
{
  if (!rule) return;
  // logistic fit of HM-16.14's own P-picture split decisions (vanilla decision path, tests/quality/make_labels_p.py +
  // p_features.py + fit_p_rule.py) on seeded pan clips of all synthetic families; weights Q10, bias and thresholds Q18
  static const int32_t w[3][10] = FHEVC_P_RULE_WEIGHTS;
  std::memcpy(rule->w, w, sizeof w);
  const int32_t ts[3] = FHEVC_P_RULE_T_SPLIT, tp[3] = FHEVC_P_RULE_T_STOP;
  std::memcpy(rule->t_split, ts, sizeof ts);
  std::memcpy(rule->t_stop, tp, sizeof tp);
  rule->window = FHEVC_P_RULE_WINDOW;
}

void fhevc_p_rule_default_wide(fhevc_p_rule* rule)
{
  if (!rule) return;
  fhevc_p_rule_default(rule);   // thresholds and window as the default rule: same score semantics (a logit)
  static const int32_t w[3][10] = FHEVC_P_RULE_WIDE_WEIGHTS;
  std::memcpy(rule->w, w, sizeof w);
}

int fhevc_p_depth_range(const fhevc_motion_node* nodes, const uint8_t* prev_depth, int valid_w, int valid_h, int qp, const fhevc_p_rule* rule,
                        uint8_t* depth_min, uint8_t* depth_max)
{
  if (!nodes || !prev_depth || !rule || !depth_min || !depth_max || valid_w < 8 || valid_w > 64 || valid_h < 8 || valid_h > 64 || qp < 0 || qp > 51)
    return FHEVC_E_INVALID;
  std::memset(depth_min, 0, 256);
  std::memset(depth_max, 0, 256);
  // split decisions of the 21 nodes, both thresholds, evaluated lazily top-down; -1 = not evaluated
  int8_t sure[21], maybe[21];
  std::memset(sure, -1, sizeof sure);
  std::memset(maybe, -1, sizeof maybe);
  auto decide = [&](int lvl, int nx, int ny) {
    const int id = kLevel[lvl].first + ny * kLevel[lvl].per_row + nx, n = 64 >> lvl;
    if (sure[id] >= 0) return id;
    if (nx * n + n > valid_w || ny * n + n > valid_h) { sure[id] = maybe[id] = 1; return id; }  // crosses the picture edge
    const int64_t s = p_split_score(nodes, prev_depth, lvl, nx, ny, qp, *rule);
    sure[id] = s > rule->t_split[lvl];
    maybe[id] = s >= -(int64_t)rule->t_stop[lvl];
    return id;
  };
  for (int uy = 0; uy * 4 < valid_h; ++uy)
    for (int ux = 0; ux * 4 < valid_w; ++ux) {
      int lo = 0, hi = 0;
      bool lo_open = true, hi_open = true;
      for (int lvl = 0; lvl < 3 && (lo_open || hi_open); ++lvl) {
        const int id = decide(lvl, ux >> (4 - lvl), uy >> (4 - lvl));
        lo_open = lo_open && sure[id];
        hi_open = hi_open && maybe[id];
        if (lo_open) lo = lvl + 1;
        if (hi_open) hi = lvl + 1;
      }
      if (rule->window < 4) {
        const int p = prev_depth[uy * 16 + ux];
        lo = std::min(3, std::max(0, std::max(lo, p - rule->window)));
        hi = std::min(3, std::max(0, std::min(hi, p + rule->window)));
        if (lo > hi) lo = hi;
      }
      depth_min[uy * 16 + ux] = (uint8_t)lo;
      depth_max[uy * 16 + ux] = (uint8_t)hi;
    }
  return FHEVC_OK;
}

// Diagnostic (not part of include/fasthevc.h): run the stamped instantiation of the depth kernel over a
// device-resident batch and return per-phase cycle sums averaged over workgroups (slots 0..7: prologue, conv1, conv2, conv3,
// barrier wait after staging, depth, heads, staging), CTUs per workgroup (8), the grid (9), and the in-kernel clock in MHz
// (10: median over the workgroups, 11: the slowest; s_memtime span / s_memrealtime span of the whole CTU loop).
int fhevc_debug_cnn_phase_cycles(fhevc_ctx* c, const void* d_luma, int sample_bytes, int stride_samples,
                                 long long frame_stride_samples, int num_frames, uint8_t* d_depth_map, double* out12)
{
  if (!c || !d_luma || !d_depth_map || !out12 || !c->have_weights) return FHEVC_E_INVALID;
  (void)hipSetDevice(c->device);
  const FhevcFrames fr = frames_of(c, d_luma, sample_bytes, stride_samples, frame_stride_samples, num_frames, 0, c->ctus_y);
  unsigned long long* d_st = nullptr;
  const int max_grid = 4 * c->num_cus;  // fhevc_launch_cnn_stamped runs at most four workgroups per CU
  const size_t slots = (size_t)max_grid * FHEVC_STAMP_SLOTS;
  HIP_TRY(c, hipMalloc(&d_st, slots * sizeof(unsigned long long)));
  HIP_TRY(c, hipMemset(d_st, 0, slots * sizeof(unsigned long long)));
  int grid = 0;
  int32_t* d_had = nullptr;  // the fused source Hadamard's output, as in the timed launch (FHEVC_DEBUG_STAMPS_NO_HADAMARD=1: without it)
  if (!std::getenv("FHEVC_DEBUG_STAMPS_NO_HADAMARD")) HIP_TRY(c, hipMalloc(&d_had, (size_t)num_frames * c->num_ctus * sizeof(int32_t)));
  HIP_TRY(c, fhevc_launch_cnn_stamped(fr, cnn_weights(c), d_depth_map, d_had, c->num_cus, c->knobs, d_st, &grid, c->stream));
  std::vector<unsigned long long> h(slots);
  HIP_TRY(c, hipMemcpyAsync(h.data(), d_st, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  (void)hipFree(d_st);
  if (d_had) (void)hipFree(d_had);
  for (int k = 0; k < 12; ++k) out12[k] = 0;
  for (int b = 0; b < grid; ++b) for (int k = 0; k < 8; ++k) out12[k] += (double)h[(size_t)b * FHEVC_STAMP_SLOTS + k] / grid;
  out12[8] = (double)num_frames * c->num_ctus / grid;
  out12[9] = grid;
  // the in-kernel clock: shader cycles per 100 MHz tick over each workgroup's whole CTU loop, median over the workgroups (MHz)
  std::vector<double> mhz;
  for (int b = 0; b < grid; ++b)
    if (h[(size_t)b * FHEVC_STAMP_SLOTS + 9]) mhz.push_back(100.0 * (double)h[(size_t)b * FHEVC_STAMP_SLOTS + 8] / (double)h[(size_t)b * FHEVC_STAMP_SLOTS + 9]);
  std::sort(mhz.begin(), mhz.end());
  out12[10] = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
  out12[11] = mhz.empty() ? 0.0 : mhz.front();
  // [12 + 9 * slot + k]: the same eight phase sums averaged over the workgroups of CU slot 0 / 1 / 2 (k = 8: how many workgroups that is) -- the caller's
  // array holds 39 doubles (FHEVC_DEBUG_STAMPS_BY_SLOT=1; the i8 form's conv-phase priority differs by slot, k_cnn.hip FHEVC_SLOT_PRIO)
  if (std::getenv("FHEVC_DEBUG_STAMPS_BY_SLOT")) {
    for (int k = 12; k < 39; ++k) out12[k] = 0;
    for (int b = 0; b < grid; ++b) {
      const int sl = (int)std::min<unsigned long long>(h[(size_t)b * FHEVC_STAMP_SLOTS + 10], 2);
      for (int k = 0; k < 8; ++k) out12[12 + 9 * sl + k] += (double)h[(size_t)b * FHEVC_STAMP_SLOTS + k];
      out12[12 + 9 * sl + 8] += 1;
    }
    for (int sl = 0; sl < 3; ++sl)
      for (int k = 0; k < 8; ++k) if (out12[12 + 9 * sl + 8] > 0) out12[12 + 9 * sl + k] /= out12[12 + 9 * sl + 8];
  }
  return FHEVC_OK;
}

int fhevc_set_cnn_arith(fhevc_ctx* c, int arith)
{
  if (!c) return FHEVC_E_INVALID;
  if (arith != FHEVC_CNN_ARITH_I8 && arith != FHEVC_CNN_ARITH_F16) return fail(c, FHEVC_E_INVALID, "arith: FHEVC_CNN_ARITH_I8 or FHEVC_CNN_ARITH_F16");
  c->cnn_i8 = arith == FHEVC_CNN_ARITH_I8;
  for (fhevc_ctx* peer : c->peers) peer->cnn_i8 = c->cnn_i8;
  return FHEVC_OK;
}

int fhevc_p_motion_compensated_depth(const fhevc_motion_node* nodes, const uint8_t* prev_map, int width, int height, int ctu, uint8_t* out)
{
  if (!nodes || !prev_map || !out || width < 8 || height < 8 || ctu < 0) return FHEVC_E_INVALID;
  const int cw = (width + 63) / 64, chh = (height + 63) / 64;
  if (ctu >= cw * chh) return FHEVC_E_INVALID;
  const int x0 = (ctu % cw) * 64, y0 = (ctu / cw) * 64;
  for (int by = 0; by < 4; ++by)
    for (int bx = 0; bx < 4; ++bx) {
      // the vector of the smallest valid node around the block: 16x16, 32x32, the CTU
      const fhevc_motion_node* cand[3] = { &nodes[5 + by * 4 + bx], &nodes[1 + (by >> 1) * 2 + (bx >> 1)], &nodes[0] };
      int mvx = 0, mvy = 0;
      for (int k = 0; k < 3; ++k)
        if (cand[k]->cost_best != 0xFFFFFFFFu) { mvx = cand[k]->mvx; mvy = cand[k]->mvy; break; }
      for (int uy = 0; uy < 4; ++uy)
        for (int ux = 0; ux < 4; ++ux) {
          const int px = std::min(std::max(x0 + bx * 16 + ux * 4 + 2 + mvx, 0), width - 1);
          const int py = std::min(std::max(y0 + by * 16 + uy * 4 + 2 + mvy, 0), height - 1);
          const int sc = (py >> 6) * cw + (px >> 6);
          out[(by * 4 + uy) * 16 + bx * 4 + ux] = prev_map[(size_t)sc * 256 + ((py & 63) >> 2) * 16 + ((px & 63) >> 2)];
        }
    }
  return FHEVC_OK;
}

int fhevc_p_node_depth(const fhevc_motion_node* nodes, const uint8_t* prev_map, int width, int height, int ctu, uint8_t* out)
{
  if (!nodes || !prev_map || !out || width < 8 || height < 8 || ctu < 0) return FHEVC_E_INVALID;
  const int cw = (width + 63) / 64, chh = (height + 63) / 64;
  if (ctu >= cw * chh) return FHEVC_E_INVALID;
  const int x0 = (ctu % cw) * 64, y0 = (ctu / cw) * 64;
  auto ref_depth = [&](int x, int y) {
    x = std::min(std::max(x, 0), width - 1); y = std::min(std::max(y, 0), height - 1);
    return (int)prev_map[(size_t)((y >> 6) * cw + (x >> 6)) * 256 + ((y & 63) >> 2) * 16 + ((x & 63) >> 2)];
  };
  auto fill = [&](int ux, int uy, int units, int depth) {
    for (int y = uy; y < uy + units; ++y) std::memset(out + y * 16 + ux, depth, (size_t)units);
  };
  struct Mv { int x, y; };
  auto vector_of = [](const fhevc_motion_node& n, Mv parent) { return n.cost_best != 0xFFFFFFFFu ? Mv{ n.mvx, n.mvy } : parent; };
  const Mv v0 = vector_of(nodes[0], Mv{ 0, 0 });
  if (ref_depth(x0 + 32 + v0.x, y0 + 32 + v0.y) == 0) { fill(0, 0, 16, 0); return FHEVC_OK; }
  for (int q = 0; q < 4; ++q) {
    const int qx = q & 1, qy = q >> 1;
    const Mv v1 = vector_of(nodes[1 + q], v0);
    if (ref_depth(x0 + qx * 32 + 16 + v1.x, y0 + qy * 32 + 16 + v1.y) <= 1) { fill(qx * 8, qy * 8, 8, 1); continue; }
    for (int b = 0; b < 4; ++b) {
      const int bx = 2 * qx + (b & 1), by = 2 * qy + (b >> 1);
      const Mv v2 = vector_of(nodes[5 + by * 4 + bx], v1);
      fill(bx * 4, by * 4, 4, ref_depth(x0 + bx * 16 + 8 + v2.x, y0 + by * 16 + 8 + v2.y) <= 2 ? 2 : 3);
    }
  }
  return FHEVC_OK;
}

int fhevc_set_motion_distortion(fhevc_ctx* c, int mode)
{
  if (!c) return FHEVC_E_INVALID;
  if (mode != FHEVC_MOTION_SATD && mode != FHEVC_MOTION_SAD) return fail(c, FHEVC_E_INVALID, "mode: FHEVC_MOTION_SATD or FHEVC_MOTION_SAD");
  c->motion_sad = mode == FHEVC_MOTION_SAD;
  return FHEVC_OK;
}

int fhevc_get_cnn_arith(const fhevc_ctx* c)
{
  if (!c) return FHEVC_E_INVALID;
  return c->cnn_i8 ? FHEVC_CNN_ARITH_I8 : FHEVC_CNN_ARITH_F16;
}

int fhevc_get_stats(fhevc_ctx* c, void* out, size_t size)
{
  if (!c || !out) return FHEVC_E_INVALID;
  time_resolve(c);
  std::memcpy(out, &c->stats, size < sizeof(fhevc_stats) ? size : sizeof(fhevc_stats));
  return FHEVC_OK;
}

}  // extern "C"
