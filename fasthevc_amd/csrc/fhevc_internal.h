// fhevc_internal.h -- shared declarations between the C-ABI layer (fhevc_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FHEVC_CTU 64

// ---- depth CNN (k_cnn.hip) -------------------------------------------------------------------------------
// Packed weight image in HBM, built once by fhevc_set_weights (fhevc_api.hip: build_weight_image):
//   frag  : MFMA A-operand fragments, one uint4 (8 bf16) per lane: conv1 [2][64], conv2 [9][64], conv3 [2][18][64], all scaled
//           by their layer's 2^-shift
//   bias  : float b1[16], b2[32], b3[64]
//   whead : int8: wh64[2][4096], wh32[2][4096], wh16[2][1024]
//   bhead : int32 bh64[2], bh32[2], bh16[2], qp_bias[3][52]
#define FHEVC_FRAG_CONV1 0
#define FHEVC_FRAG_CONV2 128
#define FHEVC_FRAG_CONV3 (128 + 9 * 64)
#define FHEVC_FRAG_HAD (128 + 9 * 64 + 2 * 18 * 64)  // the source Hadamard's constant A operands (k_cnn.hip, HAD == 2): [M tile 2][K step 5][64]
#define FHEVC_FRAG_TOTAL (FHEVC_FRAG_HAD + 2 * 5 * 64)
// conv3 of the 16-bit form: 0 = v_mfma_f32_16x16x32_f16 (M tiles of 16 channels, one output row per chain), 1 = v_mfma_f32_32x32x16_f16
// (the wave's 32 channels x two output rows 8 apart per MFMA, K step = one tap x 16 channels).  The fragment image differs:
// kernel and build_weight_image read this switch
#ifndef FHEVC_F16_CONV3_32
#define FHEVC_F16_CONV3_32 0  // measured equal (0.5624 against 0.5659 ms on one box, parity green): the 16x16x32 form stays
#endif
// the i8 variant's fragments (one uint4 = 16 signed bytes per lane): conv2 [2 row pairs][3 columns of taps][64], conv3 [2 tiles][9 taps][64]
#define FHEVC_STAMP_SLOTS 11   // per workgroup of the stamped diagnostic kernel: 8 phase sums, the CTU loop's s_memtime and s_memrealtime spans, the workgroup's slot on its CU
#define FHEVC_FRAGI8_CONV2 0
#define FHEVC_FRAGI8_CONV3 (6 * 64)
#define FHEVC_FRAGI8_TOTAL (6 * 64 + 2 * 9 * 64)

// Tuning / test switches read from the environment ONCE, at fhevc_create (fhevc_read_knobs), and kept in the context: nothing on a launch path
// calls getenv.  None of them changes results; FHEVC_CNN_ARITH / FHEVC_CNN_REQUANT / FHEVC_HADAMARD_FORM / FHEVC_CNN_PIPE / FHEVC_FUSE_HADAMARD
// select between forms that the parity suite runs side by side.
struct FhevcKnobs {
  int wg_per_cu = 0;            // FHEVC_CNN_WG_PER_CU=1..4: workgroups per CU of the depth kernel's persistent grid (0: the form's own)
  int debug_wg_per_cu = 0;      // FHEVC_DEBUG_WG_PER_CU=1..4: the same for the stamped diagnostic build only
  bool requant_general = false; // FHEVC_CNN_REQUANT=general: the i8 form's general requant instead of the short forms
  bool family_layers = false;   // FHEVC_FAMILY_LAYERS: a member with a fused kernel runs layer by layer all the same (tests)
  bool trio = false;            // FHEVC_CNN_TRIO=1: the i8 depth kernel as ONE 768-thread workgroup per CU, three groups one barrier interval apart (k_cnn.hip, TRIO)
  bool d2_requant_general = false; // FHEVC_D2_REQUANT=general: the fused two-convolution kernel's general requant instantiation whatever the blob allows (tests, A/B)
  bool fused_d2 = true;         // FHEVC_FUSED_D2=0: the two-convolutions-per-block members run layer by layer instead of through k_cnn_d2.inc (tests, A/B)
  bool layers_no_fuse = false;  // FHEVC_LAYERS_NO_FUSE: the layer path without the first convolution fused into the second (tests)
  bool layers_no_dbuf = false;  // FHEVC_LAYERS_NO_DBUF: the layer path's single-buffered staging (tests)
  size_t layers_lds_limit = 80 * 1024;  // FHEVC_LAYERS_LDS_KB (experiments)
  int layers_grid = 2048;       // FHEVC_LAYERS_GRID (experiments)
};
FhevcKnobs fhevc_read_knobs();

struct FhevcFrames {
  const void* luma;          // device pointer to sample (0,0) of frame 0
  int sample_bytes;          // 1 or 2
  int stride;                // samples
  long long frame_stride;    // samples
  int width, height, bit_depth;
  int ctus_x, ctus_y;
  int num_frames;
  int row_begin, row_end;    // CTU-row band processed by this launch
  int qp;                    // slice QP (0..51): selects the per-QP prior of the classifier heads
};

struct FhevcCnnWeights {
  const uint4* frag;
  const float* bias;
  const uint8_t* whead;
  const int32_t* bhead;      // bh64[2], bh32[2], bh16[2], then qp_bias[3][52]
  float scale[3];            // 2^-shift per conv layer (folded into the fragments; the kernel pre-scales the biases with it)
  // the i8 variant of conv2 / conv3 (v_mfma_i32_32x32x32_i8 on activations a - 128): unscaled int8 fragments, int32 biases
  // (+ 128 * the sum of the filter's weights; entries 16.. of bias_i8, laid out like bias) and the requant shifts
  const uint4* frag_i8;
  const int32_t* bias_i8;
  int shift[3];
  int requant_mode[3];       // per layer: 0 general, 1 shift <= 7 (packed 16-bit shift), 2 shift == 8 and |accumulator| < 2^23 (byte gather)
  int i8;                    // 1: run that variant
  int had_valu;              // 1: the fused source Hadamard on packed 16-bit VALU also for 8-bit content (default; FHEVC_HADAMARD_FORM=mfma: 0)
  int pipe;                  // 1: the i8 form runs as the two-stage software pipeline over CTUs (fhevc_cnn_depth_pipe_kernel; FHEVC_CNN_PIPE)
};

// a member of the reference's Bayesian-optimisation network family with one convolution per block (k_cnn_family.inc; FHW3 blob)
struct FhevcFamilyWeights {
  int c[3];                  // channel widths (multiples of 16 / 32 / 32)
  const uint4* frag1;        // conv1: [C1 / 16 groups][2 (pre-pool column)][64 lanes], bf16, scaled by 2^-shift1
  const float* bias1;        // conv1 biases, pre-scaled, - 128 * sum of weights
  const uint4* frag2;        // conv2: [M tile][K chunk][tap][64 lanes], 16 signed bytes per lane
  const uint4* frag3;        // conv3: likewise
  const int32_t* bias_i8;    // bias2[C2], bias3[C3], + 128 * sum of weights
  const uint8_t* whead;      // wh64[2][8][8][C3], wh32[2][8][8][C3], wh16[2][4][4][C3]
  const uint8_t* headm;      // MFMA image of wh32 / wh16: [position 16][chunk C3 / 64][column 16][64 B] (columns 0, 1: 16-level; 2 + 2 sub + class: 32-level)
  const int32_t* bhead;      // as FhevcCnnWeights::bhead
  int shift[3];
};
hipError_t fhevc_launch_cnn_family(const FhevcFrames& fr, const FhevcFamilyWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                                   uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, hipStream_t stream);
bool fhevc_cnn_family_supported(int c1, int c2, int c3);

// any member of the family, layer by layer through HBM (k_cnn_layers.inc)
struct FhevcLayer {
  const uint4* frag;     // [M tile][K chunk][tap 9][64 lanes] (first layer: [M tile][64]: K = the nine taps of the one input channel)
  const int32_t* bias;   // [cout_pad], + 128 * sum of the filter's weights (not for the first layer: its input is centred samples)
  int8_t* out;           // [chunk CTUs][Ho + 2] rows of ([Ho + 2][cout_pad] + out_pad bytes)
  int shift, kc, cout_pad, H, pool;   // kc = Cin_pad / 32 (0: first layer); H = input size (64 / 32 / 16)
  int rq;                             // the shortest requant form the layer's weights allow (k_cnn_layers.inc: layer_store16): 1 = shift <= 7, 2 = shift 8 and |accumulator| < 2^23, else 0
  int in_pad, out_pad, swz;           // bytes added to the row pitch of the input / output tensor; XOR mask of the LDS image (fhevc_layer_lds_image)
};
// the row-pitch padding and XOR mask that make the layer kernel's LDS reads conflict-free for an input of kc x 32 channels at H x H (tools/lds_swizzle_search.py)
void fhevc_layer_lds_image(int kc, int pool, int H, int* pad, int* mask);
struct FhevcLayersWeights {
  int num_layers, chunk, c3, c3_pad;
  int d2_short;            // the fused two-convolution kernel may run its short-requant instantiation (l[0].rq == 1, l[1..5].rq == 2; FHEVC_D2_REQUANT=general: never)
  FhevcLayer l[9];
  int8_t* in0;             // [chunk CTUs][66][66]
  const uint8_t* whead;    // wh64[2][8][8][c3_pad], wh32[2][8][8][c3_pad], wh16[2][4][4][c3_pad] (zeros behind the c3 weights)
  const int32_t* bhead;    // as FhevcCnnWeights::bhead
};

hipError_t fhevc_launch_cnn_layers(const FhevcFrames& fr, const FhevcLayersWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                                   uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, const FhevcKnobs& knobs, hipStream_t stream);

// the same members with two convolutions per block and padded widths 32 / 64 / 96 (23 / 46 / 92 x 2) as one LDS-resident kernel (k_cnn_d2.inc)
bool fhevc_cnn_d2_supported(const FhevcLayersWeights& w);
hipError_t fhevc_launch_cnn_d2(const FhevcFrames& fr, const FhevcLayersWeights& w, uint8_t* d_depth, int32_t* d_logits, uint32_t* d_flags,
                               uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, hipStream_t stream);

hipError_t fhevc_cnn_prepare_device();  // LDS opt-in of the depth kernel on the current device (once per context)
// d_depth_max / margins: soft decisions (nullptr / 0, 0 = the plain map only)
// d_had != nullptr: the per-CTU source Hadamard is computed inside the depth kernel from the samples it loads anyway (one
// pass over the frame); allowed only where fhevc_cnn_can_fuse_hadamard(fr), otherwise use fhevc_launch_src_hadamard
bool fhevc_cnn_can_fuse_hadamard(const FhevcFrames& fr);
hipError_t fhevc_launch_cnn(const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_had, int32_t* d_logits,
                            uint32_t* d_flags, uint8_t* d_depth_max, int margin_split, int margin_stop, int num_cus, const FhevcKnobs& knobs, hipStream_t stream);
hipError_t fhevc_launch_expand_flags(const FhevcFrames& fr, const uint32_t* d_flags, uint8_t* d_depth, hipStream_t stream);

hipError_t fhevc_launch_cnn_stamped(const FhevcFrames& fr, const FhevcCnnWeights& w, uint8_t* d_depth, int32_t* d_had /* null: without the fused Hadamard */, int num_cus, const FhevcKnobs& knobs,
                                    unsigned long long* d_stamps, int* grid_out, hipStream_t stream);

// ---- source Hadamard + SATD (k_hadamard.hip) ---------------------------------------------------------------
hipError_t fhevc_launch_src_hadamard(const FhevcFrames& fr, int32_t* d_out, hipStream_t stream);
hipError_t fhevc_launch_satd(const int16_t* d_org, int org_stride, const int16_t* d_cur, int cur_stride,
                             int w, int h, int bit_depth, uint32_t* d_out, hipStream_t stream);

// ---- 35-mode first pass (k_firstpass.hip) ------------------------------------------------------------------
struct FhevcNodeCost { uint32_t satd; uint32_t mode; double cost; };
// d_all (optional): every (node, mode) pair, [CTU][85][35] -- the parity output behind fhevc_intra_first_pass_all
hipError_t fhevc_launch_first_pass(const FhevcFrames& fr, double sqrt_lambda, FhevcNodeCost* d_out, FhevcNodeCost* d_all, hipStream_t stream);

// the K (<= 8) cheapest modes per node out of d_all, best first (candidate lists of fhevc_intra_first_pass_candidates)
hipError_t fhevc_launch_first_pass_topk(const FhevcNodeCost* d_all, long long nodes, int k, uint8_t* d_modes, hipStream_t stream);

// ---- source-only motion search per CU node (k_motion.hip; config 4) -----------------------------------------
#define FHEVC_NODES 85
#define FHEVC_MOTION_MAX_RANGE 8
struct FhevcMotionNode { uint32_t satd_zero, satd_best, cost_best; int16_t mvx, mvy; };
struct FhevcMvCost { uint32_t c[(2 * FHEVC_MOTION_MAX_RANGE + 1) * (2 * FHEVC_MOTION_MAX_RANGE + 1)]; };  // [dy + R][dx + R] of the window in use
// frames 1 .. num_frames-1 of fr, each searched in the frame before it; d_out: (num_frames - 1) * band CTUs * 85 nodes
// sad: SAD (HM's integer-search distortion, pinned to the reference's xPatternSearch) instead of Hadamard SATD
hipError_t fhevc_launch_motion(const FhevcFrames& fr, int range, const FhevcMvCost& mvc, FhevcMotionNode* d_out, int num_cus, bool sad, hipStream_t stream);

// the same search in its SAD mode over a window of up to +-64 (HM's SearchRange), 8-bit content (k_motion_wide.hip); d_mvtab: (2 range + 1)^2
// vector costs in raster order, in HBM
#define FHEVC_MOTION_WIDE_MAX_RANGE 64
hipError_t fhevc_launch_motion_wide(const FhevcFrames& fr, int range, const uint32_t* d_mvtab, FhevcMotionNode* d_out, int num_cus, hipStream_t stream);
// ... and for content ABOVE 8 bit (16-bit planes; round 4): fhevc_motion_kernel itself laid out for the +-64 window (k_motion.hip), the same d_mvtab
hipError_t fhevc_launch_motion_big(const FhevcFrames& fr, int range, const uint32_t* d_mvtab, FhevcMotionNode* d_out, int num_cus, hipStream_t stream);

// the shipped P-picture rule (fhevc_p_rule_default): see fasthevc.h; regenerate with tests/quality/fit_p_rule.py
#define FHEVC_P_RULE_WEIGHTS { { 3101, 188, -94, 80, 1149, 1149, 3174, -138, 15748, -351620 }, \
                               { 594, 101, 375, -8, 1078, 1078, -197, 436, 3462, 256745 },      \
                               { 317, -3, 366, -216, 745, 745, 0, 1344, 2780, -18423 } }
// thresholds (Q18) picked on the 1080p pan clip (tests/quality/eval_p.py, profiles/r02_p_slice_motion_rule.json): no split is
// ever FORCED (a wrongly forced split costs several percent of rate on P pictures), splits are FORBIDDEN below scores of
// -3 / -1 / -0.5, and the range stays within +-1 of the co-located depth
#define FHEVC_P_RULE_T_SPLIT { 25952256, 25952256, 25952256 }
#define FHEVC_P_RULE_T_STOP  { 786432, 262144, 131072 }
#define FHEVC_P_RULE_WINDOW  1

// the rule for the wide search (fhevc_p_rule_default_wide): fitted on SAD-mode features of the +-64 search with the reference picture's depths taken
// at the motion-compensated position (tests/quality/p_features_gpu.py on the MI355X, fit_p_rule.py; 176 clips with motion of 0 .. 48 samples per picture)
#define FHEVC_P_RULE_WIDE_WEIGHTS { { 6048, -127, -5256, 95, 876, 1134, 2120, 630, 1548, -447236 }, \
                                    { 1152, -30, -842, -103, 866, 1285, 415, -455, -48, 19539 },     \
                                    { 255, 36, -3, 16, 643, 1206, 0, -243, -177, 342124 } }

// ---- adaptive-QP pre-analysis (k_preanalyze.hip) -----------------------------------------------------------
// d_activity: per frame parts_per_frame doubles, layers concatenated (layer d: ceil(H/P) x ceil(W/P), P = 64 >> d)
hipError_t fhevc_launch_preanalyze(const FhevcFrames& fr, int layers, long long parts_per_frame, double* d_activity,
                                   int num_cus, hipStream_t stream);
