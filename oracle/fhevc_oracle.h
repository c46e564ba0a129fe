/*
 * fhevc_oracle.h -- CPU restatement (plain C) of the CU-partition fast-decision hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it, and only as the checker.  The product path
 * (fasthevc_amd/, include/fasthevc.h) never calls into oracle/.
 *
 * Each function cites the reference file:line (relative to /root/reference) whose algorithm it
 * restates.  Parity pins: tests/golden/ref_*.npz were produced by running the reference's own
 * functions (oracle/_ref/libhmref.so, built by oracle/Makefile from the sources where they lie)
 * through oracle/gen_golden.py; tests/test_oracle_golden.py checks this file against them.
 * The depth classifier (A15) has no reference inference code or weights to pin against
 * (SURVEY.md F4): for that part parity is "unpinned" and the oracle is the definition.
 */
#ifndef FHEVC_ORACLE_H
#define FHEVC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- A14: scan tables (TComRom.cpp:290-345 initZscanToRaster / initRasterToZscan) -------- */
/* 16x16 grid of 4x4 units inside a 64x64 CTU (MaxPartitionDepth 4). */
void fho_init_scan_tables(uint16_t raster_to_zscan[256], uint16_t zscan_to_raster[256]);

/* depth map layout helpers: raster 16x16 <-> z-order (TComDataCU::m_puhDepth, TComDataCU.h:86) */
void fho_depth_raster_to_zorder(const uint8_t raster[256], uint8_t zorder[256]);
void fho_depth_zorder_to_raster(const uint8_t zorder[256], uint8_t raster[256]);

/* Pre-order split-flag serialisation of one CTU (TComSysuCuMDTools.cpp:16-38 writeOutSplitMode,
 * :112-135 readInSplitMode).  A flag is written for every node of size > 8x8 that is visited;
 * returns the number of flags.  depth is raster 16x16, values 0..3. */
int  fho_depth_to_split_flags(const uint8_t depth_raster[256], uint8_t flags[85]);
int  fho_split_flags_to_depth(const uint8_t* flags, int nflags, uint8_t depth_raster[256]);
/* sum |delta depth| over the 256 units (TComSysuCuMDTools.cpp:48-77 compareSplitMode) */
int  fho_compare_split_mode(const uint8_t a[256], const uint8_t b[256]);

/* ---- A5: SATD (TComRdCost.cpp:1527-1824, wrapper calcHAD :297-334) ---------------------- */
uint32_t fho_had2x2(const int16_t* org, int so, const int16_t* cur, int sc);
uint32_t fho_had4x4(const int16_t* org, int so, const int16_t* cur, int sc);
uint32_t fho_had8x8(const int16_t* org, int so, const int16_t* cur, int sc);
/* == TComRdCost::calcHAD / xGetHADs: 8x8 tiles if w,h %8==0, else 4x4, else 2x2; >> (bitDepth-8) */
uint32_t fho_satd(const int16_t* org, int so, const int16_t* cur, int sc, int w, int h, int bit_depth);

/* ---- A6: source-only CTU activity (TEncCu.cpp:1230-1343) -------------------------------- */
int32_t fho_had8x8_src(const int16_t* org, int stride);                 /* xCalcHADs8x8_ISlice */
int32_t fho_ctu_src_hadamard(const int16_t* ctu_org, int stride, int w, int h); /* updateCtuDataISlice */
/* whole frame: one value per CTU in raster order; w/h per CTU = min(64, remaining) */
void    fho_frame_src_hadamard(const int16_t* luma, int stride, int width, int height, int32_t* out);

/* ---- N3: adaptive-QP pre-analysis (TEncPreanalyzer.cpp:64-152 xPreanalyze) -------------- */
/* One layer: AQ parts of part x part samples (64 >> depth), cropped at the picture edge; per part the minimum of the
 * four quadrant variances, activity = 1 + minVar.  activity: ceil(h/part) x ceil(w/part) doubles, raster order.
 * Returns the layer's average activity (sum in raster order / number of parts). */
double fho_preanalyze_layer(const int16_t* luma, int stride, int width, int height, int part, double* activity);

/* TEncCu::xComputeQP (TEncCu.cpp:1093-1117): QP of a CU from its AQ part's activity and the layer average */
int fho_aq_qp(double activity, double avg_activity, int qp_adaptation_range, int base_qp, int qp_bd_offset);

/* ---- A11: lambda (TEncSlice.cpp:433-527 calculateLambda, all-intra path) ----------------- */
double fho_lambda_intra(int qp, int bit_depth);

/* ---- A7: reference samples from the ORIGINAL plane (source-only twin of
 *          TComPattern.cpp:115-539 initIntraPatternChType / fillReferenceSamples) ----------- */
/* ref has 4N+1 samples: ref[2N] = top-left, ref[2N+1+i] = above/above-right i (0..2N-1),
 * ref[2N-1-j] = left/below-left j (0..2N-1).  A 4x4 unit is "available" iff it lies inside the
 * picture and precedes the block in coding order (CTU raster order, z-order inside a CTU) --
 * the rule HM's isAbove/Left/AboveRight/BelowLeftAvailable implement for one slice, one tile. */
void fho_fill_ref(const int16_t* luma, int stride, int width, int height,
                  int x0, int y0, int n, int bit_depth, int16_t* ref);
/* low-level form: flags[k] for the 4N/4+... units in HM order (bottom-left first .. above-right
 * last, fillReferenceSamples's bNeighborFlags); used to pin against the reference function. */
void fho_fill_ref_flags(const int16_t* roi_origin, int pic_stride, const uint8_t* flags,
                        int n, int bit_depth, int16_t* ref);
/* [1 2 1]/4 smoothing or 32x32 strong smoothing (TComPattern.cpp:196-295) */
void fho_filter_ref(const int16_t* ref, int n, int bit_depth, int strong_enabled, int16_t* out);
/* filter decision per mode (TComPattern.cpp:541-566, table TComPrediction.cpp:50-67) */
int  fho_use_filtered_ref(int mode, int n);

/* ---- A8: 35 intra predictors (TComPrediction.cpp:183-473, 731-818) ----------------------- */
/* pred is n x n, row stride n.  Applies DC edge filter / angular edge filters for n <= 16. */
void fho_pred_intra(const int16_t* ref_unfiltered, const int16_t* ref_filtered, int n, int mode,
                    int bit_depth, int16_t* pred);

/* ---- A4: 35-mode first pass per CU node, references from the original plane -------------- */
/* modeBits: 1 context-coded flag at its slice-initial state is approximated by 1 bit, then
 * MPM idx 0 -> +1, idx 1,2 -> +2, non-MPM -> +5 bypass bins (TEncSbac.cpp:643-696); MPM set is
 * the no-neighbour default {PLANAR, DC, VER} (TComDataCU.cpp:1362-1445).
 * cost = (double)satd + (double)bits * sqrt_lambda; ties keep the lower mode index
 * (TEncSearch.cpp:2288, 5385-5408). */
typedef struct { uint32_t satd; uint32_t mode; double cost; } fho_node_cost;
void fho_first_pass_node(const int16_t* luma, int stride, int width, int height,
                         int x0, int y0, int n, int bit_depth, double sqrt_lambda,
                         fho_node_cost* best, uint32_t satd_all[35]);
/* all nodes of one CTU: 1 (64) + 4 (32) + 16 (16) + 64 (8) = 85 in pre-order-by-level layout:
 * index 0 = 64x64; 1..4 = 32x32 raster; 5..20 = 16x16 raster; 21..84 = 8x8 raster.  Nodes that
 * cross the picture edge get satd = 0xFFFFFFFF, mode = 255, cost = -1. */
void fho_first_pass_ctu(const int16_t* luma, int stride, int width, int height,
                        int ctu_x, int ctu_y, int bit_depth, double sqrt_lambda, fho_node_cost out[85]);

/* per node the num modes of smallest first-pass cost, best first (fhevc_intra_first_pass_candidates); modes: 85 * num, 255 = node crosses the edge */
void fho_first_pass_candidates_ctu(const int16_t* luma, int stride, int width, int height, int ctu_x, int ctu_y, int bit_depth,
                                   double sqrt_lambda, int num, uint8_t* modes);

/* ---- A13 / N4 (config 4): source-only motion search per CU node -------------------------------
 * Twin of the integer full search TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848: raster order over the window,
 * strict "<", cost = distortion + TComRdCost::getCostOfVectorWithPredictor, TComRdCost.h:166-174) with the Hadamard
 * distortion HM uses under HADME (TComRdCost::xGetHADs, TComRdCost.cpp:1753-1824; setDistParam(..., bHadamard),
 * TEncSearch.cpp:836) -- evaluated against the PREVIOUS ORIGINAL picture instead of the reconstructed reference (that
 * one exists only inside HM's serial loop), zero motion-vector predictor, samples outside the picture replicated from
 * the border as TComPicYuv::extendPicBorder does (TComPicYuv.cpp:229-270).  The depth decision of a P picture reads
 * these costs next to the co-located depths of its reference picture (TEncFastDepth::predictPicture). */
typedef struct { uint32_t satd_zero, satd_best, cost_best; int16_t mvx, mvy; } fho_motion_node;
/* cost of vector (x, y) in integer samples: Distortion((m_motionLambda * bits) / 65536.0) with m_motionLambda =
 * 65536 * sqrt(lambda) (TComRdCost.cpp:109-114), bits = xGetExpGolombNumberOfBits(x << 2) + ...(y << 2)
 * (quarter-sample units, TComRdCost.cpp:177-190), zero predictor */
uint32_t fho_mv_cost(int x, int y, double sqrt_lambda);
/* one CTU: out[85] in the node order of fho_first_pass_ctu; nodes crossing the picture edge get 0xFFFFFFFF / mv 0.
 * range <= 8.  cost of a node at mv = (sum of its 8x8 tile SATDs >> (bit_depth - 8)) + fho_mv_cost(mv). */
#define FHO_MOTION_SATD 0
#define FHO_MOTION_SAD 1
/* dist: FHO_MOTION_SAD = HM's integer-search distortion (pinned to the reference's xPatternSearch), FHO_MOTION_SATD = Hadamard */
void fho_motion_ctu_dist(const int16_t* cur, int cur_stride, const int16_t* ref, int ref_stride, int width, int height,
                         int ctu_x, int ctu_y, int bit_depth, int range, double sqrt_lambda, int dist, fho_motion_node out[85]);
void fho_motion_ctu(const int16_t* cur, int cur_stride, const int16_t* ref, int ref_stride, int width, int height,
                    int ctu_x, int ctu_y, int bit_depth, int range, double sqrt_lambda, fho_motion_node out[85]);
/* Depth range of a P picture's CTU from its motion nodes and the co-located depths of the reference picture: the integer rule
 * that include/fasthevc.h specifies for fhevc_p_depth_range (this is its independent restatement). */
typedef struct { int32_t w[3][10]; int32_t t_split[3], t_stop[3]; int32_t window; } fho_p_rule;
int32_t fho_ilog2_q8(uint32_t x);   /* floor(256 log2 x) by integer squaring, x >= 1 */
void fho_p_motion_compensated_depth(const fho_motion_node nodes[85], const uint8_t* prev_map, int width, int height, int ctu, uint8_t out[256]);
void fho_p_node_depth(const fho_motion_node nodes[85], const uint8_t* prev_map, int width, int height, int ctu, uint8_t out[256]);
void fho_p_depth_range(const fho_motion_node nodes[85], const uint8_t prev_depth[256], int valid_w, int valid_h,
                       int qp, const fho_p_rule* rule, uint8_t depth_min[256], uint8_t depth_max[256]);

/* ---- A15: depth classifier, integer-valued restatement --------------------------------- */
/* Architecture follows matlab/dataExtraction/Train...Example.m:75-96 run convolutionally on the
 * 64x64 CTU: conv3x3x16 pad1 -> ReLU -> maxpool2 -> conv3x3x32 pad1 -> ReLU -> maxpool2 ->
 * conv3x3x64 pad1 -> ReLU -> FC(2) heads.  BN is folded; values are fixed-point integers so that
 * fp32 accumulation of bf16 operands on the GPU is exact.  See HISTORY.md section 4. */
typedef struct {
  int32_t shift[3];
  int8_t  w1[16 * 9];          /* [oc][ky][kx]            */
  int32_t b1[16];
  int8_t  w2[32 * 16 * 9];     /* [oc][ic][ky][kx]        */
  int32_t b2[32];
  int8_t  w3[64 * 32 * 9];     /* [oc][ic][ky][kx]        */
  int32_t b3[64];
  int8_t  wh64[2 * 4096];      /* [cls][y8][x8][c64] on sumpool2x2(a3) */
  int32_t bh64[2];
  int8_t  wh32[2 * 4096];      /* [cls][y8][x8][c64] on a quadrant of a3 */
  int32_t bh32[2];
  int8_t  wh16[2 * 1024];      /* [cls][y4][x4][c64] on a 4x4 window of a3 */
  int32_t bh16[2];
  int32_t qp_bias[3 * 52];     /* [level 64,32,16][slice QP]: added to the split logit */
} fho_weights;

/* The Bayesian-optimisation network FAMILY of the reference (matlab/dataExtraction/OptimizeDeepNeuralNetworksUsingBayesianOptimizationExample.m
 * :103-106 NetworkDepth in [1, 3]; :233-259 convBlock(3, F, depth) x 3 with F = round(32 / sqrt(depth)) filters, 2F, 4F; max-pool after the
 * first two blocks; :367-373 convBlock = depth x (conv3x3 pad 1 + BN + ReLU)): channel widths c1, c2, c3 and `depth` convolutions per
 * block, the same fixed-point arithmetic as fho_weights (int8 weights, int32 biases, one requant shift per convolution, activations
 * clamp((acc + b) >> s, 0, 255)), the same three heads on the last block's map.  depth = 1, widths 16 / 32 / 64 is fho_weights' network.
 *   w[b][j]: convolution j of block b, [oc][ic][ky][kx]; its input width is 1 (b = 0, j = 0), the block's own width (j > 0) or the
 *   previous block's width (j = 0).  shift[b][j], bias[b][j] likewise. */
typedef struct {
  int32_t c[3];                /* channel widths of the three blocks */
  int32_t depth;               /* convolutions per block, 1..3 */
  int32_t shift[3][3];
  const int8_t* w[3][3];
  const int32_t* b[3][3];
  const int8_t* wh64; const int8_t* wh32; const int8_t* wh16;   /* [cls][y][x][c3] as in fho_weights */
  int32_t bh64[2], bh32[2], bh16[2];
  int32_t qp_bias[3 * 52];
} fho_family;
void fho_cnn_ctu_family(const fho_family* w, const int8_t* ctu, int qp, int32_t logits[21][2]);
void fho_predict_frame_family(const fho_family* w, const int16_t* luma, int stride, int width, int height,
                              int bit_depth, int qp, uint8_t* depth_map, int32_t* logits_out);

/* ctu: 64x64 centred 8-bit samples (value-128), row stride 64, zeros outside the picture.
 * logits[21][2]: 0 = 64-level, 1..4 = 32-level quadrants (raster), 5..20 = 16-level blocks (raster). */
void fho_cnn_ctu(const fho_weights* w, const int8_t* ctu, int qp, int32_t logits[21][2]);
/* optional taps for debugging: a1 [32][32][16], a2 [16][16][32], a3 [16][16][64] (uint8) */
void fho_cnn_ctu_debug(const fho_weights* w, const int8_t* ctu, int qp, uint8_t* a1, uint8_t* a2, uint8_t* a3,
                       int32_t logits[21][2]);
/* logits -> raster 16x16 depth map.  valid_w/valid_h = in-picture part of the CTU; nodes crossing
 * the picture edge are forced to split (TEncCu.cpp:574,894 bBoundary); units outside get 0. */
void fho_depth_from_logits(const int32_t logits[21][2], int valid_w, int valid_h, uint8_t depth_raster[256]);
/* soft decisions (hm_patch soft hook): splits surer than +margin_split -> depth_min, splits not rejected by more than
 * -margin_stop -> depth_max */
void fho_depth_range_from_logits(const int32_t logits[21][2], int valid_w, int valid_h, int margin_split, int margin_stop,
                                 uint8_t depth_min[256], uint8_t depth_max[256]);
/* one margin pair per split level (64, 32, 16): the calibration of HISTORY.md section 4 */
void fho_depth_range_from_logits_levels(const int32_t logits[21][2], int valid_w, int valid_h, const int32_t margin_split[3],
                                        const int32_t margin_stop[3], uint8_t depth_min[256], uint8_t depth_max[256]);
/* The same decisions as one 21-bit word per CTU: bit 0 = 64x64 split, bits 1..4 = 32x32 quadrants (raster),
 * bits 5..20 = 16x16 blocks (raster); a bit is set only under split parents and inside the picture (forced
 * splits at the picture edge included).  This is the reference's pre-order split-flag serialisation
 * (TComSysuCuMDTools.cpp:16-38) in fixed positions; 4 bytes instead of 256 for the multi-GPU all-gather. */
uint32_t fho_flags_from_logits(const int32_t logits[21][2], int valid_w, int valid_h);
void fho_depth_from_flags(uint32_t flags, int valid_w, int valid_h, uint8_t depth_raster[256]);
/* gather + centre one CTU from the frame (8- or 10-bit Pel) */
void fho_load_ctu(const int16_t* luma, int stride, int width, int height, int ctu_x, int ctu_y,
                  int bit_depth, int8_t ctu[64 * 64]);
/* whole frame: depth maps (numCtus*256, raster per CTU) and optionally logits (numCtus*42) */
void fho_predict_frame(const fho_weights* w, const int16_t* luma, int stride, int width, int height,
                       int bit_depth, int qp, uint8_t* depth_map, int32_t* logits_out);

#ifdef __cplusplus
}
#endif
#endif
