/*
 * fasthevc.h -- C ABI of the MI355X-native CU-partition fast-decision path.
 *
 * This is the drop-in boundary for HM's source/Lib/TLibEncoder (reference: omricarmi/FastHEVC, an HM-16.14
 * fork).  The reference has no plugin/FFI layer: the boundary the path sits behind is HM's own C++ class
 * surface, TEncSlice::compressSlice -> TEncCu::compressCtu -> TEncCu::xCompressCU (TEncSlice.cpp:698-983,
 * TEncCu.cpp:252-288, 496-1058).  A patched TEncSlice/TEncCu (hm_patch/) calls these entry
 * points from three places; everything else in HM stays untouched.  Plain pointers and sizes only.
 *
 * Conventions:
 *   - return 0 on success, a negative FHEVC_E_* code on failure; the caller falls back to stock full RDO for
 *     that picture -- the library never aborts the encode.  (HM itself reports errors by assert/exit:
 *     TEncCu.cpp:1055-1057.)
 *   - all buffers are caller-owned; calls are synchronous from the single encoder thread unless the entry
 *     point takes a stream; a context is not thread-safe (HM is not re-entrant either: TEncCu.cpp:50,126-131).
 *   - there is NO CPU backend: fhevc_create fails with FHEVC_E_NO_DEVICE when no gfx950 device is usable.
 *
 * Depth-map convention: per CTU 256 bytes, raster 16x16 of 4x4 luma units, value = CU depth 0..3
 * (0 = 64x64 ... 3 = 8x8); equals TComDataCU::getDepth(g_auiRasterToZscan[r]) (TComDataCU.h:86,207-211,
 * TComRom.cpp:284-287).  Units outside the picture carry 0; CUs that cross the picture edge are marked split,
 * as HM forces them to be (TEncCu.cpp:574, 894, 915).
 */
#ifndef FASTHEVC_H
#define FASTHEVC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FHEVC_OK               0
#define FHEVC_E_INVALID       -1   /* bad argument / unsupported geometry */
#define FHEVC_E_NO_DEVICE     -2   /* no usable gfx950 device, or HIP runtime error at create */
#define FHEVC_E_HIP           -3   /* HIP runtime error during a call (fhevc_last_error has the text) */
#define FHEVC_E_WEIGHTS       -4   /* weight blob missing / malformed */
#define FHEVC_E_NOMEM         -5
#define FHEVC_E_STATE         -6   /* call not valid in this state (e.g. predict before weights are set) */

#define FHEVC_BACKEND_HIP      1

#define FHEVC_NODES_PER_CTU   85   /* 1 + 4 + 16 + 64 CU nodes of sizes 64, 32, 16, 8 */
#define FHEVC_LOGITS_PER_CTU  42   /* 21 split decisions x 2 classes */

typedef struct fhevc_ctx fhevc_ctx; /* opaque: device buffers, streams, weights, timing events */

typedef struct {
  int width, height;        /* luma picture size (TComPicYuv::getWidth/getHeight(COMPONENT_Y)) */
  int bit_depth;            /* sps.getBitDepth(CHANNEL_TYPE_LUMA): 8..12 */
  int ctu_size;             /* 64 (the reference's dumper hard-codes it: HARP_Defines.h:28-29) */
  int max_depth;            /* 3: 8x8 leaves (MaxPartitionDepth 4) */
  int num_devices;          /* devices of THIS process (<= 16): > 1 makes the host-buffer entry points (fhevc_predict_frame[_range], fhevc_predict_frames)
                               shard CTU-row bands / runs of pictures over them (fhevc_band) and gather the maps into the caller's buffer;
                               a device that fails is dropped and its share redone on another (fhevc_stats.devices_failed), never an abort.
                               Entry points that take DEVICE pointers, and the parity / feature entry points, run on device_ids[0] */
  const int* device_ids;    /* HIP device ordinals, NULL = {0}; an ordinal may repeat (two queues on one device: tests) */
  const char* weights_path; /* FHW1 blob (fasthevc_amd/weights.py), or NULL and call fhevc_set_weights */
  int backend;              /* FHEVC_BACKEND_HIP */
  int max_frames;           /* frames per batched call the context sizes its staging buffers for (>= 1) */
} fhevc_cfg;

/* per-node result of the 35-mode first pass (twin of TEncSearch::estIntraPredLumaQT's first pass,
 * TEncSearch.cpp:2271-2295, with reference samples taken from the ORIGINAL plane) */
typedef struct {
  uint32_t satd;            /* SATD of the cheapest mode (TComRdCost::xGetHADs) */
  uint32_t mode;            /* its intra mode 0..34; 255 when the node crosses the picture edge */
  double   cost;            /* satd + modeBits * sqrt(lambda); -1 for skipped nodes */
} fhevc_node_cost;

typedef struct {
  uint64_t frames;          /* frames processed since create */
  uint64_t ctus;            /* CTU depth decisions delivered */
  uint64_t bytes_h2d, bytes_d2h;
  uint64_t kernels_launched;
  double   ms_h2d, ms_kernels, ms_d2h;   /* accumulated, HIP-event timed (host-buffer entry points only) */
  double   last_cnn_ms, last_hadamard_ms, last_first_pass_ms; /* last launch of each kernel */
  uint64_t devices;         /* devices this context works with right now (cfg.num_devices minus the failed ones) */
  uint64_t devices_failed;  /* devices that could not be brought up at fhevc_create or were dropped after a failed call */
} fhevc_stats;

int  fhevc_create(fhevc_ctx** out, const fhevc_cfg* cfg);
void fhevc_destroy(fhevc_ctx* ctx);
/* weights from memory instead of cfg.weights_path (same FHW1 bytes) */
int  fhevc_set_weights(fhevc_ctx* ctx, const void* blob, size_t bytes);

/* One picture, host buffers, synchronous.  Called once per picture before the CTU loop of
 * TEncSlice::compressSlice (TEncSlice.cpp:792) with pcPic->getPicYuvOrg()->getAddr(COMPONENT_Y) / getStride
 * (TComPicYuv.h:121-147).  depth_map: numCtus*256 bytes.  ctu_src_hadamard: optional, numCtus values equal to
 * TEncCu::updateCtuDataISlice(ctu, w, h) (TEncCu.cpp:1324-1343) -- what TEncSlice::calCostSliceI sums
 * (TEncSlice.cpp:663-695).  qp is the slice QP: it selects a per-QP prior on the split decisions (the reference
 * stores QP in its labels, CShow_PredResiReco.h:93, but its MATLAB pipeline never reads it); slice_type is
 * reserved (I slices only this round). */
int  fhevc_predict_frame(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int qp, int slice_type,
                         uint8_t* depth_map, int32_t* ctu_src_hadamard);

/* Host batch ("one CTU's depth map delivered to host memory"): num_frames pictures in host memory ->
 * depth maps (and, optionally, per-CTU source Hadamards) in host memory.  luma: sample_bytes = 2 (int16 Pel planes as HM lays
 * them out) or 1 (uint8 planes of 8-bit content, e.g. straight from an 8-bit .yuv file: TVideoIOYuv.cpp:249-330 reads the
 * same bytes and widens them); frame f starts at luma + f * frame_stride_samples.  The pictures travel in chunks of
 * cfg.max_frames through two streams, so that the upload of chunk k+1 and the download of chunk k-1 overlap the kernels of
 * chunk k.  Buffers from fhevc_alloc_host (or otherwise pinned) are read / written by DMA directly; pageable buffers go
 * through the context's pinned staging ring (one extra host copy per chunk).  Synchronous. */
int  fhevc_predict_frames(fhevc_ctx* ctx, const void* luma, int sample_bytes, int stride_samples, long long frame_stride_samples,
                          int num_frames, int qp, uint8_t* depth_map /* num_frames * numCtus * 256 */,
                          int32_t* ctu_src_hadamard /* num_frames * numCtus, or NULL */);
/* Library-side luma reader, host code only -- no device work, callable without a context: the luma planes of
 * `num_frames` pictures of a planar YUV file, from picture `first_frame` on, straight into `dst` (fhevc_alloc_host memory: no intermediate
 * copy between the file and the DMA source of fhevc_predict_frames).  What TVideoIOYuv::read does for COMPONENT_Y (TVideoIOYuv.cpp:249-330,
 * 675-760): one byte per sample for file_bit_depth 8, two little-endian bytes above; chroma (chroma_format 400 / 420 / 422 / 444) is skipped
 * with a seek, never read; the plane is padded on the right and at the bottom by edge replication from file_width x file_height to
 * dst_width x dst_height (the conformance size: ConformanceWindowMode 1 pads to a multiple of the minimum CU size 8, TAppEncCfg.cpp:1310-1370),
 * and left-shifted from the file's to the internal bit depth (scalePlane, TVideoIOYuv.cpp:70-84, :730; internal < file is not supported).
 * dst_sample_bytes 2: int16 Pel samples; 1: uint8 samples (8-bit file at 8-bit internal depth only).  Returns the number of pictures read
 * (fewer than asked for at the end of the file), or FHEVC_E_INVALID / FHEVC_E_STATE (file cannot be opened / is shorter than one picture). */
int  fhevc_read_yuv_luma(const char* path, int file_width, int file_height, int file_bit_depth, int chroma_format, long long first_frame,
                         int num_frames, int dst_width, int dst_height, int internal_bit_depth, int dst_sample_bytes, void* dst,
                         long long dst_stride_samples, long long dst_frame_stride_samples);
/* pinned host memory for the batch entry point (a .yuv reader can read luma planes straight into it) */
void* fhevc_alloc_host(fhevc_ctx* ctx, size_t bytes);
void  fhevc_free_host(fhevc_ctx* ctx, void* p);

/* Soft decisions for the xCompressCU hook (hm_patch/): depth_min holds only the splits whose logit difference exceeds
 * +margin_split, depth_max every split not rejected by more than -margin_stop (both >= 0, logit units; 0, 0: both maps
 * equal the map of fhevc_predict_frame).  The hook forces a split while depth < depth_min, forbids one at
 * depth >= depth_max and leaves the depths in between to HM's own RD search (TEncCu.cpp:576-849, 892).  margin_split
 * costs almost no time (the parent CU is evaluated as well), margin_stop costs the recursion it allows. */
int  fhevc_predict_frame_range(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int qp, int slice_type,
                               int margin_split, int margin_stop, uint8_t* depth_min, uint8_t* depth_max,
                               int32_t* ctu_src_hadamard);

/* == TComRdCost::calcHAD(bitDepth, org, strideOrg, cur, strideCur, w, h) (TComRdCost.cpp:297-334) and
 * xGetHADs (:1753-1824); host buffers, w,h <= 64.  Parity entry point. */
int  fhevc_satd(fhevc_ctx* ctx, const int16_t* org, int org_stride, const int16_t* cur, int cur_stride,
                int w, int h, int bit_depth, uint32_t* out);

/* 35-mode first pass for every CU node of every CTU of one picture (host buffers).  out: numCtus * 85 entries,
 * node order 64x64, 32x32 (raster), 16x16 (raster), 8x8 (raster).  lambda as TEncSlice::calculateLambda
 * (TEncSlice.cpp:433-527) would give it for qp; pass qp. */
int  fhevc_intra_first_pass(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int qp, fhevc_node_cost* out);

/* Parity entry point: the same pass, returning EVERY (node, mode) pair -- all: numCtus * 85 * 35 entries, [CTU][node][mode]
 * with satd = xGetHADs of that mode's prediction, mode = the mode, cost = satd + modeBits * sqrt(lambda) (nodes crossing the
 * picture edge: satd 0xFFFFFFFF, mode 255, cost -1); best (optional) as fhevc_intra_first_pass.  Lets a test see the
 * predictors and SATDs of modes that never win. */
int  fhevc_intra_first_pass_all(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int qp, fhevc_node_cost* best,
                                fhevc_node_cost* all);

/* What HM's own first pass is for: the candidate list of estIntraPredLumaQT (TEncSearch.cpp:2271-2320).  Per node the num_candidates (1..8; HM: 8 for
 * 8x8 PUs, 3 above) modes of smallest cost, best first, an earlier mode ahead of a later one of equal cost (xUpdateCandList,
 * TEncSearch.cpp:5385-5408); modes: numCtus * 85 * num_candidates bytes, 255 for nodes crossing the picture edge.  The costs come from ORIGINAL
 * neighbours (HM's own pass sees reconstructed ones inside its serial loop) and the mode-bit model of fhevc_intra_first_pass; HM still appends its
 * most-probable modes itself.  The selection runs on the device (85 x num_candidates bytes per CTU come back).  hm_patch: FHEVC_FIRST_PASS=1. */
int  fhevc_intra_first_pass_candidates(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int qp, int num_candidates, uint8_t* modes);

/* first pass over a device-resident batch (layout and band arguments as fhevc_predict_frames_device below);
 * d_out: (num_frames * band CTUs) * 85 entries in HBM.  Asynchronous with respect to the host. */
int  fhevc_intra_first_pass_device(fhevc_ctx* ctx, const void* d_luma, int sample_bytes, int stride_samples,
                                   long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                   int qp, fhevc_node_cost* d_out, void* stream);

/* Device-resident batch: num_frames pictures already in HBM, CTU rows [ctu_row_begin, ctu_row_end) of each.
 * d_luma: sample_bytes = 2 -> int16 Pel plane(s) as HM lays them out, 1 -> uint8 (8-bit content);
 * frame f starts at d_luma + f * frame_stride_samples.  Outputs are device pointers, compact over the band:
 * entry ((f * band_rows + (row - ctu_row_begin)) * ctus_per_row + col).  d_hadamard / d_logits / d_flags may be NULL.
 * stream: hipStream_t.  NULL = the context's own stream, a BLOCKING stream: work on it is ordered after everything issued
 * earlier on the legacy default stream (stream 0) and before everything issued later on it, so a caller that lives on the
 * default stream needs no extra synchronisation; a caller on its own non-blocking stream passes that stream.
 * Asynchronous with respect to the host. */
int  fhevc_predict_frames_device(fhevc_ctx* ctx, const void* d_luma, int sample_bytes, int stride_samples,
                                 long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                 int qp, uint8_t* d_depth_map, int32_t* d_hadamard, int32_t* d_logits, uint32_t* d_flags,
                                 void* stream);

/* the same with soft decisions (see fhevc_predict_frame_range); d_depth_max may be NULL; d_flags follow d_depth_map */
int  fhevc_predict_frames_device_range(fhevc_ctx* ctx, const void* d_luma, int sample_bytes, int stride_samples,
                                       long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                       int qp, int margin_split, int margin_stop, uint8_t* d_depth_map, uint8_t* d_depth_max, int32_t* d_hadamard,
                                       int32_t* d_logits, uint32_t* d_flags, void* stream);

/* The 21 split decisions of a CTU as one word (bit 0 = 64x64, bits 1..4 = 32x32 quadrants, bits 5..20 = 16x16
 * blocks; set only under split parents, forced splits at the picture edge included): the optional d_flags output
 * above, 4 bytes per CTU instead of 256 -- what the ranks of a node all-gather.  This call expands gathered words
 * back into depth maps on the device: num_frames * numCtus words, whole pictures in CTU raster order. */
int  fhevc_expand_depth_flags_device(fhevc_ctx* ctx, const uint32_t* d_flags, int num_frames, uint8_t* d_depth_map, void* stream);

/* Adaptive-QP pre-analysis == TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64-152), called by TEncGOP before
 * compressSlice when --AdaptiveQP is on (TEncGOP.cpp: m_pcPreanalyzer->xPreanalyze(pcPic)).  max_aq_depth =
 * TEncPic's uiMaxAdaptiveQPDepth (1..4): layer d has parts of 64 >> d samples, ceil(height/P) x ceil(width/P) of
 * them, raster order.  activity: all layers concatenated (fhevc_aq_parts gives the offsets) = what
 * TEncQPAdaptationUnit::getActivity returns; avg_activity: max_aq_depth values = TEncPicQPAdaptationLayer::
 * getAvgActivity.  Doubles are bit-identical to HM's.  Picture width/height must be multiples of 8. */
int  fhevc_aq_parts(int width, int height, int max_aq_depth, long long* layer_offsets /* max_aq_depth + 1, may be NULL */);
int  fhevc_preanalyze(fhevc_ctx* ctx, const int16_t* luma, int stride_samples, int max_aq_depth,
                      double* activity, double* avg_activity);
/* == TEncCu::xComputeQP (TEncCu.cpp:1093-1117) for every AQ part: qp[i] = Clip3(-qp_bd_offset, 51, base_qp +
 * floor(6*log2(normalised activity) + 0.49999)); activity/avg_activity/qp in fhevc_preanalyze's layout.  Runs on
 * the host with the same libm calls HM makes (pow, log, floor), so the integers are HM's. */
int  fhevc_aq_qp(const double* activity, const double* avg_activity, int width, int height, int max_aq_depth,
                 int qp_adaptation_range, int base_qp, int qp_bd_offset, int8_t* qp);
/* device-resident batch; d_activity holds num_frames * fhevc_aq_parts() doubles in whole-picture layout, of which
 * this call writes the parts inside CTU rows [ctu_row_begin, ctu_row_end) */
int  fhevc_preanalyze_frames_device(fhevc_ctx* ctx, const void* d_luma, int sample_bytes, int stride_samples,
                                    long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                    int max_aq_depth, double* d_activity, void* stream);

/* ---- config 4 (P slices): source-only motion search per CU node ----------------------------------------------------
 * For every CU node of every CTU (node order as fhevc_node_cost): integer full search over [-search_range, search_range]^2
 * in the PREVIOUS ORIGINAL picture, raster order and strict "<" as TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848),
 * cost = distortion + TComRdCost::getCostOfVectorWithPredictor (TComRdCost.h:166-174; zero predictor, iCostScale 2, lambda of
 * slice QP qp), samples outside the picture replicated from the border (TComPicYuv::extendPicBorder).  Distortion
 * (fhevc_set_motion_distortion):
 *   FHEVC_MOTION_SAD   what HM's integer search uses (xPatternSearch's setDistParam selects DF_SAD, TComRdCost.cpp:205-236): in this
 *                      mode vector, distortion and cost equal what the reference's own xPatternSearch returns on the same planes;
 *   FHEVC_MOTION_SATD  (default) Hadamard SATD (TComRdCost::xGetHADs) at integer positions.  HM applies Hadamard to the fractional
 *                      refinement only (HadamardME, TEncSearch.cpp:836): this is the library's own choice, the one the P-picture
 *                      rule of fhevc_p_depth_range was fitted on.
 * HM's own search runs on reconstructed references inside its serial CTU loop; this is its source-only twin, available for
 * the whole picture before that loop starts.  search_range 1..8 in either mode; 9..64 (HM's cfg: SearchRange 64; xPatternSearch,
 * TEncSearch.cpp:3786-3848, is bit-depth agnostic) in the SAD mode: the same full search, same result as xPatternSearch over that
 * window -- 8-bit content on a kernel laid out for 16 641 vectors per node around v_qsad_pk_u16_u8 (k_motion_wide.hip), content above
 * 8 bit (16-bit planes; cfg/encoder_lowdelay_P_main10.cfg) on the 16-bit SAD kernel laid out for the wide window (k_motion.hip, round 4;
 * ~20 x slower than the byte kernel, still ~100 x HM's own search per core); 9..64 in the SATD mode: FHEVC_E_INVALID.
 * (HM's P configuration itself runs the TZ search, cfg/encoder_lowdelay_P_main.cfg:34 FastSearch 1 -> xPatternSearchFast,
 * TEncSearch.cpp:3850: the exhaustive twin is a superset of what TZ visits and serves as a source-only feature.) */
#define FHEVC_MOTION_SATD 0
#define FHEVC_MOTION_SAD  1
int  fhevc_set_motion_distortion(fhevc_ctx* ctx, int mode);
#define FHEVC_MOTION_MAX_RANGE 8        /* SATD mode */
#define FHEVC_MOTION_SAD_MAX_RANGE 64   /* SAD mode, any bit depth */
typedef struct {
  uint32_t satd_zero;       /* distortion (SATD or SAD) at vector (0, 0) */
  uint32_t satd_best;       /* distortion at the cheapest vector */
  uint32_t cost_best;       /* its distortion + vector cost; 0xFFFFFFFF in all three for nodes crossing the picture edge */
  int16_t  mvx, mvy;        /* the cheapest vector, integer samples */
} fhevc_motion_node;
/* one picture pair, host buffers (both planes with the same stride), synchronous; out: numCtus * 85 */
int  fhevc_motion_search(fhevc_ctx* ctx, const int16_t* cur_luma, const int16_t* ref_luma, int stride_samples, int qp,
                         int search_range, fhevc_motion_node* out);
/* device-resident batch (layout as fhevc_predict_frames_device): frame f = 1 .. num_frames-1 is searched in frame f-1;
 * d_out: (num_frames - 1) * band CTUs * 85 nodes in HBM */
int  fhevc_motion_search_device(fhevc_ctx* ctx, const void* d_luma, int sample_bytes, int stride_samples,
                                long long frame_stride_samples, int num_frames, int ctu_row_begin, int ctu_row_end,
                                int qp, int search_range, fhevc_motion_node* d_out, void* stream);

/* Depth range of every 4x4 unit of a P picture's CTU from its motion nodes and the co-located depths of its reference picture
 * ("inter-CU depth reuse", BASELINE config 4).  Host-side integer arithmetic, no device work.  Per split decision (64->32,
 * 32->16, 16->8) a linear score over nine features of the node, all in 1/256 units (L(x) = floor(256 log2 x) by integer
 * squaring, lgN = 2 * 256 * log2(node size), qn = 256 * qp / 6):
 *   f0 = L(satd_best + 1) - lgN - qn          residual per sample against the quantiser step
 *   f1 = L(cost_best - sum of the four children's cost_best (clamped at 0) + 1) - lgN - qn     what splitting the search gains
 *   f2 = L(sum of the children's satd_best + 1) - lgN - qn
 *   f3 = L(satd_zero + 1) - L(satd_best + 1)  how much motion compensation helps at all
 *   f4, f5 = 256 if the largest / smallest co-located depth of the reference picture inside the node is deeper than the node
 *   f6 = 256 if the largest co-located depth is at least two levels deeper
 *   f7 = 64 * number of children whose cheapest vector differs from the node's
 *   f8 = 8 * qp
 * score = sum w[level][i] * f_i + w[level][9]   (Q18).  depth_min follows the splits with score > t_split[level] top-down,
 * depth_max those with score >= -t_stop[level]; CUs crossing the picture edge are split in both, units outside get 0; then
 * both are clipped to the co-located depth +- window when window < 4.  The xCompressCU hook forces a split while depth <
 * depth_min and forbids one at depth >= depth_max (as for I pictures). */
typedef struct {
  int32_t w[3][10];
  int32_t t_split[3], t_stop[3];
  int32_t window;
} fhevc_p_rule;
void fhevc_p_rule_default(fhevc_p_rule* rule);   /* the shipped rule (fitted on the reference's own P-picture decisions) */
/* the same rule fitted on SAD-mode features of the +-64 search with the reference picture's depths taken at the motion-compensated position
 * (fhevc_p_motion_compensated_depth): what goes with search ranges above 8 */
void fhevc_p_rule_default_wide(fhevc_p_rule* rule);
int  fhevc_p_depth_range(const fhevc_motion_node* nodes /* 85 */, const uint8_t* prev_depth /* 256, raster */, int valid_w,
                         int valid_h, int qp, const fhevc_p_rule* rule, uint8_t* depth_min /* 256 */, uint8_t* depth_max /* 256 */);

/* "Inter-CU depth reuse" for content that moves: the reference picture's depths seen THROUGH the motion.  For every 4x4 unit of CTU `ctu`
 * (raster CTU index) the depth the reference picture's map holds where the unit's centre lands when displaced by the cheapest vector
 * of the 16x16 node the unit lies in (a node crossing the picture edge: the vector of its 32x32 node, then of the CTU, then zero),
 * positions clamped to the picture.  prev_map: numCtus * 256 depths of the reference picture (raster per CTU, as fhevc_predict_frame writes
 * them and TComDataCU::getDepth holds them); out: 256, the prev_depth argument of fhevc_p_depth_range.  With zero vectors this is the
 * co-located map.  Host-side integer logic, no device work, no context. */
int  fhevc_p_motion_compensated_depth(const fhevc_motion_node* nodes /* 85 */, const uint8_t* prev_map, int width, int height, int ctu,
                                      uint8_t* out /* 256 */);

/* The same through the CU NODES of the current picture (round 4): a displaced depth map is not aligned to the current picture's CU grid -- a 64x64 CU of the
 * reference picture lands on two half CTUs here -- and forcing such a map costs more than the co-located one (HISTORY.md section 4b).  This form asks, top-down
 * per node of CTU `ctu`, for the reference picture's depth at the node's CENTRE displaced by the node's cheapest vector (a node crossing the picture edge: its
 * parent's vector; the CTU node: zero): a node whose answer is not deeper than its own level becomes one CU of that depth, otherwise its four children are
 * asked (16x16 nodes: depth 2, or 3 when the answer is 3).  out is a quadtree-consistent partition on the CURRENT grid.  With zero vectors and a prev_map
 * that is itself a partition this is the co-located map.  Measured (profiles/r04_p_slice_node_*.json): one global pan of 32 samples per picture
 * -0.26 % BD-rate with the +-1 window where the per-unit form costs +1.76 % and the co-located map -0.10 %; elsewhere equal to the co-located map. */
int  fhevc_p_node_depth(const fhevc_motion_node* nodes /* 85 */, const uint8_t* prev_map, int width, int height, int ctu, uint8_t* out /* 256 */);

/* CTU-row band of rank `rank` out of `world`: rows [begin, end) */
int  fhevc_band(int ctu_rows, int rank, int world, int* begin, int* end);

/* average duration in ms of the dominant kernels over launches since the last reset, measured with HIP
 * events on the launch stream; which: 0 = depth CNN, 1 = source Hadamard, 2 = first pass, 3 = pre-analysis, 4 = motion search */
int  fhevc_kernel_timing(fhevc_ctx* ctx, int which, int reset, double* avg_ms, uint64_t* launches);
int  fhevc_enable_kernel_timing(fhevc_ctx* ctx, int on);

/* Arithmetic of the depth classifier's conv2 / conv3 (both forms deliver the same integers, bit for bit):
 *   FHEVC_CNN_ARITH_I8  (default): v_mfma_i32_32x32x32_i8 on activations kept as signed bytes, three workgroups per CU;
 *   FHEVC_CNN_ARITH_F16          : 16-bit MFMAs (f16 activations), two workgroups per CU.
 * The environment variable FHEVC_CNN_ARITH=i8|f16 sets the initial value at fhevc_create; a change takes effect at the next launch. */
#define FHEVC_CNN_ARITH_I8   8
#define FHEVC_CNN_ARITH_F16 16
int  fhevc_set_cnn_arith(fhevc_ctx* ctx, int arith);
int  fhevc_get_cnn_arith(const fhevc_ctx* ctx);   /* FHEVC_CNN_ARITH_*, or a negative status */

int  fhevc_get_stats(fhevc_ctx* ctx, void* out, size_t size); /* copies min(size, sizeof(fhevc_stats)) */
const char* fhevc_last_error(fhevc_ctx* ctx);
const char* fhevc_version(void);

#ifdef __cplusplus
}
#endif
#endif
