// ref_rdo_harness.cpp -- drives the REAL reference decision path for one intra picture:
//   TEncSlice::compressSlice -> TEncCu::compressCtu -> xCompressCU -> xCheckRDCostIntra -> TEncSearch::estIntraPred*QT
// (TEncSlice.cpp:698-983, TEncCu.cpp:252-1058, TEncSearch.cpp:2178-2712) without TEncTop / TEncGOP / TAppEncoder,
// which need OpenCV (Src_HARP) and are therefore not buildable in this image.
//
// TEST INFRASTRUCTURE ONLY (oracle/_ref/libhmref.so).  This file contains NO reference code: it constructs the
// reference's own objects (TEncCfg, TComSPS/PPS, TComPic, TEncSlice, TEncCu, TEncSearch, TComTrQuant, TComRdCost,
// the RD-SBAC coder arrays) and wires them the way TEncTop::create/init do (TEncTop.cpp:89-145, 183-231), with the
// parameter values of cfg/encoder_intra_main.cfg / encoder_intra_main10.cfg.  Member wiring uses the usual
// test-harness access trick (private -> public for the reference headers only); no layout changes.
//
// Used for: depth-map labels for the trainer, the "reference" CPU baseline of bench.py (HM's own xCompressCU timed
// on the host), and the RD-cost check of the xCompressCU hook (hm_patch/) through FHEVC_HOOK builds.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <list>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#define private public
#define protected public
#ifdef FHEVC_HOOK
#include "TEncFastDepth.h"
#include "TEncCu.h"  // the patched copy: oracle/_ref/hook/src precedes the reference tree on the include path
#endif
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/ContextModel.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComSlice.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComTU.h"
#include "TLibCommon/TComPattern.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibCommon/TComBitStream.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncCu.h"
#include "TLibEncoder/TEncSlice.h"
#include "TLibEncoder/TEncSearch.h"
#include "TLibEncoder/TEncSampleAdaptiveOffset.h"
#include "libmd5/MD5.h"
#include "TLibEncoder/TEncEntropy.h"
#include "TLibEncoder/TEncSbac.h"
#include "TLibEncoder/TEncBinCoderCABAC.h"
#include "TLibEncoder/TEncBinCoderCABACCounter.h"
#include "TLibEncoder/TEncRateCtrl.h"
#include "TLibEncoder/TEncPic.h"
#include "TLibEncoder/TEncPreanalyzer.h"
#undef private
#undef protected


namespace {

struct Encoder {
  int width = 0, height = 0, bit_depth = 0;
  TEncCfg cfg;
  TComSPS sps;
  TComPPS pps;
  TEncCu cu;
  TEncSlice slice;
  TEncSearch search;
  TComTrQuant trq;
  TComRdCost rd;
  TEncEntropy entropy;
  TEncSbac sbac;
  TEncBinCABAC bin;
  TEncSbac rdGoOnSbac;
#if FAST_BIT_EST
  TEncBinCABACCounter rdGoOnBin;
#else
  TEncBinCABAC rdGoOnBin;
#endif
  TEncRateCtrl rc;
  TEncSbac*** rdSbac = nullptr;
  TComPic* pic = nullptr;
  TComPic* pic0 = nullptr;   // the first picture object (pic may point to either while P pictures ping-pong)
  TComPic* pic1 = nullptr;   // second picture object (P-slice variant)
  TComPic* last = nullptr;   // the picture encoded last: reference of the next P picture
  TEncSampleAdaptiveOffset sao;   // FHREF_SAO=1: the reference's own SAO decision + filter after the deblocking pass
  bool sao_created = false;
  unsigned char slice_md5[16] = { 0 };   // md5 of the slice-data bytes the last encodeSlice wrote (FHREF_ENCODE_SLICE=1)
  bool have_md5 = false;
};

std::map<long long, Encoder*> g_encoders;

int env_int(const char* name, int dflt);
void configure(Encoder& e)
{
  TEncCfg& c = e.cfg;
  // cfg/encoder_intra_main.cfg (+ TAppEncCfg.cpp defaults for everything the cfg does not name)
  c.setSourceWidth(e.width); c.setSourceHeight(e.height);
  c.setChromaFormatIdc(CHROMA_420);
  c.setMaxCUWidth(64); c.setMaxCUHeight(64); c.setMaxTotalCUDepth(4); c.setLog2DiffMaxMinCodingBlockSize(3);
  c.setQuadtreeTULog2MaxSize(5); c.setQuadtreeTULog2MinSize(2);
  c.setQuadtreeTUMaxDepthInter(3); c.setQuadtreeTUMaxDepthIntra(3);
  c.setIntraPeriod(1); c.setGOPSize(1); c.setDecodingRefreshType(1);
  c.setBitDepth(CHANNEL_TYPE_LUMA, e.bit_depth); c.setBitDepth(CHANNEL_TYPE_CHROMA, e.bit_depth);
  c.setUseRDOQ(env_int("FHREF_RDOQ", 1)); c.setUseRDOQTS(env_int("FHREF_RDOQ", 1)); c.setUseSelectiveRDOQ(false); c.setRDpenalty(0);
  c.setUseTransformSkip(true); c.setUseTransformSkipFast(true); c.setLog2MaxTransformSkipBlockSize(2);
  c.setFastUDIUseMPMEnabled(true); c.setUseEarlyCU(false); c.setUseCbfFastMode(false); c.setUseEarlySkipDetection(false);
  c.setFastDeltaQp(false); c.setUseFastDecisionForMerge(true);
  c.setMaxDeltaQP(0); c.setMaxCuDQPDepth(0); c.setDeltaQpRD(0); c.setUseAdaptiveQP(false);
  c.setUsePCM(false); c.setPCMLog2MinSize(3); c.setPCMLog2MaxSize(5);
  c.setUseAMP(true); c.setUseSAO(true);
  c.setUseStrongIntraSmoothing(true);
  c.setSignDataHidingEnabledFlag(env_int("FHREF_SDH", 1));
  c.setTransquantBypassEnabledFlag(false); c.setCUTransquantBypassFlagForceValue(false);
  c.setCostMode(COST_STANDARD_LOSSY);
  c.setUseRateCtrl(false);
  c.setEntropyCodingSyncEnabledFlag(false);
  c.setSliceMode(NO_SLICES); c.setSliceSegmentMode(NO_SLICES);
  c.setUseScalingListId(SCALING_LIST_OFF);
  c.setCrossComponentPredictionEnabledFlag(false); c.setUseReconBasedCrossCPredictionEstimate(false);
  c.setExtendedPrecisionProcessingFlag(false); c.setHighPrecisionOffsetsEnabledFlag(false);
  c.setRdpcmEnabledFlag(RDPCM_SIGNAL_IMPLICIT, false); c.setRdpcmEnabledFlag(RDPCM_SIGNAL_EXPLICIT, false);
  c.setTransformSkipRotationEnabledFlag(false); c.setTransformSkipContextEnabledFlag(false);
  c.setPersistentRiceAdaptationEnabledFlag(false); c.setCabacBypassAlignmentEnabledFlag(false);
  c.setIntraSmoothingDisabledFlag(false);
#if ADAPTIVE_QP_SELECTION
  c.setUseAdaptQpSelect(false);
#endif
  LumaLevelToDeltaQPMapping lm; lm.mode = LUMALVL_TO_DQP_DISABLED; lm.maxMethodWeight = 0.0;
  c.setLumaLevelToDeltaQPControls(lm);
}

int env_int(const char* name, int dflt) { const char* v = std::getenv(name); return v ? std::atoi(v) : dflt; }

// the subset of TEncTop::xInitSPS / xInitPPS (TEncTop.cpp:580-900) that the CTU decision path reads
void init_parameter_sets(Encoder& e)
{
  TComSPS& sps = e.sps;
  sps.setPicWidthInLumaSamples(e.width); sps.setPicHeightInLumaSamples(e.height);
  sps.setMaxCUWidth(64); sps.setMaxCUHeight(64); sps.setMaxTotalCUDepth(4);
  sps.setChromaFormatIdc(CHROMA_420);
  sps.setLog2DiffMaxMinCodingBlockSize(3); sps.setLog2MinCodingBlockSize(3);
  sps.setPCMLog2MinSize(3); sps.setUsePCM(false); sps.setPCMLog2MaxSize(5);
  sps.setQuadtreeTULog2MaxSize(5); sps.setQuadtreeTULog2MinSize(2);
  sps.setQuadtreeTUMaxDepthInter(3); sps.setQuadtreeTUMaxDepthIntra(3);
  sps.setSPSTemporalMVPEnabledFlag(false);
  sps.setMaxTrSize(32);
  sps.setUseAMP(true);
  for (UInt ch = 0; ch < MAX_NUM_CHANNEL_TYPE; ch++) {
    sps.setBitDepth(ChannelType(ch), e.bit_depth);
#if O0043_BEST_EFFORT_DECODING
    sps.setStreamBitDepth(ChannelType(ch), e.bit_depth);
#endif
    sps.setQpBDOffset(ChannelType(ch), 6 * (e.bit_depth - 8));
    sps.setPCMBitDepth(ChannelType(ch), e.bit_depth);
  }
  sps.setUseSAO(true);
  sps.setMaxTLayers(1); sps.setTemporalIdNestingFlag(true);
  sps.setScalingListFlag(false);
  sps.setUseStrongIntraSmoothing(true);

  TComPPS& pps = e.pps;
  pps.setSPSId(0); pps.setPPSId(0);
  pps.setUseDQP(false); pps.setMaxCuDQPDepth(0);
  pps.setQpOffset(COMPONENT_Cb, 0); pps.setQpOffset(COMPONENT_Cr, 0);
  pps.setEntropyCodingSyncEnabledFlag(false);
  pps.setTilesEnabledFlag(false);
  pps.setNumTileColumnsMinus1(0); pps.setNumTileRowsMinus1(0); pps.setTileUniformSpacingFlag(false);
  pps.setUseWP(false); pps.setWPBiPred(false);
  pps.setSignDataHidingEnabledFlag(env_int("FHREF_SDH", 1));
  pps.setTransquantBypassEnabledFlag(false);
  pps.setUseTransformSkip(true);
  pps.getPpsRangeExtension().setLog2MaxTransformSkipBlockSize(2);
  pps.setDependentSliceSegmentsEnabledFlag(false);
  pps.setCabacInitPresentFlag(false);
  pps.setLoopFilterAcrossTilesEnabledFlag(true);
  pps.setScalingListPresentFlag(false);
}

Encoder* get_encoder(int w, int h, int bd)
{
  const long long key = ((long long)w << 40) | ((long long)h << 16) | bd;
  auto it = g_encoders.find(key);
  if (it != g_encoders.end()) return it->second;
  static bool rom = false;
  if (!rom) {
    initROM();
#if FAST_BIT_EST
    ContextModel::buildNextStateTable();  // done by TEncTop's constructor in the real encoder (TEncTop.cpp:73-76)
#endif
    rom = true;
  }
  Encoder* e = new Encoder();
  e->width = w; e->height = h; e->bit_depth = bd;
  configure(*e);
  init_parameter_sets(*e);

  // TEncTop::create (TEncTop.cpp:89-145)
  e->slice.create(w, h, CHROMA_420, 64, 64, 4);
  e->cu.create(4, 64, 64, CHROMA_420);
  e->rdSbac = new TEncSbac**[5];
  for (int d = 0; d < 5; d++) {
    e->rdSbac[d] = new TEncSbac*[CI_NUM];
    for (int ci = 0; ci < CI_NUM; ci++) {
      e->rdSbac[d][ci] = new TEncSbac;
#if FAST_BIT_EST
      e->rdSbac[d][ci]->init(new TEncBinCABACCounter);
#else
      e->rdSbac[d][ci]->init(new TEncBinCABAC);
#endif
    }
  }
  // TEncTop::init (TEncTop.cpp:183-231) without the GOP encoder
  e->rd.setCostMode(COST_STANDARD_LOSSY);
  const Int maxLog2TrDynamicRange[MAX_NUM_CHANNEL_TYPE] = { e->sps.getMaxLog2TrDynamicRange(CHANNEL_TYPE_LUMA),
                                                             e->sps.getMaxLog2TrDynamicRange(CHANNEL_TYPE_CHROMA) };
  e->trq.init(32, env_int("FHREF_RDOQ", 1), env_int("FHREF_RDOQ", 1), false, true, true
#if ADAPTIVE_QP_SELECTION
              , false
#endif
  );
  e->trq.setFlatScalingList(maxLog2TrDynamicRange, e->sps.getBitDepths());
  e->trq.setUseScalingList(false);
  e->sbac.init(&e->bin);
  e->rdGoOnSbac.init(&e->rdGoOnBin);
  // TEncSlice::init / TEncCu::init (TEncSlice.cpp:78-104, TEncCu.cpp:227-243) with our own owners instead of TEncTop
  e->slice.m_pcCfg = &e->cfg; e->slice.m_pcListPic = nullptr; e->slice.m_pcGOPEncoder = nullptr;
  e->slice.m_pcCuEncoder = &e->cu; e->slice.m_pcPredSearch = &e->search;
  e->slice.m_pcEntropyCoder = &e->entropy; e->slice.m_pcSbacCoder = &e->sbac; e->slice.m_pcBinCABAC = &e->bin;
  e->slice.m_pcTrQuant = &e->trq; e->slice.m_pcRdCost = &e->rd;
  e->slice.m_pppcRDSbacCoder = e->rdSbac; e->slice.m_pcRDGoOnSbacCoder = &e->rdGoOnSbac;
  e->slice.m_vdRdPicLambda.resize(1); e->slice.m_vdRdPicQp.resize(1); e->slice.m_viRdPicQp.resize(1);
  e->slice.m_pcRateCtrl = &e->rc;
  e->cu.m_pcEncCfg = &e->cfg; e->cu.m_pcPredSearch = &e->search; e->cu.m_pcTrQuant = &e->trq; e->cu.m_pcRdCost = &e->rd;
  e->cu.m_pcEntropyCoder = &e->entropy; e->cu.m_pcBinCABAC = &e->bin;
  e->cu.m_pppcRDSbacCoder = e->rdSbac; e->cu.m_pcRDGoOnSbacCoder = &e->rdGoOnSbac; e->cu.m_pcRateCtrl = &e->rc;
  e->cu.m_lumaQPOffset = 0;
  e->cu.initLumaDeltaQpLUT();
  e->cu.setSliceEncoder(&e->slice);
  e->search.init(&e->cfg, &e->trq, 64, 4, MESEARCH_DIAMOND, 64, 64, 4, &e->entropy, &e->rd, e->rdSbac, &e->rdGoOnSbac);

  e->pic = new TComPic();
  e->pic->create(e->sps, e->pps, true, true);
  e->pic0 = e->pic;
  g_encoders[key] = e;
  return e;
}

void load_picture(Encoder& e, const int16_t* luma, int stride, const int16_t* cb, const int16_t* cr)
{
  TComPicYuv* org = e.pic->getPicYuvOrg();
  for (int comp = 0; comp < 3; comp++) {
    const ComponentID id = ComponentID(comp);
    Pel* dst = org->getAddr(id);
    const int s = org->getStride(id), w = org->getWidth(id), h = org->getHeight(id);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++)
        dst[y * s + x] = (comp == 0) ? (Pel)luma[(size_t)y * stride + x]
                         : ((comp == 1 ? cb : cr) ? (Pel)(comp == 1 ? cb : cr)[(size_t)y * w + x]   // packed W/2 x H/2 plane
                                                  : (Pel)(1 << (e.bit_depth - 1)));                 // flat chroma
  }
  org->copyToPic(e.pic->getPicYuvTrueOrg());
}

// slice set-up: the parts of TEncSlice::initEncSlice (TEncSlice.cpp:159-430) that matter for one I picture
void init_slice(Encoder& e, int qp)
{
  e.pic->getPicSym()->clearSliceBuffer();
  e.pic->getPicSym()->allocateNewSlice();
  TComSlice* s = e.pic->getSlice(0);
  e.pic->setCurrSliceIdx(0);
  s->setSPS(&e.sps); s->setPPS(&e.pps); s->setPic(e.pic);
  s->initSlice();
  s->setSliceBits(0); s->setPicOutputFlag(true);
  s->setPOC(0); s->setDepth(0); s->setSliceType(I_SLICE); s->setNalUnitType(NAL_UNIT_CODED_SLICE_IDR_W_RADL);
  s->setSliceQp(qp); s->setSliceQpBase(qp); s->setSliceQpDelta(0);
  s->setSliceChromaQpDelta(COMPONENT_Cb, 0); s->setSliceChromaQpDelta(COMPONENT_Cr, 0);
  s->setUseChromaQpAdj(false);
  s->setNumRefIdx(REF_PIC_LIST_0, 0); s->setNumRefIdx(REF_PIC_LIST_1, 0);
  s->setTLayer(0); e.pic->setTLayer(0);
  s->setSliceMode(NO_SLICES); s->setSliceArgument(0); s->setSliceSegmentMode(NO_SLICES); s->setSliceSegmentArgument(0);
  s->setMaxNumMergeCand(5);
  s->setSliceCurStartCtuTsAddr(0); s->setSliceSegmentCurStartCtuTsAddr(0);
  s->setSliceCurEndCtuTsAddr(e.pic->getNumberOfCtusInFrame()); s->setSliceSegmentCurEndCtuTsAddr(e.pic->getNumberOfCtusInFrame());
  s->setDependentSliceSegmentFlag(false);
  e.pic->setPicYuvPred(&e.slice.m_picYuvPred); e.pic->setPicYuvResi(&e.slice.m_picYuvResi);
  e.slice.setSliceIdx(0);
  // lambda: TEncSlice::calculateLambda for an I slice of an all-intra configuration (GOPSize 1, modifiers 1.0)
  // = 0.57 * 2^((qp-12)/3) (TEncSlice.cpp:433-527); then the reference's own setUpLambda (:113-157)
  const double lambda = 0.57 * std::pow(2.0, (qp - 12) / 3.0);
  e.slice.setUpLambda(s, lambda, qp);
}

}  // namespace

#ifdef FHREF_RECORD_COSTS
// `make costs` build (tests/quality/record_costs_patch.py): xCompressCU reports the two RD costs its xCheckBestMode compares at every
// node it evaluates both ways.  Table: [ctu][node 0..20][0 = best non-split mode, 1 = four-way split]; node 0 = the CTU, 1 + q = its
// quadrant q, 5 + b = its 16x16 block b, q and b in z-order (zorder >> 6, zorder >> 4).  NaN = not evaluated both ways.
static std::vector<double> g_split_costs;
extern "C" void fhref_record_split(unsigned ctu, unsigned zorder, unsigned depth, double cost_no_split, double cost_split)
{
  if (depth > 2) return;
  const size_t node = depth == 0 ? 0 : depth == 1 ? 1 + (zorder >> 6) : 5 + (zorder >> 4);
  const size_t i = ((size_t)ctu * 21 + node) * 2;
  if (i + 1 < g_split_costs.size()) { g_split_costs[i] = cost_no_split; g_split_costs[i + 1] = cost_split; }
}
extern "C" int href_split_costs(double* out, int num_ctus)
{
  if ((size_t)num_ctus * 42 != g_split_costs.size()) return -1;
  std::memcpy(out, g_split_costs.data(), g_split_costs.size() * sizeof(double));
  return 0;
}
#endif

extern "C" {

// One intra picture through the reference's compressSlice.  luma: Pel samples at the internal bit depth.
// forced_depth (only in hook builds): numCtus*256 raster depth map fed to the xCompressCU hook, or NULL.
// depth_out: numCtus*256, raster per CTU = getDepth(g_auiRasterToZscan[r]).  stats: [0] bits (RD-SBAC estimate summed
// over CTUs), [1] distortion (SSE, chroma weighted as in TComRdCost), [2] RD cost, [3] seconds in compressSlice,
// [4] luma SSE of the reconstruction vs the original, [5] number of CTUs, [6] bits counted by encodeCtu, [7] luma SSE after the
// reference's own deblocking filter when FHREF_DEBLOCK=1 (else -1), [8] slice-data bits written by the reference's own encodeSlice
// (real CABAC) when FHREF_ENCODE_SLICE=1 (else -1), [9] luma SSE after deblocking AND the reference's own SAO when FHREF_SAO=1 (else -1).
int href_rdo_encode_frame_yuv(const int16_t* luma, int stride, const int16_t* cb, const int16_t* cr, int width, int height,
                              int bit_depth, int qp, const uint8_t* forced_depth, uint8_t* depth_out, double* stats);

// soft hook: depth_max map for the NEXT encode call (forced_depth is then depth_min); hook builds only
static const uint8_t* g_forced_max = nullptr;
int href_rdo_set_forced_max(const uint8_t* depth_max)
{
#ifdef FHEVC_HOOK
  g_forced_max = depth_max;
  return 0;
#else
  (void)depth_max;
  return -2;
#endif
}

// first-pass candidate lists (numCtus * 85 * 8 modes, best first) for the NEXT encode call: the hook's estIntraPredLumaQT patch reads them instead of
// running HM's 35-mode pass (TEncFastDepth::setExternalCandidates); hook builds only, one-shot
static const uint8_t* g_candidates = nullptr;
int href_rdo_set_candidates(const uint8_t* cand)
{
#ifdef FHEVC_HOOK
  g_candidates = cand;
  return 0;
#else
  (void)cand;
  return -2;
#endif
}

int href_rdo_encode_frame(const int16_t* luma, int stride, int width, int height, int bit_depth, int qp,
                          const uint8_t* forced_depth, uint8_t* depth_out, double* stats)
{
  return href_rdo_encode_frame_yuv(luma, stride, nullptr, nullptr, width, height, bit_depth, qp, forced_depth, depth_out, stats);
}

// same with 4:2:0 chroma planes (packed W/2 x H/2, internal bit depth); NULL = flat mid-grey chroma
int href_rdo_encode_frame_yuv(const int16_t* luma, int stride, const int16_t* cb, const int16_t* cr, int width, int height,
                              int bit_depth, int qp, const uint8_t* forced_depth, uint8_t* depth_out, double* stats)
{
  if ((width % 8) || (height % 8) || width < 64 || height < 64) return -1;
  Encoder* e = get_encoder(width, height, bit_depth);
  load_picture(*e, luma, stride, cb, cr);
  init_slice(*e, qp);
#ifdef FHEVC_HOOK
  // explicit depth-map feed of the hook (TEncFastDepth's public validation interface); NULL clears it
  if (forced_depth && g_forced_max) e->cu.getFastDepth().setExternalRange(forced_depth, g_forced_max, (int)e->pic->getNumberOfCtusInFrame());
  else e->cu.getFastDepth().setExternalMap(forced_depth, forced_depth ? (int)e->pic->getNumberOfCtusInFrame() : 0);
  g_forced_max = nullptr;  // one-shot
  e->cu.getFastDepth().setExternalCandidates(g_candidates, (int)e->pic->getNumberOfCtusInFrame());
  g_candidates = nullptr;
#else
  if (forced_depth) return -2;
#endif
#ifdef FHREF_RECORD_COSTS
  g_split_costs.assign((size_t)e->pic->getNumberOfCtusInFrame() * 42, std::nan(""));
#endif
  const auto t0 = std::chrono::steady_clock::now();
  e->slice.compressSlice(e->pic, false, false);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const int n = (int)e->pic->getNumberOfCtusInFrame();
  for (int c = 0; c < n; c++) {
    TComDataCU* ctu = e->pic->getCtu(c);
    for (int r = 0; r < 256; r++) depth_out[c * 256 + r] = ctu->getDepth(g_auiRasterToZscan[r]);
  }
  if (stats) {
    stats[0] = (double)e->slice.m_uiPicTotalBits;
    stats[6] = (double)e->pic->getSlice(0)->getSliceBits();  // bits counted by encodeCtu (true CABAC state)
    stats[1] = (double)e->slice.m_uiPicDist;
    stats[2] = e->slice.m_dPicRdCost;
    stats[3] = sec;
    double sse = 0;
    const Pel* o = e->pic->getPicYuvOrg()->getAddr(COMPONENT_Y);
    const Pel* r = e->pic->getPicYuvRec()->getAddr(COMPONENT_Y);
    const int so = e->pic->getPicYuvOrg()->getStride(COMPONENT_Y), sr = e->pic->getPicYuvRec()->getStride(COMPONENT_Y);
    for (int y = 0; y < height; y++)
      for (int x = 0; x < width; x++) { const double d = (double)o[y * so + x] - (double)r[y * sr + x]; sse += d * d; }
    stats[4] = sse;
    stats[5] = n;
    stats[7] = -1.0;
    stats[8] = -1.0;
    // the in-loop filters and the entropy coder in the order TEncGOP runs them (TEncGOP.cpp:1607-1619 deblocking, :1662-1685 SAO,
    // :1745 encodeSlice): the SAO syntax of a CTU is part of the slice data, so SAO has to be decided before encodeSlice
    TComSlice* slice0 = e->pic->getSlice(0);
    if (env_int("FHREF_DEBLOCK", 0)) {
      // the reference's own in-loop deblocking filter on the reconstruction (TComLoopFilter::loopFilterPic, called by TEncGOP.cpp:1607-1619
      // with the slice's default parameters: filter enabled, beta / tc offsets 0), then the luma SSE again: distortion as a decoder
      // sees it before SAO.  Modifies PicYuvRec in place (only used for I pictures that no P picture follows).
      TComLoopFilter lf;
      lf.create(4);
      lf.setCfg(true);
      lf.loopFilterPic(e->pic);
      lf.destroy();
      double sse2 = 0;
      for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { const double d = (double)o[y * so + x] - (double)r[y * sr + x]; sse2 += d * d; }
      stats[7] = sse2;
    }
    stats[9] = -1.0;
    if (env_int("FHREF_SAO", 0) && env_int("FHREF_DEBLOCK", 0)) {
      // the reference's own TEncSampleAdaptiveOffset::SAOProcess on the deblocked picture, set up as TEncTop::create does (TEncTop.cpp:98-101)
      // and called as TEncGOP does (TEncGOP.cpp:1662-1685; cfg defaults: SaoEncodingRate 0.75 / 0.5, no picture-level test, SAOLcuBoundary 0)
      if (!e->sao_created) {
        e->sao.create(width, height, CHROMA_420, 64, 64, 4, 0, 0);
        e->sao.createEncData(false);
        e->sao_created = true;
      }
      Bool sliceEnabled[MAX_NUM_COMPONENT];
      TComBitCounter tempBitCounter;
      tempBitCounter.resetBits();
      e->rdGoOnSbac.setBitstream(&tempBitCounter);
      e->sao.initRDOCabacCoder(&e->rdGoOnSbac, slice0);
      e->sao.SAOProcess(e->pic, sliceEnabled, slice0->getLambdas(), false, 0.75, 0.5, false, false);
      e->sao.PCMLFDisableProcess(e->pic);
      e->rdGoOnSbac.setBitstream(NULL);
      slice0->setSaoEnabledFlag(CHANNEL_TYPE_LUMA, sliceEnabled[COMPONENT_Y]);
      slice0->setSaoEnabledFlag(CHANNEL_TYPE_CHROMA, sliceEnabled[COMPONENT_Cb]);
      double sse3 = 0;
      for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) { const double d = (double)o[y * so + x] - (double)r[y * sr + x]; sse3 += d * d; }
      stats[9] = sse3;
    } else {
      slice0->setSaoEnabledFlag(CHANNEL_TYPE_LUMA, false);    // SAO not run: no SAO syntax in the slice data
      slice0->setSaoEnabledFlag(CHANNEL_TYPE_CHROMA, false);
    }
    e->have_md5 = false;
    if (env_int("FHREF_ENCODE_SLICE", 0)) {
      // the reference's own TEncSlice::encodeSlice (TEncSlice.cpp:985-1160: the real arithmetic coder, TEncBinCABAC) over the picture the
      // decision path has just filled in: slice-data bytes as they would stand in the bitstream (SAO syntax included when SAO ran; the slice
      // header is TEncGOP's business and not part of it).  Their md5 is kept for href_slice_md5 (SURVEY F11 at the byte level)
      TComOutputBitstream substream;
      UInt bins = 0;
      e->slice.encodeSlice(e->pic, &substream, bins);
      stats[8] = (double)substream.getNumberOfWrittenBits();
      MD5 md5;
      std::vector<uint8_t>& fifo = substream.getFIFO();
      md5.update(fifo.data(), (unsigned)fifo.size());
      md5.finalize(e->slice_md5);
      e->have_md5 = true;
    }
  }
  e->last = e->pic;  // reference of a following P picture (href_rdo_encode_next_p, P-variant builds)
  return 0;
}

// TEncPreanalyzer::xPreanalyze (TEncPreanalyzer.cpp:64-152) on a TEncPic holding `luma`: activities of all layers
// (layer d = parts of 64 >> d, raster order, layers concatenated) and the per-layer averages.
int href_preanalyze(const int16_t* luma, int stride, int width, int height, int bit_depth, int max_aq_depth,
                    double* activity, double* avg)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  TEncPic pic;
  pic.create(e->sps, e->pps, (UInt)max_aq_depth);
  Pel* dst = pic.getPicYuvOrg()->getAddr(COMPONENT_Y);
  const int s = pic.getPicYuvOrg()->getStride(COMPONENT_Y);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) dst[y * s + x] = (Pel)luma[(size_t)y * stride + x];
  TEncPreanalyzer pre;
  pre.xPreanalyze(&pic);
  size_t k = 0;
  for (int d = 0; d < max_aq_depth; d++) {
    TEncPicQPAdaptationLayer* layer = pic.getAQLayer(d);
    const size_t n = (size_t)layer->getNumAQPartInWidth() * layer->getNumAQPartInHeight();
    for (size_t i = 0; i < n; i++) activity[k++] = layer->getQPAdaptationUnit()[i].getActivity();
    avg[d] = layer->getAvgActivity();
  }
  pic.destroy();
  return (int)k;
}

// TEncCu::xComputeQP (TEncCu.cpp:1093-1117) for the CU at the origin of every AQ part of every layer, after the
// reference's own pre-analysis of `luma`: qp_out has the layout of href_preanalyze's activity array.
int href_aq_qp(const int16_t* luma, int stride, int width, int height, int bit_depth, int max_aq_depth,
               int qp_adaptation_range, int base_qp, int* qp_out)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  TEncPic pic;
  pic.create(e->sps, e->pps, (UInt)max_aq_depth);
  Pel* dst = pic.getPicYuvOrg()->getAddr(COMPONENT_Y);
  const int s = pic.getPicYuvOrg()->getStride(COMPONENT_Y);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) dst[y * s + x] = (Pel)luma[(size_t)y * stride + x];
  TEncPreanalyzer pre;
  pre.xPreanalyze(&pic);
  const bool aq_was = e->cfg.getUseAdaptiveQP();
  const int range_was = e->cfg.getQPAdaptationRange();
  e->cfg.setUseAdaptiveQP(true);
  e->cfg.setQPAdaptationRange(qp_adaptation_range);
  TComSlice slice;
  slice.setSPS(&e->sps);
  slice.setSliceQp(base_qp);
  TComDataCU cu;  // only the fields xComputeQP reads are set
  cu.m_pcPic = &pic;
  cu.m_pcSlice = &slice;
  size_t k = 0;
  for (int d = 0; d < max_aq_depth; d++) {
    const int part = 64 >> d;
    for (int y = 0; y < height; y += part)
      for (int x = 0; x < width; x += part) {
        cu.m_uiCUPelX = (UInt)x;
        cu.m_uiCUPelY = (UInt)y;
        qp_out[k++] = e->cu.xComputeQP(&cu, (UInt)d);
      }
  }
  cu.m_pcPic = nullptr;
  cu.m_pcSlice = nullptr;
  e->cfg.setUseAdaptiveQP(aq_was);
  e->cfg.setQPAdaptationRange(range_was);
  pic.destroy();
  return (int)k;
}

#ifdef FHEVC_PVAR
// ---- config 4 (encoder_lowdelay_P_main.cfg): the NEXT picture as a P slice predicted from the reconstruction of the picture
// encoded last in this geometry (by href_rdo_encode_frame[_yuv] = I slice, or by this function), through the reference's own
// compressSlice with HM-16.14's inter checks restored (hm_patch/restore_inter.py; as shipped the reference aborts on any
// non-I slice: SURVEY F6).  Differences to the full encoder, all documented in DESIGN.md: one reference picture, no
// in-loop filters on it, TMVP off, QP of the P picture given by the caller (TEncSlice::initEncSlice derives QP + QPOffset +
// model offset from the GOP entry), lambda = 0.57 * 2^((qp-12)/3) as under LambdaFromQpEnable.
// forced_min/forced_max: optional depth range (hook), NULL = full RDO.  stats as href_rdo_encode_frame, [7] = skipped share.
int href_rdo_encode_next_p(const int16_t* luma, const int16_t* cb, const int16_t* cr, int stride, int width, int height,
                           int bit_depth, int qp, int poc, const uint8_t* forced_min, const uint8_t* forced_max,
                           uint8_t* depth_out, double* stats)
{
  if ((width % 8) || (height % 8) || width < 64 || height < 64) return -1;
  Encoder* e = get_encoder(width, height, bit_depth);
  if (!e->last) return -3;  // nothing encoded yet in this geometry
  e->cfg.setUseHADME(true); e->cfg.setFastInterSearchMode(FASTINTERSEARCH_DISABLED);
  e->cfg.setDisableIntraPUsInInterSlices(false); e->cfg.setRestrictMESampling(false);
  e->cfg.setClipForBiPredMeEnabled(false); e->cfg.setFastMEAssumingSmootherMVEnabled(false);
  e->cfg.setSearchRange(64); e->cfg.setMinSearchWindow(8); e->cfg.setBipredSearchRange(4);
  const int n = (int)e->pic->getNumberOfCtusInFrame();
  // the last reconstruction becomes the reference picture: borders extended for motion search, motion field compressed
  TComPic* ref = e->last;
  ref->getPicYuvRec()->setBorderExtension(false);  // the object is re-used: its borders are from an older picture
  ref->getPicYuvRec()->extendPicBorder();
  ref->compressMotion();
  ref->setReconMark(true);
  ref->getSlice(0)->setReferenced(true);
  if (!e->pic1) { e->pic1 = new TComPic(); e->pic1->create(e->sps, e->pps, true, true); }
  TComPic* other = (ref == e->pic0) ? e->pic1 : e->pic0;  // ping-pong between the two picture objects
  e->pic = other;                   // load_picture / init_slice work on e->pic
  load_picture(*e, luma, stride, cb, cr);
  init_slice(*e, qp);
  TComSlice* s = e->pic->getSlice(0);
  s->setPOC(poc); s->setSliceType(P_SLICE); s->setNalUnitType(NAL_UNIT_CODED_SLICE_TRAIL_R);
  s->setNumRefIdx(REF_PIC_LIST_0, 1); s->setNumRefIdx(REF_PIC_LIST_1, 0);
  s->setRefPic(ref, REF_PIC_LIST_0, 0);
  s->setRefPOCList(); s->setList1IdxToList0Idx();
  s->setEnableTMVPFlag(false); s->setColFromL0Flag(true); s->setColRefIdx(0);
  s->setMaxNumMergeCand(5); s->setDepth(0);
  s->setLFCrossSliceBoundaryFlag(true);
  e->slice.setSearchRange(s);
  e->cu.getFastDepth().readKnobs();  // the harness changes the environment between pictures
  if (forced_min && forced_max) e->cu.getFastDepth().setExternalRange(forced_min, forced_max, n);
  else e->cu.getFastDepth().setExternalMap(forced_min, forced_min ? n : 0);
  const auto t0 = std::chrono::steady_clock::now();
  e->slice.compressSlice(e->pic, false, false);
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  long skipped = 0;
  for (int c = 0; c < n; c++) {
    TComDataCU* ctu = e->pic->getCtu(c);
    for (int r = 0; r < 256; r++) depth_out[c * 256 + r] = ctu->getDepth(g_auiRasterToZscan[r]);
    for (int z = 0; z < 256; z++) skipped += ctu->isSkipped(z) ? 1 : 0;
  }
  stats[0] = (double)e->slice.m_uiPicTotalBits; stats[1] = (double)e->slice.m_uiPicDist; stats[2] = e->slice.m_dPicRdCost;
  stats[3] = sec; stats[5] = n; stats[6] = (double)e->pic->getSlice(0)->getSliceBits(); stats[7] = (double)skipped / (n * 256.0);
  double sse = 0;
  const Pel* o = e->pic->getPicYuvOrg()->getAddr(COMPONENT_Y);
  const Pel* r = e->pic->getPicYuvRec()->getAddr(COMPONENT_Y);
  const int so = e->pic->getPicYuvOrg()->getStride(COMPONENT_Y), sr = e->pic->getPicYuvRec()->getStride(COMPONENT_Y);
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) { const double d = (double)o[y * so + x] - (double)r[y * sr + x]; sse += d * d; }
  stats[4] = sse;
  e->last = e->pic;
  return 0;
}
#endif

// ---- A7 goldens: the reference's OWN TComPrediction::initIntraPatternChType (TComPattern.cpp:115-320: availability
// isAbove/Left/AboveRight/BelowLeftAvailable :568-746, fillReferenceSamples :322-539, [1 2 1] / strong smoothing :196-295) on live CUs of
// the picture encoded last in this geometry.  from_original != 0 first copies the original picture into the reconstruction
// plane the function reads (the GPU first pass takes its neighbours from the original: source-only twin), so that only the
// availability rule, the substitution walk and the smoothing are under test.  The node is the CU of size 64 >> depth whose
// top-left 4x4 unit has raster index part_raster inside CTU `ctu`.  Lines in the oracle's layout: line[2N] = top-left,
// line[2N + 1 + i] = above / above-right i, line[2N - 1 - j] = left / below-left j  (4N + 1 samples each).
int href_intra_lines(int width, int height, int bit_depth, int ctu, int depth, int part_raster, int from_original,
                     int16_t* unfiltered, int16_t* filtered)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  if (!e->last) return -3;
  TComPic* pic = e->last;
  if (ctu < 0 || ctu >= (int)pic->getNumberOfCtusInFrame() || depth < 0 || depth > 3 || part_raster < 0 || part_raster > 255) return -1;
  if (from_original) pic->getPicYuvOrg()->copyToPic(pic->getPicYuvRec());
  TComDataCU* cu = pic->getCtu(ctu);
  const UInt z = g_auiRasterToZscan[part_raster];
  const int n = 64 >> depth;
  // the TU rectangle comes from the CU's stored size at that partition (TComTU.cpp:63-75): set it for the call, restore after
  const UChar w0 = cu->getWidth()[z], h0 = cu->getHeight()[z], d0 = cu->getDepth()[z];
  cu->getWidth()[z] = (UChar)n; cu->getHeight()[z] = (UChar)n; cu->getDepth()[z] = (UChar)depth;
  {
    TComTURecurse tu(cu, z, (UInt)depth);
    e->search.initIntraPatternChType(tu, COMPONENT_Y, true);
  }
  cu->getWidth()[z] = w0; cu->getHeight()[z] = h0; cu->getDepth()[z] = d0;
  const int stride = 2 * n + 1;
  const Pel* bufs[2] = { e->search.m_piYuvExt[COMPONENT_Y][PRED_BUF_UNFILTERED], e->search.m_piYuvExt[COMPONENT_Y][PRED_BUF_FILTERED] };
  int16_t* outs[2] = { unfiltered, filtered };
  for (int b = 0; b < 2; b++) {
    outs[b][2 * n] = bufs[b][0];
    for (int i = 0; i < 2 * n; i++) outs[b][2 * n + 1 + i] = bufs[b][1 + i];
    for (int j = 0; j < 2 * n; j++) outs[b][2 * n - 1 - j] = bufs[b][(1 + j) * stride];
  }
  return 4 * n + 1;
}

// ---- first-pass mode bits: the reference's OWN TEncSearch::xModeBitsIntra (TEncSearch.cpp:5353-5381) for the CU at the
// origin of CTU 0 of a fresh I slice at slice QP qp (CABAC contexts at their slice-initial state, TEncSlice.cpp:719-720, 826;
// no neighbours: the default MPM set).  bits[35]: what estIntraPredLumaQT adds times sqrt(lambda) (TEncSearch.cpp:2288).
int href_mode_bits_origin(int width, int height, int bit_depth, int qp, int depth, unsigned* bits)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  if (!e->last || depth < 0 || depth > 3) return -3;
  e->pic = e->last;
  init_slice(*e, qp);
  TComSlice* slice = e->pic->getSlice(0);
  e->entropy.setEntropyCoder(e->rdSbac[0][CI_CURR_BEST]);
  e->entropy.resetEntropy(slice);
  for (int d = 1; d < 5; d++) e->rdSbac[d][CI_CURR_BEST]->load(e->rdSbac[0][CI_CURR_BEST]);
  e->entropy.setEntropyCoder(&e->rdGoOnSbac);
  TComBitCounter counter;                       // as compressSlice's tempBitCounter (TEncSlice.cpp:726, 827-828)
  e->entropy.setBitstream(&counter);
  counter.resetBits();
  e->rdGoOnSbac.load(e->rdSbac[0][CI_CURR_BEST]);
  TComDataCU* cu = e->pic->getCtu(0);
  cu->initCtu(e->pic, 0);
  cu->setPartSizeSubParts(SIZE_2Nx2N, 0, (UInt)depth);
  cu->setPredModeSubParts(MODE_INTRA, 0, (UInt)depth);
  for (int m = 0; m < 35; m++) bits[m] = e->search.xModeBitsIntra(cu, (UInt)m, 0, (UInt)depth, CHANNEL_TYPE_LUMA);
  e->rdGoOnSbac.setBitstream(NULL);             // stop use of the local counter (TEncSlice.cpp:971-972)
  return 35;
}

// ---- config 4: the reference's OWN integer full search, TEncSearch::xPatternSearch (TEncSearch.cpp:3786-3848), on ORIGINAL planes.
// cur / ref: luma planes (stride in samples); the reference plane is copied into a TComPicYuv with HM's margins and extended by the
// reference's own TComPicYuv::extendPicBorder (TComPicYuv.cpp).  Per block (x0, y0, n): TComPattern on the current plane, the window
// [-range, +range]^2 around zero handed over as xMotionEstimation would after xSetSearchRange, distortion set up by the 4-argument
// TComRdCost::setDistParam (-> DF_SAD, TComRdCost.cpp:205-236), vector cost getCostOfVectorWithPredictor (TComRdCost.h:166-174) with a
// zero predictor, iCostScale 2 and m_motionLambda = m_dLambdaMotionSAD[0] = 65536 sqrt(lambda) (setLambda, selectMotionLambda(true, ..)).
// out per block: mvx, mvy, the SAD the function returns (ruiSAD: best cost minus its vector cost), and the best cost.
int href_pattern_search(const int16_t* cur, const int16_t* ref, int stride, int width, int height, int bit_depth, double lambda, int range,
                        int nblocks, const int* blocks /* x0, y0, n per block */, int* out /* mvx, mvy, sad, cost per block */)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  TComPicYuv refYuv;
  refYuv.create(width, height, CHROMA_420, 64, 64, 4, true);
  Pel* dst = refYuv.getAddr(COMPONENT_Y);
  const int rs = refYuv.getStride(COMPONENT_Y);
  for (int y = 0; y < height; y++) std::memcpy(dst + (size_t)y * rs, ref + (size_t)y * stride, (size_t)width * sizeof(Pel));
  refYuv.extendPicBorder();
  e->rd.setLambda(lambda, e->sps.getBitDepths());
  e->rd.selectMotionLambda(true, 0, false);
  TComMv zero(0, 0);
  e->rd.setPredictor(zero);
  e->rd.setCostScale(2);
  for (int b = 0; b < nblocks; b++) {
    const int x0 = blocks[3 * b], y0 = blocks[3 * b + 1], n = blocks[3 * b + 2];
    if (x0 < 0 || y0 < 0 || x0 + n > width || y0 + n > height || range < 0 || range > 64) { refYuv.destroy(); return -1; }
    TComPattern pattern;
    pattern.initPattern(const_cast<Pel*>(cur) + (size_t)y0 * stride + x0, n, n, stride, bit_depth);
    TComMv lt(-range, -range), rb(range, range), mv;
    Distortion sad = 0;
    e->search.xPatternSearch(&pattern, dst + (size_t)y0 * rs + x0, rs, &lt, &rb, mv, sad);
    out[4 * b] = mv.getHor(); out[4 * b + 1] = mv.getVer(); out[4 * b + 2] = (int)sad;
    out[4 * b + 3] = (int)(sad + e->rd.getCostOfVectorWithPredictor(mv.getHor(), mv.getVer()));
  }
  refYuv.destroy();
  return nblocks;
}

// md5 (16 bytes) of the slice-data bytes written by the reference's encodeSlice in the last href_rdo_encode_frame* call of that geometry
// (FHREF_ENCODE_SLICE=1); -1 when there is none
int href_slice_md5(int width, int height, int bit_depth, unsigned char* digest16)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  if (!e->have_md5) return -1;
  std::memcpy(digest16, e->slice_md5, 16);
  return 0;
}

// debugging aid: histograms of the decisions of the last encoded picture of that geometry
int href_rdo_debug_hist(int width, int height, int bit_depth, int* modes35, int* part2, int* trdepth4, int* tskip2, int* cbf2)
{
  Encoder* e = get_encoder(width, height, bit_depth);
  const int n = (int)e->pic->getNumberOfCtusInFrame();
  for (int c = 0; c < n; c++) {
    TComDataCU* ctu = e->pic->getCtu(c);
    for (int z = 0; z < 256; z++) {
      modes35[ctu->getIntraDir(CHANNEL_TYPE_LUMA, z)]++;
      part2[ctu->getPartitionSize(z) == SIZE_NxN ? 1 : 0]++;
      trdepth4[std::min(3, (int)ctu->getTransformIdx(z))]++;
      tskip2[ctu->getTransformSkip(z, COMPONENT_Y) ? 1 : 0]++;
      cbf2[ctu->getCbf(z, COMPONENT_Y, ctu->getTransformIdx(z)) ? 1 : 0]++;
    }
  }
  return n;
}

#ifdef FHEVC_HOOK
int href_has_hook(void) { return 1; }
#else
int href_has_hook(void) { return 0; }
#endif

}  // extern "C"
