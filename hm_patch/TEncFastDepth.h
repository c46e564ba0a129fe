/* TEncFastDepth.h -- host-side glue between HM's TEncCu/TEncSlice and the C ABI of include/fasthevc.h.
 *
 * New file for source/Lib/TLibEncoder (nothing in HM is replaced).  One instance lives in TEncCu.  Per picture,
 * TEncSlice::compressSlice calls predictPicture() before its CTU loop (TEncSlice.cpp:792); per CU node,
 * TEncCu::xCompressCU asks forcedRange() right after bBoundary is known (TEncCu.cpp:574).  If the library is
 * disabled, missing or returns an error, forcedRange() is false and HM runs its stock full RDO: never aborts.
 *
 * Knobs arrive through the environment so that TAppEncCfg stays untouched:
 *   FHEVC_ENABLE=1            turn the path on
 *   FHEVC_WEIGHTS=<file>      weight blob: FHW1 (the 16 / 32 / 64 network) or FHW3 (any member of the reference's network family, e.g.
 *                             depthnet_family_d2.fhw = 23 / 46 / 92 x 2, the best classifier shipped and the recommended one)
 *   FHEVC_DEVICE=<ordinal>    HIP device (default 0)
 *   FHEVC_MARGIN=<int>        soft decisions: logit margin inside which a split decision is left to HM's RDO;
 *   FHEVC_MARGIN_SPLIT / FHEVC_MARGIN_STOP set the two sides separately (not forcing unsure splits is almost free,
 *                             not forbidding unsure ones costs the recursion it allows).  Defaults: split 100000, stop 64000 (every content
 *                             family measured stays within 1 % BD-rate: <= +0.47 % with depthnet_family_d2.fhw); split 64000, stop 32000 for
 *                             content like the training set;
 *                             FHEVC_MARGIN=0 gives hard decisions
 *   FHEVC_P_MODE=window|motion  P/B pictures whose first reference picture was inter coded (default: off = stock RDO):
 *                             window = co-located depth of that picture +- FHEVC_P_WINDOW levels (host logic only, independent of
 *                             FHEVC_ENABLE); motion = GPU motion search of every CU node in the reference's ORIGINAL picture
 *                             (FHEVC_P_RANGE = window radius 1..64, default 4; above 8: SAD search, and the displaced depths of FHEVC_P_MC=node come on with it; FHEVC_P_MC=1: the per-unit form of round 3; FHEVC_P_MC=0 keeps co-located depths and the default rule) + fhevc_p_depth_range (needs FHEVC_ENABLE=1);
 *                             FHEVC_P_THRESH overrides the six thresholds of the rule, FHEVC_P_WINDOW adds the +- clip to it
 *   FHEVC_FIRST_PASS=1        intra pictures: the candidate list of TEncSearch::estIntraPredLumaQT (the numModesForFullRD modes its 35-mode
 *                             Hadamard pass would pick) comes from the GPU's first pass over the ORIGINAL picture for PUs of 8x8 and larger
 *                             (fhevc_intra_first_pass_candidates); HM's own 35-mode loop is skipped for them, its MPM handling stays
 */
#ifndef __TENCFASTDEPTH__
#define __TENCFASTDEPTH__

#include <vector>

class TComPic;
class TComDataCU;
struct fhevc_ctx;

class TEncFastDepth
{
public:
  TEncFastDepth();
  ~TEncFastDepth();
  void readKnobs();   ///< (re-)read the environment knobs; the constructor calls it

  /// one GPU pass over the original luma plane of pcPic; false -> this picture runs stock RDO
  bool predictPicture(TComPic* pcPic, int sliceQp, int sliceType);
  /// depth range [dmin, dmax] (each 0..3) of the CU whose top-left 4x4 unit is (ctuRsAddr, zorderIdx): the caller
  /// forces a split while uiDepth < dmin and forbids one once uiDepth >= dmax; false -> no prediction
  bool forcedRange(const TComDataCU* pcCU, int& dmin, int& dmax) const;
  /// explicit maps from another source (e.g. a validation harness) instead of the GPU, until cleared with NULL
  void setExternalMap(const unsigned char* map, int numCtus) { setExternalRange(map, map, numCtus); }
  void setExternalRange(const unsigned char* mapMin, const unsigned char* mapMax, int numCtus);
  const std::vector<unsigned char>& depthMap() const { return m_depth; }
  /// first-pass candidate lists from another source (validation harness): numCtus * 85 * 8 modes, best first; NULL clears
  void setExternalCandidates(const unsigned char* cand, int numCtus);
  /// the candidate list of the square PU of `size` samples at (xInCtu, yInCtu) of CTU ctuRsAddr: numModes (<= 8) modes into list;
  /// false -> no list (HM runs its own 35-mode pass).  Static: TEncSearch has no path to the TEncCu that owns the instance
  static bool candidateList(unsigned ctuRsAddr, int xInCtu, int yInCtu, int size, int numModes, unsigned* list);
  /// FHEVC_FIRST_PASS_EXTRA=<n>: that many more candidates than HM's numModesForFullRD go to the full RD check where the list comes from the
  /// GPU (the list is built from original, not reconstructed, neighbours: a wider list buys the difference back); total clamped to 8
  static int  candidateExtra();

private:
  bool       m_enabled, m_valid, m_external;
  const TComPic* m_cachePic;               // the picture m_depth belongs to (+ POC, slice QP, slice type)
  int        m_cachePoc, m_cacheQp, m_cacheType;
  unsigned long long m_cacheFp;
  fhevc_ctx* m_ctx;
  int        m_width, m_height, m_bitDepth, m_marginSplit, m_marginStop, m_pWindow;
  enum PMode { P_OFF, P_WINDOW, P_MOTION };
  int        m_pMode, m_pRange;
  bool       m_pMotionCompensated;  ///< reference depths taken at the motion-compensated position (FHEVC_P_MC)
  bool       m_pNodeForm;           ///< ... per CU node of the current grid (FHEVC_P_MC=node: fhevc_p_node_depth) instead of per 4x4 unit
  int        m_firstPassExtra;
  bool       m_firstPass, m_candValid, m_candExternal;   ///< FHEVC_FIRST_PASS; m_cand holds this picture's lists; lists fed by a harness
  std::vector<unsigned char> m_cand;    // numCtus * 85 * 8: the eight cheapest modes per node, best first (255: node crosses the picture edge)
  static TEncFastDepth* s_active;       // the instance whose lists candidateList() reads: the one predictPicture() ran on last
  int        m_pRule[37];                     // fhevc_p_rule (include/fasthevc.h), kept opaque so that this header needs no library header
  bool       ensureContext(TComPic* pcPic);   // (re-)create the GPU context for this picture geometry
  std::vector<unsigned char> m_depth;     // numCtus * 256, raster 16x16 per CTU: depth_min
  std::vector<unsigned char> m_depthMax;  // depth_max (== m_depth when the margin is 0)
};

#endif
