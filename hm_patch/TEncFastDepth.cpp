/* TEncFastDepth.cpp -- see TEncFastDepth.h.  Links against libfasthevc_hip.so (include/fasthevc.h). */
#include "TEncFastDepth.h"

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComRom.h"

#ifndef FHEVC_HOOK_NO_GPU
#include "fasthevc.h"
static_assert(sizeof(fhevc_p_rule) == 37 * sizeof(int), "TEncFastDepth::m_pRule holds a fhevc_p_rule");
#define P_RULE (reinterpret_cast<fhevc_p_rule*>(m_pRule))
#endif

TEncFastDepth::TEncFastDepth()
  : m_enabled(false), m_valid(false), m_external(false), m_cachePic(NULL), m_cachePoc(-1), m_cacheQp(-1), m_cacheType(-1), m_cacheFp(0), m_ctx(NULL), m_width(0), m_height(0), m_bitDepth(0), m_marginSplit(0), m_marginStop(0), m_pWindow(1), m_pMode(P_OFF), m_pRange(4),
    m_pMotionCompensated(false), m_pNodeForm(false), m_firstPassExtra(0), m_firstPass(false), m_candValid(false), m_candExternal(false)
{
  readKnobs();
}

void TEncFastDepth::readKnobs()
{
  const char* en = std::getenv("FHEVC_ENABLE");
  m_enabled = en != NULL && en[0] == '1';
  const char* mg = std::getenv("FHEVC_MARGIN");
  const char* ms = std::getenv("FHEVC_MARGIN_SPLIT");
  const char* mt = std::getenv("FHEVC_MARGIN_STOP");
  // defaults: the calibration that keeps EVERY content family within 1 % BD-rate: splits are forced only above +100000, forbidden only below
  // -64000, HM's own search decides in between.  With the recommended blob depthnet_family_d2.fhw: ten families, three never in a training
  // label, -0.26 .. +0.47 % at 1.2-2.1x less decision time (profiles/r04_bdrate_family_d2_ten_families.json); with depthnet_v2.fhw <= +0.83 %
  // (profiles/r03_bdrate_generalization.json).  On content like the classifier's training set FHEVC_MARGIN_SPLIT=64000 FHEVC_MARGIN_STOP=32000
  // keeps nine of the ten families at or below +0.91 % at 1.4-3.8x.
  m_marginSplit = ms ? std::atoi(ms) : (mg ? std::atoi(mg) : 100000);
  m_marginStop  = mt ? std::atoi(mt) : (mg ? std::atoi(mg) : 64000);
  if (m_marginSplit < 0) m_marginSplit = 0;
  if (m_marginStop < 0) m_marginStop = 0;
  const char* fpk = std::getenv("FHEVC_FIRST_PASS");
  m_firstPass = fpk != NULL && std::atoi(fpk) != 0;
  const char* fpe = std::getenv("FHEVC_FIRST_PASS_EXTRA");
  m_firstPassExtra = fpe != NULL ? std::atoi(fpe) : 0;
  if (m_firstPassExtra < 0) m_firstPassExtra = 0;
  if (m_firstPassExtra > 5) m_firstPassExtra = 5;
  const char* pw = std::getenv("FHEVC_P_WINDOW");
  const char* pm = std::getenv("FHEVC_P_MODE");
  m_pWindow = pw ? std::atoi(pw) : 1;
  m_pMode = P_OFF;
  if (pm != NULL && std::strcmp(pm, "window") == 0) m_pMode = P_WINDOW;
  if (pm != NULL && std::strcmp(pm, "motion") == 0) m_pMode = P_MOTION;
  if (pm == NULL && pw != NULL && m_pWindow >= 0) m_pMode = P_WINDOW;   // FHEVC_P_WINDOW alone keeps its round-1 meaning
  const char* pr = std::getenv("FHEVC_P_RANGE");
  m_pRange = pr ? std::atoi(pr) : 4;
  if (m_pRange < 1) m_pRange = 1;
  if (m_pRange > 64) m_pRange = 64;   // above 8: HM's own integer search (SAD, xPatternSearch) over the window, 8-bit content (fasthevc.h)
  // FHEVC_P_MC: the reference picture's depths are taken where the motion search says the content came from instead of co-located.
  //   node (or 2): per CU node of the CURRENT grid, at the node's displaced centre (fhevc_p_node_depth, round 4): a partition on this picture's grid; under a
  //                global pan of 32 samples per picture -0.26 % where the co-located map gives -0.10 % and the per-unit form +1.76 %, elsewhere equal to
  //                the co-located map (profiles/r04_p_slice_node_*.json);
  //   1:           per 4x4 unit (fhevc_p_motion_compensated_depth, round 3): not aligned to the CU grid, decides worse than the co-located map;
  //   0 / unset:   co-located.
  const char* pc = std::getenv("FHEVC_P_MC");
  m_pMotionCompensated = pc != NULL && (std::atoi(pc) != 0 || pc[0] == 'n');
  m_pNodeForm = pc != NULL && (pc[0] == 'n' || std::atoi(pc) == 2);
  // The wide rule (fhevc_p_rule_default_wide) was fitted -- and measured, profiles/r03_p_slice_wide_*.json -- on SAD features of the +-64 search WITH the
  // reference picture's depths taken at the displaced position: FHEVC_P_RANGE > 8 therefore turns the displaced depths on unless FHEVC_P_MC says
  // otherwise, and with FHEVC_P_MC=0 the wide search feeds the DEFAULT rule (fitted on co-located depths) instead of a rule it was not fitted for
  if (m_pRange > 8 && pc == NULL) { m_pMotionCompensated = true; m_pNodeForm = true; }
#ifndef FHEVC_HOOK_NO_GPU
  if (m_pRange > 8 && m_pMotionCompensated) fhevc_p_rule_default_wide(P_RULE); else fhevc_p_rule_default(P_RULE);
  const char* pt = std::getenv("FHEVC_P_THRESH");   // "split64,split32,split16,stop64,stop32,stop16" in score units (1.0 = 2^18)
  if (pt != NULL)
  {
    double v[6];
    if (std::sscanf(pt, "%lf,%lf,%lf,%lf,%lf,%lf", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5]) == 6)
      for (int l = 0; l < 3; l++) { P_RULE->t_split[l] = (int)(v[l] * 262144.0); P_RULE->t_stop[l] = (int)(v[3 + l] * 262144.0); }
  }
  if (pm != NULL && std::strcmp(pm, "motion") == 0 && pw != NULL) P_RULE->window = m_pWindow;   // overrides the rule's own +-1 clip (4 = off)
#endif
}

TEncFastDepth::~TEncFastDepth()
{
  if (s_active == this) s_active = NULL;
#ifndef FHEVC_HOOK_NO_GPU
  if (m_ctx != NULL) fhevc_destroy(m_ctx);
#endif
}

void TEncFastDepth::setExternalRange(const unsigned char* mapMin, const unsigned char* mapMax, int numCtus)
{
  m_external = mapMin != NULL && mapMax != NULL;
  m_valid = m_external;
  if (m_external)
  {
    m_depth.assign(mapMin, mapMin + (size_t)numCtus * 256);
    m_depthMax.assign(mapMax, mapMax + (size_t)numCtus * 256);
  }
}

TEncFastDepth* TEncFastDepth::s_active = NULL;

void TEncFastDepth::setExternalCandidates(const unsigned char* cand, int numCtus)
{
  m_candExternal = cand != NULL;
  if (m_candExternal) m_cand.assign(cand, cand + (size_t)numCtus * 85 * 8);
  s_active = this;
}

int TEncFastDepth::candidateExtra()
{
  return s_active != NULL ? s_active->m_firstPassExtra : 0;
}

bool TEncFastDepth::candidateList(unsigned ctuRsAddr, int xInCtu, int yInCtu, int size, int numModes, unsigned* list)
{
  const TEncFastDepth* f = s_active;
  if (f == NULL || !(f->m_candExternal || f->m_candValid) || numModes < 1 || numModes > 8) return false;
  int lvl, first;
  switch (size) { case 64: lvl = 0; first = 0; break; case 32: lvl = 1; first = 1; break; case 16: lvl = 2; first = 5; break; case 8: lvl = 3; first = 21; break; default: return false; }
  if ((xInCtu | yInCtu) < 0 || xInCtu + size > 64 || yInCtu + size > 64 || (xInCtu % size) != 0 || (yInCtu % size) != 0) return false;
  const size_t node = (size_t)ctuRsAddr * 85 + first + (size_t)(yInCtu / size) * (1 << lvl) + xInCtu / size;
  if ((node + 1) * 8 > f->m_cand.size()) return false;
  const unsigned char* c = &f->m_cand[node * 8];
  if (c[0] > 34) return false;   // 255: the node crosses the picture edge
  for (int i = 0; i < numModes; i++) list[i] = c[i];
  return true;
}

bool TEncFastDepth::predictPicture(TComPic* pcPic, int sliceQp, int sliceType)
{
  s_active = this;
  if (m_external) return true;       // validation feed wins
  // cheap early-outs first: pictures this hook can do nothing for never pay for the fingerprint below
  // the depth-map layout (16x16 units of a 64x64 CTU, depths 0..3) is what the library produces and forcedRange() reads:
  // any other CTU geometry runs stock RDO
  const TComSPS& sps = pcPic->getPicSym()->getSPS();
  const bool geometryOk = sps.getMaxCUWidth() == 64 && sps.getMaxCUHeight() == 64 && sps.getLog2DiffMaxMinCodingBlockSize() == 3;
  bool possible = geometryOk && (sliceType == I_SLICE ? m_enabled : m_pMode != P_OFF);
#ifdef FHEVC_HOOK_NO_GPU
  if (sliceType == I_SLICE || m_pMode != P_WINDOW) possible = false;   // the CPU-test build of the hook has the temporal window only
#endif
  if (!possible) { m_valid = false; m_candValid = false; m_cachePic = NULL; return false; }
  // compressSlice runs once per slice and once more per precompressSlice iteration (DeltaQpRD): the map of a picture is
  // computed once per (picture object, POC, slice QP, slice type, fingerprint of the original luma) and kept.  The fingerprint
  // (FNV-1a over EVERY luma sample: ~2 M multiplies per 1080p picture, nothing beside HM's seconds per picture) tells a recycled
  // TComPic with a repeated POC (IDR-only streams) from the same picture; a sparse hash could miss a change in the rows it skips
  unsigned long long fp = 1469598103934665603ULL;
  {
    const TComPicYuv* o = pcPic->getPicYuvOrg();
    const Pel* p = o->getAddr(COMPONENT_Y);
    const int w = o->getWidth(COMPONENT_Y), h = o->getHeight(COMPONENT_Y), st = o->getStride(COMPONENT_Y);
    for (int y = 0; y < h; y++)
      for (int x = 0; x < w; x++) { fp ^= (unsigned long long)(unsigned short)p[(size_t)y * st + x]; fp *= 1099511628211ULL; }
  }
  if (m_valid && pcPic == m_cachePic && pcPic->getPOC() == m_cachePoc && sliceQp == m_cacheQp && sliceType == m_cacheType && fp == m_cacheFp) return true;
  m_valid = false;
  m_candValid = false;
  m_cachePic = pcPic; m_cachePoc = pcPic->getPOC(); m_cacheQp = sliceQp; m_cacheType = sliceType; m_cacheFp = fp;
  if (sliceType != I_SLICE)
  {
    // Config 4, "inter-CU depth reuse": a P/B picture takes its depth range from the co-located depths of its first reference
    // picture -- but only when that picture was itself inter coded (intra depths say little about inter depths).
    //   FHEVC_P_MODE=window : those depths +- FHEVC_P_WINDOW levels; host logic only (the depths sit in the DPB)
    //   FHEVC_P_MODE=motion : the GPU searches every CU node of the picture in the ORIGINAL of the reference picture
    //                         (fhevc_motion_search) and fhevc_p_depth_range turns node costs + co-located depths into ranges
    if (m_pMode == P_OFF) return false;
    TComSlice* slice = pcPic->getSlice(pcPic->getCurrSliceIdx());
    if (slice->getNumRefIdx(REF_PIC_LIST_0) < 1) return false;
    TComPic* ref = slice->getRefPic(REF_PIC_LIST_0, 0);
    if (ref == NULL || ref->getSlice(0)->getSliceType() == I_SLICE) return false;
    const int numCtus = (int)pcPic->getNumberOfCtusInFrame();
    m_depth.resize((size_t)numCtus * 256);
    m_depthMax.resize(m_depth.size());
    std::vector<unsigned char> prev((size_t)numCtus * 256);
    for (int c = 0; c < numCtus; c++)
    {
      const TComDataCU* ctu = ref->getCtu(c);
      for (int r = 0; r < 256; r++) prev[(size_t)c * 256 + r] = ctu->getDepth(g_auiRasterToZscan[r]);
    }
    if (m_pMode == P_WINDOW)
    {
      for (size_t i = 0; i < prev.size(); i++)
      {
        const int d = (int)prev[i];
        m_depth[i]    = (unsigned char)(d - m_pWindow < 0 ? 0 : d - m_pWindow);
        m_depthMax[i] = (unsigned char)(d + m_pWindow > 3 ? 3 : d + m_pWindow);
      }
      m_valid = true;
      return true;
    }
#ifdef FHEVC_HOOK_NO_GPU
    return false;
#else
    if (!ensureContext(pcPic)) return false;
    TComPicYuv* org = pcPic->getPicYuvOrg();
    TComPicYuv* rorg = ref->getPicYuvOrg();
    if (rorg == NULL || rorg->getStride(COMPONENT_Y) != org->getStride(COMPONENT_Y)) return false;
    std::vector<fhevc_motion_node> nodes((size_t)numCtus * FHEVC_NODES_PER_CTU);
    fhevc_set_motion_distortion(m_ctx, m_pRange > 8 ? FHEVC_MOTION_SAD : FHEVC_MOTION_SATD);
    const int rc = fhevc_motion_search(m_ctx, org->getAddr(COMPONENT_Y), rorg->getAddr(COMPONENT_Y), org->getStride(COMPONENT_Y), sliceQp,
                                       m_pRange, &nodes[0]);
    if (rc != FHEVC_OK)
    {
      std::fprintf(stderr, "[fasthevc] P picture falls back to full RDO: %s\n", fhevc_last_error(m_ctx));
      return false;
    }
    const int w = org->getWidth(COMPONENT_Y), h = org->getHeight(COMPONENT_Y), cw = (w + 63) / 64;
    for (int c = 0; c < numCtus; c++)
    {
      const int vw = std::min(64, w - (c % cw) * 64), vh = std::min(64, h - (c / cw) * 64);
      unsigned char seen[256];
      const unsigned char* prevCtu = &prev[(size_t)c * 256];
      if (m_pMotionCompensated)
      {
        if ((m_pNodeForm ? fhevc_p_node_depth(&nodes[(size_t)c * FHEVC_NODES_PER_CTU], &prev[0], w, h, c, seen)
                         : fhevc_p_motion_compensated_depth(&nodes[(size_t)c * FHEVC_NODES_PER_CTU], &prev[0], w, h, c, seen)) != FHEVC_OK) return false;
        prevCtu = seen;
      }
      if (fhevc_p_depth_range(&nodes[(size_t)c * FHEVC_NODES_PER_CTU], prevCtu, vw, vh, sliceQp, P_RULE,
                              &m_depth[(size_t)c * 256], &m_depthMax[(size_t)c * 256]) != FHEVC_OK) return false;
    }
    m_valid = true;
    return true;
#endif
  }
  if (!m_enabled) return false;
#ifdef FHEVC_HOOK_NO_GPU
  (void)pcPic; (void)sliceQp; (void)sliceType;
  return false;
#else
  if (!ensureContext(pcPic)) return false;
  TComPicYuv* org = pcPic->getPicYuvOrg();
  m_depth.resize((size_t)pcPic->getNumberOfCtusInFrame() * 256);
  m_depthMax.resize(m_depth.size());
  const int rc = fhevc_predict_frame_range(m_ctx, org->getAddr(COMPONENT_Y), org->getStride(COMPONENT_Y), sliceQp, sliceType,
                                           m_marginSplit, m_marginStop, &m_depth[0], &m_depthMax[0], NULL);
  if (rc != FHEVC_OK)
  {
    std::fprintf(stderr, "[fasthevc] picture falls back to full RDO: %s\n", fhevc_last_error(m_ctx));
    return false;
  }
  if (m_firstPass)
  {
    // the candidate lists of estIntraPredLumaQT for every node of the picture, from the same original plane (one more kernel + 85 x 8 bytes per CTU)
    m_cand.resize((size_t)pcPic->getNumberOfCtusInFrame() * 85 * 8);
    if (fhevc_intra_first_pass_candidates(m_ctx, org->getAddr(COMPONENT_Y), org->getStride(COMPONENT_Y), sliceQp, 8, &m_cand[0]) == FHEVC_OK) m_candValid = true;
    else std::fprintf(stderr, "[fasthevc] first-pass candidates unavailable (HM's own pass runs): %s\n", fhevc_last_error(m_ctx));
  }
  m_valid = true;
  return true;
#endif
}

bool TEncFastDepth::ensureContext(TComPic* pcPic)
{
#ifdef FHEVC_HOOK_NO_GPU
  (void)pcPic;
  return false;
#else
  if (!m_enabled) return false;
  TComPicYuv* org = pcPic->getPicYuvOrg();
  const int w = org->getWidth(COMPONENT_Y), h = org->getHeight(COMPONENT_Y);
  const int bd = pcPic->getPicSym()->getSPS().getBitDepth(CHANNEL_TYPE_LUMA);
  if (m_ctx != NULL && w == m_width && h == m_height && bd == m_bitDepth) return true;
  if (m_ctx != NULL) { fhevc_destroy(m_ctx); m_ctx = NULL; }
  // FHEVC_DEVICES=0,1,2,...: the MI355X devices this (single-threaded, single-process) encoder may use; the library shards the
  // CTU rows of every picture over them and gathers the maps into m_depth (fasthevc.h: fhevc_cfg.num_devices).  FHEVC_DEVICE=<n>: one device
  int devices[16];
  int numDevices = 0;
  if (const char* list = std::getenv("FHEVC_DEVICES"))
  {
    for (const char* p = list; *p != 0 && numDevices < 16;)
    {
      char* end = NULL;
      const long v = std::strtol(p, &end, 10);
      if (end == p) break;
      devices[numDevices++] = (int)v;
      p = (*end == ',') ? end + 1 : end;
    }
  }
  if (numDevices == 0)
  {
    const char* dev = std::getenv("FHEVC_DEVICE");
    devices[numDevices++] = dev ? std::atoi(dev) : 0;
  }
  fhevc_cfg cfg;
  cfg.width = w; cfg.height = h; cfg.bit_depth = bd; cfg.ctu_size = 64; cfg.max_depth = 3;
  cfg.num_devices = numDevices; cfg.device_ids = devices; cfg.weights_path = std::getenv("FHEVC_WEIGHTS");
  cfg.backend = FHEVC_BACKEND_HIP; cfg.max_frames = 1;
  if (cfg.weights_path == NULL || fhevc_create(&m_ctx, &cfg) != FHEVC_OK)
  {
    std::fprintf(stderr, "[fasthevc] disabled: cannot create the GPU context (weights/device)\n");
    m_enabled = false; m_ctx = NULL;
    return false;
  }
  m_width = w; m_height = h; m_bitDepth = bd;
  return true;
#endif
}

bool TEncFastDepth::forcedRange(const TComDataCU* pcCU, int& dmin, int& dmax) const
{
  if (!m_valid) return false;
  const size_t base = (size_t)pcCU->getCtuRsAddr() * 256;
  if (base + 256 > m_depth.size() || base + 256 > m_depthMax.size()) return false;
  // the CU's area in 4x4 units: a node must split if ANY unit inside asks for a deeper CU, and may split if any unit allows
  // it (for the maps the library produces -- constant inside every CU -- this equals reading the top-left unit; it keeps
  // per-unit ranges from other sources, e.g. the temporal window, consistent over the node)
  const int raster = (int)g_auiZscanToRaster[pcCU->getZorderIdxInCtu()];
  const int ux = raster & 15, uy = raster >> 4, n = 16 >> (int)pcCU->getDepth(0);
  int lo = 0, hi = 0;
  for (int y = uy; y < uy + n && y < 16; y++)
    for (int x = ux; x < ux + n && x < 16; x++)
    {
      const int a = (int)m_depth[base + (size_t)y * 16 + x], b = (int)m_depthMax[base + (size_t)y * 16 + x];
      if (a > lo) lo = a;
      if (b > hi) hi = b;
    }
  dmin = lo;
  dmax = hi;
  return dmax >= dmin;
}
