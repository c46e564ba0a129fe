// ref_harness.cpp -- thin extern "C" driver around the REAL reference functions on the hot path.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libhmref.so, together with
// objects compiled from the reference's own sources where they lie under /root/reference
// (TLibCommon/*.cpp, TLibEncoder/*.cpp except TEncGOP.cpp, libmd5).  No reference source is copied,
// patched or stubbed: TEncGOP.cpp / TAppEncoder need OpenCV, which this image lacks, so the full
// encoder is unbuildable here and is not built (see HISTORY.md section 3).
//
// This file only CALLS reference code; it contains none of it.  It is used by oracle/gen_golden.py
// to produce tests/golden/ref_*.npz and by tests/test_reference_rdo.py when oracle/_ref exists.
#include <cstring>
#include <cstdint>
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComPrediction.h"

// defined with external linkage in TEncCu.cpp:1230 (no header declares it)
Int xCalcHADs8x8_ISlice(Pel* piOrg, Int iStrideOrg);
// forward-declared and defined in TComPattern.cpp:51,322
Void fillReferenceSamples(const Int bitDepth, const Pel* piRoiOrigin, Pel* piIntraTemp, const Bool* bNeighborFlags,
                          const Int iNumIntraNeighbor, const Int unitWidth, const Int unitHeight,
                          const Int iAboveUnits, const Int iLeftUnits, const UInt uiWidth, const UInt uiHeight,
                          const Int iPicStride);

namespace {
struct PredProbe : public TComPrediction {  // exposes the protected predictor kernels
  using TComPrediction::xPredIntraAng;
  using TComPrediction::xPredIntraPlanar;
  using TComPrediction::xDCPredFiltering;
};
bool g_rom_ready = false;
void ensure_rom()
{
  if (g_rom_ready) return;
  initROM();
  // TEncCu::create (TEncCu.cpp:126-131) initialises the scan tables like this:
  UInt* p = &g_auiZscanToRaster[0];
  initZscanToRaster(5, 1, 0, p);
  initRasterToZscan(64, 64, 5);
  initRasterToPelXY(64, 64, 5);
  g_rom_ready = true;
}
}  // namespace

extern "C" {

const char* href_version() { return NV_VERSION; }

// TComRdCost::calcHAD (TComRdCost.cpp:297-334)
uint32_t href_calc_had(int bit_depth, const int16_t* a, int sa, const int16_t* b, int sb, int w, int h)
{
  ensure_rom();
  TComRdCost rd;
  return rd.calcHAD(bit_depth, a, sa, b, sb, w, h);
}

// setDistParam(..., bHadamard=true) + DistFunc == xGetHADs (TComRdCost.cpp:282-295, 1753-1824)
uint32_t href_get_hads(int bit_depth, const int16_t* a, int sa, const int16_t* b, int sb, int w, int h)
{
  ensure_rom();
  TComRdCost rd;
  DistParam dp;
  rd.setDistParam(dp, bit_depth, a, sa, b, sb, w, h, true);
  dp.bApplyWeight = false;
  return dp.DistFunc(&dp);
}

// xCalcHADs8x8_ISlice (TEncCu.cpp:1230-1322)
int32_t href_had8x8_islice(const int16_t* org, int stride)
{
  return xCalcHADs8x8_ISlice(const_cast<Pel*>(org), stride);
}

// the loop of TEncCu::updateCtuDataISlice (TEncCu.cpp:1334-1342) over the reference's block function
int32_t href_ctu_src_hadamard(const int16_t* org, int stride, int width, int height)
{
  int32_t sum = 0;
  for (int y = 0; (y + 8) <= height; y += 8)
    for (int x = 0; (x + 8) <= width; x += 8) sum += xCalcHADs8x8_ISlice(const_cast<Pel*>(org) + stride * y + x, stride);
  return sum;
}

// g_auiZscanToRaster / g_auiRasterToZscan after TEncCu::create's init sequence
void href_scan_tables(uint32_t* raster_to_zscan, uint32_t* zscan_to_raster)
{
  ensure_rom();
  for (int i = 0; i < 256; i++) { raster_to_zscan[i] = g_auiRasterToZscan[i]; zscan_to_raster[i] = g_auiZscanToRaster[i]; }
}

// fillReferenceSamples (TComPattern.cpp:322-539).  flags: HM order, 2N/4 left units bottom-to-top, TL,
// 2N/4 above units.  out: the (2N+1) x (2N+1) ROI buffer, row-major (only row 0 and column 0 are written).
void href_fill_ref(int bit_depth, const int16_t* roi_origin, int pic_stride, const uint8_t* flags, int n, int16_t* out)
{
  const int units = 2 * n / 4;
  Bool bf[4 * 16 + 1];
  int navail = 0;
  for (int i = 0; i < 2 * units + 1; i++) { bf[i] = flags[i] != 0; navail += bf[i] ? 1 : 0; }
  fillReferenceSamples(bit_depth, roi_origin, out, bf, navail, 4, 4, units, units, 2 * n + 1, 2 * n + 1, pic_stride);
}

// TComPrediction::filteringIntraReferenceSamples (TComPattern.cpp:541-566), luma, 4:2:0
int href_use_filtered(int mode, int n)
{
  return TComPrediction::filteringIntraReferenceSamples(COMPONENT_Y, (UInt)mode, (UInt)n, (UInt)n, CHROMA_420, false) ? 1 : 0;
}

// One predictor from a (2N+1)x(2N+1) ROI buffer laid out as HM's m_piYuvExt (row 0 = TL + above, column 0 = TL + left),
// following predIntraAng (TComPrediction.cpp:448-470): planar, or xPredIntraAng (+ xDCPredFiltering for DC).
void href_pred_intra(const int16_t* roi, int n, int mode, int bit_depth, int16_t* pred)
{
  ensure_rom();
  static PredProbe* probe = new PredProbe();
  const int sw = 2 * n + 1;
  const Pel* src = roi + sw + 1;
  if (mode == PLANAR_IDX) {
    probe->xPredIntraPlanar(src, sw, pred, n, n, n);
  } else {
    probe->xPredIntraAng(bit_depth, src, sw, pred, n, n, n, CHANNEL_TYPE_LUMA, (UInt)mode, true);
    if (mode == DC_IDX) probe->xDCPredFiltering(src, sw, pred, n, n, n, CHANNEL_TYPE_LUMA);
  }
}

}  // extern "C"
