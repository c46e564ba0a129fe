#!/usr/bin/env python3
"""Apply the fast-depth hook to a copy of HM-16.14's TEncCu.h / TEncCu.cpp / TEncSlice.cpp / TEncSearch.cpp.

usage: apply_hook.py <TLibEncoder dir of the HM tree> <output dir>
Reads the four files, inserts the lines below at anchors (regular expressions over HM's own identifiers) and writes
the patched copies to <output dir>; nothing else of HM is touched.  The script carries no HM source text.
"""
import os
import re
import sys


def sub_once(text, pattern, repl, what):
    new, n = re.subn(pattern, repl, text, count=1, flags=re.M)
    if n != 1:
        raise SystemExit(f"anchor not found: {what}")
    return new


def patch_cu_h(s):
    s = sub_once(s, r'(#include "TEncSearch.h"\n)', r'\1#include "TEncFastDepth.h"\n', "TEncCu.h include")
    s = sub_once(s, r'(^\s*TEncRateCtrl\*\s+m_pcRateCtrl;\n)', r'\1  TEncFastDepth           m_fastDepth;      ///< per-picture depth map from the GPU path\n', "TEncCu.h member")
    s = sub_once(s, r'(^\s*Void\s+setFastDeltaQp\s*\(.*\n)', r'\1  TEncFastDepth& getFastDepth()                   { return m_fastDepth; }\n', "TEncCu.h accessor")
    return s


def patch_cu_cpp(s):
    # 1. after bBoundary is computed: the two forcing flags
    s = sub_once(s, r'(^\s*const Bool bBoundary = .*;\n)',
                 r'\1\n    Int iDepthMin = 0, iDepthMax = 0;                                        // predicted depth range of this CU\n'
                 r'    const Bool bHaveRange   = m_fastDepth.forcedRange( rpcBestCU, iDepthMin, iDepthMax );\n'
                 r'    // (a split may only be forced where the recursion below really runs -- the condition of the "further split" branch --\n'
                 r'    //  otherwise rpcBestCU would stay at MAX_DOUBLE and the assert at the end of xCompressCU fires)\n'
                 r'    const Bool bCanRecurse  = uiDepth < sps.getLog2DiffMaxMinCodingBlockSize() && ( !getFastDeltaQp() || uiWidth > fastDeltaQPCuMaxSize );\n'
                 r'    const Bool bForceSplit  = bHaveRange && !bBoundary && bCanRecurse && iDepthMin > (Int)uiDepth;   // skip the mode loop at this depth\n'
                 r'    const Bool bForceStop   = bHaveRange && !bBoundary && iDepthMax <= (Int)uiDepth;  // do not recurse below it\n',
                 "bBoundary")
    # 2. the mode loop runs only when the node is not forced to split
    s = sub_once(s, r'^(\s*)if \( !bBoundary \)\n', r'\1if ( !bBoundary && !bForceSplit )\n', "mode loop guard")
    # 3. no recursion below a forced leaf
    s = sub_once(s, r'(^\s*const Bool bSubBranch = )(.*);\n', r'\1( \2 ) && !bForceStop;\n', "bSubBranch")
    return s


def patch_search_cpp(s):
    # the candidate list of estIntraPredLumaQT from the GPU's first pass (FHEVC_FIRST_PASS): HM's own 35-mode Hadamard loop runs only where
    # no list exists (PUs of 4x4, nodes crossing the picture edge, library off); its MPM handling after the loop stays as it is
    s = sub_once(s, r'(#include "TEncSearch.h"\n)', r'\1#include "TEncFastDepth.h"\n', "TEncSearch.cpp include")
    s = sub_once(s, r'^(\s*)for\( Int modeIdx = 0; modeIdx < numModesAvailable; modeIdx\+\+ \)\n',
                 r'\1const Int  iGpuModes = std::min( 8, numModesForFullRD + TEncFastDepth::candidateExtra() );   // FHEVC_FIRST_PASS_EXTRA\n'
                 r'\1const Bool bGpuList = puRect.width == puRect.height && puRect.width >= 8 &&\n'
                 r'\1  TEncFastDepth::candidateList( pcCU->getCtuRsAddr(), Int(pcCU->getCUPelX() & 63) + Int(puRect.x0), Int(pcCU->getCUPelY() & 63) + Int(puRect.y0),\n'
                 r'\1                                Int(puRect.width), iGpuModes, uiRdModeList );\n'
                 r'\1if ( bGpuList ) { numModesForFullRD = iGpuModes; CandNum = numModesForFullRD; }\n'
                 r'\1for( Int modeIdx = 0; !bGpuList && modeIdx < numModesAvailable; modeIdx++ )\n',
                 "first-pass mode loop")
    return s


def patch_slice_cpp(s):
    # once per picture, before the CTU loop of compressSlice
    s = sub_once(s, r'(^\s*m_pcCuEncoder->setFastDeltaQp\(bFastDeltaQP\);\n)',
                 r'\1  m_pcCuEncoder->getFastDepth().predictPicture( pcPic, pcSlice->getSliceQp(), Int(pcSlice->getSliceType()) );\n',
                 "compressSlice")
    return s


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(dst, exist_ok=True)
    for name, fn in (("TEncCu.h", patch_cu_h), ("TEncCu.cpp", patch_cu_cpp), ("TEncSlice.cpp", patch_slice_cpp), ("TEncSearch.cpp", patch_search_cpp)):
        if not os.path.exists(os.path.join(src, name)) and name == "TEncSearch.cpp":
            continue   # a staging directory that holds only the files another patch touched: the caller passes TEncSearch.cpp separately
        with open(os.path.join(src, name)) as f:
            text = f.read()
        with open(os.path.join(dst, name), "w") as f:
            f.write(fn(text))
    print("patched TEncCu.h TEncCu.cpp TEncSlice.cpp TEncSearch.cpp ->", dst)


if __name__ == "__main__":
    main()
