"""Input wire format: planar 4:2:0 / 4:0:0 YUV files, 8 or 16 bits per sample.

Mirrors what TVideoIOYuv::read does for the LUMA plane (TVideoIOYuv.cpp:249-380, 675-760): little-endian 16-bit samples
when the file bit depth exceeds 8, right/bottom padding by edge replication up to the conformance size (a multiple
of the minimum CU size, 8), and the InputBitDepth -> InternalBitDepth left shift (scalePlane, :70-84, 730).  Chroma is
skipped with a seek, never read: the GPU path needs luma only, and for 8-bit files the plane is uploaded as uint8
(half the PCIe and HBM bytes of HM's int16 Pel plane; fhevc_predict_frames_device(sample_bytes = 1)).
"""
import os

import numpy as np


class YuvLumaReader:
    def __init__(self, path, width, height, file_bit_depth=8, chroma_format="420"):
        self.path, self.width, self.height, self.file_bit_depth = path, width, height, file_bit_depth
        self.bps = 2 if file_bit_depth > 8 else 1
        chroma = {"400": 0, "420": (width // 2) * (height // 2) * 2, "422": (width // 2) * height * 2,
                  "444": width * height * 2}[chroma_format]
        self.luma_bytes = width * height * self.bps
        self.frame_bytes = self.luma_bytes + chroma * self.bps
        size = os.path.getsize(path)
        self.num_frames = size // self.frame_bytes
        if self.num_frames < 1:
            raise ValueError(f"{path}: shorter than one {width}x{height} frame")
        self._mm = np.memmap(path, dtype=np.uint8, mode="r")  # memory-mapped: large files are never loaded whole

    def padded_size(self, min_cu=8):
        """conformance size HM pads to (ConformanceWindowMode 1: next multiple of the minimum CU size)"""
        return -(-self.width // min_cu) * min_cu, -(-self.height // min_cu) * min_cu

    def luma(self, frame, internal_bit_depth=None, pad=True, as_pel=False):
        """Luma plane of `frame`: uint8 for 8-bit files at 8-bit internal depth (unless as_pel), int16 Pel otherwise."""
        if not 0 <= frame < self.num_frames:
            raise IndexError(frame)
        ibd = internal_bit_depth or self.file_bit_depth
        off = frame * self.frame_bytes
        raw = self._mm[off:off + self.luma_bytes]
        if self.bps == 2:
            y = raw.view("<u2").reshape(self.height, self.width).astype(np.int16)
        else:
            y = np.asarray(raw).reshape(self.height, self.width)
        if pad:
            pw, ph = self.padded_size()
            if (pw, ph) != (self.width, self.height):
                y = np.pad(y, ((0, ph - self.height), (0, pw - self.width)), mode="edge")
        shift = ibd - self.file_bit_depth
        if shift < 0:
            raise ValueError("internal bit depth below the file bit depth is not supported on this path")
        if self.bps == 1 and shift == 0 and not as_pel:
            return np.ascontiguousarray(y)
        return np.ascontiguousarray(y.astype(np.int16) << shift)

    def read_luma_into(self, out, first=0):
        """Fill out[f] (uint8 [count, H, W], e.g. pinned memory from Context.alloc_host) with the luma planes of frames
        first .. first + count - 1 of an 8-bit file whose size needs no padding: one copy file -> destination, chroma is
        never touched.  This is what feeds fhevc_predict_frames."""
        assert self.bps == 1 and out.dtype == np.uint8 and out.shape[1:] == (self.height, self.width)
        assert self.padded_size() == (self.width, self.height), "pad through luma() instead"
        if first < 0 or first + out.shape[0] > self.num_frames:
            raise IndexError((first, out.shape[0]))
        for f in range(out.shape[0]):
            off = (first + f) * self.frame_bytes
            out[f].reshape(-1)[:] = self._mm[off:off + self.luma_bytes]
        return out

    def gop_uint8(self, first, count):
        """`count` luma planes of an 8-bit file as one [count, H, W] uint8 array (what a GOP upload hands the device)."""
        assert self.bps == 1
        return np.stack([self.luma(first + f, pad=True) for f in range(count)])
