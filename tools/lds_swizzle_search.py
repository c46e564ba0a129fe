#!/usr/bin/env python3
"""The LDS image of the layer path (k_cnn_layers.inc): which row-pitch padding and XOR mask make the B-fragment reads (ds_read_b128, lane (n, h) = 16 bytes
of position n) free of bank conflicts.  Model: /opt/skills/guides/MI355X_MICROARCH.md, LDS table: a ds_read_b128 is served in four groups of 16 lanes, a
16-byte slot = 4 of the 64 banks, N distinct addresses on one slot within a group = N cycles.  Image: byte a of the staged map lives at
a ^ (((a >> 8) & mask) << 4).  Prints the table fhevc_api.hip (layer_lds_image) carries.  usage: python tools/lds_swizzle_search.py"""
GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          [32, 33, 34, 35] + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]


def cycles(addrs):
    tot = 0
    for g in GROUPS:
        slots = {}
        for lane in g:
            slots.setdefault((addrs[lane] >> 4) & 15, set()).add(addrs[lane] >> 4)
        tot += max(len(v) for v in slots.values())
    return tot


def score(kc, pool, H, mask, pad):
    px, Ho, S = kc * 32, (H // 2 if pool else H), (2 if pool else 1)
    pitch = (H + 2) * px + pad
    tot = cnt = 0
    for t in range(Ho * Ho // 32):
        for k in range(kc):
            for ty in range(4 if pool else 3):
                for tx in range(4 if pool else 3):
                    addrs = []
                    for lane in range(64):
                        p = t * 32 + (lane & 31)
                        a = ((p // Ho) * S + ty) * pitch + ((p % Ho) * S + tx) * px + k * 32 + 16 * (lane >> 5)
                        addrs.append(a ^ (((a >> 8) & mask) << 4))
                    tot += cycles(addrs)
                    cnt += 1
    return tot / cnt


if __name__ == "__main__":
    print("kc pool  H : plain -> best (pad bytes, mask)   [4.00 = conflict-free]")
    for kc in (1, 2, 3, 4):
        for pool in (0, 1):
            for H in (16, 32, 64):
                best = None
                for pad in range(0, 256, 16):
                    for mask in (0, 1, 3, 7, 15):
                        sc = score(kc, pool, H, mask, pad)
                        if best is None or sc < best[0] - 1e-9:
                            best = (sc, pad, mask)
                print(f"{kc}  {pool}   {H:2d} : {score(kc, pool, H, 0, 0):5.2f} -> {best[0]:5.2f}  ({best[1]}, {best[2]})")
