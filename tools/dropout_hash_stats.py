"""Statistics of two per-element mixers for the dropout mask on sequential indices, CPU only: lowbias32 (csrc/common.h, the product's) and dropmix24 (the same
shape on 24-bit multiplies: built and measured in round 6, no gain, not adopted — profiles/r06_w_dropout_hash_ab.txt):
drop rate at p = 0.1 / 0.5, largest serial correlation of the mask over lags 1 ... 4096, chi-square (255 d.o.f.) of the top and bottom byte
of the 24-bit value, four (inner, start) draws of 4 M elements each.
    python tools/dropout_hash_stats.py"""
import numpy as np
m32, m24 = np.uint64(0xFFFFFFFF), np.uint64(0xFFFFFF)


def lowbias32(v):
    v = v & m32
    v ^= v >> np.uint64(16); v = (v * np.uint64(0x7feb352d)) & m32
    v ^= v >> np.uint64(15); v = (v * np.uint64(0x846ca68b)) & m32
    v ^= v >> np.uint64(16)
    return v


def dropmix24(v):
    v = v & m32
    v ^= v >> np.uint64(16); v = ((v & m24) * np.uint64(0xD35A2D)) & m32
    v ^= v >> np.uint64(15); v = ((v & m24) * np.uint64(0xB97F4B)) & m32
    v ^= v >> np.uint64(16)
    return v


for name, f in (("lowbias32", lowbias32), ("dropmix24", dropmix24)):
    rng = np.random.default_rng(0)
    print(name)
    for trial in range(4):
        inner = np.uint64(rng.integers(0, 2 ** 32))
        i = np.arange(1 << 22, dtype=np.uint64) + np.uint64(rng.integers(0, 2 ** 31))
        h = f((i & m32) ^ inner) >> np.uint64(8)
        row = []
        for p in (0.1, 0.5):
            keep = (h >= np.uint64(int(p * 16777216))).astype(np.float64)
            k = keep - keep.mean()
            ac = max(abs(float((k[:-l] * k[l:]).mean() / k.var())) for l in (1, 2, 3, 4, 8, 16, 64, 1024, 4096))
            row.append(f"p={p}: drop rate {1 - keep.mean():.5f}, max |serial corr| {ac:.4f}")
        e = len(h) / 256
        chi_hi = float((((np.bincount((h >> np.uint64(16)).astype(np.int64), minlength=256) - e) ** 2) / e).sum())
        chi_lo = float((((np.bincount((h & np.uint64(0xff)).astype(np.int64), minlength=256) - e) ** 2) / e).sum())
        print("   " + "; ".join(row) + f"; chi2 top byte {chi_hi:.0f}, bottom byte {chi_lo:.0f}")
