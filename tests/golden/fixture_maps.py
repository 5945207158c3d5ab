"""Procedural maps shared by gen_golden.py and the tests (no RNG, integer jitter only).

A fixture stores a coarse [C,16,26] fp32 array; the full [C,128,208] map is
    full[c,y,x] = coarse[c, y//8, x//8] + jitter * ((31*y + 17*x + 7*c) % 97)
so that neighbouring pixels differ (index errors show up) while the committed
fixture stays a few KB.
"""
import numpy as np


def expand(coarse, jitter):
    coarse = np.asarray(coarse, np.float32)
    C = coarse.shape[0]
    full = np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)
    c = np.arange(C, dtype=np.int64)[:, None, None]
    y = np.arange(full.shape[1], dtype=np.int64)[None, :, None]
    x = np.arange(full.shape[2], dtype=np.int64)[None, None, :]
    j = ((31 * y + 17 * x + 7 * c) % 97).astype(np.float32)
    return (full + np.float32(jitter) * j).astype(np.float32)
