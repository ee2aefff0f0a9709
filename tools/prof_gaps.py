"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace sqlite result (graph replays included):
usage: python tools/prof_gaps.py results.db [min_burst_kernels]"""
import sqlite3
import sys

import numpy as np

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
st = np.array([r[0] for r in rows], dtype=np.float64)
en = np.array([r[1] for r in rows], dtype=np.float64)
gap = st[1:] - np.maximum.accumulate(en)[:-1]
# bursts = runs of kernels separated by < 50 us (one graph replay / one image)
cut = np.where(gap > 50e3)[0]
bounds = np.concatenate([[0], cut + 1, [len(st)]])
print("%d kernels, %d bursts" % (len(st), len(bounds) - 1))
for a, b in zip(bounds[:-1], bounds[1:]):
    if b - a < (int(sys.argv[2]) if len(sys.argv) > 2 else 1000):
        continue
    span = en[a:b].max() - st[a]
    busy = (en[a:b] - st[a:b]).sum()
    g = gap[a:b - 1]
    print("burst of %5d kernels: span %8.3f ms, sum of kernel durations %8.3f ms, idle between kernels %7.3f ms (median gap %.2f us, p90 %.2f us), overlapped (negative gaps) %7.3f ms"
          % (b - a, span / 1e6, busy / 1e6, g[g > 0].sum() / 1e6, np.median(g) / 1e3, np.percentile(g, 90) / 1e3, -g[g < 0].sum() / 1e6))
