"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite result (the rocpd *.db): count, total, avg, share.
usage: python tools/prof_summary.py results.db [images]

Load-time work is NOT per-image work: the runtime's own copy / fill kernels (`__amd_rocclr_*`, one per state-dict tensor when the
model is uploaded) and the weight-repack / finalize kernels (`permute_kernel`) are listed in a section of their own and excluded
from the per-image figures (0.82 ms of "25.43 ms per image" was ~918 load-time copyBuffer launches)."""
import re
import sqlite3
import sys

LOAD_TIME = re.compile(r"__amd_rocclr_|permute_kernel")

db = sqlite3.connect(sys.argv[1])
images = float(sys.argv[2]) if len(sys.argv) > 2 else None
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
work = [r for r in rows if not LOAD_TIME.search(r[0])]
load = [r for r in rows if LOAD_TIME.search(r[0])]
tot = sum(r[2] for r in work)
n = sum(r[1] for r in work)
print("GPU kernel time of the path %.3f ms over %d launches" % (tot / 1e6, n) + (" = %.3f ms / %.0f launches per image" % (tot / 1e6 / images, n / images) if images else "")
      + "   (excluded as load-time: %.3f ms over %d launches, listed at the end)" % (sum(r[2] for r in load) / 1e6, sum(r[1] for r in load)))
print("%-86s %7s %11s %7s %9s %9s %9s" % ("kernel", "calls", "total_us", "share", "avg_us", "min_us", "max_us"))


def show(r, denom):
    name = re.sub(r"\(.*", "", r[0])
    name = re.sub(r"^void ", "", name)[:86]
    print("%-86s %7d %11.1f %6.1f%% %9.2f %9.2f %9.2f" % (name, r[1], r[2] / 1e3, 100.0 * r[2] / denom, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))


for r in work:
    show(r, tot)
if load:
    print("# load-time kernels (model upload / weight repack), not part of the per-image figures; share = of the path's time")
    for r in load:
        show(r, tot)
