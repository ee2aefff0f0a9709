"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite result (the rocpd *.db): count, total, avg, share.
usage: python tools/prof_summary.py results.db [images]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
images = float(sys.argv[2]) if len(sys.argv) > 2 else None
cur = db.cursor()
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
n = sum(r[1] for r in rows)
print("total GPU kernel time %.3f ms over %d launches" % (tot / 1e6, n) + (" = %.3f ms / %.0f launches per image" % (tot / 1e6 / images, n / images) if images else ""))
print("%-86s %7s %11s %7s %9s %9s %9s" % ("kernel", "calls", "total_us", "share", "avg_us", "min_us", "max_us"))
for r in rows:
    name = re.sub(r"\(.*", "", r[0])
    name = re.sub(r"^void ", "", name)[:86]
    print("%-86s %7d %11.1f %6.1f%% %9.2f %9.2f %9.2f" % (name, r[1], r[2] / 1e3, 100.0 * r[2] / tot, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3))
