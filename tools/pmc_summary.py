"""Summarise a rocprofv3 --pmc sqlite result: per kernel name, mean of every counter. usage: pmc_summary.py results.db [filter]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("tables:", tabs)
    sys.exit(0)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from %s group by kernel_name, counter_name" % view).fetchall()
out = {}
for k, c, v, n in rows:
    if flt and flt not in k:
        continue
    out.setdefault(re.sub(r"\(.*", "", k)[:80], {})[c] = (v, n)
for k, d in out.items():
    print(k)
    for c, (v, n) in sorted(d.items()):
        print("    %-28s %16.1f  (n=%d)" % (c, v, n))
