"""Per-kernel sums of whatever counters a rocprofv3 --pmc pass collected (rocpd database).  usage: python tools/pmc_generic.py <results.db> [name filter]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name").fetchall()
k = {}
for name, cn, n, s in rows:
    if flt in name:
        k.setdefault(name, {})[cn] = (n, s)
for name, d in k.items():
    print(name[:70])
    for cn, (n, s) in sorted(d.items()):
        print('    %-28s launches %4d  sum %16.0f  per launch %14.0f' % (cn, n, s, s / n))
