"""Per-dispatch values of one counter for the kernels whose name contains a filter (rocpd database of a rocprofv3 --pmc pass).
usage: python tools/pmc_per_dispatch.py <results.db> <counter> <name filter>   (raw counter units: KB for FETCH_SIZE / WRITE_SIZE; the
gfx950 correction of tools/pmc_to_json.py is NOT applied here -- compare dispatches with each other, not with the step table)"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cn, flt = sys.argv[2], sys.argv[3]
rows = db.execute("select dispatch_id, kernel_name, grid_size, sum(value) from counters_collection where counter_name = ? "
                  "group by dispatch_id, kernel_name, grid_size order by dispatch_id", (cn,)).fetchall()
for did, name, grid, v in rows:
    if flt in name:
        print('%6d  grid %8d  %14.0f  %s' % (did, grid, v, name[:60]))
