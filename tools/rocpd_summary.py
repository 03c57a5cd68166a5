"""Summarise a rocprofv3 rocpd sqlite database (ROCm 7.2 default output) as a per-kernel stats table
(the equivalent of `--stats` CSV output) and, for --pmc runs, per-kernel counter sums.
usage: python tools/rocpd_summary.py <results.db> [--skip-first N_dispatches_fraction]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    print('# columns of view `kernels`:', ','.join(cols))
    print('%-72s %7s %12s %12s %12s %12s %6s' % ('Name', 'Calls', 'TotalNs', 'AvgNs', 'MinNs', 'MaxNs', 'Pct'))
    for n, cnt, tot, avg, mn, mx in rows:
        print('%-72s %7d %12d %12.0f %12d %12d %6.2f' % (n[:72], cnt, tot, avg, mn, mx, 100.0 * tot / total))
    print('TOTAL kernel ns', total)
    try:
        pm = c.execute("select k.name, p.name, count(*), sum(e.value), avg(e.value) from pmc_events e "
                       "join kernels k on 1=0 join pmc_info p on 1=0").fetchall()
    except Exception:
        pm = None
    try:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        if ccols:
            print('# counters_collection columns:', ','.join(ccols))
            q = c.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                          "group by kernel_name, counter_name order by sum(value) desc").fetchall()
            print('%-60s %-14s %7s %16s %16s' % ('Kernel', 'Counter', 'N', 'Sum', 'Avg/dispatch'))
            for k, cn, n, s, a in q:
                print('%-60s %-14s %7d %16.1f %16.1f' % (k[:60], cn, n, s, a))
    except Exception as e:
        print('# no counters:', e)


if __name__ == '__main__':
    main()
