"""Kernel sequence of ONE training step from a rocprofv3 rocpd database: name, grid, duration, in launch order.
usage: python tools/rocpd_timeline.py <results.db> [step_index]   (steps are delimited by clip_adam_kernel)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    grid = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    q = "select name, start, end%s from kernels order by start" % ((', ' + grid) if grid else '')
    rows = c.execute(q).fetchall()
    want = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    step, out = 0, []
    for r in rows:
        if step == want:
            out.append(r)
        if 'clip_adam' in r[0]:
            step += 1
    t0 = out[0][1]
    for r in out:
        name = r[0].replace('(anonymous namespace)::', '').replace('void ', '')
        print('%9.1f us  %8.1f us  %-8s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3] if grid else '', name[:70]))


if __name__ == '__main__':
    main()
