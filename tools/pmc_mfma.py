"""Matrix-pipe utilisation per kernel from a rocprofv3 --pmc pass (rocpd database) with the counters
SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES.
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): share of SIMD-cycles with the matrix pipe busy
while the kernel ran (GRBM_GUI_ACTIVE is reported per XCD and summed).
usage: python tools/pmc_mfma.py <results.db> "<build hash / note>" """
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                      "group by kernel_name, counter_name").fetchall()
    k = {}
    for name, cn, n, s in rows:
        d = k.setdefault(name, {})
        d[cn] = s
        d['n'] = n
    print('# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -- python bench.py --steps 2 --warmup 1 '
          '--no-cpu-baseline --no-inference --no-extras')
    print('# build', sys.argv[2] if len(sys.argv) > 2 else '')
    print('# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)')
    print('%-64s %8s %18s %18s %9s' % ('kernel', 'launches', 'MFMA_BUSY_CYCLES', 'GUI_ACTIVE(sum8)', 'MfmaUtil'))
    tot_b = tot_g = 0.0
    for name, d in sorted(k.items(), key=lambda kv: -kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)):
        b, g = d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), d.get('GRBM_GUI_ACTIVE', 0.0)
        util = b / (g / 8.0 * 1024.0) if g else 0.0
        if b > 0:
            tot_b += b
            tot_g += g
        print('%-64s %8d %18.0f %18.0f %8.1f%%' % (name[:64], d['n'], b, g, 100 * util))
    if tot_g:
        print('# all kernels that use the matrix pipe: %.1f%%' % (100 * tot_b / (tot_g / 8.0 * 1024.0)))


if __name__ == '__main__':
    main()
