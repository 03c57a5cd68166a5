"""Per-shape counter sums of a rocprofv3 --pmc pass over tools/gemm_pmc.py.  usage: python tools/gemm_pmc_read.py <results.db> <order.json>
Dispatches whose kernel name contains 'gemm' are assigned to the shapes of order.json in launch order."""
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
order = json.load(open(sys.argv[2]))
rows = db.execute("select dispatch_id, kernel_name, counter_name, sum(value), min(start), max(end) from counters_collection "
                  "group by dispatch_id, counter_name order by dispatch_id").fetchall()
disp = {}
for did, kn, cn, v, st, en in rows:
    if 'gemm' not in kn: continue
    d = disp.setdefault(did, {'kernel': kn, 'ns': en - st})
    d[cn] = v
ids = sorted(disp)
pos = 0
print('# dispatches with gemm in the name: %d; expected %d' % (len(ids), sum(o['n'] for o in order)))
for o in order:
    mine = [disp[i] for i in ids[pos:pos + o['n']]][1:]   # first launch of a shape = warm-up
    pos += o['n']
    if not mine: continue
    keys = sorted(k for k in mine[0] if k not in ('kernel', 'ns'))
    avg = {k: sum(m.get(k, 0) for m in mine) / len(mine) for k in keys}
    ns = sum(m['ns'] for m in mine) / len(mine)
    line = '%-36s %-28s %8.1f us(pmc)' % (o['name'], mine[0]['kernel'].split('::')[-1][:28], ns / 1e3)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in avg and 'GRBM_GUI_ACTIVE' in avg and avg['GRBM_GUI_ACTIVE'] > 0:
        line += '  MfmaUtil %5.1f%%  clk %.2f GHz' % (100 * avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] / 8 * 1024),
                                                     avg['GRBM_GUI_ACTIVE'] / 8 / ns)
    if 'SQ_WAVE_CYCLES' in avg and avg['SQ_WAVE_CYCLES'] > 0:
        wc = avg['SQ_WAVE_CYCLES']
        line += '  ' + ' '.join('%s %.1f%%' % (k.replace('SQ_', ''), 100 * avg[k] / wc) for k in keys if k != 'SQ_WAVE_CYCLES')
    print(line)
