"""Whole-step HBM-side traffic per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases) over
`bench.py --steps S --warmup W --no-cpu-baseline --no-inference --no-extras` (S + W train steps and nothing else).
FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md: the counter tallies 128-byte requests at 64 bytes); both counters
sit on the L2's fabric side, so Infinity-Cache hits are included -- this is traffic leaving the XCD L2s, an upper bound on HBM.
usage: python tools/pmc_step.py <fetch.db> <write.db> <steps_in_run> <out.json> <build hash>"""
import json
import sqlite3
import sys

FAMILIES = (('decoder (persistent fwd + BPTT)', ('decoder3_', 'decoder_fwd', 'decoder_bwd')),
            ('MFMA GEMM family', ('conv_gemm', 'gemm_tn', 'highway_stack', 'mlp2_kernel')),
            ('bi-GRU recurrences', ('bigru_',)),
            ('elementwise / layout / optimiser', ('',)))
# algorithmic bytes of one S1 train step (parameters + Adam slots + activations written once and read once)
ALGORITHMIC_STEP_BYTES = 0.8e9
# launches of the run that are NOT part of a train step: torch's zero-fills of the freshly allocated buffers (workspace, Adam slots),
# the runtime's own copy / fill kernels and bench.py's box probes.  They used to be divided by the step count and added to the
# "elementwise" family (0.31 GB "per step" in profiles/r06_pmc_step.txt before this correction); reported separately now.
SETUP = ('at::native::', '__amd_rocclr_', 'clock_probe_kernel', 'fabric_probe_kernel')


def per_kernel(dbfile, counter):
    db = sqlite3.connect(dbfile)
    return {name: (n, s) for name, n, s in db.execute(
        "select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name", (counter,))}


def family_of(name):
    for fam, keys in FAMILIES:
        if any(k in name for k in keys):
            return fam
    return FAMILIES[-1][0]


def main():
    fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
    steps = int(sys.argv[3])
    fam = {f: {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches_per_step': 0.0} for f, _ in FAMILIES}
    kern = {}
    setup = {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches': 0}
    for name in set(fetch) | set(write):
        f = 2.0 * fetch.get(name, (0, 0.0))[1] * 1024.0 / steps
        w = write.get(name, (0, 0.0))[1] * 1024.0 / steps
        n = max(fetch.get(name, (0, 0))[0], write.get(name, (0, 0))[0]) / steps
        if any(k in name for k in SETUP):
            setup['fetch_bytes'] += f * steps
            setup['write_bytes'] += w * steps
            setup['launches'] += int(round(n * steps))
            continue
        d = fam[family_of(name)]
        d['fetch_bytes'] += f
        d['write_bytes'] += w
        d['launches_per_step'] += n
        short = name.replace('(anonymous namespace)::', '').replace('void ', '')
        cut = short.find('(')
        short = (short[:cut] if cut > 0 else short)[:70]
        e = kern.setdefault(short, {'fetch_bytes': 0.0, 'write_bytes': 0.0, 'launches_per_step': 0.0})
        e['fetch_bytes'] += f
        e['write_bytes'] += w
        e['launches_per_step'] += n
    for d in fam.values():
        d['bytes'] = d['fetch_bytes'] + d['write_bytes']
    total = sum(d['bytes'] for d in fam.values())
    res = {'build': sys.argv[5] if len(sys.argv) > 5 else '', 'steps_in_run': steps, 'step_bytes': total,
           'algorithmic_step_bytes': ALGORITHMIC_STEP_BYTES, 'ratio': total / ALGORITHMIC_STEP_BYTES, 'families': fam,
           'setup_not_in_a_step': setup,
           'kernels': dict(sorted(kern.items(), key=lambda kv: -(kv[1]['fetch_bytes'] + kv[1]['write_bytes']))),
           'note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH doubled (gfx950); L2 fabric-side counters: '
                   'Infinity-Cache hits included'}
    json.dump(res, open(sys.argv[4], 'w'), indent=1)
    print('step total %.3f GB (algorithmic %.1f GB: %.2fx)' % (total / 1e9, ALGORITHMIC_STEP_BYTES / 1e9, total / ALGORITHMIC_STEP_BYTES))
    for f, d in fam.items():
        print('  %-36s %8.3f GB  (fetch %.3f, write %.3f; %.0f launches)' % (f, d['bytes'] / 1e9, d['fetch_bytes'] / 1e9, d['write_bytes'] / 1e9,
                                                                             d['launches_per_step']))
    print('  (set-up launches of the run, not part of a step: %d launches, %.3f GB in total)' %
          (setup['launches'], (setup['fetch_bytes'] + setup['write_bytes']) / 1e9))
    for k, d in list(res['kernels'].items())[:14]:
        print('    %-60s %8.1f MB fetch %8.1f MB write  x%.1f' % (k, d['fetch_bytes'] / 1e6, d['write_bytes'] / 1e6, d['launches_per_step']))


if __name__ == '__main__':
    main()
