"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd databases) into profiles/*.json:
per decoder kernel, fabric-side bytes per launch with the gfx950 FETCH_SIZE correction (x2, MI355X_MICROARCH.md).
usage: python tools/pmc_to_json.py <fetch.db> <write.db> <out.json> "<build note>" """
import json
import sqlite3
import sys


def per_launch(dbfile, counter):
    db = sqlite3.connect(dbfile)
    out = {}
    for name, n, s in db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? "
                                 "group by kernel_name", (counter,)):
        out[name] = s / n
    return out


def main():
    fetch = per_launch(sys.argv[1], 'FETCH_SIZE')
    write = per_launch(sys.argv[2], 'WRITE_SIZE')
    res = {}
    for key in ('decoder3_fwd_kernel', 'decoder3_bwd_kernel', 'decoder_fwd_kernel', 'decoder_bwd_kernel'):
        f = [v for k, v in fetch.items() if key in k]
        w = [v for k, v in write.items() if key in k]
        if not f or not w:
            continue
        res[key] = {
            'FETCH_SIZE_KB_per_launch_raw': f[0], 'WRITE_SIZE_KB_per_launch_raw': w[0],
            'hbm_bytes_per_launch': (2.0 * f[0] + w[0]) * 1024.0,
            'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (bench.py --steps 2 --warmup 1); FETCH_SIZE '
                    'doubled per the gfx950 correction in MI355X_MICROARCH.md; fabric-side counters (Infinity-Cache hits included)',
        }
    res['build'] = sys.argv[4] if len(sys.argv) > 4 else ''      # bench.source_hash(): ties the counters to a kernel build
    res['build_note'] = sys.argv[5] if len(sys.argv) > 5 else ''
    json.dump(res, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
