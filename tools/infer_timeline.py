"""Kernel sequence of ONE inference call (B given on the command line) from a rocprofv3 rocpd database.
GPU box:  rocprofv3 --kernel-trace -d gpurun_out/inf -o inf -- python tools/infer_timeline.py run 1
here:     python tools/infer_timeline.py show gpurun_out/inf/.../inf_results.db
(calls are delimited by the persistent decoder kernel, which runs once per call)"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(Bi):
    import torch
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
    b = synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40)
    m = Tacotron(c, b, train=False, seed=0)
    for _ in range(6):
        m.run()
    torch.cuda.synchronize()
    m.check()


def show(path, want=4):
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    grid = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
    wg = 'workgroup_x' if 'workgroup_x' in cols else ('workgroup_size_x' if 'workgroup_size_x' in cols else None)
    extra = ''.join(', ' + x for x in (grid, wg) if x)
    rows = c.execute("select name, start, end%s from kernels order by start" % extra).fetchall()
    call, out, seen_dec = 0, [], False
    # a call = everything from the first kernel after a decoder kernel's successor chain; split at 'embedding'
    for r in rows:
        if 'embed' in r[0] and seen_dec:
            call += 1
            seen_dec = False
        if 'dec' in r[0] and 'infer' in r[0]:
            seen_dec = True
        if call == want:
            out.append(r)
    if not out:
        out = rows[-80:]
    t0 = out[0][1]
    busy = 0.0
    for r in out:
        name = r[0].replace('(anonymous namespace)::', '').replace('void ', '')
        g = (r[3] // r[4]) if (grid and wg and r[4]) else ''
        busy += (r[2] - r[1]) / 1e3
        print('%9.1f us  %8.1f us  wg=%-6s %s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, g, name[:90]))
    print('span %.1f us  busy %.1f us  launches %d' % ((out[-1][2] - t0) / 1e3, busy, len(out)))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]))
    else:
        show(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
