"""Reproduces bench.py's leg order and prints the decoder cluster width after each leg.  usage: python tools/seq_probe.py [legs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tacotron_amd import lib
from tacotron_amd.config import Config
from tacotron_amd.data import synthetic_batch
from tacotron_amd.model import Tacotron
legs = sys.argv[1] if len(sys.argv) > 1 else 'TFSVI'
def train(Td, S, tag):
    c = Config(); c.r, c.vocab_size, c.num_speakers = 2, 60, S
    m = Tacotron(c, synthetic_batch(32, 200, Td, 2, 60, num_speakers=S), train=True, seed=0)
    for _ in range(3): m.step()
    torch.cuda.synchronize()
    print(tag, 'train P fwd/bwd', lib.last_cluster(0), lib.last_cluster(1), flush=True)
    return m
for ch in legs:
    if ch == 'T':
        m = train(180, 1, 'S1')
    elif ch == 'F':
        lib.profile_enable(0b1100); m.step(); torch.cuda.synchronize(); lib.profile_enable(0)
        print('family', len(lib.profile_read(2)), len(lib.profile_read(3)), flush=True)
    elif ch == 'E':
        del m; torch.cuda.empty_cache()
    elif ch == 'S':
        m2 = train(500, 1, 'S2'); del m2; torch.cuda.empty_cache()
    elif ch == 'V':
        m3 = train(180, 109, 'VCTK'); del m3; torch.cuda.empty_cache()
    elif ch == 'A':   # allocate the inference models now, run them later ('R')
        c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
        held = {Bi: Tacotron(c, synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40), train=False, seed=0) for Bi in (1, 32)}
    elif ch in 'IR':
        c = Config(); c.r, c.vocab_size, c.max_decode_iter = 2, 60, 180
        for Bi in (1, 32):
            mi = held[Bi] if ch == 'R' else Tacotron(c, synthetic_batch(Bi, 140, 180, 2, 60, seed=77, min_len=40), train=False, seed=0)
            print('  params @%x ws @%x' % (mi.params.flat.data_ptr(), mi.workspace.data_ptr()))
            for _ in range(2): mi.run()
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(5): mi.run()
            torch.cuda.synchronize()
            eoff = [o for n, o, sz, d in lib.workspace_table(mi.shape, False) if n == 'dec.err'][0]
            cen = mi.workspace[eoff:eoff + 16].view(torch.int32)[4:12].tolist()
            print('infer B=%d %.2f ms P=%d  census (blockIdx - xcc) mod 8 over 7 launches: %s' % (Bi, (time.perf_counter() - t) / 5 * 1e3, lib.last_cluster(0), cen), flush=True)
