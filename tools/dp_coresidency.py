"""Data-parallel co-residency probe (VERDICT r2 #1a): does a collective-sized kernel on a second stream disturb the
persistent decoder BPTT launch (256 workgroups x 512 threads, up to 158 KB LDS each -- one per CU, every CU), and the other way
round?  RCCL cannot run with one GPU, so an RCCL-footprint stand-in is used: 64 workgroups x 256 threads x 64 KB LDS spinning
for a fixed time (taco_debug_spin), enqueued exactly where GradReducer enqueues the post-net segment's all-reduce:

  bptt_ms      BPTT kernel launch -> end (HIP events on its stream; includes any wait for CUs): decoder3_bwd_kernel fills every
               CU and therefore must never run beside a collective -- the post-net segment is announced AFTER it
  spin_ms      stand-in enqueue-point -> end on the communication stream (nominal = its spin time; more = it waited for CUs)
  bwd_ms       whole taco_backward on the main stream
  err          decoder error words (exchange time-outs)

usage: python tools/dp_coresidency.py  [> profiles/r03_dp_coresidency.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

SPIN_BLOCKS, SPIN_THREADS, SPIN_LDS = 64, 256, 64 * 1024


def build_model(B=32, Tt=200, Td=180):
    from tacotron_amd.config import Config
    from tacotron_amd.data import synthetic_batch
    from tacotron_amd.model import Tacotron
    c = Config()
    c.r, c.vocab_size = 2, 60
    m = Tacotron(c, synthetic_batch(B, Tt, Td, 2, 60), train=True, seed=0)
    return m, m.draw_masks()


def one(m, masks, order, spin_us, comm):
    """order: None (no stand-in), 'segment' (on the communication stream behind taco_wait_grad_segment(post-net), i.e. exactly where
    GradReducer enqueues the post-net all-reduce)."""
    from tacotron_amd import lib
    m.forward(masks)
    torch.cuda.synchronize()
    lib.profile_read(1)
    lib.profile_enable(2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    b0.record(main)
    m.backward()
    b1.record(main)
    if order == 'segment':
        with torch.cuda.stream(comm):
            lib.wait_grad_segment(len(lib.grad_segments(m.shape)) - 2, comm)   # the post-net segment: the first one announced
            e0.record(comm)
            lib.debug_spin(SPIN_BLOCKS, SPIN_THREADS, SPIN_LDS, spin_us, comm)
            e1.record(comm)
    torch.cuda.synchronize()
    lib.profile_enable(0)
    bptt = lib.profile_read(1)
    err = m._err.tolist()
    res = {'bptt_ms': bptt[0], 'bwd_ms': b0.elapsed_time(b1), 'err': err,
           'spin_ms': e0.elapsed_time(e1) if order else None,
           'spin_start_after_bwd_start_ms': b0.elapsed_time(e0) if order else None}
    if err[0] or err[1]:
        lib.clear_error(m.shape, True, m.workspace)
    return res


def measure(spin_us=2000, reps=3, verbose=True):
    from tacotron_amd import lib
    m, masks = build_model()
    comm = torch.cuda.Stream(priority=-1)   # as GradReducer's communication stream
    out = {}
    for _ in range(2):
        one(m, masks, None, 0, comm)

    def med(xs, k):
        v = sorted(x[k] for x in xs if x[k] is not None)
        return v[len(v) // 2] if v else None

    cases = [('solo', None, 0),
             ('default: post-net segment announced after the BPTT kernel', 'segment', spin_us)]
    for name, order, us in cases:
        runs = [one(m, masks, order, us, comm) for _ in range(reps)]
        r = {k: med(runs, k) for k in ('bptt_ms', 'bwd_ms', 'spin_ms', 'spin_start_after_bwd_start_ms')}
        r['err'] = [max(x['err'][0] for x in runs), max(x['err'][1] for x in runs)]
        r['spin_nominal_ms'] = us / 1e3
        out[name] = r
        if verbose:
            print('%-72s bptt %.2f ms  backward %.2f ms  stand-in %s (nominal %.1f, enqueue point %s ms after backward start)  err %s' %
                  (name, r['bptt_ms'], r['bwd_ms'], '%.2f ms' % r['spin_ms'] if r['spin_ms'] is not None else '-', us / 1e3,
                   '%.2f' % r['spin_start_after_bwd_start_ms'] if r['spin_start_after_bwd_start_ms'] is not None else '-', r['err']))
    return out


if __name__ == '__main__':
    print('stand-in: %d workgroups x %d threads x %d KB LDS; S1 shape (B=32, Tt=200, Td=180); medians of 3' %
          (SPIN_BLOCKS, SPIN_THREADS, SPIN_LDS // 1024))
    measure()
