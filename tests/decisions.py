"""The HIP path's DISCRETE decisions (ReLU on/off, max-pool winner), read back from the workspace tensors it stashes for its
own backward pass, in the site naming of oracle.taco_torch.Decisions.  Used to exhibit fp32-vs-fp64 decision flips and to
compare gradients with the HIP path's decisions imposed on the fp64 graph (tests/test_gpu_sizes.py)."""
import numpy as np
import torch

K_ST_P1, K_ST_P2 = 0, 256   # csrc/kernels.h: pre-net slots of the decoder stash record


def _pool_decisions(bank, pool, gamma, beta, B, T):
    """z = bank * (gamma / sqrt(1 + eps)) + beta exactly as the kernels form it (gemm2.hip pooled epilogues: fp32, the scale
    premultiplied in fp32, one fused multiply-add), checked bit for bit against the pooled tensor the forward pass wrote;
    decision True = row t+1 beats row t (strictly)."""
    rs = np.float32(1.0) / np.sqrt(np.float32(1.0) + np.float32(1e-3))
    sc = (gamma.astype(np.float32) * rs).astype(np.float32)
    be = beta.astype(np.float32)
    x = bank.reshape(B, T, -1)
    cands = {'fma': (x.astype(np.float64) * sc.astype(np.float64) + be.astype(np.float64)).astype(np.float32),
             'mul+add': (x * sc).astype(np.float32) + be}
    want = pool.reshape(B, T, -1)
    for how, z in cands.items():
        nxt = np.concatenate([z[:, 1:], z[:, -1:]], 1)
        if np.array_equal(np.maximum(z, nxt), want):
            take = nxt > z
            take[:, -1] = False
            return take, how
    raise AssertionError('the pooled tensor is reproduced by neither the fused nor the two-rounding form of the BN affine')


def hip_decisions(R, p, masks, B, Tt, Td, r, speakers):
    """R: tests.test_gpu_model.Runner after forward().  Returns (decisions, comparable): name -> bool array; `comparable` marks
    where a decision is observable (a unit dropped by dropout reads 0 whatever the ReLU decided)."""
    d, ok = {}, {}
    how = {}
    for pre, T in (('enc', Tt), ('post', Td * r)):
        prefix = 'encoder/cbhg/' if pre == 'enc' else 'post/cbhg/'
        bank = R.wsget(pre + '.bank')
        d[prefix + 'bank'] = (bank > 0).reshape(B, T, -1)
        d[prefix + 'pool'], how[pre] = _pool_decisions(bank, R.wsget(pre + '.pool'), p[prefix + 'bank_bn/gamma'],
                                                       p[prefix + 'bank_bn/beta'], B, T)
        d[prefix + 'proj1'] = (R.wsget(pre + '.pj1pre') > 0).reshape(B, T, -1)
        for l in range(4):
            d[prefix + 'highway_%d/H' % l] = (R.wsget(pre + '.th%d' % l)[:, 128:] > 0).reshape(B, T, -1)
    if speakers > 1:
        for l in range(4):
            d['encoder/cbhg/highway_%d/spk' % l] = R.wsget('enc.sv_h0')[l] > 0   # (sv[0..3] | h0: one (5,B,128) block since round 6)
        d['encoder/cbhg/gru_init'] = R.wsget('enc.sv_h0')[4] > 0
    d['encoder/pre_net/l1'] = (R.wsget('enc.p1') > 0).reshape(B, Tt, -1)
    d['encoder/pre_net/l2'] = (R.wsget('enc.p2') > 0).reshape(B, Tt, -1)
    ok['encoder/pre_net/l1'] = np.asarray(masks['enc_keep1']) > 0
    ok['encoder/pre_net/l2'] = np.asarray(masks['enc_keep2']) > 0
    st = R.wsget('dec.stash').reshape(B, Td, -1)
    k1, k2 = np.asarray(masks['dec_keep1']) > 0, np.asarray(masks['dec_keep2']) > 0
    for t in range(Td):
        d['decoder/pre_net/l1@%d' % t] = st[:, t, K_ST_P1:K_ST_P1 + 256] > 0
        d['decoder/pre_net/l2@%d' % t] = st[:, t, K_ST_P2:K_ST_P2 + 128] > 0
        ok['decoder/pre_net/l1@%d' % t] = k1[:, t]
        ok['decoder/pre_net/l2@%d' % t] = k2[:, t]
    return d, ok, how


def flips(hip, ok, dec):
    """dec: a taco_torch.Decisions in record mode after the fp64 forward.  Returns a list of (site, index tuple, fp64 margin)
    for every observable decision the HIP path took differently."""
    out = []
    for name, h in hip.items():
        ref = dec.rec[name].numpy()
        diff = h != ref
        if name in ok:
            diff &= ok[name]
        if diff.any():
            mg = dec.margin[name].numpy()
            for idx in zip(*np.nonzero(diff)):
                out.append((name, tuple(int(i) for i in idx), float(mg[idx])))
    return out


def as_force(hip):
    return {k: torch.tensor(v, dtype=torch.float64) for k, v in hip.items()}
