"""CPU: oracle/unwired_ops_cpu.py against golden vectors from the reference's own modules."""
import os

import numpy as np
import torch
import torch.nn as nn

from conftest import GOLDEN
from oracle import unwired_ops_cpu as U


def test_unwired_oracle_matches_reference():
    g = np.load(os.path.join(GOLDEN, 'unwired.npz'))
    outs, mean, unb = U.sync_bn_forward([torch.from_numpy(g['sbn_xa']), torch.from_numpy(g['sbn_xb'])],
                                        torch.from_numpy(g['sbn_w']), torch.from_numpy(g['sbn_b']))
    assert np.abs(outs[0].numpy() - g['sbn_ya']).max() < 1e-5
    assert np.allclose(mean.numpy(), g['sbn_mean'], atol=1e-6)
    assert np.allclose(0.1 * unb.numpy() + 0.9, g['sbn_running_var'], rtol=1e-5)
    torch.manual_seed(32)
    m = U.UpConvCPU(16, 8).train()
    assert np.abs(m(torch.from_numpy(g['up_x'])).detach().numpy() - g['up_y']).max() < 1e-6
    for tag in ('xr_a', 'xr_b'):
        c, hw = [int(v) for v in g[tag + '_cfg']]
        torch.manual_seed(33)
        m = U.XResidualBlockCPU(c, c).train()
        assert sum(p.numel() for p in m.parameters()) == int(g[tag + '_nparams'])
        x = torch.from_numpy(g[tag + '_x']).requires_grad_(True)
        y = m(x)
        assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 1e-5
        y.backward(torch.from_numpy(g[tag + '_dy']))
        assert np.abs(x.grad.numpy() - g[tag + '_dx']).max() < 1e-5
    w, u, v = torch.from_numpy(g['sn_w_orig']), torch.from_numpy(g['sn_u0']), torch.from_numpy(g['sn_v0'])
    wsn, u1, v1, sigma = U.spectral_norm_step(w, u, v, 1)
    assert np.abs(u1.numpy() - g['sn_u1']).max() < 1e-6 and np.abs(v1.numpy() - g['sn_v1']).max() < 1e-6
    assert np.abs(wsn.numpy() - g['sn_w1']).max() < 1e-6
    for tag in ('mb_a', 'mb_b', 'mb_c', 'mb_d'):
        k, s, inp, out, e, hw = [int(v) for v in g[tag + '_cfg']]
        torch.manual_seed(35)
        m = U.MBConvCPU(k, s, inp, out, e, 0.25, 224, stride_literal=[s]).train()      # fixtures use stride=[s] (decoded form)
        x = torch.from_numpy(g[tag + '_x']).requires_grad_(True)
        y = m(x)
        assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 1e-5, tag
        y.backward(torch.from_numpy(g[tag + '_dy']))
        assert np.abs(x.grad.numpy() - g[tag + '_dx']).max() < 1e-5 * max(1.0, np.abs(g[tag + '_dx']).max()), tag
