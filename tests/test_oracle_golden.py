"""CPU: the oracle (oracle/seg_gan_cpu.py) against the golden vectors that oracle/gen_golden.py
produced from the REFERENCE's own modules.  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import seg_gan_cpu as O


def _digests(params, grads=False):
    rows = []
    for p in params:
        t = (p.grad if grads else p).detach().double()
        rows.append([t.sum().item(), t.abs().sum().item(), (t * t).sum().sqrt().item()])
    return np.array(rows)


def test_oracle_init_and_two_steps_match_reference():
    g = np.load(os.path.join(GOLDEN, 'step_n2_64.npz'))
    G, D, og, od = O.make_models()
    assert list(G.state_dict().keys()) == [str(k) for k in g['state_keys_G']]
    assert list(D.state_dict().keys()) == [str(k) for k in g['state_keys_D']]
    assert len(G.state_dict()) == 269 and len(D.state_dict()) == 55            # SURVEY.md 5 (checkpoint format)
    assert np.allclose(_digests(G.parameters()), g['init_G'], rtol=1e-10, atol=1e-12)
    assert np.allclose(_digests(D.parameters()), g['init_D'], rtol=1e-10, atol=1e-12)
    inp, tgt = O.synthetic_batch(2, 64, 64)
    assert np.array_equal(inp.numpy(), g['input']) and np.array_equal(tgt.numpy(), g['target'])
    r = O.gan_step(G, D, og, od, inp, tgt)
    assert np.abs(r['out'].numpy() - g['s0_logits']).max() < 1e-6
    got = np.array([r[k] for k in ('loss', 'closs', 'adv_g', 'adv_d', 'iou', 'dice')])
    assert np.abs(got - g['s0_scalars']).max() < 1e-6
    assert np.allclose(_digests(G.parameters())[:, 1], g['s0_g_step_G'][:, 1], rtol=1e-6, atol=1e-7)
    assert np.allclose(_digests(D.parameters())[:, 1], g['s0_d_step_D'][:, 1], rtol=1e-6, atol=1e-7)
    r = O.gan_step(G, D, og, od, inp, tgt)                                     # second step: state carry-over
    got = np.array([r[k] for k in ('loss', 'closs', 'adv_g', 'adv_d', 'iou', 'dice')])
    assert np.abs(got - g['s1_scalars']).max() < 2e-4
    assert np.median(np.abs(r['out'].numpy() - g['s1_logits'])) < 1e-5


def test_oracle_blocks_match_reference():
    g = np.load(os.path.join(GOLDEN, 'blocks.npz'))
    for tag in ('bb_a', 'bb_b', 'bb_c'):
        cin, cout, hw = [int(v) for v in g[tag + '_cfg']]
        torch.manual_seed(11)
        m = O.ResBlockCPU(cin, cout).train()
        x = torch.from_numpy(g[tag + '_x']).requires_grad_(True)
        y = m(x)
        assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 1e-6
        y.backward(torch.from_numpy(g[tag + '_dy']))
        assert np.abs(x.grad.numpy() - g[tag + '_dx']).max() < 1e-6
    for tag in ('sp_a', 'sp_b'):
        c, hw = [int(v) for v in g[tag + '_cfg']]
        torch.manual_seed(12)
        m = O.SelfSpadeCPU(c, 3, c / 16).train()
        x = torch.from_numpy(g[tag + '_x']).requires_grad_(True)
        y = m(x)
        assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 1e-6
        y.backward(torch.from_numpy(g[tag + '_dy']))
        assert np.abs(x.grad.numpy() - g[tag + '_dx']).max() < 1e-6
    for tag in ('d_96', 'd_64'):
        torch.manual_seed(14)
        m = O.DiscriminatorCPU(3, 3, 8, 8, 1024).train()
        x = torch.from_numpy(g[tag + '_x']).requires_grad_(True)
        y = m(x)
        assert np.abs(y.detach().numpy() - g[tag + '_y']).max() < 1e-6
    x = torch.from_numpy(g['loss_x']); t = torch.from_numpy(g['loss_t'])
    assert abs(O.bce_dice_loss(x, t).item() - float(g['loss_val'])) < 1e-7
    assert abs(O.stable_bce(x, t).item() - float(g['loss_bce'])) < 1e-7
    assert abs(O.iou_score(x[:, 1:].clone(), t[:, 1:].clone()) - float(g['loss_iou'])) < 1e-9
    assert abs(O.dice_coef(x[:, 1:].clone(), t[:, 1:].clone()) - float(g['loss_dice'])) < 1e-7


def test_oracle_config1_step():
    """BASELINE.json configs[0]: 4 x 3x256x256, one G+D step on the CPU."""
    g = np.load(os.path.join(GOLDEN, 'step_n4_256.npz'))
    torch.set_num_threads(8)
    G, D, og, od = O.make_models()
    inp, tgt = O.synthetic_batch(4, 256, 256)
    r = O.gan_step(G, D, og, od, inp, tgt)
    got = np.array([r[k] for k in ('loss', 'closs', 'adv_g', 'adv_d', 'iou', 'dice')])
    assert np.abs(got - g['s0_scalars']).max() < 1e-5
    assert np.abs(r['out'].numpy()[:, :, ::8, ::8] - g['s0_logits_ds']).max() < 1e-5


def test_nan_and_inf_guards():
    """train_seg_gan.py:190 and losses.py:297-300 fallbacks."""
    x = torch.zeros(1, 3, 4, 4); t = torch.ones(1, 3, 4, 4)
    x[0, 0, 0, 0] = float('inf')
    l = O.bce_dice_loss(x, t)
    p = torch.sigmoid(x).view(1, -1); tt = t.view(1, -1)
    dice = 1 - ((2 * (p * tt).sum(1) + 1e-5) / (p.sum(1) + tt.sum(1) + 1e-5)).sum()
    assert torch.isfinite(l) and abs(l.item() - 2 * dice.item()) < 1e-6
    y = torch.tensor([[float('nan'), 1.0]])
    assert O.iou_score(y, torch.tensor([[1.0, 1.0]])) == pytest.approx((1 + 1e-5) / (2 + 1e-5))


def test_oracle_stage1_matches_reference():
    """Stage-1 step (train.py:79-115): weight clamp between forward and backward, Adam with weight decay."""
    g = np.load(os.path.join(GOLDEN, 'stage1_n2_64.npz'))
    torch.manual_seed(41)
    model = O.UNetRSSv2CPU(3, 3, False).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-7)
    inp, tgt = O.synthetic_batch(2, 64, 64)
    r = O.stage1_step(model, opt, inp, tgt)
    assert np.abs(np.array([r['loss'], r['iou'], r['dice']]) - g['s0_scalars']).max() < 1e-6
    assert np.abs(r['out'].numpy() - g['s0_logits']).max() < 1e-6
    assert np.allclose(_digests(model.parameters())[:, 1], g['s0_params'][:, 1], rtol=1e-6, atol=1e-6)
    r = O.stage1_step(model, opt, inp, tgt)
    assert np.abs(np.array([r['loss'], r['iou'], r['dice']]) - g['s1_scalars']).max() < 1e-3


def test_oracle_data_parallel_step_matches_reference():
    """"sync-BN-at-W == single-process batch W*b".  The fixture drives the REFERENCE's SynchronizedBatchNorm2d
    (parallel branch + its own _compute_mean_std) over the concatenated batch of 2 ranks x 2 tiles; the oracle's restatement
    of that arithmetic must reproduce it."""
    g = np.load(os.path.join(GOLDEN, 'step_dp_w2_n4_64.npz'))
    G, D, _, _ = O.make_models()
    G = O.convert_sync_batchnorm(G); D = O.convert_sync_batchnorm(D)
    og = torch.optim.Adam(G.parameters(), lr=2e-5); od = torch.optim.Adam(D.parameters(), lr=2e-5)
    assert [k for k, _ in G.named_parameters()] == [str(k) for k in g['param_names_G']]
    assert sum(isinstance(m, O.SyncBatchNorm2dCPU) for m in list(G.modules()) + list(D.modules())) == int(g['n_sync_bn'])
    inp, tgt = O.synthetic_batch(4, 64, 64)
    assert np.array_equal(inp.numpy(), g['input'])
    r = O.gan_step(G, D, og, od, inp, tgt)
    assert np.abs(r['out'].numpy() - g['s0_logits']).max() < 2e-6
    got = np.array([r[k] for k in ('loss', 'closs', 'adv_g', 'adv_d', 'iou', 'dice')])
    assert np.abs(got - g['s0_scalars']).max() < 2e-6
    assert np.allclose(_digests(G.parameters())[:, 1], g['s0_g_step_G'][:, 1], rtol=1e-6, atol=1e-7)
    assert np.allclose(_digests(D.parameters())[:, 1], g['s0_d_step_D'][:, 1], rtol=1e-6, atol=1e-7)
    bufs = np.stack([[b.double().sum().item(), b.double().abs().sum().item(), (b.double() ** 2).sum().sqrt().item()] for b in G.buffers()])
    assert np.allclose(bufs[:, 1], g['s0_bufs_G'][:, 1], rtol=1e-6, atol=1e-7)
    # and it is NOT the unsynchronised formula: the same step with stock batch norm gives other logits
    G2, D2, og2, od2 = O.make_models()
    r2 = O.gan_step(G2, D2, og2, od2, inp, tgt)
    assert np.abs(r2['out'].numpy() - g['s0_logits']).max() > 1e-5
