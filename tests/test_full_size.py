"""GPU: the two BASELINE configurations bench.py times, checked END TO END at their own sizes.

  configs[1]  C_NETWORK.eval() forward + bound / mask-apply / subtract at complex64 [16,256,2000]
              (reference c_network.py:187-226, network_functions.py:240-243) — LSTM sequence 500, CBN over 2 M
              pixels, the batched attention tables and every inference tile plan in one pass
  configs[2]  one optimisation step at [32,256,256] (network_functions.py:210-258 + Adam/AMSGrad) — the split-K
              plans, occupancy-derived slab counts, deferred batched reduces and the pack plan as the bench runs them

against the CPU oracle run on the box's host cores (seconds per pass), eager launches AND hipGraph replay.
"""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.cnet_oracle import C_NETWORK_Oracle                  # noqa: E402
from oracle.nf_oracle import dcs_train_losses, mask_apply_subtract  # noqa: E402
from oracle.seeded_state import fill_state, seeded_input           # noqa: E402


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(n, 32)))


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from dcsnet import _lib
    _lib.load()
    _threads()
    return torch.device('cuda:0')


def _hip_net(dev, seed):
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    return fill_state(C_NETWORK(config, hp, seed), seed).to(dev)


def _bias_before_bn(n):
    """A conv bias in front of a batch-statistics CBN has an analytically zero gradient: both sides are rounding noise."""
    return n.endswith('.bias') and (('.0.conv_r' in n or '.0.conv_i' in n) and n.startswith('encoder')
                                    or ('.0.conv_tran_' in n and n.startswith('decoder')))


def test_inference_at_bench_size_against_oracle_eager_and_graph(dev):
    """BASELINE configs[1] as `bench.py --mode infer` runs it."""
    from dcsnet import functional as F
    from dcsnet.config import hparams
    seed, B, T = 11, 16, 2000
    x = seeded_input(B, 256, T, seed=seed, scale=0.5)
    oracle = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).eval()
    with torch.no_grad():
        m_ref = oracle(x)
        M_ref, N_ref, S_ref = mask_apply_subtract(x, m_ref)
    del oracle
    net = _hip_net(dev, seed).eval()
    xd = x.to(dev)

    def run():
        with torch.no_grad():
            m_raw = net(xd)
            return (m_raw, *F.bound_mask_apply_complex(xd, m_raw, hparams['atan2_eps']))

    eager = run()
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static = run()
    g.replay()
    torch.cuda.synchronize()
    for what, outs in (('eager', eager), ('graph', static)):
        m, M, N, S = (t.cpu() for t in outs)
        assert m.shape == (B, 256, T)
        assert torch.isfinite(torch.view_as_real(m)).all(), what
        for name, got, want, tol in (('mask', m, m_ref, 2e-4), ('bounded mask', M, M_ref, 2e-4),
                                     ('N_hat', N, N_ref, 2e-4 * 3), ('S_hat', S, S_ref, 2e-4 * 3)):
            err = float((got - want).abs().max())
            assert err <= tol, (what, name, err)
    # replay reproduces the eager launches bit for bit (same kernels, same plans)
    assert torch.equal(eager[0], static[0])


@pytest.mark.parametrize('use_graph', [False, True])
def test_train_step_at_bench_size_against_oracle(dev, use_graph):
    """BASELINE configs[2] as `bench.py` runs it (dropout off for comparability): loss, all 222 gradient norms and
    a set of full gradient tensors that covers the split-K layers (enc5 / enc6 / dec0), the 16-column kernel (dec5),
    the tap-sum stage (dec6), the small-channel kernels (enc0, attention convs), CBN, LSTM and Linear."""
    from dcsnet.dp import TrainStep
    seed, B, T = 3, 32, 256
    clean, noise = seeded_input(B, 256, T, 1, 0.1), seeded_input(B, 256, T, 2, 0.05)
    noisy = clean + noise
    ref = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).train()
    loss_ref = dcs_train_losses(ref, noise, noisy, clean)[2]
    loss_ref.backward()
    want = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in ref.named_parameters()}
    loss_ref = float(loss_ref)
    del ref

    net = _hip_net(dev, seed).train()
    # lr = 0: the captured graph is replayed on UNCHANGED parameters, so the replayed step is the oracle's step too
    net.hparams['lr'] = 0.0
    net.hparams['optim_weight_decay'] = 0.0
    ts = TrainStep(net, use_graph=use_graph, graph_warmup=1)
    batch = (noise.to(dev), noisy.to(dev), clean.to(dev), list(range(B)))
    for _ in range(3 if use_graph else 1):
        loss = ts(batch)
    torch.cuda.synchronize()
    if use_graph:
        assert ts._graph is not None, 'capture did not happen (fell back to eager)'
    loss = float(loss)
    assert abs(loss - loss_ref) <= 2e-3 * abs(loss_ref), (loss, loss_ref)

    pd = dict(net.named_parameters())
    assert sorted(pd) == sorted(want)
    checked = 0
    for n, w in want.items():
        g = pd[n].grad
        if w is None:                                 # decoder_attention.12 / .13: built, never run
            assert g is None, n
            continue
        gn, wn = float(g.norm()), float(w.norm())
        if _bias_before_bn(n):
            assert gn <= 2e-3 * max(1.0, wn) + 1e-3, (n, gn, wn)
            continue
        assert abs(gn - wn) <= 2e-3 * wn + 1e-7, (n, gn, wn)
        checked += 1
    assert checked >= 200

    full = ['encoder.0.0.conv_r.weight', 'encoder.1.0.conv_i.weight', 'encoder.5.0.conv_r.weight',
            'encoder.6.0.conv_i.weight', 'encoder.6.1.weight', 'decoder.0.0.conv_tran_r.weight',
            'decoder.1.0.conv_tran_i.weight', 'decoder.3.1.bias', 'decoder.5.0.conv_tran_r.weight',
            'decoder.6.conv_tran_r.weight', 'decoder.6.conv_tran_i.bias', 'lstm.real_lstm.weight_hh_l0',
            'lstm.imag_lstm.weight_ih_l1_reverse', 'fc.fc_r.weight', 'skip_attention.1.conv1.conv_r.weight',
            'skip_attention.12.fc.0.conv_i.weight', 'decoder_attention.5.conv1.conv_i.weight',
            'decoder_attention.0.fc.2.conv_r.weight', 'initial_batchnorm.weight']
    for n in full:
        g, w = pd[n].grad.cpu(), want[n]
        err = float((g - w).abs().max())
        assert err <= 2e-3 * float(w.abs().max()) + 1e-7, (n, err, float(w.abs().max()))


def test_complex_lstm_at_inference_sequence_length(dev):
    """S = 500 = the LSTM sequence of [.,256,2000] inputs (c_network.py:200): the persistent recurrence kernel's
    fast tanh / sigmoid over 500 dependent steps against nn.LSTM on the CPU."""
    from dcsnet.c_network import ComplexLSTM
    from oracle.cnet_oracle import ComplexLSTM as OracleLSTM
    torch.manual_seed(0)
    B, S = 4, 500
    ref = OracleLSTM(128, 64, 2, True)
    mod = ComplexLSTM(128, 64, 2, True, True)
    mod.load_state_dict(ref.state_dict())
    z = seeded_input(B, S, 128, seed=4, scale=0.7)
    with torch.no_grad():
        want = ref(z)
        got = mod.to(dev)(z.to(dev)).cpu()
    err = float((got - want).abs().max())
    assert err <= 1e-4 * max(1.0, float(want.abs().max())), err
