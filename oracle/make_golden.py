"""ORACLE — TEST INFRASTRUCTURE ONLY.  Fixture generator; runs ONLY in the build
container where /root/reference is mounted.  Its outputs (tests/golden/*.npz)
are plain numeric vectors; no reference source travels.

What it does:
  1. imports the reference's own ``network_functions.py`` (trainer / metric
     packages that the image lacks are replaced by empty stubs in sys.modules —
     none of them is on the hot path) and records its outputs on seeded inputs
     -> tests/golden/nf_vectors.npz
  2. imports the reference's own ``c_network.py`` with ``oracle.cpt_oracle``
     registered under the name of the absent third-party package
     ``complexPyTorch`` and records seeded input -> mask, plus gradients of a
     fixed scalar functional of the mask -> tests/golden/cnet_*.npz.
     This pins the WIRING of c_network.py:12-226 (and autograd through it);
     the layer arithmetic underneath is cpt_oracle's (parity unpinned there).

Usage:  python -m oracle.make_golden   (from the repo root)
"""
import os
import sys
import types
import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')


def _install_stubs():
    from oracle import cpt_oracle

    pl = types.ModuleType('pytorch_lightning')

    class _LM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._hp = {}

        @property
        def hparams(self):
            return self._hp

        def save_hyperparameters(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    pl.seed_everything = lambda s: torch.manual_seed(s)
    pl.Callback = object
    pl.LightningModule = _LM
    core = types.ModuleType('pytorch_lightning.core')
    lightning = types.ModuleType('pytorch_lightning.core.lightning')
    lightning.LightningModule = _LM
    core.lightning = lightning
    pl.core = core
    sys.modules['pytorch_lightning'] = pl
    sys.modules['pytorch_lightning.core'] = core
    sys.modules['pytorch_lightning.core.lightning'] = lightning
    for name, attr in (('pypesq', 'pesq'), ('pystoi', 'stoi')):
        m = types.ModuleType(name)
        setattr(m, attr, lambda *a, **k: float('nan'))
        sys.modules[name] = m

    pkg = types.ModuleType('complexPyTorch')
    layers = types.ModuleType('complexPyTorch.complexLayers')
    funcs = types.ModuleType('complexPyTorch.complexFunctions')
    for n in ('ComplexConv2d', 'ComplexConvTranspose2d', 'ComplexBatchNorm2d', 'ComplexLinear', 'ComplexReLU'):
        setattr(layers, n, getattr(cpt_oracle, n))
    layers.ComplexAvgPool2d = type('ComplexAvgPool2d', (torch.nn.Module,), {})  # deleted at c_network.py:6
    funcs.complex_upsample = cpt_oracle.complex_upsample
    funcs.complex_relu = cpt_oracle.complex_relu
    pkg.complexLayers, pkg.complexFunctions = layers, funcs
    sys.modules['complexPyTorch'] = pkg
    sys.modules['complexPyTorch.complexLayers'] = layers
    sys.modules['complexPyTorch.complexFunctions'] = funcs


def _ref_config(nf):
    """Literal geometry of config.py:83-106 (config.py itself needs torchaudio)."""
    c = types.SimpleNamespace()
    c.kernel_sizeE = [7, 7, 5, 5, 3, 3, 3]
    c.kernel_sizeD = [3] * 7
    c.paddingE = [k // 2 for k in c.kernel_sizeE]
    c.paddingD = [k // 2 for k in c.kernel_sizeD]
    c.strideE = [(2, 2), (2, 2), (2, 2), (2, 1), (2, 1), (2, 1), (2, 1)]
    c.strideD = (1, 1)
    from oracle import cpt_oracle
    c.CactivationE = cpt_oracle.ComplexReLU
    c.CactivationD = nf.ComplexLReLU
    c.upsample_scale_factor = [(2, 1), (2, 1), (2, 1), (2, 1), (2, 2), (2, 2), (2, 2)]
    c.upsampling_mode = 'nearest'
    return c


def _ref_hparams(dropout):
    return {'lr': 10e-5, 'initialisation_distribution': torch.nn.init.xavier_uniform_,
            'speech_alpha': 0.7, 'no_of_layers': 7,
            'channels': [1, 16, 32, 64, 128, 256, 256, 256],
            'lstm_layers': 2, 'lstm_bidir': True, 'noise_loss_type': 6, 'speech_loss_type': 0,
            'dropout': True, 'dropout_conv': 0.1 if dropout else 0.0, 'dropout_fc': 0.2 if dropout else 0.0,
            'batch_size': 32, 'optim_eps': 10e-7, 'atan2_eps': 10e-7, 'optim_weight_decay': 10e-5,
            'optim_amsgrad': True, 'channel_attention_reduction_ratio': 16,
            'spatial_attention_kernel_size': 7}


def _c(x):
    return x.detach().cpu().numpy()


def nf_vectors(nf):
    from oracle.seeded_state import seeded_input
    out = {}
    hp = {'atan2_eps': 10e-7}
    for tag, (B, F, T) in (('small', (2, 8, 16)), ('mid', (2, 64, 32))):
        M = seeded_input(B, F, T, seed=1, scale=1.5)
        Y = seeded_input(B, F, T, seed=2, scale=0.7)
        S = seeded_input(B, F, T, seed=3, scale=0.5)
        if tag == 'small':   # edge cases: origin, negative real axis, tiny and huge magnitudes
            M.view(-1)[:6] = torch.tensor([0 + 0j, -1 + 0j, -1e-7 + 0j, 1e-20 + 1e-20j, 50 - 70j, -3 + 1e-9j])
            Y.view(-1)[:2] = torch.tensor([0 + 0j, 1e-5 - 1e-5j])
        out[f'{tag}_M'], out[f'{tag}_Y'], out[f'{tag}_S'] = _c(M), _c(Y), _c(S)
        b1 = nf.bound_cRM(M, hp)
        b2 = nf.bound_cRM(b1, hp)
        out[f'{tag}_bound1'], out[f'{tag}_bound2'] = _c(b1), _c(b2)
        out[f'{tag}_cRM'] = _c(nf.cRM(S, Y))
        nhat = nf.complex_mat_mult(Y, b2)
        out[f'{tag}_nhat'], out[f'{tag}_shat'] = _c(nhat), _c(Y - nhat)
        out[f'{tag}_lrelu'] = _c(nf.complex_lrelu(M))
        out[f'{tag}_sigmoid'] = _c(nf.complex_sigmoid(M))
        x4 = M.view(B, 2, F // 2, T)
        out[f'{tag}_avgpool'] = _c(nf.complex_adaptive_avg_pool2d(x4, output_size=1))
        out[f'{tag}_maxpool'] = _c(nf.complex_adaptive_max_pool2d(x4, output_size=1))
        a, b = torch.view_as_real(S).reshape(B, -1), torch.view_as_real(Y).reshape(B, -1)
        out[f'{tag}_sisnr'] = _c(nf.SiSNR()(a, b))
    return out


def cnet_vectors(cn, nf):
    from oracle.seeded_state import fill_state, seeded_input
    cfg = _ref_config(nf)
    res = {}
    for tag, (B, T, seed) in (('b2t32', (2, 32, 0)), ('b1t16', (1, 16, 1)), ('b3t8', (3, 8, 2))):
        net = cn.C_NETWORK(cfg, _ref_hparams(dropout=False), seed)
        fill_state(net, seed)
        x = seeded_input(B, 256, T, seed=seed)
        res[f'{tag}_x'] = _c(x)
        net.eval()
        with torch.no_grad():
            res[f'{tag}_eval'] = _c(net(x))
        net.train()
        out = net(x)
        res[f'{tag}_train'] = _c(out)
        # running stats after exactly one training-mode forward
        sd = net.state_dict()
        for k in ('initial_batchnorm.running_mean', 'initial_batchnorm.running_covar',
                  'encoder.3.1.running_mean', 'encoder.3.1.running_covar',
                  'decoder.2.1.running_mean', 'decoder.2.1.running_covar'):
            res[f'{tag}_after_{k}'] = _c(sd[k])
        if tag == 'b2t32':
            # gradient of a fixed scalar functional of the (train-mode) mask
            g = torch.Generator().manual_seed(77)
            w = torch.rand(out.shape, generator=g)
            loss = (w * (out.real ** 2 + 0.5 * out.imag ** 2 + 0.25 * out.real * out.imag)).sum()
            net.zero_grad()
            loss.backward()
            res[f'{tag}_loss'] = _c(loss)
            res[f'{tag}_loss_w'] = _c(w)
            names, norms = [], []
            for n, p in net.named_parameters():
                names.append(n)
                norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
            res[f'{tag}_grad_names'] = np.array(names)
            res[f'{tag}_grad_norms'] = np.array(norms, dtype=np.float64)
            keep = ('encoder.0.0.conv_r.weight', 'encoder.0.0.conv_i.bias', 'encoder.0.1.weight',
                    'encoder.2.1.bias', 'initial_batchnorm.weight', 'decoder.6.conv_tran_i.weight',
                    'decoder.5.0.conv_tran_r.bias', 'decoder.3.1.weight', 'fc.fc_r.bias',
                    'skip_attention.11.conv1.conv_r.weight', 'skip_attention.8.fc.0.conv_i.weight',
                    'decoder_attention.10.fc.2.conv_r.weight', 'decoder_attention.7.conv1.conv_i.weight',
                    'lstm.real_lstm.bias_hh_l0')
            pd = dict(net.named_parameters())
            for k in keep:
                res[f'{tag}_grad_{k}'] = _c(pd[k].grad)
    return res


def rnet_vectors(rn, nf):
    """BASELINE configs[0]: R_NETWORK (DR-Net), B=1, real magnitude input, CPU."""
    from oracle.seeded_state import seeded_input, fill_state_stream
    cfg = _ref_config(nf)
    cfg.RactivationE, cfg.RactivationD = torch.nn.ReLU, torch.nn.LeakyReLU
    res = {}
    for tag, (B, T) in (('b1t256', (1, 256)), ('b2t32', (2, 32))):
        net = rn.R_NETWORK(cfg, _ref_hparams(dropout=False), 0)
        fill_state_stream(net, 5)
        x = seeded_input(B, 256, T, seed=9).abs()
        res[f'{tag}_x'] = _c(x)
        net.eval()
        with torch.no_grad():
            res[f'{tag}_eval'] = _c(net(x))
        net.train()
        with torch.no_grad():
            res[f'{tag}_train'] = _c(net(x))
    return res


def rnet_grad_vectors(rn, nf):
    """Gradients of the reference's own R_NETWORK (r_network.py: stock torch.nn layers, no stand-in anywhere): train-mode
    forward on a seeded magnitude input, a fixed scalar functional of the mask, every parameter's gradient norm and a
    set of full gradient tensors."""
    from oracle.seeded_state import seeded_input, fill_state_stream
    cfg = _ref_config(nf)
    cfg.RactivationE, cfg.RactivationD = torch.nn.ReLU, torch.nn.LeakyReLU
    res = {}
    B, T = 2, 32
    net = rn.R_NETWORK(cfg, _ref_hparams(dropout=False), 0)
    fill_state_stream(net, 5)
    net.train()
    x = seeded_input(B, 256, T, seed=9).abs()
    out = net(x)
    g = torch.Generator().manual_seed(78)
    w = torch.rand(out.shape, generator=g)
    loss = (w * (out ** 2 + 0.3 * out)).sum()
    net.zero_grad()
    loss.backward()
    res['x'], res['out'], res['loss_w'], res['loss'] = _c(x), _c(out), _c(w), _c(loss)
    names, norms = [], []
    for n, p in net.named_parameters():
        names.append(n)
        norms.append(float(p.grad.norm()) if p.grad is not None else -1.0)
    res['grad_names'] = np.array(names)
    res['grad_norms'] = np.array(norms, dtype=np.float64)
    keep = ('encoder.0.0.weight', 'encoder.0.0.bias', 'encoder.1.0.weight', 'encoder.3.0.weight', 'encoder.2.1.weight',
            'encoder.5.1.bias', 'initial_batchnorm.weight', 'initial_batchnorm.bias', 'decoder.0.0.weight',
            'decoder.4.0.weight', 'decoder.5.0.bias', 'decoder.6.weight', 'decoder.6.bias', 'decoder.2.1.weight',
            'lstm.weight_hh_l0', 'lstm.weight_ih_l1_reverse', 'lstm.bias_ih_l0', 'fc.weight', 'fc.bias',
            'skip_attention.0.fc.0.weight', 'skip_attention.5.conv1.weight', 'decoder_attention.2.fc.2.weight',
            'decoder_attention.9.conv1.weight')
    pd = dict(net.named_parameters())
    for k in keep:                                   # large tensors: every 37th element (the test samples likewise)
        gk = pd[k].grad
        res[f'grad_{k}'] = _c(gk if gk.numel() <= 20000 else gk.flatten()[::37])
    sd = net.state_dict()
    for k in ('initial_batchnorm.running_mean', 'initial_batchnorm.running_var', 'encoder.2.1.running_mean',
              'encoder.2.1.running_var', 'decoder.1.1.running_var'):
        res[f'after_{k}'] = _c(sd[k])
    return res


def loss_vectors(nf):
    """The reference's calc_loss (network_functions.py:168-208) for every noise_loss_type 0-6 on seeded masks and
    signals (sys.argv[1] must read 'dcs' while it runs)."""
    from oracle.seeded_state import seeded_input
    res = {}
    B, L = 3, 480
    g = torch.Generator().manual_seed(5)
    sig = {k: torch.randn(B, L, generator=g) * s_ for k, s_ in (('noisy_audio', 0.2), ('noise_audio', 0.1),
                                                                 ('clean_audio', 0.15), ('predict_noise_audio', 0.1),
                                                                 ('predict_clean_audio', 0.15))}
    sig['target_noise_mask'] = seeded_input(B, 8, 16, seed=21, scale=0.6)
    sig['predict_noise_mask'] = seeded_input(B, 8, 16, seed=22, scale=0.6)
    for k, v in sig.items():
        res[k] = _c(v)
    cfg = types.SimpleNamespace(L1=torch.nn.L1Loss(), mse=torch.nn.MSELoss(), SiSNR=nf.SiSNR(), wSDR=nf.wSDR())
    for t in range(7):
        me = types.SimpleNamespace(hparams={'noise_loss_type': t, 'speech_loss_type': 0, 'speech_alpha': 0.7}, config=cfg)
        noise_loss, speech_loss, total = nf.calc_loss(me, **sig)
        res[f'type{t}'] = np.array([float(noise_loss), float(speech_loss), float(total)], dtype=np.float64)
    return res


def main():
    if not os.path.isdir(REF):
        raise SystemExit('reference not mounted: fixtures can only be generated in the build container')
    sys.path.insert(0, REPO)
    _install_stubs()
    sys.path.insert(0, REF)
    argv = sys.argv
    sys.argv = ['train.py', 'dcs', '0']          # network_functions.py / c_network.py read sys.argv[1]
    import network_functions as nf
    import c_network as cn
    import r_network as rn
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(1)                      # bit-stable reductions
    np.savez_compressed(os.path.join(OUT, 'loss_vectors.npz'), **loss_vectors(nf))
    np.savez_compressed(os.path.join(OUT, 'rnet_grad_vectors.npz'), **rnet_grad_vectors(rn, nf))
    sys.argv = argv
    if '--only-new' in argv:                      # leave the fixtures of earlier rounds untouched
        return
    np.savez_compressed(os.path.join(OUT, 'nf_vectors.npz'), **nf_vectors(nf))
    np.savez_compressed(os.path.join(OUT, 'cnet_vectors.npz'), **cnet_vectors(cn, nf))
    np.savez_compressed(os.path.join(OUT, 'rnet_vectors.npz'), **rnet_vectors(rn, nf))
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
