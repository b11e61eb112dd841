"""R_NETWORK (DR-Net / DRS-Net) — the reference's real-valued twin of C_NETWORK (r_network.py:8-173), forward pass on
the HIP path.  BASELINE.json configs[0]; SURVEY.md §8(f) rank 4.

Same module tree, constructor and state_dict keys as the reference (stock torch.nn containers hold the parameters:
`encoder.{i}.0.weight`, `encoder.{i}.1.running_mean`, `lstm.weight_ih_l0`, `skip_attention.{2i}.fc.0.weight`, ...), so a
reference checkpoint loads.  `forward` does not run those containers; it runs

  * every Conv2d / ConvTranspose2d on the complex path's fp32 MFMA implicit-GEMM kernel: a real channels-last activation
    with an even channel count IS an interleaved complex one with half the channels, and a real conv is that kernel's
    GEMM with a B panel that lacks the complex 2x2 block structure (`dcs_rconv2d_fwd`, panel packed here); cat +
    nearest-upsample are resolved in the kernel's gather exactly as for C_NETWORK;
  * enc0 (1 -> 16 channels) and dec6 (32 -> 1) as COMPLEX convs on the existing kernels: pairing the 16 real filters
    into 8 complex ones reproduces enc0 on an input with zero imaginary part; dec6's single real output is the real part
    of a 16 -> 1 complex conv with weights (w_even - j w_odd);
  * BatchNorm2d + ReLU / LeakyReLU as the CBN apply kernel with per-real-channel coefficients (a diagonal 2x2 block per
    complex pair), batch statistics (train mode) from two reductions of the activation;
  * both attentions of a block in one C-ABI call (dcs_rattention_fwd, csrc/r_attention.hip): channel max pool, FC,
    per-pixel (mean, max) of ca*x, the 7x7 conv (2 -> 1 real = 1 -> 1 complex with the same pairing trick, direct
    kernel) and the broadcast multiply;

  * the LSTM recurrence on the persistent kernel of the complex path, instantiated for hidden size 128 (one weight set);
    its input projections and the Linear are plain rocBLAS GEMMs.

Plumbing that stays in ATen on this path: the one-channel initial BatchNorm (a scalar affine), the batch statistics of
train-mode BatchNorm, the final sigmoid.  Forward only: there is no hand-written backward for the real path (training DR-Net is out of scope).

Quirks kept (r_network.py): channel attention = sigmoid(fc(max_pool)) only (:23-24); dropout_fc gated by
hparams['dropout'] (:152) while dropout_conv is not; torch.squeeze drops the batch dimension at B = 1 (:171).
"""
import torch

from . import ops
from . import functional as F
from ._lib import DcsHipError
from ._pl_compat import LightningModule, seed_everything


class RealChannelAttention(torch.nn.Module):          # r_network.py:8-26
    def __init__(self, no_channels, reduction_ratio):
        super().__init__()
        hidden = max(no_channels // reduction_ratio, 1)
        self.avg_pool = torch.nn.AdaptiveAvgPool2d(1)
        self.max_pool = torch.nn.AdaptiveMaxPool2d(1, return_indices=False)
        self.fc = torch.nn.Sequential(torch.nn.Conv2d(no_channels, hidden, 1, bias=False), torch.nn.ReLU(),
                                      torch.nn.Conv2d(hidden, no_channels, 1, bias=False))
        self.sigmoid = torch.nn.Sigmoid()

    def hip(self, x):
        """x: [B,H,W,C] real channels-last -> ca [B,C]."""
        mx = x.amax(dim=(1, 2))
        w1, w2 = self.fc[0].weight.flatten(1), self.fc[2].weight.flatten(1)
        return torch.sigmoid(torch.relu(mx @ w1.t()) @ w2.t())


class RealSpatialAttention(torch.nn.Module):          # r_network.py:29-42
    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv1 = torch.nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = torch.nn.Sigmoid()

    def hip(self, z):
        """z: [B,H,W,C] -> sa [B,H,W,1]: sigmoid(conv7x7(cat(mean_c z, max_c z))) as a 1 -> 1 complex conv whose real part
        is w_mean * mean + w_max * max (weights w_mean - j w_max)."""
        pooled = torch.stack([z.mean(dim=-1), z.amax(dim=-1)], dim=-1).unsqueeze(3)      # [B,H,W,1,2]
        w = self.conv1.weight                                                            # [1, 2, k, k]
        wp, bias = _packed(self, 'sa', (w,), lambda: ops.pack_conv_weight(w[:, 0:1].contiguous(), (-w[:, 1:2]).contiguous()))
        k = self.kernel_size
        sa = ops.cconv2d(pooled.contiguous(), None, wp, bias, (k, k), (1, 1), (k // 2, k // 2), (1, 1), F.ACT_SIGMOID)
        return sa[..., 0]                                                                 # real part: [B,H,W,1]


_PACKS = {}


def _packed(owner, tag, tensors, make):
    """Inference-time constant derived from parameters: cached per (module, tag), guarded by tensor identity/version and
    the global state generation (functional.state_generation: bumped by anything that rewrites parameters in place)."""
    key = (F.state_generation(), tuple((id(t), t.data_ptr(), t._version) for t in tensors))
    ent = _PACKS.get((id(owner), tag))
    if ent is not None and ent[0] == key and ent[1]() is owner:
        return ent[2]
    import weakref
    val = make()
    if not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
        if len(_PACKS) > 512:
            _PACKS.clear()
        _PACKS[(id(owner), tag)] = (key, weakref.ref(owner), val)
    return val


def pack_real_panel(w_corr):
    """Correlation kernel w_corr [Cout_r, Cin_r, kh, kw] -> the MFMA B panel of dcs_rconv2d_fwd (include/dcsnet_hip.h):
    element (tap, kg, nt, lane = 32*kk + j, e) = B[tap][8*kg + 4*kk + e][32*nt + j], B[tap][k][n] = w_corr[n, k, dy, dx]."""
    co, ci, kh, kw = w_corr.shape
    if ci % 16 or co % 16:
        raise DcsHipError(f'real MFMA conv needs channel counts in multiples of 16, got {ci} -> {co}')
    taps, nt = kh * kw, (co + 31) // 32
    b = w_corr.permute(2, 3, 1, 0).reshape(taps, ci, co)                     # [tap][k][n]
    if nt * 32 != co:
        b = torch.nn.functional.pad(b, (0, nt * 32 - co))
    b = b.reshape(taps, ci // 8, 2, 4, nt, 32).permute(0, 1, 4, 2, 5, 3)     # [tap, kg, nt, kk, j, e]
    return b.contiguous().reshape(-1)


def rconv2d(x1, x2, panel, bias, cout, ksize, stride, pad, up=(1, 1), act=F.ACT_NONE):
    """Real conv over cat(x1, x2) (channels-last [B,H,W,Cr]), nearest-upsampled by `up`, on the MFMA kernel."""
    from . import _lib
    lib = _lib.load()
    B, H, W, c1 = x1.shape
    c2 = 0 if x2 is None else x2.shape[3]
    geo = (B, H, W, c1, c2, up[0], up[1], cout, ksize[0], ksize[1], stride[0], stride[1], pad[0], pad[1])
    ho = (H * up[0] + 2 * pad[0] - ksize[0]) // stride[0] + 1
    wo = (W * up[1] + 2 * pad[1] - ksize[1]) // stride[1] + 1
    y = torch.empty((B, ho, wo, cout), dtype=torch.float32, device=x1.device)
    nbytes = lib.dcs_rconv2d_fwd_workspace_bytes(*geo)
    if nbytes < 0:
        raise DcsHipError(f'rconv2d: unsupported geometry {geo}')
    ws = ops._workspace(nbytes, x1.device) if nbytes > 0 else None
    _lib.check(lib.dcs_rconv2d_fwd(_lib.ptr(x1), _lib.ptr(x2), _lib.ptr(panel), _lib.ptr(bias), _lib.ptr(y), _lib.ptr(ws),
                                   0 if ws is None else ws.numel(), *geo, act, _lib.cur_stream()), 'dcs_rconv2d_fwd')
    return y


class R_NETWORK(LightningModule):
    def __init__(self, config, hparams, seed):
        super().__init__()
        seed_everything(seed)
        self.config = config
        self.hparams.update(hparams)
        self.save_hyperparameters(self.hparams)
        hp = self.hparams
        ch, L = hp['channels'], hp['no_of_layers']

        self.encoder = torch.nn.ModuleList()          # registration order: r_network.py:52-55
        self.decoder = torch.nn.ModuleList()
        self.decoder_attention = torch.nn.ModuleList()
        self.skip_attention = torch.nn.ModuleList()

        self.initial_batchnorm = torch.nn.BatchNorm2d(ch[0])
        for i in range(L):
            self.encoder.append(torch.nn.Sequential(
                torch.nn.Conv2d(1 if i == 0 else ch[i], ch[i + 1], kernel_size=config.kernel_sizeE[i],
                                stride=config.strideE[i], padding=config.paddingE[i]),
                torch.nn.BatchNorm2d(ch[i + 1]), config.RactivationE()))
        self.lstm = torch.nn.LSTM(input_size=ch[5], hidden_size=ch[4], num_layers=hp['lstm_layers'],
                                  bidirectional=hp['lstm_bidir'], batch_first=True)
        self.fc = torch.nn.Linear(ch[5], ch[5])
        self.dropout_conv = torch.nn.Dropout(hp['dropout_conv'])
        self.dropout_fc = torch.nn.Dropout(hp['dropout_fc'])
        ratio, sk = hp['channel_attention_reduction_ratio'], hp['spatial_attention_kernel_size']
        for i in range(L):
            cin, cout = ch[L - i], max(ch[L - 1 - i], 1)
            convt = torch.nn.ConvTranspose2d(2 * cin, cout, kernel_size=config.kernel_sizeD[i], stride=config.strideD,
                                             padding=config.paddingD[i])
            if i == L - 1:
                self.decoder.append(convt)
            else:
                self.decoder.append(torch.nn.Sequential(convt, torch.nn.BatchNorm2d(ch[L - 1 - i]), config.RactivationD()))
            self.skip_attention.append(RealChannelAttention(cin, ratio))
            self.skip_attention.append(RealSpatialAttention(sk))
            self.decoder_attention.append(RealChannelAttention(cout, ratio))
            self.decoder_attention.append(RealSpatialAttention(sk))
        self.weights_init()

    def weights_init(self):                            # r_network.py:126-137
        init = self.hparams['initialisation_distribution']
        for m in self.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)):
                init(m.weight)

    # ---- pieces ---------------------------------------------------------------------------------------------------
    def _bn_act(self, bn, x, act):
        """BatchNorm2d (+ activation) of a channels-last real tensor [B,H,W,Cr] with even Cr through the CBN apply kernel:
        per complex pair (2k, 2k+1) the coefficients are the diagonal block (a0, 0, 0, a3 | c0, c1)."""
        B, H, W, C = x.shape
        if self.training or not bn.track_running_stats:
            flat = x.reshape(-1, C)
            mean, var = flat.mean(0), flat.var(0, unbiased=False)
            if self.training and bn.track_running_stats:
                n = flat.shape[0]
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                    m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
                    bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
                    bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
        else:
            mean, var = bn.running_mean, bn.running_var
        scale = bn.weight * torch.rsqrt(var + bn.eps)
        shift = bn.bias - mean * scale
        z = torch.zeros_like(scale[0::2])
        coef = torch.stack([scale[0::2], z, z, scale[1::2], shift[0::2], shift[1::2]], dim=1).contiguous()   # [C/2, 6]
        stats = torch.zeros((C // 2, 8), dtype=torch.float32, device=x.device)
        y, _, _ = ops.cbn(x.view(B, H, W, C // 2, 2), None, None, None, None, bn.eps, None, False, act,
                          coef_cached=(stats, coef))
        return y.view(B, H, W, C)

    def _conv(self, conv, x1, x2=None, up=(1, 1), transposed=False):
        w = conv.weight
        if transposed:                                 # stride-1 ConvTranspose2d = correlation with the flipped, swapped kernel
            kh, kw = w.shape[2:]
            pad = (kh - 1 - conv.padding[0], kw - 1 - conv.padding[1])
            panel = _packed(conv, 'panel', (w,), lambda: pack_real_panel(w.flip(2, 3).transpose(0, 1)))
            cout, stride = w.shape[1], (1, 1)
        else:
            pad, stride, cout = conv.padding, conv.stride, w.shape[0]
            panel = _packed(conv, 'panel', (w,), lambda: pack_real_panel(w))
        return rconv2d(x1, x2, panel, conv.bias, cout, tuple(w.shape[2:]), tuple(stride), tuple(pad), up)

    def _enc0(self, conv, x):
        """1 -> 16 real channels as a 1 -> 8 complex conv on (x, 0): filters (w[2k] + j w[2k+1]); the complex layer adds
        (b_r - b_i) + j (b_r + b_i), so b_r = (b[2k] + b[2k+1]) / 2, b_i = (b[2k+1] - b[2k]) / 2."""
        w, b = conv.weight, conv.bias
        def make():
            be, bo = b[0::2], b[1::2]
            return ops.pack_conv_weight(w[0::2].contiguous(), w[1::2].contiguous(), ((be + bo) / 2).contiguous(),
                                        ((bo - be) / 2).contiguous())
        wp, bias = _packed(conv, 'enc0', (w, b), make)
        xc = torch.stack([x, torch.zeros_like(x)], dim=-1).unsqueeze(3)                    # [B,F,T,1,2]
        y = ops.cconv2d(xc, None, wp, bias, tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))
        return y.flatten(3)                                                                 # [B,F,T,16]

    def _dec_last(self, convt, d, skip, up):
        """(C1 + C2) -> 1 real channels: real part of a complex conv with weights (w[2k] - j w[2k+1]) over cat(d, skip)."""
        w, b = convt.weight, convt.bias                                                     # [Cin_r, 1, kh, kw]
        def make():
            return ops.pack_conv_weight(w[0::2].contiguous(), (-w[1::2]).contiguous(), b.contiguous(),
                                        torch.zeros_like(b), transposed=True, up=up)
        wp, bias = _packed(convt, 'dec6', (w, b), make)
        B, H, W, c1 = d.shape
        kh, kw = convt.kernel_size
        pad = (kh - 1 - convt.padding[0], kw - 1 - convt.padding[1])
        y = ops.cconv2d(d.view(B, H, W, c1 // 2, 2), skip.view(B, H, W, skip.shape[3] // 2, 2), wp, bias, (kh, kw), (1, 1),
                        pad, up)
        return y[..., 0, 0]                                                                 # [B, Hout, Wout]

    def _lstm(self, x):
        """torch.nn.LSTM (bidirectional, batch_first) forward: per layer one input-projection GEMM for all time steps
        (rocBLAS) + the hand-written recurrence (dcs_lstm_layer_fwd, hidden size 128, one weight set)."""
        lstm = self.lstm
        if not (lstm.bidirectional and lstm.batch_first and lstm.hidden_size in (64, 128)):
            raise DcsHipError('R_NETWORK: LSTM geometry outside the HIP recurrence (bidirectional, batch_first, hidden 64/128)')
        B, S, _ = x.shape
        Hh = lstm.hidden_size
        inp = x.reshape(B * S, -1)
        for layer in range(lstm.num_layers):
            names = [f'_l{layer}', f'_l{layer}_reverse']
            ps = [getattr(lstm, k + n) for n in names for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            def make():
                w_ih = torch.cat([getattr(lstm, 'weight_ih' + n) for n in names])                       # [8H, in]
                bias = torch.cat([getattr(lstm, 'bias_ih' + n) + getattr(lstm, 'bias_hh' + n) for n in names])
                w_hh = torch.stack([getattr(lstm, 'weight_hh' + n) for n in names]).unsqueeze(0).contiguous()   # [1,2,4H,H]
                return w_ih, bias, w_hh
            w_ih, bias, w_hh = _packed(lstm, f'lstm{layer}', ps, make)
            gx = torch.addmm(bias, inp, w_ih.t())                                                        # [B*S, 2*4H]
            out, _, _ = ops.lstm_layer(gx, w_hh, 1, B, S, (0, S * 8 * Hh, 8 * Hh), False)               # [B, S, 2H]
            inp = out.reshape(B * S, 2 * Hh)
        return inp.view(B, S, 2 * Hh)

    @staticmethod
    def _attend(ca_m, sa_m, x):
        """sa (.) ca (.) x (r_network.py:155-158 / :166-167) in one C-ABI call: dcs_rattention_fwd."""
        from . import _lib
        lib = _lib.load()
        B, H, W, C = x.shape
        w1, w2, w = ca_m.fc[0].weight, ca_m.fc[2].weight, sa_m.conv1.weight
        wp, bias = _packed(sa_m, 'sa', (w,), lambda: ops.pack_conv_weight(w[:, 0:1].contiguous(), (-w[:, 1:2]).contiguous()))
        nbytes = lib.dcs_rattention_workspace_bytes(B, H * W, C)
        if nbytes < 0:
            raise DcsHipError(f'R_NETWORK attention: unsupported channel count {C}')
        ws = ops._workspace(nbytes, x.device)
        ca = torch.empty((B, C), dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        _lib.check(lib.dcs_rattention_fwd(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(ca),
                                          _lib.ptr(y), _lib.ptr(ws), ws.numel(), B, H, W, C, w1.shape[0], sa_m.kernel_size,
                                          _lib.cur_stream()), 'dcs_rattention_fwd')
        return y

    # ---- forward (r_network.py:140-173) ---------------------------------------------------------------------------
    def forward(self, x):
        hp, cfg = self.hparams, self.config
        L = hp['no_of_layers']
        if x.dim() != 3 or x.dtype != torch.float32 or not x.is_cuda:
            raise DcsHipError(f'R_NETWORK.forward expects CUDA float32 [B,F,T], got {x.dtype} {tuple(x.shape)} on {x.device}')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise DcsHipError('R_NETWORK: the HIP path is forward-only (call under torch.no_grad())')
        B, Fb, T = x.shape
        bn0 = self.initial_batchnorm                     # one channel: a scalar affine (no pair to hand the CBN kernel)
        if self.training:
            mean, var = x.mean(), x.var(unbiased=False)
            if bn0.track_running_stats:
                n = x.numel()
                bn0.num_batches_tracked += 1
                m = bn0.momentum if bn0.momentum is not None else 1.0 / float(bn0.num_batches_tracked)
                bn0.running_mean.mul_(1 - m).add_(mean * m)
                bn0.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)) * m)
        else:
            mean, var = bn0.running_mean[0], bn0.running_var[0]
        e = (x - mean) * torch.rsqrt(var + bn0.eps) * bn0.weight[0] + bn0.bias[0]
        feats = [e]                                                                          # [B,F,T] (C = 1)
        drop = self.dropout_conv if self.training and self.dropout_conv.p > 0 else None
        for i in range(L):
            conv, bn = self.encoder[i][0], self.encoder[i][1]
            y = self._enc0(conv, feats[0]) if i == 0 else self._conv(conv, feats[i])
            y = self._bn_act(bn, y, F.ACT_RELU)
            feats.append(drop(y) if drop is not None else y)
        lat = feats[L]
        _, F7, T7, C7 = lat.shape
        z = self.fc(self._lstm(lat.reshape(B, F7 * T7, C7)))
        if hp['dropout'] and self.training:
            z = self.dropout_fc(z)
        d = z.reshape(B, F7, T7, C7)
        for i in range(L):
            skip = self._attend(self.skip_attention[2 * i], self.skip_attention[2 * i + 1], feats[L - i])
            stage = self.decoder[i]
            up = tuple(cfg.upsample_scale_factor[i])
            if i == L - 1:
                d = self._dec_last(stage, d, skip, up)
            else:
                y = self._conv(stage[0], d, skip, up, transposed=True)
                y = self._bn_act(stage[1], y, F.ACT_LRELU)
                d = self._attend(self.decoder_attention[2 * i], self.decoder_attention[2 * i + 1], y)
            if drop is not None:
                d = drop(d)
        return torch.sigmoid(torch.squeeze(d))
