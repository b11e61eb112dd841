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

Training (r_network.py:176-363 hooks, network_functions.py:224-232): with autograd on, `forward` runs the same kernels
behind autograd nodes —
  * _RConvFn: forward = the real MFMA conv; data gradient = the same kernel over the zero-inserted cotangent with the
    flipped, in/out-swapped panel (dcs_rconv2d_bwd_data) + dcs_upsample_cat_bwd; weight gradient = the COMPLEX
    weight-gradient kernels run twice, on x and on conj(x): with D_qr = sum_p g_q x_r they return
    (D_rr + D_ii) + j (D_ir - D_ri) and (D_rr - D_ii) + j (D_ir + D_ri), i.e. all four real blocks of the gradient;
  * _RBnFn: BatchNorm2d (all of them, the one-channel initial one included) on the CBN statistics / apply / backward
    kernels with diagonal coefficients (dcs_rbn_fwd / dcs_rbn_bwd);
  * enc0 / dec6 / the 7x7 attention convs: the complex conv nodes of the DCS path over weight views of the real
    parameters (autograd maps the gradients back through the pairing);
  * the LSTM recurrence forward + BPTT kernels at hidden size 128, its projections and the Linear as rocBLAS GEMMs.
  * _RAttendFn: an attention block as one node — dcs_rattention_pool_fwd / _apply_fwd and their _bwd twins for the pools, the
    FC and the broadcast products (first-maximum gradient routing, as torch.max / AdaptiveMaxPool2d), the 7x7 conv through
    dcs_cconv2d_fwd / _bwd_data / _bwd_weight on the (mean, max) pair as one complex channel.
In ATen under autograd: dropout, the final sigmoid, and small glue (sigmoid' of the spatial map, weight-gradient views).

Quirks kept (r_network.py): channel attention = sigmoid(fc(max_pool)) only (:23-24); dropout_fc gated by
hparams['dropout'] (:152) while dropout_conv is not; torch.squeeze drops the batch dimension at B = 1 (:171).
"""
import os
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


_CONJ = {}


def _conj_vec(device):
    """(1, -1) on the device: made by device-side fills (a host-to-device copy is not permitted while a stream is being
    captured into a hipGraph, which is where TrainStep's captured step calls this)."""
    v = _CONJ.get(device)
    if v is None:
        v = torch.ones(2, dtype=torch.float32, device=device)
        v[1].fill_(-1.0)
        if not (device.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
            _CONJ[device] = v
    return v


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


def rconv2d_bwd_data(gy, panel_bwd, Hv, Wv, cin, ksize, stride, pad):
    """Gradient of the virtual (upsampled, concatenated) input [B,Hv,Wv,cin] of a real conv: dcs_rconv2d_bwd_data."""
    from . import _lib
    lib = _lib.load()
    B, Ho, Wo, cout = gy.shape
    geo = (B, Hv, Wv, cin, cout, ksize[0], ksize[1], stride[0], stride[1], pad[0], pad[1])
    nbytes = lib.dcs_rconv2d_bwd_data_workspace_bytes(*geo)
    if nbytes < 0:
        raise DcsHipError(f'rconv2d_bwd_data: unsupported geometry {geo}')
    ws = ops._workspace(nbytes, gy.device) if nbytes > 0 else None
    gxv = torch.empty((B, Hv, Wv, cin), dtype=torch.float32, device=gy.device)
    _lib.check(lib.dcs_rconv2d_bwd_data(_lib.ptr(gy), _lib.ptr(panel_bwd), _lib.ptr(gxv), _lib.ptr(ws),
                                        0 if ws is None else ws.numel(), *geo, _lib.cur_stream()), 'dcs_rconv2d_bwd_data')
    return gxv


def upsample_cat_bwd(gxv, H, W, c1, c2, up):
    """[B,H*uf,W*ut,c1+c2] -> (g_x1 [B,H,W,c1], g_x2 [B,H,W,c2] | None): block sum + channel split (real channel counts)."""
    from . import _lib
    B = gxv.shape[0]
    g1 = torch.empty((B, H, W, c1), dtype=torch.float32, device=gxv.device)
    g2 = torch.empty((B, H, W, c2), dtype=torch.float32, device=gxv.device) if c2 else None
    _lib.check(_lib.load().dcs_upsample_cat_bwd(_lib.ptr(gxv), _lib.ptr(g1), _lib.ptr(g2), B, H, W, c1 // 2, c2 // 2,
                                                up[0], up[1], _lib.cur_stream()), 'dcs_upsample_cat_bwd')
    return g1, g2


class _RConvFn(torch.autograd.Function):
    """Real Conv2d / stride-1 ConvTranspose2d over upsample(cat(x1, x2)) on the MFMA kernel, with its gradients (see the
    module docstring).  w: the module's weight ([Cout,Cin,kh,kw], or [Cin,Cout,kh,kw] when transposed)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b, transposed, stride, pad, up):
        wc = w.detach()
        w_corr = wc.flip(2, 3).transpose(0, 1).contiguous() if transposed else wc      # correlation kernel [Cout,Cin,kh,kw]
        kh, kw = w_corr.shape[2:]
        cpad = (kh - 1 - pad[0], kw - 1 - pad[1]) if transposed else tuple(pad)
        cstride = (1, 1) if transposed else tuple(stride)
        y = rconv2d(x1, x2, pack_real_panel(w_corr), None if b is None else b.detach(), w_corr.shape[0], (kh, kw), cstride,
                    cpad, tuple(up))
        ctx.cfg = (bool(transposed), cstride, cpad, tuple(up), b is not None)
        ctx.save_for_backward(x1, x2, w_corr)
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, w_corr = ctx.saved_tensors
        transposed, stride, pad, up, has_bias = ctx.cfg
        gy = gy.contiguous()
        B, H, W, c1 = x1.shape
        c2 = 0 if x2 is None else x2.shape[3]
        cout, cin, kh, kw = w_corr.shape
        g1 = g2 = gw = gb = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            panel_b = pack_real_panel(w_corr.flip(2, 3).transpose(0, 1).contiguous())
            gxv = rconv2d_bwd_data(gy, panel_b, H * up[0], W * up[1], cin, (kh, kw), stride, pad)
            if up != (1, 1) or c2:
                g1, g2 = upsample_cat_bwd(gxv, H, W, c1, c2, up)
            else:
                g1 = gxv
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            cplx = lambda t: None if t is None else t.view(*t.shape[:3], t.shape[3] // 2, 2)
            conj = _conj_vec(gy.device)
            gyc = cplx(gy)
            shape_c = (cout // 2, cin // 2, kh, kw)
            a_r, a_i, ab_r, ab_i = ops.cconv2d_bwd_weight(cplx(x1), cplx(x2), gyc, shape_c, has_bias, (kh, kw), stride, pad, up)
            c_r, c_i, _, _ = ops.cconv2d_bwd_weight(cplx(x1) * conj, None if x2 is None else cplx(x2) * conj, gyc, shape_c,
                                                    False, (kh, kw), stride, pad, up)
            g_corr = torch.empty_like(w_corr)
            g_corr[0::2, 0::2] = 0.5 * (a_r + c_r)          # D_rr
            g_corr[1::2, 1::2] = 0.5 * (a_r - c_r)          # D_ii
            g_corr[1::2, 0::2] = 0.5 * (a_i + c_i)          # D_ir: imaginary output part x real input part
            g_corr[0::2, 1::2] = 0.5 * (c_i - a_i)          # D_ri
            gw = g_corr.flip(2, 3).transpose(0, 1).contiguous() if transposed else g_corr
            if has_bias:                                     # complex bias = (b_r - b_i) + j (b_r + b_i)
                gb = torch.empty(cout, dtype=torch.float32, device=gy.device)
                gb[0::2] = 0.5 * (ab_r - ab_i)
                gb[1::2] = 0.5 * (ab_r + ab_i)
        return g1, g2, gw, gb, None, None, None, None


class _RBnFn(torch.autograd.Function):
    """BatchNorm2d (+ activation) of a real channels-last tensor on the CBN kernels (dcs_rbn_fwd / dcs_rbn_bwd).
    x: [B,H,W,Cr] with even Cr, or [B,F,T] for the one-channel initial BatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, use_batch, act):
        from . import _lib
        lib = _lib.load()
        Cr = x.shape[3] if x.dim() == 4 else 1
        P = x.numel() // Cr
        C, Pc = (1, P // 2) if Cr == 1 else (Cr // 2, P)
        y = torch.empty_like(x)
        stats = torch.empty((C, 8), dtype=torch.float32, device=x.device)
        coef = torch.empty((C, 6), dtype=torch.float32, device=x.device)
        nbytes = lib.dcs_cbn_workspace_bytes(Pc, C)
        if nbytes < 0:
            raise DcsHipError(f'BatchNorm2d: unsupported channel count {Cr}')
        ws = ops._workspace(nbytes, x.device)
        d = lambda t: None if t is None else t.detach()
        _lib.check(lib.dcs_rbn_fwd(_lib.ptr(x), _lib.ptr(y), _lib.ptr(d(weight)), _lib.ptr(d(bias)), _lib.ptr(running_mean),
                                   _lib.ptr(running_var), _lib.ptr(stats), _lib.ptr(coef), _lib.ptr(ws), ws.numel(), P, Cr,
                                   eps, -1.0 if momentum is None else momentum, int(bool(use_batch)), act,
                                   _lib.cur_stream()), 'dcs_rbn_fwd')
        ctx.cfg = (P, Cr, C, Pc, bool(use_batch), act, weight is not None)
        ctx.save_for_backward(x, stats, coef)
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        lib = _lib.load()
        x, stats, coef = ctx.saved_tensors
        P, Cr, C, Pc, use_batch, act, affine = ctx.cfg
        g = g.contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty(Cr, dtype=torch.float32, device=x.device) if affine else None
        gb = torch.empty(Cr, dtype=torch.float32, device=x.device) if affine else None
        ws = ops._workspace(lib.dcs_cbn_bwd_workspace_bytes(Pc, C), x.device)
        _lib.check(lib.dcs_rbn_bwd(_lib.ptr(x), _lib.ptr(g), _lib.ptr(gx), _lib.ptr(stats), _lib.ptr(coef), _lib.ptr(gw),
                                   _lib.ptr(gb), _lib.ptr(ws), ws.numel(), P, Cr, int(use_batch), act, _lib.cur_stream()),
                   'dcs_rbn_bwd')
        return gx, gw, gb, None, None, None, None, None, None


class _RAttendFn(torch.autograd.Function):
    """The real CBAM pair y = sa (.) ca (.) x (r_network.py:8-42, :155-158) with hand-written gradients: the pools, the FC
    and the broadcast products through dcs_rattention_{pool,apply}_{fwd,bwd} (csrc/r_attention.hip), the k x k conv through
    the complex path's entries on the (mean, max) pair read as one complex channel with weights w_mean - j w_max.  The
    reference differentiates the same graph with autograd; torch.max / AdaptiveMaxPool2d send each maximum's gradient to
    its first position, and so do the kernels."""

    @staticmethod
    def forward(ctx, x, w1, w2, w7, k):
        from . import _lib
        lib = _lib.load()
        B, H, W, C = x.shape
        Ch = w1.shape[0]
        x = x.contiguous()
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        nbytes = lib.dcs_rattention_train_workspace_bytes(B, H * W, C, Ch)
        if nbytes < 0:
            raise DcsHipError(f'R_NETWORK attention: unsupported channel count {C}')
        ws = ops._workspace(nbytes, x.device)
        dev = x.device
        ca = torch.empty((B, C), dtype=torch.float32, device=dev)
        mx = torch.empty((B, C), dtype=torch.float32, device=dev)
        hid = torch.empty((B, Ch), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, H, W, 1, 2), dtype=torch.float32, device=dev)
        _lib.check(lib.dcs_rattention_pool_fwd(_lib.ptr(x), _lib.ptr(w1c), _lib.ptr(w2c), _lib.ptr(ca), _lib.ptr(mx), _lib.ptr(hid),
                                               _lib.ptr(pooled), _lib.ptr(ws), ws.numel(), B, H, W, C, Ch, _lib.cur_stream()),
                   'dcs_rattention_pool_fwd')
        wd = w7.detach()
        wp, bias = ops.pack_conv_weight(wd[:, 0:1].contiguous(), (-wd[:, 1:2]).contiguous())
        sa = ops.cconv2d(pooled, None, wp, bias, (k, k), (1, 1), (k // 2, k // 2), (1, 1), F.ACT_SIGMOID)    # [B,H,W,1,2]
        y = torch.empty_like(x)
        _lib.check(lib.dcs_rattention_apply_fwd(_lib.ptr(x), _lib.ptr(ca), _lib.ptr(sa), _lib.ptr(y), B, H, W, C,
                                                _lib.cur_stream()), 'dcs_rattention_apply_fwd')
        ctx.k = k
        ctx.save_for_backward(x, w1c, w2c, wp, ca, mx, hid, pooled, sa)
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import _lib
        lib = _lib.load()
        x, w1, w2, wp, ca, mx, hid, pooled, sa = ctx.saved_tensors
        k = ctx.k
        B, H, W, C = x.shape
        Ch = w1.shape[0]
        dev = x.device
        gy = gy.contiguous()
        ws = ops._workspace(lib.dcs_rattention_train_workspace_bytes(B, H * W, C, Ch), dev)
        gx = torch.empty_like(x)
        g_sa = torch.empty_like(sa)
        g_ca = torch.empty((B, C), dtype=torch.float32, device=dev)
        _lib.check(lib.dcs_rattention_apply_bwd(_lib.ptr(gy), _lib.ptr(x), _lib.ptr(ca), _lib.ptr(sa), _lib.ptr(gx), _lib.ptr(g_sa),
                                                _lib.ptr(g_ca), _lib.ptr(ws), ws.numel(), B, H, W, C, _lib.cur_stream()),
                   'dcs_rattention_apply_bwd')
        g_u = g_sa * sa * (1.0 - sa)                       # sigmoid on the parts; the imaginary cotangent is zero
        ksz, one, pad = (k, k), (1, 1), (k // 2, k // 2)
        g_pooled, _ = ops.cconv2d_bwd_data(g_u, ops.pack_conv_weight_bwd(wp, ksz, one, pad, one), (H, W, 1), ksz, one, pad, one, 1)
        gw_r, gw_i, _, _ = ops.cconv2d_bwd_weight(pooled, None, g_u, (1, 1, k, k), False, ksz, one, pad, one)
        g_w7 = torch.cat([gw_r, -gw_i], dim=1)             # conv1.weight [1, 2, k, k] = (w_mean, w_max) = (w_r, -w_i)
        gw1, gw2 = torch.empty_like(w1), torch.empty_like(w2)
        _lib.check(lib.dcs_rattention_pool_bwd(_lib.ptr(g_pooled), _lib.ptr(g_ca), _lib.ptr(x), _lib.ptr(ca), _lib.ptr(mx),
                                               _lib.ptr(hid), _lib.ptr(w1), _lib.ptr(w2), _lib.ptr(gx), _lib.ptr(gw1), _lib.ptr(gw2),
                                               1, _lib.ptr(ws), ws.numel(), B, H, W, C, Ch, _lib.cur_stream()),
                   'dcs_rattention_pool_bwd')
        return gx, gw1.view_as(w1), gw2.view_as(w2), g_w7, None


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

    # ---- trainer hooks (r_network.py:176-363): the same bodies as C_NETWORK's with dtype "real" ---------------------
    _step_dtype = 'real'

    def _hooks():
        from .c_network import C_NETWORK
        return {k: getattr(C_NETWORK, k) for k in ('configure_optimizers', 'training_step', '_eval_step', 'validation_step',
                                                   'test_step', '_epoch_end', 'validation_epoch_end', 'test_epoch_end',
                                                   'on_after_backward')}
    locals().update(_hooks())
    del _hooks

    def weights_init(self):                            # r_network.py:126-137
        init = self.hparams['initialisation_distribution']
        for m in self.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)):
                init(m.weight)

    # ---- pieces ---------------------------------------------------------------------------------------------------
    def _bn_act(self, bn, x, act):
        """BatchNorm2d (+ activation) of a channels-last real tensor ([B,H,W,Cr] with even Cr, or [B,F,T]: one channel) on
        the CBN kernels (_RBnFn): statistics, running-statistic update, normalisation and activation; differentiable."""
        use_batch = self.training or not bn.track_running_stats
        momentum = None
        if self.training and bn.track_running_stats:
            bn.num_batches_tracked += 1
            momentum = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            F.note_state_update()
        return _RBnFn.apply(x.contiguous(), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, momentum,
                            use_batch, act)

    def _conv(self, conv, x1, x2=None, up=(1, 1), transposed=False):
        w = conv.weight
        if torch.is_grad_enabled() and (w.requires_grad or x1.requires_grad):
            return _RConvFn.apply(x1, x2, w, conv.bias, transposed, tuple(conv.stride), tuple(conv.padding), tuple(up))
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
        xc = torch.stack([x, torch.zeros_like(x)], dim=-1).unsqueeze(3)                    # [B,F,T,1,2]
        if torch.is_grad_enabled() and (w.requires_grad or x.requires_grad):
            be, bo = b[0::2], b[1::2]                          # weight views: autograd maps the gradients back
            y = F.cconv2d(xc, None, w[0::2].contiguous(), w[1::2].contiguous(), (be + bo) / 2, (bo - be) / 2, False,
                          tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))
            return y.flatten(3)
        def make():
            be, bo = b[0::2], b[1::2]
            return ops.pack_conv_weight(w[0::2].contiguous(), w[1::2].contiguous(), ((be + bo) / 2).contiguous(),
                                        ((bo - be) / 2).contiguous())
        wp, bias = _packed(conv, 'enc0', (w, b), make)
        y = ops.cconv2d(xc, None, wp, bias, tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding))
        return y.flatten(3)                                                                 # [B,F,T,16]

    def _dec_last(self, convt, d, skip, up):
        """(C1 + C2) -> 1 real channels: real part of a complex conv with weights (w[2k] - j w[2k+1]) over cat(d, skip)."""
        w, b = convt.weight, convt.bias                                                     # [Cin_r, 1, kh, kw]
        if torch.is_grad_enabled() and (w.requires_grad or d.requires_grad):
            B, H, W, c1 = d.shape
            kh, kw = convt.kernel_size
            pad = (kh - 1 - convt.padding[0], kw - 1 - convt.padding[1])
            y = F.cconv_single_output(d.view(B, H, W, c1 // 2, 2), skip.view(B, H, W, skip.shape[3] // 2, 2),
                                      w[0::2].contiguous(), (-w[1::2]).contiguous(), b, torch.zeros_like(b), (kh, kw), pad, up)
            return y[..., 0, 0]
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
        if torch.is_grad_enabled() and (x.requires_grad or lstm.weight_ih_l0.requires_grad):
            for layer in range(lstm.num_layers):             # projections: rocBLAS; recurrence + BPTT: lstm.hip
                names = [f'_l{layer}', f'_l{layer}_reverse']
                w_ih = torch.cat([getattr(lstm, 'weight_ih' + n) for n in names])
                bias = torch.cat([getattr(lstm, 'bias_ih' + n) + getattr(lstm, 'bias_hh' + n) for n in names])
                w_hh = torch.stack([getattr(lstm, 'weight_hh' + n) for n in names]).unsqueeze(0)
                gx = torch.addmm(bias, inp, w_ih.t())
                out = F._LstmRecFn.apply(gx.view(1, B, S, 2, 4 * Hh), w_hh.contiguous())
                inp = out.reshape(B * S, 2 * Hh)
            return inp.view(B, S, 2 * Hh)
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
    def _attend_autograd(ca_m, sa_m, x):
        """The same block under autograd: pools, FC and broadcast multiplies as differentiable torch ops, the 7x7 conv as
        the complex path's conv node over the (mean, max) pair read as ONE complex channel with weights w_mean - j w_max
        (its real part is the answer; the complex sigmoid epilogue acts on the parts separately)."""
        mx = x.amax(dim=(1, 2))                                                          # [B, C]
        w1, w2, w = ca_m.fc[0].weight.flatten(1), ca_m.fc[2].weight.flatten(1), sa_m.conv1.weight
        ca = torch.sigmoid(torch.relu(mx @ w1.t()) @ w2.t())
        z = x * ca[:, None, None, :]
        pooled = torch.stack([z.mean(dim=-1), z.amax(dim=-1)], dim=-1).unsqueeze(3).contiguous()   # [B,H,W,1,2]
        k = sa_m.kernel_size
        sa = F.cconv2d(pooled, None, w[:, 0:1].contiguous(), (-w[:, 1:2]).contiguous(), None, None, False, (k, k), (1, 1),
                       (k // 2, k // 2), (1, 1), F.ACT_SIGMOID)[..., 0]                  # [B,H,W,1]
        return z * sa

    @staticmethod
    def _attend(ca_m, sa_m, x):
        """sa (.) ca (.) x (r_network.py:155-158 / :166-167): one C-ABI call at inference (dcs_rattention_fwd), one autograd
        node over the training entries otherwise (_RAttendFn)."""
        if torch.is_grad_enabled() and (x.requires_grad or sa_m.conv1.weight.requires_grad):
            if os.environ.get('DCS_RATTENTION_ATEN', '0') == '1':          # (the ATen formulation: a test reference)
                return R_NETWORK._attend_autograd(ca_m, sa_m, x)
            return _RAttendFn.apply(x, ca_m.fc[0].weight, ca_m.fc[2].weight, sa_m.conv1.weight, sa_m.kernel_size)
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
        B, Fb, T = x.shape
        if (B * Fb * T) % 4:
            raise DcsHipError(f'R_NETWORK.forward: B*F*T must be a multiple of 4, got {tuple(x.shape)}')
        e = self._bn_act(self.initial_batchnorm, x, F.ACT_NONE)    # one channel: the P values as P/2 (re, im) pairs
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
