"""Tensor-level wrappers over the C ABI (include/dcsnet_hip.h).

Activations are float32 CUDA tensors of shape [B, H, W, C, 2] — channels-last interleaved
complex, the memory of a torch complex64 [B, C, H, W] tensor in channels_last format.  Every
function validates device / dtype / contiguity and raises on a non-zero return code; there is
no eager or CPU fallback.
"""
import os as _os
import torch

from . import _lib
from ._lib import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, check, ptr, cur_stream  # noqa: F401


def _chk(t, name, dims=None, act=False):
    """act=True: an ACTIVATION tensor — float32, or bfloat16 where the activations are stored in bf16 (the _h entry points)."""
    if t is None:
        return
    if not t.is_cuda:
        raise _lib.DcsHipError(f'{name}: expected a CUDA (HIP) tensor; the HIP path has no CPU fallback')
    if t.dtype != torch.float32 and not (act and t.dtype == torch.bfloat16):
        raise _lib.DcsHipError(f'{name}: expected float32{" or bfloat16" if act else ""}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.DcsHipError(f'{name}: expected a contiguous tensor')
    if dims is not None and t.dim() != dims:
        raise _lib.DcsHipError(f'{name}: expected {dims} dims, got shape {tuple(t.shape)}')


def _sym(name, *acts):
    """The entry point for these activation tensors: `name` for fp32, `name`_h for bf16 storage; mixing them is an error."""
    dts = {t.dtype for t in acts if t is not None}
    if len(dts) > 1:
        raise _lib.DcsHipError(f'{name}: activation tensors of different dtypes {sorted(str(d) for d in dts)}')
    bf = dts == {torch.bfloat16}
    if bf and conv_precision() != 'bf16' and 'conv' in name:
        raise _lib.DcsHipError(f'{name}: bf16 activations need set_conv_precision("bf16") (bf16 weight panels)')
    return getattr(_lib.load(), name + ('_h' if bf else ''))


CONV_TIMER = None        # set by bench.py: object with begin(flops, tag=None, executed=1.0) -> token / end(token)

# Device-side dropout seed offset (int64 tensor with one element, or None).  Every dropout-capable
# kernel adds it to its by-value seed, so a captured hipGraph draws a fresh mask per replay once the
# owner (dp.TrainStep) increments it in-graph.  Forward and backward of a step see the same value.
SEED_STATE = None

_workspaces = {}
_retired = []


def set_conv_precision(mode):
    """'bf16x6' (default: fp32 emulated on the bf16 MFMA — exact three-way bf16 splits of both operands, the six leading
    cross products accumulated in fp32; product error < 2^-24, measured error against fp64 below the native
    instruction's), 'f32' (native fp32 MFMA) or 'bf16' (bf16-rounded operands, fp32 accumulate: BASELINE configs[4]) in
    the MFMA conv forward / data gradient (dcs_set_conv_precision).  Every packed weight made under the other mode becomes invalid: the caches
    are cleared here, a recorded pack plan must be re-recorded by its owner."""
    code = {'f32': 0, 'fp32': 0, 'bf16': 1, 'bf16x6': 2}[mode]
    check(_lib.load().dcs_set_conv_precision(code), 'dcs_set_conv_precision')
    global BF16_OPERANDS_ON_PURPOSE
    if code != 1:
        BF16_OPERANDS_ON_PURPOSE = False                       # a switch away from 'bf16' ends the choice
    elif not _IN_SET_ACTIVATION_DTYPE:
        BF16_OPERANDS_ON_PURPOSE = True                        # chosen by the caller; a later set_activation_dtype('bf16') keeps it
    from . import functional
    functional._pack_cache.clear()
    pack_plan_drop()


# 'bf16' chosen on purpose: through set_conv_precision directly (not as a side effect of set_activation_dtype), or as the
# process preset DCS_CONV_PRECISION=1 (csrc/api.hip) — a preset process runs fp32-storage networks on bf16
# operands because it was asked to, not because another network switched the mode
BF16_OPERANDS_ON_PURPOSE = _os.environ.get('DCS_CONV_PRECISION', '').strip() == '1'
_IN_SET_ACTIVATION_DTYPE = False


def conv_precision():
    return ('f32', 'bf16', 'bf16x6')[_lib.load().dcs_get_conv_precision()]


class PackPlan:
    """Python side of a dcs_pack_plan: owns the destination tensors the plan re-packs in place and serves
    them to packed-weight lookups while `valid` (= dcs_pack_plan_run has run since the last parameter
    update).  fwd: key -> [parameter weakrefs, (wp, bias), parameter versions]; bwd: key -> data-gradient weight."""

    def __init__(self):
        self.handle, self.recording, self.valid = None, True, False
        self.fwd, self.bwd, self.keep = {}, {}, []

    def stats(self):
        import ctypes
        nj, nl = ctypes.c_int(0), ctypes.c_int(0)
        check(_lib.load().dcs_pack_plan_jobs(self.handle, ctypes.byref(nj), ctypes.byref(nl)), 'dcs_pack_plan_jobs')
        return nj.value, nl.value

    def __del__(self):
        try:
            if self.handle:
                _lib.load().dcs_pack_plan_destroy(self.handle)
        except Exception:      # noqa: BLE001 - interpreter shutdown
            pass


PLAN = None                # the active PackPlan (dp.TrainStep owns it), or None


def pack_plan_begin():
    global PLAN
    check(_lib.load().dcs_pack_plan_begin(), 'dcs_pack_plan_begin')
    PLAN = PackPlan()
    return PLAN


def pack_plan_end():
    import ctypes
    h = ctypes.c_void_p()
    plan = PLAN
    check(_lib.load().dcs_pack_plan_end(ctypes.byref(h)), 'dcs_pack_plan_end')
    plan.handle, plan.recording = h.value, False
    _plan_snapshot(plan)
    plan.valid = True          # everything recorded was also executed
    return plan


def _plan_snapshot(plan):
    for e in plan.fwd.values():
        e[2] = tuple(None if r is None or r() is None else r()._version for r in e[0])


def pack_plan_run(plan):
    global PLAN
    PLAN = plan                # lookups consult the plan that ran last
    check(_lib.load().dcs_pack_plan_run(plan.handle, cur_stream()), 'dcs_pack_plan_run')
    _plan_snapshot(plan)
    plan.valid = True


def pack_plan_invalidate():
    if PLAN is not None:
        PLAN.valid = False


def pack_plan_drop():
    global PLAN
    PLAN = None



def _workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream); reused across calls on the same stream."""
    key = (device.index, cur_stream())
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired.append(buf)       # a captured hipGraph may hold its address: never hand it back to the allocator
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def pack_conv_weight(w_r, w_i, b_r=None, b_i=None, transposed=False, up=(1, 1)):
    """(conv_r.weight, conv_i.weight[, biases]) -> (wp [k*k,Cin,Cout,2], bias [Cout,2]).
    `up`: the upsample factors of the cconv2d call the weight is for (adds the folded panels)."""
    for n, t in (('w_r', w_r), ('w_i', w_i), ('b_r', b_r), ('b_i', b_i)):
        _chk(t, n)
    if transposed:
        Cin, Cout, kh, kw = w_r.shape
    else:
        Cout, Cin, kh, kw = w_r.shape
    lib = _lib.load()
    # one buffer, several panels: [taps,Cin,Cout] complex for the direct kernels, then (when the shape is
    # MFMA-eligible) the fragment-ordered real-embedded panel and the upsample-folded sub-kernels; `wp`
    # views the first, the library finds the others behind it
    buf = torch.empty(lib.dcs_packed_weight_floats(Cout, Cin, kh, kw, up[0], up[1]), dtype=torch.float32,
                      device=w_r.device)
    wp = buf[:kh * kw * Cin * Cout * 2].view(kh * kw, Cin, Cout, 2)
    bias = torch.empty((Cout, 2), dtype=torch.float32, device=w_r.device)
    check(lib.dcs_pack_conv_weight(ptr(w_r), ptr(w_i), ptr(b_r), ptr(b_i), ptr(wp), ptr(bias),
                                   Cout, Cin, kh, kw, int(bool(transposed)), up[0], up[1], cur_stream()),
          'dcs_pack_conv_weight')
    return wp, bias


def pack_tap_rows(w_r, w_i, ct):
    """conv_tran_r / conv_tran_i.weight [Cin,1,kh,kw] -> the 1x1 packed weight (wp [1,Cin,ct,2], zero bias [ct,2]) of the
    tap-sum factorisation (dcs_pack_tap_rows)."""
    _chk(w_r, 'w_r', 4)
    _chk(w_i, 'w_i', 4)
    Cin, one, kh, kw = w_r.shape
    if one != 1 or ct < kh * kw:
        raise _lib.DcsHipError(f'pack_tap_rows: expected [Cin,1,kh,kw] and ct >= kh*kw, got {tuple(w_r.shape)}, ct={ct}')
    lib = _lib.load()
    buf = torch.empty(lib.dcs_packed_weight_floats(ct, Cin, 1, 1, 1, 1), dtype=torch.float32, device=w_r.device)
    wp = buf[:Cin * ct * 2].view(1, Cin, ct, 2)
    bias = torch.empty((ct, 2), dtype=torch.float32, device=w_r.device)
    check(lib.dcs_pack_tap_rows(ptr(w_r), ptr(w_i), ptr(wp), ptr(bias), Cin, kh, kw, ct, cur_stream()), 'dcs_pack_tap_rows')
    return wp, bias


def tap_rows_scatter(gt_r, gt_i, w_shape, outs=None):
    """Weight gradient of the 1x1 tap conv [ct,Cin,1,1] -> gradient of conv_tran_r/_i.weight [Cin,1,kh,kw]; `outs`:
    optional (g_r, g_i) destinations that are ADDED to (gradient sinks)."""
    Cin, _, kh, kw = w_shape
    acc = outs is not None and outs[0] is not None and outs[1] is not None
    g_r = outs[0] if acc else torch.empty(w_shape, dtype=torch.float32, device=gt_r.device)
    g_i = outs[1] if acc else torch.empty(w_shape, dtype=torch.float32, device=gt_r.device)
    check(_lib.load().dcs_tap_rows_wgrad_scatter(ptr(gt_r), ptr(gt_i), ptr(g_r), ptr(g_i), Cin, kh, kw, int(acc),
                                                 cur_stream()), 'dcs_tap_rows_wgrad_scatter')
    return g_r, g_i


def cconv2d(x1, x2, wp, bias, ksize, stride=(1, 1), pad=(0, 0), up=(1, 1), act=ACT_NONE, coef=None):
    """Complex correlation over the virtual input upsample(cat(x1, x2)); see dcs_cconv2d_fwd.  coef [Cout, 6]: an eval-mode
    CBN's coefficients folded in between bias and activation (dcs_cconv2d_fwd_affine)."""
    _chk(x1, 'x1', 5, act=True)
    _chk(x2, 'x2', 5, act=True)
    _chk(wp, 'wp', 4)
    _chk(bias, 'bias', 2)
    B, Hin, Win, C1, _ = x1.shape
    C2 = 0
    if x2 is not None:
        if x2.shape[:3] != x1.shape[:3]:
            raise _lib.DcsHipError(f'cconv2d: x1 {tuple(x1.shape)} and x2 {tuple(x2.shape)} differ in B/H/W')
        C2 = x2.shape[3]
    kh, kw = ksize
    if wp.shape[0] != kh * kw or wp.shape[1] != C1 + C2:
        raise _lib.DcsHipError(f'cconv2d: packed weight {tuple(wp.shape)} does not match k={ksize}, Cin={C1 + C2}')
    Cout = wp.shape[2]
    Hout = (Hin * up[0] + 2 * pad[0] - kh) // stride[0] + 1
    Wout = (Win * up[1] + 2 * pad[1] - kw) // stride[1] + 1
    y = torch.empty((B, Hout, Wout, Cout, 2), dtype=x1.dtype, device=x1.device)
    lib = _lib.load()
    # split-K scratch for layers with too few output tiles to fill the chip (0 bytes for most geometries)
    nbytes = lib.dcs_cconv2d_fwd_workspace_bytes(B, Hin, Win, C1, C2, up[0], up[1], Cout, kh, kw, stride[0], stride[1],
                                                 pad[0], pad[1])
    ws = _workspace(nbytes, x1.device) if nbytes > 0 else None
    nbytes = max(nbytes, 0)
    # bench.py's live roofline probe: 8 real flops per complex MAC
    # (strided forward convs are exactly the encoder's ComplexConv2d stack: tagged for bench.py's encoder roofline)
    ev = (CONV_TIMER.begin(8.0 * B * Hout * Wout * Cout * (C1 + C2) * kh * kw, 'enc_fwd' if max(stride) > 1 else None,
                           executed=_fold_fraction(C1, C1 + C2, Cout, ksize, stride, pad, up),
                           emulated=_emulated(C1 + C2, Cout, kh * kw if tuple(up) == (1, 1) else 0) and (C1 % 2 == 0 or C1 + C2 == 1),
                           nbytes=_conv_bytes(x1, B * Hin * Win * (C1 + C2), B * Hout * Wout * Cout, kh * kw * (C1 + C2) * Cout))
          if CONV_TIMER is not None else None)
    if coef is not None:
        _chk(coef, 'coef', 2)
        if tuple(coef.shape) != (Cout, 6):
            raise _lib.DcsHipError(f'cconv2d: coef {tuple(coef.shape)} does not match Cout={Cout}')
    check(_sym('dcs_cconv2d_fwd_affine', x1, x2)(ptr(x1), ptr(x2), ptr(wp), ptr(bias), ptr(coef), ptr(y), ptr(ws), nbytes, B, Hin, Win,
                                     C1, C2, up[0], up[1], Cout, kh, kw, stride[0], stride[1], pad[0], pad[1], act,
                                     cur_stream()), 'dcs_cconv2d_fwd_affine')
    if ev is not None:
        CONV_TIMER.end(ev)
    return y


def cconv2d_stats(x1, x2, wp, bias, ksize, stride=(1, 1), pad=(0, 0), up=(1, 1)):
    """cconv2d (no activation) that also leaves the training-mode CBN statistics of its raw output in its epilogue
    (dcs_cconv2d_fwd_stats).  Returns (y, stat): stat = (part float[Cout, 5, rows capacity], rows, pivot = bias) for cbn(stat=...),
    or None when this geometry has no statistics epilogue (the plain conv ran: the CBN makes its own statistics pass)."""
    import ctypes
    _chk(x1, 'x1', 5, act=True)
    _chk(x2, 'x2', 5, act=True)
    _chk(wp, 'wp', 4)
    _chk(bias, 'bias', 2)
    B, Hin, Win, C1, _ = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    kh, kw = ksize
    Cout = wp.shape[2]
    lib = _lib.load()
    geo = (B, Hin, Win, C1, C2, up[0], up[1], Cout, kh, kw, stride[0], stride[1], pad[0], pad[1])
    rows = lib.dcs_cconv2d_fwd_stats_rows(*geo) if STATS_EPILOGUE else 0
    if rows < 1:
        return cconv2d(x1, x2, wp, bias, ksize, stride, pad, up, ACT_NONE), None
    if x2 is not None and x2.shape[:3] != x1.shape[:3]:
        raise _lib.DcsHipError(f'cconv2d: x1 {tuple(x1.shape)} and x2 {tuple(x2.shape)} differ in B/H/W')
    if wp.shape[0] != kh * kw or wp.shape[1] != C1 + C2:
        raise _lib.DcsHipError(f'cconv2d: packed weight {tuple(wp.shape)} does not match k={ksize}, Cin={C1 + C2}')
    Hout = (Hin * up[0] + 2 * pad[0] - kh) // stride[0] + 1
    Wout = (Win * up[1] + 2 * pad[1] - kw) // stride[1] + 1
    y = torch.empty((B, Hout, Wout, Cout, 2), dtype=x1.dtype, device=x1.device)
    part = torch.empty((Cout, 5, rows), dtype=torch.float32, device=x1.device)
    nbytes = max(lib.dcs_cconv2d_fwd_workspace_bytes(*geo), 0)
    ws = _workspace(nbytes, x1.device) if nbytes > 0 else None
    ev = (CONV_TIMER.begin(8.0 * B * Hout * Wout * Cout * (C1 + C2) * kh * kw, 'enc_fwd' if max(stride) > 1 else None,
                           executed=_fold_fraction(C1, C1 + C2, Cout, ksize, stride, pad, up),
                           emulated=_emulated(C1 + C2, Cout, kh * kw if tuple(up) == (1, 1) else 0) and (C1 % 2 == 0 or C1 + C2 == 1),
                           nbytes=_conv_bytes(x1, B * Hin * Win * (C1 + C2), B * Hout * Wout * Cout, kh * kw * (C1 + C2) * Cout))
          if CONV_TIMER is not None else None)
    used = ctypes.c_int(0)
    check(_sym('dcs_cconv2d_fwd_stats', x1, x2)(ptr(x1), ptr(x2), ptr(wp), ptr(bias), ptr(y), ptr(part), rows, ctypes.byref(used), ptr(ws),
                                    nbytes, *geo, cur_stream()), 'dcs_cconv2d_fwd_stats')
    if ev is not None:
        CONV_TIMER.end(ev)
    return y, (part, used.value, bias)


import os as _os
FUSE_APPLY_POOL = _os.environ.get('DCS_FUSE_APPLY_POOL', '1') != '0'       # 0 (A/B runs): CBN apply and channel pool as separate launches
STATS_EPILOGUE = _os.environ.get('DCS_STATS_EPILOGUE', '1') != '0'       # 0 (A/B runs): every training-mode CBN makes its own statistics pass


def _conv_bytes(act, n_in, n_out, n_w):
    """ALGORITHMIC HBM bytes of one conv launch (every activation read once and written once, the weights
    once): n_in complex values read + n_out written at the activations' storage size (8 B fp32, 4 B bf16), n_w complex
    weights at the packed panel's size (fp32 8 B; bf16 panels 4 B)."""
    e = 4.0 if act.dtype == torch.bfloat16 else 8.0
    w = 4.0 if conv_precision() == 'bf16' else 8.0
    return e * (n_in + n_out) + w * n_w


def _emulated(k_channels, n_channels, taps=0):
    """Whether the MFMA GEMM with K = 2 * k_channels, N = 2 * n_channels runs on the bf16 MFMA in fp32-emulation mode
    (conv::mfma_precision, conv_mfma.hip: 16-channel chunks, 32-column tiles) — for bench.py's instruction accounting."""
    if conv_precision() == 'bf16x6' and k_channels == 1 and n_channels == 8 and taps == 49:
        return True                       # the first encoder conv (conv_enc0.hip: cconv_enc0b_kernel, 16x16x32 bf16 MFMAs)
    return (conv_precision() == 'bf16x6' and n_channels % 8 == 0 and
            (k_channels % 16 == 0 or (k_channels == 8 and taps in (49, 16) and n_channels != 8)))


def _wgrad_emulated(C1, Cin, Cout, ksize):
    """Whether the weight gradient of this geometry runs on the emulated kernel (conv_wgrad_mfma.hip, launch<V>: the MFMA
    variants 1x1, 3x3 incl. its folded forms, 5x5 and the 7x7 / 16-channel layer — under precision mode
    'bf16x6' unless DCS_WGRAD_X6=0) — for bench.py's instruction accounting."""
    import os
    if conv_precision() != 'bf16x6' or os.environ.get('DCS_WGRAD_X6', '1') == '0':
        return False
    if Cin % 8 or Cout % 8 or C1 % 2 or ksize[0] != ksize[1]:
        return False
    return ksize[0] in (1, 3, 5) or (ksize[0] == 7 and 16 <= Cout < 32)


def _fold_fraction(C1, Cin, Cout, ksize, stride, pad, up):
    """Share of a conv's ALGORITHMIC multiply-accumulates (what the reference computes: k*k taps per output pixel) that
    the library actually issues.  A 3x3 / stride-1 / pad-1 conv over a nearest-upsampled input runs in folded form
    (csrc/conv_pack.hip: per output-parity class a 2-tap kernel along every upsampled axis, taps pre-summed): 6/9 of
    the taps for a (2,1) upsample, 4/9 for (2,2) — forward, data gradient and weight gradient alike.  Everything else
    (strided encoder convs, their class-decomposed data gradients, 1x1) issues exactly the algorithmic count."""
    if (tuple(ksize) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and tuple(up) in ((2, 1), (1, 2), (2, 2))
            and Cin % 8 == 0 and Cout % 8 == 0 and C1 % 2 == 0):
        return (2 if up[0] == 2 else 3) * (2 if up[1] == 2 else 3) / 9.0
    return 1.0


def pack_conv_weight_bwd(wp, ksize, stride=(1, 1), pad=(0, 0), up=(1, 1)):
    """Forward packed weight [taps,Cin,Cout,2] -> data-gradient weight [taps,Cout,Cin,2] (+ derived panels
    for the forward geometry it will be used with)."""
    _chk(wp, 'wp', 4)
    taps, Cin, Cout, _ = wp.shape
    kh, kw = ksize
    plan = PLAN
    key = (wp.data_ptr(), tuple(ksize), tuple(stride), tuple(pad), tuple(up))
    if plan is not None and plan.valid and not plan.recording:
        hit = plan.bwd.get(key)            # plan-owned wp: its address cannot be reused by another tensor
        if hit is not None:
            return hit
    lib = _lib.load()
    geo = (Cout, Cin, kh, kw, stride[0], stride[1], pad[0], pad[1], up[0], up[1])
    buf = torch.empty(lib.dcs_packed_weight_bwd_floats(*geo), dtype=torch.float32, device=wp.device)
    wpb = buf[:taps * Cin * Cout * 2].view(taps, Cout, Cin, 2)
    check(lib.dcs_pack_conv_weight_bwd(ptr(wp), ptr(wpb), *geo, cur_stream()), 'dcs_pack_conv_weight_bwd')
    if plan is not None and plan.recording:
        plan.bwd[key] = wpb
        plan.keep.append((wp, buf))
    return wpb


def cconv2d_bwd_data(gy, wp_bwd, in_shape, ksize, stride, pad, up=(1, 1), C1=None):
    """gy [B,Hout,Wout,Cout,2] -> (g_x1, g_x2) for the forward call with x1 [B,Hin,Win,C1,2] (+ x2).
    in_shape = (Hin, Win, Cin_total); wp_bwd from pack_conv_weight_bwd with the same geometry."""
    _chk(gy, 'gy', 5, act=True)
    _chk(wp_bwd, 'wp_bwd', 4)
    B, Hout, Wout, Cout, _ = gy.shape
    Hin, Win, Cin = in_shape
    C1 = Cin if C1 is None else C1
    C2 = Cin - C1
    gx1 = torch.empty((B, Hin, Win, C1, 2), dtype=gy.dtype, device=gy.device)
    gx2 = torch.empty((B, Hin, Win, C2, 2), dtype=gy.dtype, device=gy.device) if C2 else None
    lib = _lib.load()
    geo = (B, Hin, Win, C1, C2, up[0], up[1], Cout, ksize[0], ksize[1], stride[0], stride[1], pad[0], pad[1])
    nbytes = lib.dcs_cconv2d_bwd_data_workspace_bytes(*geo)
    if nbytes < 0:
        raise _lib.DcsHipError(f'cconv2d_bwd_data: unsupported geometry {geo}')
    ws = _workspace(nbytes, gy.device) if nbytes else None
    ev = (CONV_TIMER.begin(8.0 * B * Hout * Wout * Cout * Cin * ksize[0] * ksize[1],
                           executed=_fold_fraction(C1, Cin, Cout, ksize, stride, pad, up), emulated=_emulated(Cout, Cin),
                           nbytes=_conv_bytes(gy, B * Hout * Wout * Cout, B * Hin * Win * Cin, ksize[0] * ksize[1] * Cin * Cout))
          if CONV_TIMER is not None else None)
    check(_sym('dcs_cconv2d_bwd_data', gy)(ptr(gy), ptr(wp_bwd), ptr(gx1), ptr(gx2), ptr(ws), ws.numel() if ws is not None else 0,
                                   *geo, cur_stream()), 'dcs_cconv2d_bwd_data')
    if ev is not None:
        CONV_TIMER.end(ev)
    return gx1, gx2


def cconv2d_bwd_weight(x1, x2, gy, w_shape, has_bias, ksize, stride, pad, up=(1, 1), transposed=False, outs=None, immediate=False):
    """Gradients in the reference's parameter layout: (gw_r, gw_i, gb_r, gb_i).  `outs`: optional
    pre-existing destinations (e.g. views of a flat gradient bucket) for any of the four."""
    _chk(x1, 'x1', 5, act=True)
    _chk(x2, 'x2', 5, act=True)
    _chk(gy, 'gy', 5, act=True)
    B, Hin, Win, C1, _ = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    Cout = gy.shape[3]
    dev = gy.device
    o = outs or (None, None, None, None)
    gw_r = o[0] if o[0] is not None else torch.empty(w_shape, dtype=torch.float32, device=dev)
    gw_i = o[1] if o[1] is not None else torch.empty(w_shape, dtype=torch.float32, device=dev)
    gb_r = (o[2] if o[2] is not None else torch.empty(Cout, dtype=torch.float32, device=dev)) if has_bias else None
    gb_i = (o[3] if o[3] is not None else torch.empty(Cout, dtype=torch.float32, device=dev)) if has_bias else None
    for n, t in (('gw_r', gw_r), ('gw_i', gw_i), ('gb_r', gb_r), ('gb_i', gb_i)):
        _chk(t, n)
    lib = _lib.load()
    geo = (B, Hin, Win, C1, C2, up[0], up[1], Cout, ksize[0], ksize[1], stride[0], stride[1], pad[0], pad[1])
    nbytes = lib.dcs_cconv2d_bwd_weight_workspace_bytes(*geo)
    if nbytes < 0:
        raise _lib.DcsHipError(f'cconv2d_bwd_weight: unsupported geometry {geo}')
    # Inside a deferred-reduce scope (wgrad_defer_begin) the slab reduce of this call is postponed to the flush when ALL
    # its results land in caller-owned destinations (`outs`): the slabs then need their own buffer, kept until the flush.
    # A call whose results are consumed right away suspends deferral for its duration.
    scope = WGRAD_DEFER
    all_out = outs is not None and o[0] is not None and o[1] is not None and (not has_bias or (o[2] is not None and o[3] is not None))
    if scope is not None and all_out:
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
        scope.append((ws, x1, x2, gy))                # a deferred KERNEL (the small-channel one) still reads its inputs at the flush
    else:
        ws = _workspace(nbytes, dev)
    # immediate (the side-stream calls of a train step: functional._CConv2dFn.backward): the slab reduce follows its kernel on
    # the same stream instead of waiting for the flush — the side stream has slack, the end of the step has none; the slabs
    # still live until the flush (they are read on a stream that is joined only there)
    suspend = scope is not None and (not all_out or immediate)
    if suspend:
        lib.dcs_wgrad_defer_suspend(1)
    ev = None
    if CONV_TIMER is not None:
        ev = CONV_TIMER.begin(8.0 * B * gy.shape[1] * gy.shape[2] * Cout * (C1 + C2) * ksize[0] * ksize[1],
                              executed=_fold_fraction(C1, C1 + C2, Cout, ksize, stride, pad, up),
                              emulated=_wgrad_emulated(C1, C1 + C2, Cout, ksize),
                              nbytes=(_conv_bytes(x1, B * Hin * Win * (C1 + C2) + B * gy.shape[1] * gy.shape[2] * Cout, 0, 0) +
                                      8.0 * ksize[0] * ksize[1] * (C1 + C2) * Cout))
    try:
        check(_sym('dcs_cconv2d_bwd_weight', x1, x2, gy)(ptr(x1), ptr(x2), ptr(gy), ptr(gw_r), ptr(gw_i), ptr(gb_r), ptr(gb_i), ptr(ws),
                                         ws.numel(), *geo, int(bool(transposed)), cur_stream()), 'dcs_cconv2d_bwd_weight')
    finally:
        if suspend:
            lib.dcs_wgrad_defer_suspend(0)
    if ev is not None:
        CONV_TIMER.end(ev)
    return gw_r, gw_i, gb_r, gb_i


WGRAD_DEFER = None         # keep-alive list of slab workspaces while a deferred-reduce scope is open, else None


def wgrad_defer_begin():
    """Open a scope in which weight-gradient slab reductions are recorded instead of launched (dcs_wgrad_defer_begin)."""
    global WGRAD_DEFER
    check(_lib.load().dcs_wgrad_defer_begin(), 'dcs_wgrad_defer_begin')
    WGRAD_DEFER = []


def wgrad_defer_flush(partial=False):
    """Run every recorded reduction as one batched launch on the current stream and release the slabs.  partial: the scope
    stays open (a new recording starts) and slabs / operands stay alive until the final flush — the caller runs this on the
    weight-gradient side stream in the middle of backward, which is joined only in front of the final flush."""
    global WGRAD_DEFER
    ev = CONV_TIMER.begin(0.0) if CONV_TIMER is not None else None     # the reduces belong to the conv family's time
    try:
        check(_lib.load().dcs_wgrad_defer_flush(cur_stream()), 'dcs_wgrad_defer_flush')
    finally:
        if not partial:
            WGRAD_DEFER = None
    if partial:
        check(_lib.load().dcs_wgrad_defer_begin(), 'dcs_wgrad_defer_begin')
    if ev is not None:
        CONV_TIMER.end(ev)



def cbn(x, weight, bias, running_mean, running_covar, eps=1e-5, momentum=0.1, use_batch_stats=True,
        act=ACT_NONE, drop_p=0.0, seed=0, out=None, coef_cached=None, stat=None):
    """ComplexBatchNorm2d (+act +dropout).  running_mean: float [C,2] view of the complex buffer.
    Returns (y, stats [C,8], coef [C,6]).  coef_cached = (stats, coef) of an earlier eval-mode call with the same
    parameters and running statistics: only the apply kernel runs.  stat = (part, rows, pivot) from cconv2d_stats (batch
    statistics only): the partial sums the producing conv left — no statistics pass over x (dcs_cbn_fwd_slabs)."""
    _chk(x, 'x', 5, act=True)
    for n, t in (('weight', weight), ('bias', bias), ('running_mean', running_mean), ('running_covar', running_covar)):
        _chk(t, n)
    B, H, W, C, _ = x.shape
    P = B * H * W
    y = torch.empty_like(x) if out is None else out
    lib = _lib.load()
    if stat is not None and use_batch_stats:
        part, rows, pivot = stat
        _chk(part, 'stat part', 3)
        _chk(pivot, 'stat pivot', 2)
        if part.shape[2] < rows or tuple(part.shape[:2]) != (C, 5) or tuple(pivot.shape) != (C, 2):
            raise _lib.DcsHipError(f'cbn: statistics slabs {tuple(part.shape)} / pivot {tuple(pivot.shape)} do not match C={C}')
        stats = torch.empty((C, 8), dtype=torch.float32, device=x.device)
        coef = torch.empty((C, 6), dtype=torch.float32, device=x.device)
        check(_sym('dcs_cbn_fwd_slabs', x, y)(ptr(x), ptr(y), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_covar), ptr(stats),
                                    ptr(coef), ptr(part), int(rows), int(part.shape[2]), ptr(pivot), P, C, eps,
                                    -1.0 if momentum is None else momentum, act, float(drop_p), int(seed), ptr(SEED_STATE),
                                    cur_stream()), 'dcs_cbn_fwd_slabs')
        return y, stats, coef
    nbytes = lib.dcs_cbn_workspace_bytes(P, C)
    if nbytes < 0:
        raise _lib.DcsHipError(f'cbn: unsupported channel count C={C}')
    ws = _workspace(nbytes, x.device)
    mode = int(bool(use_batch_stats))
    if coef_cached is not None:            # eval mode, coefficients of an earlier call with the same parameters (caller's cache)
        if mode != 0:
            raise _lib.DcsHipError('cbn: cached coefficients are an eval-mode feature')
        stats, coef = coef_cached
        mode = 2
    else:
        stats = torch.empty((C, 8), dtype=torch.float32, device=x.device)
        coef = torch.empty((C, 6), dtype=torch.float32, device=x.device)
    check(_sym('dcs_cbn_fwd', x, y)(ptr(x), ptr(y), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_covar),
                          ptr(stats), ptr(coef), ptr(ws), ws.numel(), P, C, eps,
                          -1.0 if momentum is None else momentum, mode, act,
                          float(drop_p), int(seed), ptr(SEED_STATE), cur_stream()), 'dcs_cbn_fwd')
    return y, stats, coef


def cbn_channel_attention(x, weight, bias, running_mean, running_covar, eps, momentum, act, stat, w1, w2):
    """A decoder stage's CBN (batch statistics from the producing conv: stat of cconv2d_stats) + activation whose apply pass
    also pools its output for the channel attention that follows, and that attention's FC (dcs_cbn_fwd_slabs_pool +
    dcs_channel_attention_fc_fwd: three launches).  Returns (a, stats, coef, ca, pooled, hidden)."""
    _chk(x, 'x', 5, act=True)
    for n, t in (('weight', weight), ('bias', bias), ('running_mean', running_mean), ('running_covar', running_covar),
                 ('w1', w1), ('w2', w2)):
        _chk(t, n)
    part, rows, pivot = stat
    B, H, W, C, _ = x.shape
    HW = H * W
    Ch = w1.shape[-2]
    lib = _lib.load()
    nch = lib.dcs_ca_pool_chunks(HW, C)
    if nch < 1 or part.shape[2] < rows or tuple(part.shape[:2]) != (C, 5) or tuple(pivot.shape) != (C, 2):
        raise _lib.DcsHipError(f'cbn_channel_attention: unsupported C={C} or mismatching statistics slabs {tuple(part.shape)}')
    dev = x.device
    a = torch.empty_like(x)
    new = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    stats, coef, ca, pooled, hidden = new(C, 8), new(C, 6), new(B, C, 2), new(B, C, 2), new(B, Ch, 2)
    nbytes = B * nch * C * 2 * 8
    ws = _workspace(nbytes, dev)
    check(_sym('dcs_cbn_fwd_slabs_pool', x, a)(ptr(x), ptr(a), ptr(weight), ptr(bias), ptr(running_mean), ptr(running_covar),
                                              ptr(stats), ptr(coef), ptr(part), int(rows), int(part.shape[2]), ptr(pivot),
                                              ptr(ws), ws.numel(), B, HW, C, eps, -1.0 if momentum is None else momentum, act,
                                              cur_stream()), 'dcs_cbn_fwd_slabs_pool')
    check(lib.dcs_channel_attention_fc_fwd(ptr(ws), ptr(w1), ptr(w2), ptr(ca), ptr(pooled), ptr(hidden), B, HW, C, Ch,
                                           cur_stream()), 'dcs_channel_attention_fc_fwd')
    return a, stats, coef, ca, pooled, hidden


def cbn_bwd(x, g_out, weight, stats, coef, use_batch_stats, act, drop_p=0.0, seed=0, affine=True, outs=None,
            g_add=None, g_out2=None, need_gx=True):
    """Backward of cbn(): returns (g_x, g_weight [C,3], g_bias [C,2]).  `outs`: optional destinations for
    (g_weight, g_bias).  g_add [B,C,2]: the cotangent is g_out + g_add[b, c] / (H*W) (attention_bwd's g_pooled).
    g_out2: a second cotangent of y (another consumer's), added on the fly.  need_gx = False (the network input's CBN: nothing
    upstream wants g_x): parameter gradients only, the apply pass is not launched and g_x is None."""
    _chk(x, 'x', 5, act=True)
    _chk(g_out, 'g_out', 5, act=True)
    _chk(g_out2, 'g_out2', 5, act=True)
    if g_out2 is not None and g_out2.shape != g_out.shape:
        raise _lib.DcsHipError(f'cbn_bwd: g_out {tuple(g_out.shape)} vs g_out2 {tuple(g_out2.shape)}')
    B, H, W, C, _ = x.shape
    P = B * H * W
    g_x = torch.empty_like(x) if need_gx else None
    o = outs or (None, None)
    g_w = (o[0] if o[0] is not None else torch.empty((C, 3), dtype=torch.float32, device=x.device)) if affine else None
    g_b = (o[1] if o[1] is not None else torch.empty((C, 2), dtype=torch.float32, device=x.device)) if affine else None
    lib = _lib.load()
    nbytes = lib.dcs_cbn_bwd_workspace_bytes(P, C)
    if nbytes < 0:
        raise _lib.DcsHipError(f'cbn_bwd: unsupported channel count C={C}')
    ws = _workspace(nbytes, x.device)
    if g_add is not None:
        _chk(g_add, 'g_add', 3)
    check(_sym('dcs_cbn_bwd_add', x, g_out, g_out2)(ptr(x), ptr(g_out), ptr(g_x), ptr(weight), ptr(stats), ptr(coef), ptr(g_w), ptr(g_b),
                              ptr(ws), ws.numel(), P, C, int(bool(use_batch_stats)), act, float(drop_p), int(seed),
                              ptr(SEED_STATE), ptr(g_add), 1.0 / (H * W), H * W, ptr(g_out2), cur_stream()), 'dcs_cbn_bwd_add')
    return g_x, g_w, g_b


def channel_attention(x, w1, w2):
    """x [B,H,W,C,2]; w1 packed [1,C,Ch,2]; w2 packed [1,Ch,C,2] -> (ca [B,C,2], pooled, hidden)."""
    _chk(x, 'x', 5, act=True)
    _chk(w1, 'w1')
    _chk(w2, 'w2')
    B, H, W, C, _ = x.shape
    Ch = w1.shape[-2]
    ca = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
    pooled = torch.empty((B, C, 2), dtype=torch.float32, device=x.device)
    hidden = torch.empty((B, Ch, 2), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.dcs_ca_workspace_bytes(B, H * W, C)
    if nbytes < 0:
        raise _lib.DcsHipError(f'channel_attention: unsupported channel count C={C}')
    ws = _workspace(nbytes, x.device)
    check(_sym('dcs_channel_attention_fwd', x)(ptr(x), ptr(w1), ptr(w2), ptr(ca), ptr(pooled), ptr(hidden), ptr(ws),
                                        ws.numel(), B, H * W, C, Ch, cur_stream()), 'dcs_channel_attention_fwd')
    return ca, pooled, hidden


def spatial_pool(x, ca=None):
    _chk(x, 'x', 5, act=True)
    _chk(ca, 'ca', 3)
    B, H, W, C, _ = x.shape
    pooled = torch.empty((B, H, W, 2, 2), dtype=torch.float32, device=x.device)
    check(_sym('dcs_spatial_pool_fwd', x)(ptr(x), ptr(ca), ptr(pooled), B, H * W, C, cur_stream()),
          'dcs_spatial_pool_fwd')
    return pooled


def attention_apply(x, ca=None, sa=None, drop_p=0.0, seed=0, out=None):
    _chk(x, 'x', 5, act=True)
    _chk(ca, 'ca', 3)
    _chk(sa, 'sa')
    B, H, W, C, _ = x.shape
    y = torch.empty_like(x) if out is None else out
    check(_sym('dcs_attention_apply_fwd', x, y)(ptr(x), ptr(ca), ptr(sa), ptr(y), B, H * W, C, float(drop_p),
                                              int(seed), ptr(SEED_STATE), cur_stream()), 'dcs_attention_apply_fwd')
    return y


def attention_bwd(x, g_out, ca, sa, sp, pooled, hidden, w1, w2, wsa, ksize, drop_p=0.0, seed=0, outs=None,
                  split_pool=False, defer_fc=False):
    """Backward of the fused attention block.  Returns (g_x, g_fc0_r, g_fc0_i, g_fc2_r, g_fc2_i,
    g_conv1_r, g_conv1_i) with the weight gradients in the reference's parameter layout.  split_pool: g_x lacks the
    average pool's broadcast term and an 8th result g_pooled [B,C,2] is returned for the consumer to add
    (cbn_bwd(g_add=...)).  defer_fc (with split_pool): the FC weight-gradient launch is left out; a 9th result is the job
    attention_bwd_fc_weights() runs later (it owns the workspace the per-sample cotangents live in)."""
    _chk(x, 'x', 5, act=True)
    _chk(g_out, 'g_out', 5, act=True)
    B, H, W, C, _ = x.shape
    HW = H * W
    Ch = hidden.shape[1]
    dev = x.device
    lib = _lib.load()
    g_pre = torch.empty((B, H, W, 1, 2), dtype=torch.float32, device=dev)
    check(_sym('dcs_attention_bwd_sa', x, g_out)(ptr(x), ptr(g_out), ptr(ca), ptr(sa), ptr(g_pre), B, HW, C, float(drop_p),
                                   int(seed), ptr(SEED_STATE), cur_stream()), 'dcs_attention_bwd_sa')
    k, pad = (ksize, ksize), (ksize // 2, ksize // 2)
    g_sp, _ = cconv2d_bwd_data(g_pre, pack_conv_weight_bwd(wsa, k, (1, 1), pad), (H, W, 2), k, (1, 1), pad)
    o = outs or (None,) * 6
    g_c1r, g_c1i, _, _ = cconv2d_bwd_weight(sp, None, g_pre, (1, 2, ksize, ksize), False, k, (1, 1), pad,
                                            outs=(o[4], o[5], None, None))
    g_x = torch.empty_like(x)
    new = lambda shape: torch.empty(shape, dtype=torch.float32, device=dev)
    g0r = o[0] if o[0] is not None else new((Ch, C, 1, 1))
    g0i = o[1] if o[1] is not None else new((Ch, C, 1, 1))
    g2r = o[2] if o[2] is not None else new((C, Ch, 1, 1))
    g2i = o[3] if o[3] is not None else new((C, Ch, 1, 1))
    nbytes = lib.dcs_attention_bwd_workspace_bytes(B, HW, C, Ch)
    if nbytes < 0:
        raise _lib.DcsHipError(f'attention_bwd: unsupported channel count C={C}')
    defer_fc = bool(defer_fc and split_pool)
    # (deferred: a buffer of its own — the cached workspace is the next block's as well)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if defer_fc else _workspace(nbytes, dev)
    g_pooled = new((B, C, 2)) if split_pool else None
    fcp = (None,) * 4 if defer_fc else (g0r, g0i, g2r, g2i)
    check(_sym('dcs_attention_bwd_x', x, g_out)(ptr(x), ptr(g_out), ptr(ca), ptr(sa), ptr(g_sp), ptr(pooled), ptr(hidden), ptr(w1),
                                  ptr(w2), ptr(g_x), ptr(fcp[0]), ptr(fcp[1]), ptr(fcp[2]), ptr(fcp[3]), ptr(g_pooled), ptr(ws),
                                  ws.numel(), B, HW, C, Ch, float(drop_p), int(seed), ptr(SEED_STATE), cur_stream()),
          'dcs_attention_bwd_x')
    if defer_fc:
        return g_x, g0r, g0i, g2r, g2i, g_c1r, g_c1i, g_pooled, (ws, pooled, hidden, g0r, g0i, g2r, g2i, B, HW, C, Ch)
    if split_pool:
        return g_x, g0r, g0i, g2r, g2i, g_c1r, g_c1i, g_pooled
    return g_x, g0r, g0i, g2r, g2i, g_c1r, g_c1i


def attention_bwd_fc_weights(job):
    """The FC weight gradients attention_bwd(defer_fc=True) left out, on the current stream."""
    ws, pooled, hidden, g0r, g0i, g2r, g2i, B, HW, C, Ch = job
    check(_lib.load().dcs_attention_bwd_fc_weights(ptr(ws), ws.numel(), ptr(pooled), ptr(hidden), ptr(g0r), ptr(g0i), ptr(g2r),
                                                   ptr(g2i), B, HW, C, Ch, cur_stream()), 'dcs_attention_bwd_fc_weights')


ATTENTION_BATCH_MAX = 8


def _attention_items(blocks):
    items = (_lib.AttentionItem * len(blocks))()
    for it, blk in zip(items, blocks):
        for k, v in blk.items():
            setattr(it, k, v if isinstance(v, int) else ptr(v))
    return items


def attention_blocks_fwd(xs, w1s, w2s, wsas, biases):
    """Several independent attention blocks (no dropout, 7x7 spatial kernel) in one set of launches
    (dcs_attention_fwd_batched).  xs[i] [B,H_i,W_i,C_i,2]; returns per block (y, ca, pooled, hidden, sp, sa)."""
    lib = _lib.load()
    B, dev = xs[0].shape[0], xs[0].device
    new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
    blocks, outs = [], []
    for x, w1, w2, wsa, bias in zip(xs, w1s, w2s, wsas, biases):
        _chk(x, 'x', 5, act=True)
        _, H, W, C, _ = x.shape
        Ch = w1.shape[-2]
        o = dict(y=torch.empty_like(x), ca=new(B, C, 2), pooled=new(B, C, 2), hidden=new(B, Ch, 2), sp=new(B, H, W, 2, 2),
                 sa=new(B, H, W, 1, 2))
        blocks.append(dict(x=x, w1=w1, w2=w2, wsa=wsa, sa_bias=bias, H=H, W=W, C=C, Ch=Ch, **o))
        outs.append(o)
    items = _attention_items(blocks)
    n = len(blocks)
    nbytes = lib.dcs_attention_fwd_batched_workspace_bytes(n, items, B)
    if nbytes < 0:
        raise _lib.DcsHipError('attention_blocks_fwd: unsupported block geometry')
    ws = _workspace(nbytes, dev)
    check(_sym('dcs_attention_fwd_batched', *xs)(n, items, ptr(ws), ws.numel(), B, cur_stream()), 'dcs_attention_fwd_batched')
    return outs


def attention_blocks_bwd(saved, g_outs, wsa_bwds, fc_outs, split_pool=None):
    """Backward of attention_blocks_fwd.  saved[i]: dict(x, w1, w2, ca, pooled, hidden, sa) of block i; fc_outs[i]: the four
    FC weight-gradient destinations.  Returns per block (g_x, g_pre, g_pooled); the 7x7 conv's weight gradient is the caller's
    (cconv2d_bwd_weight(sp, g_pre)).  split_pool[i]: block i's g_x lacks the average pool's broadcast term and g_pooled [B,C,2]
    is returned for the consumer to add (cbn_bwd(g_add=...)); else g_pooled is None."""
    lib = _lib.load()
    B, dev = g_outs[0].shape[0], g_outs[0].device
    blocks, res = [], []
    split_pool = split_pool or (False,) * len(saved)
    for (sv, g, wb, fo), split in zip(zip(saved, g_outs, wsa_bwds, fc_outs), split_pool):
        _chk(g, 'g_out', 5, act=True)
        x = sv['x']
        _, H, W, C, _ = x.shape
        g_pre = torch.empty((B, H, W, 1, 2), dtype=torch.float32, device=dev)
        g_sp = torch.empty((B, H, W, 2, 2), dtype=torch.float32, device=dev)
        g_x = torch.empty_like(x)
        g_pooled = torch.empty((B, C, 2), dtype=torch.float32, device=dev) if split else None
        blocks.append(dict(x=x, w1=sv['w1'], w2=sv['w2'], ca=sv['ca'], pooled=sv['pooled'], hidden=sv['hidden'], sa=sv['sa'],
                           g_out=g, wsa_bwd=wb, g_pre=g_pre, g_sp=g_sp, g_x=g_x, g_fc0_r=fo[0], g_fc0_i=fo[1], g_fc2_r=fo[2],
                           g_fc2_i=fo[3], H=H, W=W, C=C, Ch=sv['hidden'].shape[1], **({'g_pooled': g_pooled} if split else {})))
        res.append((g_x, g_pre, g_pooled))
    items = _attention_items(blocks)
    n = len(blocks)
    nbytes = lib.dcs_attention_bwd_batched_workspace_bytes(n, items, B)
    if nbytes < 0:
        raise _lib.DcsHipError('attention_blocks_bwd: unsupported block geometry')
    ws = _workspace(nbytes, dev)
    check(_sym('dcs_attention_bwd_batched', *g_outs, *[sv['x'] for sv in saved])(n, items, ptr(ws), ws.numel(), B, cur_stream()), 'dcs_attention_bwd_batched')
    return res


def lstm_layer(gx, w_hh, n_sets, seqs_per_set, S, strides, save=False, save_hprev=False, bias=None):
    """Recurrent half of one LSTM layer for every (set, sequence, direction); see dcs_lstm_layer_fwd.
    gx: float pre-activations addressed by `strides` = (stride_set, stride_n, stride_t) in floats;
    w_hh: float [n_sets, 2, 4H, H].  Returns (out [n_sets*seqs_per_set, S, 2H], gates, c[, hprev])."""
    _chk(gx, 'gx')
    _chk(w_hh, 'w_hh', 4)
    H = w_hh.shape[-1]
    NS = n_sets * seqs_per_set
    out = torch.empty((NS, S, 2 * H), dtype=torch.float32, device=gx.device)
    gates = torch.empty((NS, S, 2, 4 * H), dtype=torch.float32, device=gx.device) if save else None
    c = torch.empty((NS, S, 2, H), dtype=torch.float32, device=gx.device) if save else None
    hprev = torch.empty((NS, S, 2, H), dtype=torch.float32, device=gx.device) if save and save_hprev else None
    ba, bb = (None, None) if bias is None else bias      # gate biases [n_sets, 2*4H] added inside the recurrence
    _chk(ba, 'bias_a')
    _chk(bb, 'bias_b')
    check(_lib.load().dcs_lstm_layer_fwd_bias(ptr(gx), ptr(w_hh), ptr(ba), ptr(bb), ptr(out), ptr(gates), ptr(c), ptr(hprev),
                                              n_sets, seqs_per_set, S, H, strides[0], strides[1], strides[2], cur_stream()),
          'dcs_lstm_layer_fwd_bias')
    return (out, gates, c, hprev) if save_hprev else (out, gates, c)


def lstm_layer_bwd(g_out, gates, c, w_hh, n_sets, seqs_per_set, S, bias_part=False):
    """-> g_pre [NS,S,2,4H] (and, bias_part=True, its per-(sequence, direction) time sums [NS,2,4H])."""
    _chk(g_out, 'g_out', 3)
    _chk(gates, 'gates', 4)
    _chk(c, 'c', 4)
    _chk(w_hh, 'w_hh', 4)
    g_pre = torch.empty_like(gates)
    part = torch.empty((gates.shape[0], 2, gates.shape[-1]), dtype=torch.float32, device=gates.device) if bias_part else None
    check(_lib.load().dcs_lstm_layer_bwd(ptr(g_out), ptr(gates), ptr(c), ptr(w_hh), ptr(g_pre), ptr(part), n_sets,
                                         seqs_per_set, S, w_hh.shape[-1], cur_stream()), 'dcs_lstm_layer_bwd')
    return (g_pre, part) if bias_part else g_pre


def dropout(x, drop_p, seed, out=None):
    _chk(x, 'x')
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().dcs_dropout_fwd(ptr(x), ptr(y), x.numel(), float(drop_p), int(seed), ptr(SEED_STATE), cur_stream()),
          'dcs_dropout_fwd')
    return y


def complex_act(x, act, out=None):
    _chk(x, 'x')
    y = torch.empty_like(x) if out is None else out
    check(_lib.load().dcs_complex_act_fwd(ptr(x), ptr(y), x.numel(), act, cur_stream()), 'dcs_complex_act_fwd')
    return y


def complex_upsample(x, up):
    _chk(x, 'x', 5)
    B, H, W, C, _ = x.shape
    y = torch.empty((B, H * up[0], W * up[1], C, 2), dtype=torch.float32, device=x.device)
    check(_lib.load().dcs_complex_upsample_fwd(ptr(x), ptr(y), B, H, W, C, up[0], up[1], cur_stream()),
          'dcs_complex_upsample_fwd')
    return y


def tapsum(z, ksize, up, pad, backward=False, grad=None, bias=None, bias_grad=None, out_dtype=torch.float32):
    """Spatial half of a Cout = 1 conv (dcs_tapsum_fwd / _bwd).  Forward: z [B,Hs,Ws,CT,2] -> y
    [B,Hs*uf,Ws*ut,1,2], plus the layer's bias if bias = (b_r, b_i) (one float each).  backward=True: grad
    [B,Ho,Wo,1,2] -> gz shaped like z (pass z's shape via `z`); bias_grad = (gb_r, gb_i) destinations (written)."""
    lib = _lib.load()
    if not backward:
        _chk(z, 'z', 5)
        B, Hs, Ws, CT, _ = z.shape
        y = torch.empty((B, Hs * up[0], Ws * up[1], 1, 2), dtype=torch.float32, device=z.device)
        b_r, b_i = bias if bias is not None else (None, None)
        check(lib.dcs_tapsum_fwd(ptr(z), ptr(y), ptr(b_r), ptr(b_i), B, Hs, Ws, CT, ksize[0], ksize[1], up[0], up[1],
                                 pad[0], pad[1], cur_stream()), 'dcs_tapsum_fwd')
        return y
    _chk(grad, 'grad', 5)
    B, Hs, Ws, CT, _ = z
    gz = torch.empty((B, Hs, Ws, CT, 2), dtype=out_dtype, device=grad.device)      # (bf16 where the activations are: _h)
    gb_r, gb_i = bias_grad if bias_grad is not None else (None, None)
    ws = _workspace(lib.dcs_tapsum_bwd_workspace_bytes(), grad.device) if gb_r is not None else None
    check(_sym('dcs_tapsum_bwd', gz)(ptr(grad), ptr(gz), ptr(gb_r), ptr(gb_i), ptr(ws), 0 if ws is None else ws.numel(), B, Hs, Ws, CT,
                             ksize[0], ksize[1], up[0], up[1], pad[0], pad[1], cur_stream()), 'dcs_tapsum_bwd')
    return gz


def cconv_up2_single(x1, x2, wt, b_r, b_i):
    """3x3 / stride 1 / one output channel over the 2x2 upsample of cat(x1, x2) in one kernel (dcs_cconv_up2_single_fwd).
    wt: tap-rows panel [1, 16, ct, 2] of pack_tap_rows."""
    _chk(x1, 'x1', 5, act=True)
    _chk(x2, 'x2', 5, act=True)
    _chk(wt, 'wt', 4)
    B, Hs, Ws, C1, _ = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    y = torch.empty((B, 2 * Hs, 2 * Ws, 1, 2), dtype=torch.float32, device=x1.device)
    check(_sym('dcs_cconv_up2_single_fwd', x1, x2)(ptr(x1), ptr(x2), ptr(wt), ptr(b_r), ptr(b_i), ptr(y), B, Hs, Ws, C1, C2,
                                               wt.shape[2], cur_stream()), 'dcs_cconv_up2_single_fwd')
    return y


def cconv_up2_single_bwd_data(gy, wt, C1, C2, out_dtype=torch.float32):
    """Data gradient of cconv_up2_single straight from the cotangent gy [B, 2 Hs, 2 Ws, 1, 2] (dcs_cconv_up2_single_bwd_data):
    -> (gx1 [B,Hs,Ws,C1,2], gx2 [B,Hs,Ws,C2,2] or None) in out_dtype (bf16 where the activations are stored in bf16)."""
    _chk(gy, 'gy', 5)
    _chk(wt, 'wt', 4)
    B, Ho, Wo = gy.shape[:3]
    Hs, Ws = Ho // 2, Wo // 2
    gx1 = torch.empty((B, Hs, Ws, C1, 2), dtype=out_dtype, device=gy.device)
    gx2 = torch.empty((B, Hs, Ws, C2, 2), dtype=out_dtype, device=gy.device) if C2 else None
    # bench.py's conv family: counted with the flops of the launch it replaces (the factored form's 1x1 tap conv over 16 tap
    # channels, an emulated-MFMA launch), so that the family's totals stay comparable across rounds
    ev = (CONV_TIMER.begin(8.0 * B * Hs * Ws * 16 * (C1 + C2), emulated=True,
                           nbytes=gy.numel() * 4.0 + B * Hs * Ws * (C1 + C2) * 2.0 * gx1.element_size())
          if CONV_TIMER is not None else None)
    check(_sym('dcs_cconv_up2_single_bwd_data', gx1, gx2)(ptr(gy), ptr(wt), ptr(gx1), ptr(gx2), B, Hs, Ws, C1, C2, wt.shape[2],
                                                          cur_stream()), 'dcs_cconv_up2_single_bwd_data')
    if ev is not None:
        CONV_TIMER.end(ev)
    return gx1, gx2


def cconv_up2_single_bwd_weight(gy, x1, x2, w_shape, outs=None, bias_outs=None):
    """Weight and bias gradient of cconv_up2_single (dcs_cconv_up2_single_bwd_weight): -> (g_r, g_i) shaped like conv_tran_r/_i.weight
    [16,1,3,3]; `outs`: optional (g_r, g_i) destinations that are ADDED to (gradient sinks); bias_outs: (gb_r, gb_i) one-float
    destinations (written) or None."""
    _chk(gy, 'gy', 5)
    _chk(x1, 'x1', 5, act=True)
    _chk(x2, 'x2', 5, act=True)
    B, Hs, Ws, C1, _ = x1.shape
    C2 = 0 if x2 is None else x2.shape[3]
    acc = outs is not None and outs[0] is not None and outs[1] is not None
    g_r = outs[0] if acc else torch.empty(w_shape, dtype=torch.float32, device=gy.device)
    g_i = outs[1] if acc else torch.empty(w_shape, dtype=torch.float32, device=gy.device)
    gb_r, gb_i = bias_outs if bias_outs is not None else (None, None)
    lib = _lib.load()
    ws = _workspace(lib.dcs_cconv_up2_single_bwd_weight_workspace_bytes(), gy.device)
    ev = (CONV_TIMER.begin(8.0 * B * Hs * Ws * 16 * (C1 + C2), emulated=True,            # (as in cconv_up2_single_bwd_data)
                           nbytes=gy.numel() * 4.0 + B * Hs * Ws * (C1 + C2) * 2.0 * x1.element_size())
          if CONV_TIMER is not None else None)
    check(_sym('dcs_cconv_up2_single_bwd_weight', x1, x2)(ptr(gy), ptr(x1), ptr(x2), ptr(g_r), ptr(g_i), ptr(gb_r), ptr(gb_i), int(acc),
                                                            ptr(ws), ws.numel(), B, Hs, Ws, C1, C2, cur_stream()),
          'dcs_cconv_up2_single_bwd_weight')
    if ev is not None:
        CONV_TIMER.end(ev)
    return g_r, g_i


def bound_crm(M, eps=10e-7, out=None):
    """M: float [..., 2] interleaved complex."""
    _chk(M, 'M')
    y = torch.empty_like(M) if out is None else out
    check(_lib.load().dcs_bound_crm_fwd(ptr(M), ptr(y), M.numel() // 2, eps, cur_stream()), 'dcs_bound_crm_fwd')
    return y


def bound_mask_apply(Y, M_in, eps=10e-7):
    _chk(Y, 'Y')
    _chk(M_in, 'M_in')
    if Y.shape != M_in.shape:
        raise _lib.DcsHipError(f'bound_mask_apply: Y {tuple(Y.shape)} vs M {tuple(M_in.shape)}')
    M, N, S = torch.empty_like(Y), torch.empty_like(Y), torch.empty_like(Y)
    check(_lib.load().dcs_bound_mask_apply_fwd(ptr(Y), ptr(M_in), ptr(M), ptr(N), ptr(S), Y.numel() // 2, eps,
                                               cur_stream()), 'dcs_bound_mask_apply_fwd')
    return M, N, S


def bound_mask_apply_pair(Y, M_in, eps=10e-7):
    """The same launch with the two estimates STACKED: returns (M, NS) with NS[0] = Y (.) M, NS[1] = Y - Y (.) M in one
    [2, *Y.shape] buffer, so that the waveform synthesis and the SiSNR pair downstream run once over 2B signals."""
    _chk(Y, 'Y')
    _chk(M_in, 'M_in')
    if Y.shape != M_in.shape:
        raise _lib.DcsHipError(f'bound_mask_apply: Y {tuple(Y.shape)} vs M {tuple(M_in.shape)}')
    M = torch.empty_like(Y)
    NS = torch.empty((2,) + tuple(Y.shape), dtype=Y.dtype, device=Y.device)
    check(_lib.load().dcs_bound_mask_apply_fwd(ptr(Y), ptr(M_in), ptr(M), ptr(NS[0]), ptr(NS[1]), Y.numel() // 2, eps,
                                               cur_stream()), 'dcs_bound_mask_apply_fwd')
    return M, NS


def bound2_mask_apply(Y, D_raw, eps=10e-7, pair=False, keep_m1=False, drop_p=0.0, seed=0):
    """Both bound_cRM applications + multiply + subtract over the network's RAW last-stage output (dcs_bound2_mask_apply_fwd).
    Returns (M1 or None, M, N, S) — or (M1 or None, M, NS) with the two estimates stacked when pair=True."""
    _chk(Y, 'Y')
    _chk(D_raw, 'D_raw')
    if Y.shape != D_raw.shape:
        raise _lib.DcsHipError(f'bound2_mask_apply: Y {tuple(Y.shape)} vs D {tuple(D_raw.shape)}')
    M = torch.empty_like(Y)
    M1 = torch.empty_like(Y) if keep_m1 else None
    if pair:
        NS = torch.empty((2,) + tuple(Y.shape), dtype=Y.dtype, device=Y.device)
        N, S = NS[0], NS[1]
    else:
        N, S = torch.empty_like(Y), torch.empty_like(Y)
    check(_lib.load().dcs_bound2_mask_apply_fwd(ptr(Y), ptr(D_raw), ptr(M1), ptr(M), ptr(N), ptr(S), Y.numel() // 2, eps,
                                                float(drop_p), int(seed), ptr(SEED_STATE), cur_stream()),
          'dcs_bound2_mask_apply_fwd')
    return (M1, M, NS) if pair else (M1, M, N, S)


def bound2_mask_apply_bwd(Y, D_raw, g_M1, g_M, g_N, g_S, eps=10e-7, drop_p=0.0, seed=0):
    _chk(D_raw, 'D_raw')
    for n, t in (('Y', Y), ('g_M1', g_M1), ('g_M', g_M), ('g_N', g_N), ('g_S', g_S)):
        _chk(t, n)
    g = torch.empty_like(D_raw)
    check(_lib.load().dcs_bound2_mask_apply_bwd(ptr(Y), ptr(D_raw), ptr(g_M1), ptr(g_M), ptr(g_N), ptr(g_S), ptr(g),
                                                D_raw.numel() // 2, eps, float(drop_p), int(seed), ptr(SEED_STATE),
                                                cur_stream()), 'dcs_bound2_mask_apply_bwd')
    return g


def bound_mask_apply_bwd(Y, M_in, g_M, g_N, g_S, eps=10e-7):
    """Cotangent of M_in; any of g_M / g_N / g_S (and Y when only g_M is given) may be None."""
    _chk(M_in, 'M_in')
    for n, t in (('Y', Y), ('g_M', g_M), ('g_N', g_N), ('g_S', g_S)):
        _chk(t, n)
    g = torch.empty_like(M_in)
    check(_lib.load().dcs_bound_mask_apply_bwd(ptr(Y), ptr(M_in), ptr(g_M), ptr(g_N), ptr(g_S), ptr(g),
                                               M_in.numel() // 2, eps, cur_stream()), 'dcs_bound_mask_apply_bwd')
    return g


def bound2_apply_polar_frames(Y, D_raw, Fp, eps=10e-7, drop_p=0.0, seed=0, want_mask=False, grad=None, g_M=None, hermitian=False):
    """dcs_bound2_apply_polar_frames_fwd / _bwd: Y, D_raw float [B,F,T,2].  Forward (grad None): (M or None, out [2B,T,Fp,2]) — the
    frame-major polar-turned spectra of Y (.) M (rows [0, B)) and Y - Y (.) M (rows [B, 2B)).  With grad [2B,T,Fp,2] (and optionally
    g_M): the cotangent of D_raw."""
    _chk(Y, 'Y', 4)
    _chk(D_raw, 'D_raw', 4)
    if Y.shape != D_raw.shape:
        raise _lib.DcsHipError(f'bound2_apply_polar_frames: Y {tuple(Y.shape)} vs D {tuple(D_raw.shape)}')
    B, F, T, _ = Y.shape
    lib = _lib.load()
    if grad is None:
        out = torch.empty((2 * B, T, Fp, 2), dtype=torch.float32, device=Y.device)
        M = torch.empty_like(Y) if want_mask else None
        check(lib.dcs_bound2_apply_polar_frames_fwd(ptr(Y), ptr(D_raw), ptr(M), ptr(out), B, F, Fp, T, eps, float(drop_p), int(seed),
                                                    ptr(SEED_STATE), cur_stream()), 'dcs_bound2_apply_polar_frames_fwd')
        return M, out
    _chk(grad, 'grad', 4)
    _chk(g_M, 'g_M', 4)
    if tuple(grad.shape) != (2 * B, T, Fp, 2):
        raise _lib.DcsHipError(f'bound2_apply_polar_frames: grad {tuple(grad.shape)} for B={B}, T={T}, Fp={Fp}')
    g = torch.empty_like(D_raw)
    check(lib.dcs_bound2_apply_polar_frames_bwd(ptr(Y), ptr(D_raw), ptr(grad), ptr(g_M), ptr(g), B, F, Fp, T, eps, int(bool(hermitian)),
                                                float(drop_p), int(seed), ptr(SEED_STATE), cur_stream()),
          'dcs_bound2_apply_polar_frames_bwd')
    return g


def polar_frames(z, Fp, eps=10e-7, grad=None, hermitian=False):
    """z: float [B,F,T,2].  Forward (grad None): FRAME-MAJOR [B,T,Fp,2] = |z| unit(z_r+eps, z_i), zero bins F..Fp-1.
    With grad [B,T,Fp,2]: the cotangent of z (hermitian: grad is the plain rfft of an unnormalised irfft's output
    cotangent; the one-sided x2 weighting is applied in the kernel)."""
    _chk(z, 'z', 4)
    B, F, T, _ = z.shape
    lib = _lib.load()
    if grad is None:
        out = torch.empty((B, T, Fp, 2), dtype=torch.float32, device=z.device)
        check(lib.dcs_polar_frames_fwd(ptr(z), ptr(out), B, F, Fp, T, eps, cur_stream()), 'dcs_polar_frames_fwd')
        return out
    _chk(grad, 'grad', 4)
    gz = torch.empty_like(z)
    check(lib.dcs_polar_frames_bwd(ptr(z), ptr(grad), ptr(gz), B, F, Fp, T, eps, int(bool(hermitian)), cur_stream()),
          'dcs_polar_frames_bwd')
    return gz


def rfft512_ola(gy, window, inv_env, T, hop, scale):
    """rfft512(istft_ola backward frames of gy) without storing the frames (dcs_rfft512_ola_frames): gy float [B, hop (T - 1)] ->
    [B, T, 257, 2]."""
    _chk(gy, 'gy', 2)
    B, L = gy.shape
    if L != hop * (T - 1) or window.numel() != 512:
        raise _lib.DcsHipError(f'rfft512_ola: gy {tuple(gy.shape)} for T={T}, hop={hop}, window {window.numel()}')
    G = torch.empty((B, T, 257, 2), dtype=torch.float32, device=gy.device)
    check(_lib.load().dcs_rfft512_ola_frames(ptr(gy), ptr(window), ptr(inv_env), ptr(G), B, T, hop, float(scale), cur_stream()),
          'dcs_rfft512_ola_frames')
    return G


def irfft512_ola(X, window, inv_env, hop, scale):
    """istft_ola(irfft512(X)) without storing the frames (dcs_irfft512_ola_frames): X float [B, T, 257, 2] -> [B, hop (T - 1)]."""
    _chk(X, 'X', 4)
    B, T = X.shape[:2]
    if X.shape[-2:] != (257, 2) or window.numel() != 512:
        raise _lib.DcsHipError(f'irfft512_ola: X {tuple(X.shape)}, window {window.numel()}')
    y = torch.empty((B, hop * (T - 1)), dtype=torch.float32, device=X.device)
    check(_lib.load().dcs_irfft512_ola_frames(ptr(X), ptr(window), ptr(inv_env), ptr(y), B, T, hop, float(scale), cur_stream()),
          'dcs_irfft512_ola_frames')
    return y


def irfft512_ola_ok(T, hop):
    return T >= 2 and hop in (64, 128, 256)


def irfft512(X):
    """X float [..., 257, 2] one-sided spectra -> [..., 512] unnormalised inverse real FFT (dcs_irfft512_frames)."""
    _chk(X, 'X')
    if X.shape[-2:] != (257, 2):
        raise _lib.DcsHipError(f'irfft512: expected [..., 257, 2], got {tuple(X.shape)}')
    y = torch.empty((*X.shape[:-2], 512), dtype=torch.float32, device=X.device)
    check(_lib.load().dcs_irfft512_frames(ptr(X), ptr(y), X.numel() // 514, cur_stream()), 'dcs_irfft512_frames')
    return y


def rfft512(g):
    """g float [..., 512] -> [..., 257, 2] forward real FFT, no scaling (dcs_rfft512_frames)."""
    _chk(g, 'g')
    if g.shape[-1] != 512:
        raise _lib.DcsHipError(f'rfft512: expected [..., 512], got {tuple(g.shape)}')
    G = torch.empty((*g.shape[:-1], 257, 2), dtype=torch.float32, device=g.device)
    check(_lib.load().dcs_rfft512_frames(ptr(g), ptr(G), g.numel() // 512, cur_stream()), 'dcs_rfft512_frames')
    return G


def istft_envelope(window, T, hop):
    """1 / (squared-window overlap-add envelope) of torch.istft(center=True) for T frames: float [hop*(T-1)]."""
    _chk(window, 'window', 1)
    n_fft = window.numel()
    inv_env = torch.empty(hop * (T - 1), dtype=torch.float32, device=window.device)
    check(_lib.load().dcs_istft_envelope(ptr(window), ptr(inv_env), T, n_fft, hop, cur_stream()), 'dcs_istft_envelope')
    return inv_env


def istft_ola(frames, window, inv_env, hop, scale=1.0, grad=None):
    """frames: float [B,T,n_fft] (irfft output).  Forward: windowed overlap-add / envelope / trim -> [B, hop*(T-1)].
    With grad [B, hop*(T-1)]: the cotangent of frames (pass the frames' shape via `frames`)."""
    B, T, n_fft = frames if grad is not None else frames.shape
    _chk(window, 'window', 1)
    _chk(inv_env, 'inv_env', 1)
    if window.numel() != n_fft or inv_env.numel() != hop * (T - 1):
        raise _lib.DcsHipError(f'istft_ola: window {window.numel()} / envelope {inv_env.numel()} do not match '
                               f'n_fft={n_fft}, T={T}, hop={hop}')
    lib = _lib.load()
    if grad is None:
        _chk(frames, 'frames', 3)
        y = torch.empty((B, hop * (T - 1)), dtype=torch.float32, device=frames.device)
        check(lib.dcs_istft_ola_fwd(ptr(frames), ptr(window), ptr(inv_env), ptr(y), B, T, n_fft, hop, float(scale),
                                    cur_stream()), 'dcs_istft_ola_fwd')
        return y
    _chk(grad, 'grad', 2)
    gf = torch.empty((B, T, n_fft), dtype=torch.float32, device=grad.device)
    check(lib.dcs_istft_ola_bwd(ptr(grad), ptr(window), ptr(inv_env), ptr(gf), B, T, n_fft, hop, float(scale),
                                cur_stream()), 'dcs_istft_ola_bwd')
    return gf


def stft_frames(clean, noisy, window, T, hop):
    """Windowed, reflect-padded frames of (clean, noisy - clean, noisy): float [3,B,T,n_fft] (dcs_stft_frames_fwd)."""
    _chk(clean, 'clean', 2)
    _chk(noisy, 'noisy', 2)
    _chk(window, 'window', 1)
    if clean.shape != noisy.shape:
        raise _lib.DcsHipError(f'stft_frames: clean {tuple(clean.shape)} vs noisy {tuple(noisy.shape)}')
    B, L = clean.shape
    n_fft = window.numel()
    frames = torch.empty((3, B, T, n_fft), dtype=torch.float32, device=clean.device)
    check(_lib.load().dcs_stft_frames_fwd(ptr(clean), ptr(noisy), ptr(window), ptr(frames), B, L, T, n_fft, hop, cur_stream()),
          'dcs_stft_frames_fwd')
    return frames


def stft_bins(spec, scale):
    """spec float [S,B,T,F+1,2] (rfft of the frames) -> float [S,B,F,T,2]: DC bin dropped, scaled, transposed."""
    _chk(spec, 'spec', 5)
    S, B, T, F1, _ = spec.shape
    out = torch.empty((S, B, F1 - 1, T, 2), dtype=torch.float32, device=spec.device)
    check(_lib.load().dcs_stft_bins_fwd(ptr(spec), ptr(out), S * B, T, F1 - 1, float(scale), cur_stream()), 'dcs_stft_bins_fwd')
    return out


def sisnr(clean, est, eps=1e-8):
    """Per-utterance SiSNR [B] of float [B,L] signals and the [B,2] coefficients its backward reads."""
    _chk(clean, 'clean', 2)
    _chk(est, 'est', 2)
    if clean.shape != est.shape:
        raise _lib.DcsHipError(f'sisnr: clean {tuple(clean.shape)} vs estimate {tuple(est.shape)}')
    B, L = est.shape
    snr = torch.empty(B, dtype=torch.float32, device=est.device)
    coef = torch.empty((B, 2), dtype=torch.float32, device=est.device)
    check(_lib.load().dcs_sisnr_fwd(ptr(clean), ptr(est), ptr(snr), ptr(coef), B, L, float(eps), cur_stream()), 'dcs_sisnr_fwd')
    return snr, coef


def sisnr_losses(snr_speech, snr_noise, alpha):
    """[noise_loss, speech_loss, total] (float [3]) of the reference's SiSNR loss configuration."""
    _chk(snr_speech, 'snr_speech', 1)
    _chk(snr_noise, 'snr_noise', 1)
    out = torch.empty(3, dtype=torch.float32, device=snr_speech.device)
    check(_lib.load().dcs_sisnr_losses_fwd(ptr(snr_speech), ptr(snr_noise), ptr(out), snr_speech.numel(), float(alpha),
                                           cur_stream()), 'dcs_sisnr_losses_fwd')
    return out


def lstm_combine(o, B):
    """o float [2 sets, 2B rows, S, W] -> complex64 [B, S, W] = (L_r(x_r) - L_i(x_i)) + j (L_r(x_i) + L_i(x_r))."""
    _chk(o, 'o', 4)
    _, B2, S, W = o.shape
    out = torch.empty((B, S, W, 2), dtype=torch.float32, device=o.device)
    check(_lib.load().dcs_lstm_combine_fwd(ptr(o), ptr(out), B * S * W, cur_stream()), 'dcs_lstm_combine_fwd')
    return torch.view_as_complex(out)


def lstm_combine_bwd(g):
    """g float [B, S, W, 2] (cotangent of the complex output) -> g_o float [2, 2B, S, W]."""
    _chk(g, 'g', 4)
    B, S, W, _ = g.shape
    g_o = torch.empty((2, 2 * B, S, W), dtype=torch.float32, device=g.device)
    check(_lib.load().dcs_lstm_combine_bwd(ptr(g), ptr(g_o), B * S * W, cur_stream()), 'dcs_lstm_combine_bwd')
    return g_o


def lstm_whh_grad(g_pre, h_prev, NT, CK, H):
    """part float [2 dirs, 2*CK, 4H, H]: the chunked g_pre^T h_prev products of one layer (both sets, both directions)."""
    _chk(g_pre, 'g_pre')
    _chk(h_prev, 'h_prev')
    part = torch.empty((2, 2 * CK, 4 * H, H), dtype=torch.float32, device=g_pre.device)
    check(_lib.load().dcs_lstm_whh_grad(ptr(g_pre), ptr(h_prev), ptr(part), NT, CK, H, cur_stream()), 'dcs_lstm_whh_grad')
    return part


def atb_chunks_acc(A, B, out, nsets, lda, ldb, M, N, NT, CK, b_shared=False, reduce=True):
    """out[s] (float [M, N], contiguous per set) += A[s]^T B[s] over the NT rows (A: [nsets, NT, lda-pitched M columns],
    B: [nsets or 1, NT, N]) as CK row chunks on the MFMA pipe + one fixed-order chunk sum.  reduce=False: the products only;
    returns (part [nsets * CK, M, N], CK) for a later sum (lstm_param_grads(..., ih=...))."""
    for n, t in (('A', A), ('B', B), ('out', out)):
        _chk(t, n)
    lib = _lib.load()
    part = torch.empty((nsets * CK, M, N), dtype=torch.float32, device=A.device)
    check(lib.dcs_atb_chunks(ptr(A), ptr(B), ptr(part), NT * lda, 0, 0 if b_shared else NT * ldb, 0, nsets, 1, lda, ldb, M, N,
                             NT // CK, CK, cur_stream()), 'dcs_atb_chunks')
    if not reduce:
        return part, CK
    check(lib.dcs_chunk_sum_acc(ptr(part), ptr(out), M * N, 0, nsets, 1, CK, M * N, cur_stream()), 'dcs_chunk_sum_acc')


LSTM_GEMM = _os.environ.get('DCS_LSTM_GEMM', '1') != '0'      # 0: the LSTM projections through torch.mm / bmm (rocBLAS), for A/B runs
# launches above this much work go to the library: the in-tree kernel's 32 x 64 wave tiles are sized for the train shapes
# (csrc/gemm.hip; same-box: inference pass 2.99 ms with the library's 256 x 256 tiles, 3.06 ms in-tree)
LSTM_GEMM_MAX_GFLOP = float(_os.environ.get('DCS_LSTM_GEMM_MAX_GFLOP', '1.5'))


def gemm_ok(M, N, K, launches=1, train=False):
    """Shapes dcs_gemm_f32 takes and is the faster choice for (the LSTM projections of the train shapes all are); N, K: one
    launch's columns and contraction length, launches: batches x segments.  train: a launch of a TRAIN step — the work cap does
    not apply there: a train step's backward runs the weight-gradient MFMA kernels on a side stream (dp.TrainStep), and a library
    GEMM (or the ATen copy kernels around it) co-resident with them is code the packed-FMA scan of tests/test_host_cpu.py never
    sees (profiles/r03_pk_fma_op_sel_hazard.txt) — so every train-step GEMM of this shape family stays in-tree,
    whatever its size (B = 64 in bf16 storage: 2.1 GFLOP per launch)."""
    return (LSTM_GEMM and M >= 1 and N >= 64 and N % 64 == 0 and K >= 32 and K % 32 == 0
            and (train or 2e-9 * M * N * K * launches <= LSTM_GEMM_MAX_GFLOP))


def gemm_f32(A, B, C, M, N, K, lda, ldb, ldc, b_transposed, nseg=1, a_seg=0, b_seg=0, nbatch=1, a_batch=0, b_batch=0,
             c_batch=0, a_planes=0, c_planes=0):
    """C_b = sum_s A_{b,s} op(B_{b,s}) on the fp32 MFMA pipe (include/dcsnet_hip.h: dcs_gemm_f32); strides in floats."""
    for n, t in (('A', A), ('B', B), ('C', C)):
        _chk(t, n)
    check(_lib.load().dcs_gemm_f32(ptr(A), ptr(B), ptr(C), M, N, K, lda, ldb, ldc, 1 if b_transposed else 0, nseg, a_seg,
                                   b_seg, nbatch, a_batch, b_batch, c_batch, a_planes, c_planes, cur_stream()), 'dcs_gemm_f32')
    return C


def atb_chunks_acc_planes(A, zr, out, nsets, lda, M, N, R0, CK, reduce=True):
    """atb_chunks_acc whose shared B is the {re rows | im rows} stacking of a complex-interleaved zr float[R0, N, 2], read in
    place: out[s] += A[s]^T [zr.re ; zr.im], A: [nsets, 2 * R0, lda-pitched M columns]; CK chunks per part."""
    for n, t in (('A', A), ('zr', zr), ('out', out)):
        _chk(t, n)
    lib = _lib.load()
    part = torch.empty((nsets * 2 * CK, M, N), dtype=torch.float32, device=A.device)
    # batches: lo = part (real / imaginary rows), hi = set -> part index (set * 2 + part) * CK + c: 2 CK chunks per set
    check(lib.dcs_atb_chunks_strided(ptr(A), ptr(zr), ptr(part), R0 * lda, 2 * R0 * lda, 1, 0, 2, nsets, lda, 2 * N, 2, M, N,
                                     R0 // CK, CK, cur_stream()), 'dcs_atb_chunks_strided')
    if not reduce:
        return part, 2 * CK
    check(lib.dcs_chunk_sum_acc(ptr(part), ptr(out), M * N, 0, nsets, 1, 2 * CK, M * N, cur_stream()), 'dcs_chunk_sum_acc')


def lstm_param_grads(part, b_part, g_whh, g_bih, g_bhh, CK, seqs, H, ih=None):
    """Accumulate one layer's recurrent-weight and bias gradients from the backward's partial products (in place).
    ih = (part_ih, CK_ih, g_wih [nsets, M, N]): the input-projection weight gradient's chunk sum in the same launch."""
    for n, t in (('part', part), ('b_part', b_part), ('g_whh', g_whh), ('g_bih', g_bih), ('g_bhh', g_bhh)):
        _chk(t, n)
    if ih is None:
        check(_lib.load().dcs_lstm_param_grads(ptr(part), ptr(b_part), ptr(g_whh), ptr(g_bih), ptr(g_bhh), CK, seqs, H,
                                               cur_stream()), 'dcs_lstm_param_grads')
        return
    part_ih, ck_ih, g_wih = ih
    _chk(part_ih, 'part_ih')
    _chk(g_wih, 'g_wih', 3)
    check(_lib.load().dcs_lstm_param_grads_ih(ptr(part), ptr(b_part), ptr(g_whh), ptr(g_bih), ptr(g_bhh), CK, seqs, H, ptr(part_ih),
                                              ptr(g_wih), ck_ih, g_wih.shape[1] * g_wih.shape[2], g_wih.shape[0], cur_stream()),
          'dcs_lstm_param_grads_ih')


def sisnr_losses_guard(snr_speech, snr_noise, alpha, skip):
    """sisnr_losses with the train step's NaN flag written by the same launch (skip: 1-element float tensor or None)."""
    _chk(snr_speech, 'snr_speech', 1)
    _chk(snr_noise, 'snr_noise', 1)
    _chk(skip, 'skip')
    out = torch.empty(3, dtype=torch.float32, device=snr_speech.device)
    check(_lib.load().dcs_sisnr_losses_guard_fwd(ptr(snr_speech), ptr(snr_noise), ptr(out), snr_speech.numel(), float(alpha),
                                                 ptr(skip), cur_stream()), 'dcs_sisnr_losses_guard_fwd')
    return out


def sisnr_pair_bwd(target, est, coef, g_noise, g_speech, g_total, alpha):
    """Cotangent of est [2B, L] (rows [0,B) noise, [B,2B) speech) for the configured loss pair; g_*: device scalars or None."""
    B2, L = est.shape
    for n, t in (('g_noise', g_noise), ('g_speech', g_speech), ('g_total', g_total)):
        _chk(t, n)
    g_est = torch.empty_like(est)
    check(_lib.load().dcs_sisnr_pair_bwd(ptr(target), ptr(est), ptr(coef), ptr(g_noise), ptr(g_speech), ptr(g_total),
                                         float(alpha), ptr(g_est), B2 // 2, L, cur_stream()), 'dcs_sisnr_pair_bwd')
    return g_est


def sisnr_bwd(clean, est, coef, g, scale):
    """g: 0-dim / 1-element float tensor on the device (upstream gradient of the batch mean)."""
    B, L = est.shape
    _chk(g, 'g')
    g_est = torch.empty_like(est)
    check(_lib.load().dcs_sisnr_bwd(ptr(clean), ptr(est), ptr(coef), ptr(g), float(scale), ptr(g_est), B, L, cur_stream()),
          'dcs_sisnr_bwd')
    return g_est


def crm(S, Y, eps=1e-8):
    _chk(S, 'S')
    _chk(Y, 'Y')
    M = torch.empty_like(S)
    check(_lib.load().dcs_crm_fwd(ptr(S), ptr(Y), ptr(M), S.numel() // 2, eps, cur_stream()), 'dcs_crm_fwd')
    return M


# ---- complex <-> channels-last float views -------------------------------------------------

def to_nhwc(z):
    """complex64 [B,C,H,W] (any strides) -> float32 [B,H,W,C,2] contiguous (no copy if the input
    is already in channels_last memory format)."""
    if z.dtype != torch.complex64:
        raise _lib.DcsHipError(f'expected complex64, got {z.dtype}')
    return torch.view_as_real(z.permute(0, 2, 3, 1).contiguous())


def from_nhwc(x):
    """float32 [B,H,W,C,2] -> complex64 logical [B,C,H,W] in channels_last memory (a view)."""
    return torch.view_as_complex(x).permute(0, 3, 1, 2)
