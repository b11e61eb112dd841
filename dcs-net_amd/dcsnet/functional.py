"""Functional layer between the nn.Module surface and the C-ABI wrappers (ops.py).

Everything here takes / returns channels-last float32 views ``[B,H,W,C,2]`` unless it says
"complex" in its name.  Packed weights are cached per parameter version, so inference packs
once and training re-packs after each optimizer step.
"""
import os
import weakref

import torch

from . import ops
from .ops import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, to_nhwc, from_nhwc  # noqa: F401
from ._lib import DcsHipError

_pack_cache = {}


_param_generation = 0


def bump_param_generation():
    """Called by optimizers that update parameters outside torch's version counters (the fused
    HIP Adam, a replayed train-step graph): every packed weight and every inference-time constant derived from
    parameters or running statistics becomes stale."""
    global _param_generation
    _param_generation += 1
    _pack_cache.clear()
    ops.pack_plan_invalidate()


def note_state_update():
    """A kernel changed module state behind torch's version counters (train-mode CBN: running statistics)."""
    global _param_generation
    _param_generation += 1


def state_generation():
    return _param_generation


def begin_pack_plan():
    """Start recording every weight pack of the coming step (they still execute)."""
    _pack_cache.clear()
    return ops.pack_plan_begin()


def end_pack_plan():
    return ops.pack_plan_end()


def run_pack_plan(plan):
    """Re-derive every recorded packed weight from the current parameters (4 launches) and serve them to
    packed_weight() / ops.pack_conv_weight_bwd until the next parameter update."""
    ops.pack_plan_run(plan)


def _ver(t):
    return (0, 0) if t is None else (id(t), t._version)


def packed_weight(w_r, w_i, b_r, b_i, transposed, up=(1, 1), tap_rows=0):
    """Packed (wp, bias) for a weight pair; cached per tensor OBJECT and version (the weakrefs
    guard against a recycled id()).  1x1 / Linear weights may be passed 2-D.  `up`: upsample factors
    of the conv call the weight is for.  tap_rows = ct > 0: the 1x1 "tap channel" weight of the tap-sum factorisation
    instead (ops.pack_tap_rows)."""
    plan = ops.PLAN
    if plan is not None:
        pkey = (id(w_r), id(w_i), id(b_r), id(b_i), bool(transposed), tuple(up), int(tap_rows))
        if plan.valid and not plan.recording:
            e = plan.fwd.get(pkey)
            if e is not None and e[0][0]() is w_r and e[0][1]() is w_i and e[2] == (
                    w_r._version, w_i._version, None if b_r is None else b_r._version,
                    None if b_i is None else b_i._version):
                return e[1]
    key = (_ver(w_r), _ver(w_i), _ver(b_r), _ver(b_i), bool(transposed), tuple(up), int(tap_rows))
    hit = _pack_cache.get(key)
    if hit is not None and hit[0]() is w_r and hit[1]() is w_i:
        return hit[2]
    if len(_pack_cache) > 512:
        _pack_cache.clear()
    d = lambda t: None if t is None else t.detach()
    wr, wi = d(w_r), d(w_i)
    if wr.dim() == 2:
        wr, wi = wr.view(*wr.shape, 1, 1), wi.view(*wi.shape, 1, 1)
    if tap_rows:
        packed = ops.pack_tap_rows(wr, wi, int(tap_rows))
    else:
        packed = ops.pack_conv_weight(wr, wi, d(b_r), d(b_i), transposed, tuple(up))
    _pack_cache[key] = (weakref.ref(w_r), weakref.ref(w_i), packed)
    if plan is not None and plan.recording:
        wref = lambda t: None if t is None else weakref.ref(t)
        plan.fwd[pkey] = [(wref(w_r), wref(w_i), wref(b_r), wref(b_i)), packed, None]
    return packed


ATTENTION_BATCH_MAX = ops.ATTENTION_BATCH_MAX

FLUSH_AT_NEXT_FORK = False   # set by the batched skip-attention backward: the next conv fork also flushes the recorded 7x7 problems
WGRAD_SIDE = None      # the side stream the deferred weight-gradient kernels of a train step run on (dp.TrainStep._backward), else None
sink_hits = 0          # diagnostics: how many parameter gradients were routed to a sink


def _sink(p):
    """Destination registered by dp.FlatBucket for this parameter's gradient (a view of the flat gradient
    bucket that IS p.grad), or None.  Backward kernels then write there directly and return None to
    autograd: no temporary, no per-parameter accumulate kernel (~200 tiny launches per step)."""
    if p is None or not p.requires_grad:
        return None
    s = getattr(p, '_dcs_grad_sink', None)
    if s is None or p.grad is None or p.grad.data_ptr() != s.data_ptr():
        return None
    global sink_hits
    sink_hits += 1
    return s


DEFER_FC = os.environ.get('DCS_DEFER_FC', '1') != '0'      # 0: the attention blocks' FC weight gradients stay on the main chain (A/B)
PENDING_SIDE = []          # launches nothing downstream waits for, queued by backward nodes for the next fork of the side stream


def _run_pending_side():
    """Issue the queued launches on the CURRENT stream (the side stream right after a fork, or — dp.TrainStep._backward — whatever
    is left before the join); their operands stay alive until the join (ops.WGRAD_DEFER)."""
    while PENDING_SIDE:
        job = PENDING_SIDE.pop(0)
        ops.attention_bwd_fc_weights(job)
        if ops.WGRAD_DEFER is not None:
            ops.WGRAD_DEFER.append(job)


class _CConv2dFn(torch.autograd.Function):
    """dcs_cconv2d_fwd with its hand-written data / weight gradients."""

    @staticmethod
    def forward(ctx, x1, x2, w_r, w_i, b_r, b_i, transposed, ksize, stride, pad, up, act, holder=None):
        wp, bias = packed_weight(w_r, w_i, b_r, b_i, transposed, up)
        if holder is not None:                 # a training-mode CBN follows: its statistics come out of this conv's epilogue
            y, holder['stat'] = ops.cconv2d_stats(x1, x2, wp, bias, ksize, stride, pad, up)
        else:
            y = ops.cconv2d(x1, x2, wp, bias, ksize, stride, pad, up, act)
        ctx.geom = (transposed, tuple(ksize), tuple(stride), tuple(pad), tuple(up), act, tuple(w_r.shape),
                    b_r is not None)
        ctx.sinks = (_sink(w_r), _sink(w_i), _sink(b_r), _sink(b_i))
        ctx.save_for_backward(x1, x2, wp, y if act == ACT_SIGMOID else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x1, x2, wp, y = ctx.saved_tensors
        transposed, ksize, stride, pad, up, act, w_shape, has_bias = ctx.geom
        gy = gy.contiguous()
        if act == ACT_SIGMOID:
            gy = gy * y * (1.0 - y)
        elif act != ACT_NONE:
            raise DcsHipError('cconv2d backward: only ACT_NONE / ACT_SIGMOID epilogues are differentiable here')
        gx1 = gx2 = gw_r = gw_i = gb_r = gb_i = None
        need = ctx.needs_input_grad
        want_w = need[2] or need[3] or need[4] or need[5]
        # The weight gradient feeds nothing before the optimizer, the data gradient feeds the whole rest of the backward pass:
        # inside TrainStep's deferred-reduce scope (every result lands in the gradient bucket, slabs and operands are kept
        # alive until the flush) the weight-gradient kernel goes to a SIDE stream that forks here, BEFORE the data gradient
        # is queued, and joins once, in front of the flush (dp.TrainStep._backward) — it runs beside this layer's data
        # gradient and under the launch-bound CBN / attention backward kernels of the next layer, which leave most CUs idle.
        side = None
        if (want_w and WGRAD_SIDE is not None and ops.WGRAD_DEFER is not None and gy.is_cuda and
                ctx.sinks[0] is not None and ctx.sinks[1] is not None and (not has_bias or (ctx.sinks[2] is not None and ctx.sinks[3] is not None))):
            side = WGRAD_SIDE
            side.wait_stream(torch.cuda.current_stream())
        if need[0] or need[1]:
            C1 = x1.shape[3]
            Cin = C1 + (x2.shape[3] if x2 is not None else 0)
            gx1, gx2 = ops.cconv2d_bwd_data(gy, ops.pack_conv_weight_bwd(wp, ksize, stride, pad, up),
                                            (x1.shape[1], x1.shape[2], Cin), ksize, stride, pad, up, C1)
        if want_w:
            if side is not None:
                with torch.cuda.stream(side):
                    _run_pending_side()
                    g = ops.cconv2d_bwd_weight(x1, x2, gy, w_shape, has_bias, ksize, stride, pad, up, transposed, ctx.sinks,
                                               immediate=True)
                    global FLUSH_AT_NEXT_FORK
                    if FLUSH_AT_NEXT_FORK:
                        # every 7x7 attention-conv problem of the step is recorded by now (the batched skip attentions came last):
                        # their one batched kernel and its reductions ride on this fork and run beside the LSTM backward
                        FLUSH_AT_NEXT_FORK = False
                        ops.wgrad_defer_flush(partial=True)
            else:
                g = ops.cconv2d_bwd_weight(x1, x2, gy, w_shape, has_bias, ksize, stride, pad, up, transposed, ctx.sinks)
            # gradients written straight into their sink are not handed back to autograd
            gw_r, gw_i, gb_r, gb_i = (None if sk is not None else t for t, sk in zip(g, ctx.sinks))
        return gx1, gx2, gw_r, gw_i, gb_r, gb_i, None, None, None, None, None, None, None


def cconv2d(x1, x2, w_r, w_i, b_r, b_i, transposed, ksize, stride, pad, up=(1, 1), act=ACT_NONE):
    return _CConv2dFn.apply(x1, x2, w_r, w_i, b_r, b_i, transposed, tuple(ksize), tuple(stride), tuple(pad),
                            tuple(up), act)


def cconv2d_with_stats(x1, x2, w_r, w_i, b_r, b_i, transposed, ksize, stride, pad, up=(1, 1)):
    """(y, stat): the conv with no activation, and the batch statistics of y its epilogue left for the training-mode
    ComplexBatchNorm2d that follows (c_network.py:107-114: conv and CBN back to back) — stat goes to cbn(..., stat=stat);
    None when the geometry has no statistics epilogue.  Not differentiable through the statistics (the CBN's backward is
    the closed form over x and its saved moments, whichever way the moments were summed)."""
    holder = {}
    y = _CConv2dFn.apply(x1, x2, w_r, w_i, b_r, b_i, transposed, tuple(ksize), tuple(stride), tuple(pad), tuple(up),
                         ACT_NONE, holder)
    return y, holder.get('stat')


def cconv2d_cbn_eval(x1, x2, w_r, w_i, b_r, b_i, transposed, ksize, stride, pad, up, coef, act):
    """Inference only: conv + eval-mode CBN (its cached coefficients, complexLayers.eval_coef) + activation as ONE kernel
    (dcs_cconv2d_fwd_affine) — the CBN's separate pass over the activation disappears."""
    wp, bias = packed_weight(w_r, w_i, b_r, b_i, transposed, tuple(up))
    return ops.cconv2d(x1, x2, wp, bias, tuple(ksize), tuple(stride), tuple(pad), tuple(up), act, coef=coef)


class _CbnFn(torch.autograd.Function):
    """dcs_cbn_fwd / dcs_cbn_bwd.  Saves only x and 14 floats per channel."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act, drop_p, seed, stat=None):
        y, stats, coef = ops.cbn(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats,
                                 act, drop_p, seed, stat=stat)
        ctx.cfg = (bool(use_batch_stats), act, float(drop_p), int(seed), weight is not None)
        ctx.sinks = (_sink(weight), _sink(bias))
        ctx.save_for_backward(x, weight, stats, coef)
        return y

    @staticmethod
    def backward(ctx, g_out):
        x, weight, stats, coef = ctx.saved_tensors
        use_batch, act, drop_p, seed, affine = ctx.cfg
        g_x, g_w, g_b = ops.cbn_bwd(x, g_out.contiguous(), weight, stats, coef, use_batch, act, drop_p, seed, affine,
                                    ctx.sinks, need_gx=ctx.needs_input_grad[0])
        g_w = None if ctx.sinks[0] is not None else g_w
        g_b = None if ctx.sinks[1] is not None else g_b
        return g_x, g_w, g_b, None, None, None, None, None, None, None, None, None


def cbn(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act=ACT_NONE,
        drop_p=0.0, seed=0, stat=None):
    return _CbnFn.apply(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act,
                        drop_p, seed, stat)


class _CbnTwoFn(torch.autograd.Function):
    """_CbnFn whose output is handed out TWICE (two tensors over one storage) for a stage output with two consumers
    (the next encoder conv and a skip attention, c_network.py:193-197 / :208-211).  Autograd then delivers the two
    cotangents separately and the CBN backward kernels add them on the fly (dcs_cbn_bwd_add g_out2) instead of an
    element-wise add launch over the activation per stage."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act, drop_p, seed, stat=None):
        y, stats, coef = ops.cbn(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats,
                                 act, drop_p, seed, stat=stat)
        ctx.cfg = (bool(use_batch_stats), act, float(drop_p), int(seed), weight is not None)
        ctx.sinks = (_sink(weight), _sink(bias))
        ctx.save_for_backward(x, weight, stats, coef)
        ctx.set_materialize_grads(False)
        y2 = torch.empty(0, dtype=y.dtype, device=y.device).set_(y.untyped_storage(), y.storage_offset(), y.shape, y.stride())
        return y, y2

    @staticmethod
    def backward(ctx, g_a, g_b):
        x, weight, stats, coef = ctx.saved_tensors
        use_batch, act, drop_p, seed, affine = ctx.cfg
        if g_a is None and g_b is None:
            return (None,) * 12
        if g_a is None:
            g_a, g_b = g_b, None
        # a skip attention's cotangent may come without its average pool's broadcast term (attention_blocks, _POOL_ADD): the
        # kernels add g_pooled[b][c] / HW on the fly instead of that block's read-modify-write pass over its g_x
        g_add = None
        for g_ in (g_a, g_b):
            if g_ is not None and _POOL_ADD:
                hit = _POOL_ADD.pop(g_.data_ptr(), None)
                if hit is not None:
                    if g_add is not None:
                        raise DcsHipError('cbn_two backward: two split pool terms for one stage output')
                    g_add = hit
        g_x, g_w, g_b_ = ops.cbn_bwd(x, g_a.contiguous(), weight, stats, coef, use_batch, act, drop_p, seed, affine,
                                     ctx.sinks, g_add=g_add, g_out2=None if g_b is None else g_b.contiguous())
        g_w = None if ctx.sinks[0] is not None else g_w
        g_b_ = None if ctx.sinks[1] is not None else g_b_
        return g_x, g_w, g_b_, None, None, None, None, None, None, None, None, None


def cbn_two(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act=ACT_NONE,
            drop_p=0.0, seed=0, stat=None):
    """(y, y') — the same values, for two different consumers (see _CbnTwoFn)."""
    y, y2 = _CbnTwoFn.apply(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act,
                            drop_p, seed, stat)
    y2._dcs_adds_pool_term = True         # its backward adds an attention block's pool term itself (_POOL_ADD): attention_blocks may split it off
    return y, y2


def channel_attention(x, fc0_r, fc0_i, fc2_r, fc2_i):
    """ComplexChannelAttention (c_network.py:53-69).  fc*_r/_i: the 1x1 conv weights."""
    w1, _ = packed_weight(fc0_r, fc0_i, None, None, False)     # [1, C, Ch, 2]
    w2, _ = packed_weight(fc2_r, fc2_i, None, None, False)     # [1, Ch, C, 2]
    ca, _, _ = ops.channel_attention(x, w1, w2)
    return ca


def spatial_attention(x, ca, conv_r, conv_i, ksize):
    """ComplexSpatialAttention (c_network.py:71-84) of z = ca * x, without materialising z."""
    pooled = ops.spatial_pool(x, ca)
    wp, bias = packed_weight(conv_r, conv_i, None, None, False)
    k = (ksize, ksize)
    return ops.cconv2d(pooled, None, wp, bias, k, (1, 1), (ksize // 2, ksize // 2), (1, 1), ACT_SIGMOID)


def attention_apply(x, ca, sa, drop_p=0.0, seed=0):
    return ops.attention_apply(x, ca, sa, drop_p, seed)


class _AttentionFn(torch.autograd.Function):
    """out = dropout(sa (.) ca (.) x): channel attention -> spatial attention -> apply, fused
    (c_network.py:208-211 / :219-222), with the hand-written backward of attention_bwd.hip."""

    @staticmethod
    def forward(ctx, x, fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p, seed):
        w1, _ = packed_weight(fc0_r, fc0_i, None, None, False)
        w2, _ = packed_weight(fc2_r, fc2_i, None, None, False)
        wsa, zero_bias = packed_weight(c1_r, c1_i, None, None, False)
        ca, pooled, hidden = ops.channel_attention(x, w1, w2)
        sp = ops.spatial_pool(x, ca)
        sa = ops.cconv2d(sp, None, wsa, zero_bias, (ksize, ksize), (1, 1), (ksize // 2, ksize // 2), (1, 1),
                         ACT_SIGMOID)
        out = ops.attention_apply(x, ca, sa, drop_p, seed)
        ctx.cfg = (ksize, float(drop_p), int(seed))
        ctx.sinks = tuple(_sink(t) for t in (fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i))
        ctx.save_for_backward(x, ca, sa, sp, pooled, hidden, w1, w2, wsa)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, ca, sa, sp, pooled, hidden, w1, w2, wsa = ctx.saved_tensors
        ksize, drop_p, seed = ctx.cfg
        g = ops.attention_bwd(x, g_out.contiguous(), ca, sa, sp, pooled, hidden, w1, w2, wsa, ksize, drop_p, seed,
                              ctx.sinks)
        gw = tuple(None if sk is not None else t for t, sk in zip(g[1:], ctx.sinks))
        return (g[0], *gw, None, None, None)


def attention_block(x, fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p=0.0, seed=0):
    return _AttentionFn.apply(x, fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p, seed)


class _CbnAttentionFn(torch.autograd.Function):
    """A decoder stage's tail as ONE node: a = CBN(x) + activation, out = dropout(sa (.) ca (.) a)
    (c_network.py:148-150, :219-222).  Same kernels as _CbnFn followed by _AttentionFn; being one node lets the
    backward hand the average pool's broadcast term (g_pooled / HW, constant per sample and channel) to the CBN
    backward kernels as an additive input instead of a read-modify-write pass over the attention's g_x."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act,
                fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p, seed, stat=None):
        w1, _ = packed_weight(fc0_r, fc0_i, None, None, False)
        w2, _ = packed_weight(fc2_r, fc2_i, None, None, False)
        wsa, zero_bias = packed_weight(c1_r, c1_i, None, None, False)
        if stat is not None and use_batch_stats and weight is not None and ops.FUSE_APPLY_POOL:
            # statistics from the conv, apply + channel pool in one pass, FC: 3 launches (5 for the chain below)
            a, stats, coef, ca, pooled, hidden = ops.cbn_channel_attention(x, weight, bias, running_mean, running_covar, eps,
                                                                           momentum, act, stat, w1, w2)
        else:
            a, stats, coef = ops.cbn(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act, 0.0, 0,
                                     stat=stat)
            ca, pooled, hidden = ops.channel_attention(a, w1, w2)
        sp = ops.spatial_pool(a, ca)
        sa = ops.cconv2d(sp, None, wsa, zero_bias, (ksize, ksize), (1, 1), (ksize // 2, ksize // 2), (1, 1), ACT_SIGMOID)
        out = ops.attention_apply(a, ca, sa, drop_p, seed)
        ctx.cfg = (bool(use_batch_stats), act, weight is not None, ksize, float(drop_p), int(seed))
        ctx.sinks = tuple(_sink(t) for t in (weight, bias, fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i))
        ctx.save_for_backward(x, weight, stats, coef, a, ca, sa, sp, pooled, hidden, w1, w2, wsa)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, weight, stats, coef, a, ca, sa, sp, pooled, hidden, w1, w2, wsa = ctx.saved_tensors
        use_batch, act, affine, ksize, drop_p, seed = ctx.cfg
        sk = ctx.sinks
        # inside TrainStep's two-stream backward the block's FC weight gradients (a launch nothing downstream waits for) are
        # queued for the side stream: they ride the next conv layer's fork (_run_pending_side)
        defer = DEFER_FC and WGRAD_SIDE is not None and ops.WGRAD_DEFER is not None and all(s_ is not None for s_ in sk[2:6])
        g = ops.attention_bwd(a, g_out.contiguous(), ca, sa, sp, pooled, hidden, w1, w2, wsa, ksize, drop_p, seed, sk[2:],
                              split_pool=True, defer_fc=defer)
        if defer:
            PENDING_SIDE.append(g[8])
        g_x, g_w, g_b = ops.cbn_bwd(x, g[0], weight, stats, coef, use_batch, act, 0.0, 0, affine, sk[:2], g_add=g[7])
        full = (g_w, g_b, *g[1:7])
        gp = tuple(None if s_ is not None else t for t, s_ in zip(full, sk))
        return (g_x, gp[0], gp[1], None, None, None, None, None, None, *gp[2:], None, None, None, None)


def cbn_attention(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act,
                  fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p=0.0, seed=0, stat=None):
    return _CbnAttentionFn.apply(x, weight, bias, running_mean, running_covar, eps, momentum, use_batch_stats, act,
                                 fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i, ksize, drop_p, seed, stat)


class _AttentionBlocksFn(torch.autograd.Function):
    """Several INDEPENDENT attention blocks (the skip attentions, c_network.py:208-211: each a function of one
    encoder output only) as one autograd node: forward = one set of five launches for all blocks
    (dcs_attention_fwd_batched), backward likewise once every block's output gradient has arrived.
    Inputs: n, ksize, then x_0..x_{n-1}, then six parameters per block (fc.0 r/i, fc.2 r/i, conv1 r/i)."""

    @staticmethod
    def forward(ctx, n, ksize, split, *args):
        xs, params = args[:n], [args[n + 6 * i:n + 6 * i + 6] for i in range(n)]
        ctx.split = tuple(split)
        w1s, w2s, wsas, biases = [], [], [], []
        for fc0_r, fc0_i, fc2_r, fc2_i, c1_r, c1_i in params:
            w1s.append(packed_weight(fc0_r, fc0_i, None, None, False)[0])
            w2s.append(packed_weight(fc2_r, fc2_i, None, None, False)[0])
            wsa, zero_bias = packed_weight(c1_r, c1_i, None, None, False)
            wsas.append(wsa)
            biases.append(zero_bias)
        outs = ops.attention_blocks_fwd(xs, w1s, w2s, wsas, biases)
        ctx.n, ctx.ksize = n, ksize
        ctx.sinks = [tuple(_sink(t) for t in pr) for pr in params]
        ctx.shapes = [(tuple(pr[0].shape), tuple(pr[2].shape)) for pr in params]
        saved = []
        for x, o, w1, w2, wsa in zip(xs, outs, w1s, w2s, wsas):
            saved += [x, o['ca'], o['sa'], o['sp'], o['pooled'], o['hidden'], w1, w2, wsa]
        ctx.save_for_backward(*saved)
        return tuple(o['y'] for o in outs)

    @staticmethod
    def backward(ctx, *g_outs):
        n, ksize = ctx.n, ctx.ksize
        t = ctx.saved_tensors
        k, pad = (ksize, ksize), (ksize // 2, ksize // 2)
        saved, gs, wbs, fcs, sps = [], [], [], [], []
        for i in range(n):
            x, ca, sa, sp, pooled, hidden, w1, w2, wsa = t[9 * i:9 * i + 9]
            saved.append(dict(x=x, ca=ca, sa=sa, pooled=pooled, hidden=hidden, w1=w1, w2=w2))
            sps.append(sp)
            g = g_outs[i]
            gs.append(torch.zeros_like(x) if g is None else g.contiguous())
            wbs.append(ops.pack_conv_weight_bwd(wsa, k, (1, 1), pad))
            sk, (s0, s2) = ctx.sinks[i], ctx.shapes[i]
            new = lambda shape: torch.empty(shape, dtype=torch.float32, device=x.device)
            fcs.append(tuple(sk[j] if sk[j] is not None else new(s0 if j < 2 else s2) for j in range(4)))
        split = tuple(sp_ and ctx.needs_input_grad[3 + i] for i, sp_ in enumerate(ctx.split))
        res = ops.attention_blocks_bwd(saved, gs, wbs, fcs, split)
        if any(split):
            # (entries of a backward pass that raised before its end-of-pass callback would survive here, and the caching
            # allocator reuses addresses: a later cotangent landing on a stale key would get a foreign pool term added.  One
            # attention_blocks node per backward pass in the networks of this repo: start from an empty table.)
            _POOL_ADD.clear()
            for (g_x, _, g_pooled), sp_ in zip(res, split):
                if sp_:
                    _POOL_ADD[g_x.data_ptr()] = g_pooled
            torch.autograd.Variable._execution_engine.queue_callback(_pool_add_consumed)
        grads_x, grads_p = [], []
        for i in range(n):
            g_x, g_pre, _ = res[i]
            sk = ctx.sinks[i]
            g_c1r, g_c1i, _, _ = ops.cconv2d_bwd_weight(sps[i], None, g_pre, (1, 2, ksize, ksize), False, k, (1, 1), pad,
                                                        outs=(sk[4], sk[5], None, None))
            full = (*fcs[i], g_c1r, g_c1i)
            grads_x.append(g_x)
            grads_p += [None if s_ is not None else g_ for g_, s_ in zip(full, sk)]
        global FLUSH_AT_NEXT_FORK
        if WGRAD_SIDE is not None and ops.WGRAD_DEFER is not None:
            ops.WGRAD_DEFER.append((sps, [r[1] for r in res]))
            FLUSH_AT_NEXT_FORK = True
        return (None, None, None, *grads_x, *grads_p)


SPLIT_SKIP_POOL = os.environ.get('DCS_SPLIT_SKIP_POOL', '1') != '0'     # 0: the skip attentions add their pool term into g_x themselves (A/B)
_POOL_ADD = {}             # data_ptr of a block's g_x -> its g_pooled [B,C,2], between _AttentionBlocksFn.backward and _CbnTwoFn.backward


def _pool_add_consumed():
    """End of the backward pass: every split-off pool term must have been picked up by its consumer (a silent miss would be a
    wrong gradient)."""
    if _POOL_ADD:
        _POOL_ADD.clear()
        raise DcsHipError('attention_blocks backward: a split average-pool term was not consumed by a cbn_two backward')


def attention_blocks(xs, params, ksize):
    """params[i] = (fc0_r, fc0_i, fc2_r, fc2_i, conv1_r, conv1_i) of block i; returns the tuple of block outputs.
    An input that is the second output of cbn_two (marked by it) gets its cotangent WITHOUT the average pool's broadcast term:
    cbn_two's backward adds it on the fly (one read-modify-write pass over every encoder output's gradient less)."""
    flat = [p for pr in params for p in pr]
    split = tuple(bool(SPLIT_SKIP_POOL and getattr(x, '_dcs_adds_pool_term', False) and torch.is_grad_enabled() and x.requires_grad)
                  for x in xs)
    return _AttentionBlocksFn.apply(len(xs), ksize, split, *xs, *flat)


# ---- complex-tensor conveniences for the drop-in layer surface ---------------------------------

def _as_real(z):
    if z.dtype != torch.complex64:
        raise DcsHipError(f'expected complex64, got {z.dtype}')
    return torch.view_as_real(z.contiguous() if not _dense(z) else z)


def _dense(z):
    # dense in SOME permutation (e.g. channels_last): element-wise kernels may run in place order
    return z.is_contiguous() or (z.dim() == 4 and z.is_contiguous(memory_format=torch.channels_last))


def _elementwise(z, act):
    if z.dim() == 4 and z.is_contiguous(memory_format=torch.channels_last) and not z.is_contiguous():
        x = to_nhwc(z)
        return from_nhwc(ops.complex_act(x, act))
    x = torch.view_as_real(z.contiguous())
    return torch.view_as_complex(ops.complex_act(x, act))


def complex_relu(z):
    return _elementwise(z, ACT_RELU)


def complex_lrelu(z):
    return _elementwise(z, ACT_LRELU)


def complex_sigmoid(z):
    return _elementwise(z, ACT_SIGMOID)


def complex_upsample(z, scale_factor):
    sf = (scale_factor, scale_factor) if isinstance(scale_factor, (int, float)) else tuple(scale_factor)
    up = (int(sf[0]), int(sf[1]))
    if up != tuple(sf) or min(up) < 1:
        raise DcsHipError(f'complex_upsample: integer scale factors only, got {scale_factor}')
    return from_nhwc(ops.complex_upsample(to_nhwc(z), up))


def complex_linear(z, w_r, w_i, b_r, b_i):
    """ComplexLinear (complexPyTorch apply_complex over two nn.Linear) as a 1x1 complex conv."""
    in_f = z.shape[-1]
    lead = z.shape[:-1]
    x = torch.view_as_real(z.reshape(-1, in_f).contiguous())          # [N, in, 2]
    N = x.shape[0]
    W = 16 if N % 16 == 0 else 1
    x = x.view(1, N // W, W, in_f, 2)
    y = cconv2d(x, None, w_r, w_i, b_r, b_i, False, (1, 1), (1, 1), (0, 0))
    return torch.view_as_complex(y.view(N, -1, 2)).view(*lead, -1)


class _LstmRecFn(torch.autograd.Function):
    """Recurrent half of one LSTM layer (dcs_lstm_layer_fwd / _bwd).  gx: [sets, seqs, S, 2, 4H]
    pre-activations of the input projection; w_hh: [sets, 2, 4H, H]."""

    @staticmethod
    def forward(ctx, gx, w_hh):
        n_sets, seqs, S, _, G4 = gx.shape
        need = gx.requires_grad or w_hh.requires_grad
        out, gates, c = ops.lstm_layer(gx, w_hh, n_sets, seqs, S, (seqs * S * 2 * G4, S * 2 * G4, 2 * G4), need)
        ctx.dims = (n_sets, seqs, S)
        if need:
            ctx.save_for_backward(out, gates, c, w_hh)
        return out

    @staticmethod
    def backward(ctx, g_out):
        out, gates, c, w_hh = ctx.saved_tensors
        n_sets, seqs, S = ctx.dims
        H = w_hh.shape[-1]
        g_pre = ops.lstm_layer_bwd(g_out.contiguous(), gates, c, w_hh, n_sets, seqs, S)
        g_pre5 = g_pre.view(n_sets, seqs, S, 2, 4 * H)
        g_whh = None
        if ctx.needs_input_grad[1]:
            o = out.view(n_sets, seqs, S, 2, H)
            h_prev = torch.zeros_like(o)
            h_prev[:, :, 1:, 0] = o[:, :, :-1, 0]           # forward direction: h_{t-1}
            h_prev[:, :, :-1, 1] = o[:, :, 1:, 1]           # reverse direction: h_{t+1}
            g_whh = torch.einsum('sntdj,sntdk->sdjk', g_pre5, h_prev)
        return g_pre5, g_whh


def _project(inp, w_ih, shared, per_set=False, train=True):
    """Input projection of one LSTM layer for both parameter sets, all time steps at once (nn.LSTM's x_t W_ih^T, c_network.py:24-31):
    shared: inp [M, in] read by both sets -> [M, 2 * 8H] against the stacked weight [2 * 8H, in] (per_set: -> [2, M, 8H], the
    same rows read once per set); else inp [2, M, in] -> [2, M, 8H].  In-tree fp32 MFMA kernel (dcs_gemm_f32); shapes it does
    not take go to the library GEMM."""
    G8, K = w_ih.shape[1], w_ih.shape[2]
    M = inp.shape[-2]
    if not (ops.gemm_ok(M, G8, K, 2, train) and inp.is_contiguous() and w_ih.is_contiguous()):
        if shared and not per_set:
            return torch.mm(inp, w_ih.reshape(2 * G8, -1).t())
        return torch.matmul(inp, w_ih.transpose(1, 2))
    if shared and not per_set:
        gx = torch.empty((M, 2 * G8), dtype=torch.float32, device=inp.device)
        return ops.gemm_f32(inp, w_ih, gx, M, 2 * G8, K, K, K, 2 * G8, True)
    gx = torch.empty((2, M, G8), dtype=torch.float32, device=inp.device)
    return ops.gemm_f32(inp, w_ih, gx, M, G8, K, K, K, G8, True, nbatch=2, a_batch=0 if shared else M * K, b_batch=G8 * K,
                        c_batch=M * G8)


def _project_bwd(g_gx, w_ih, shared):
    """Data gradient of _project: g_gx [2, M, 8H], w_ih [2, 8H, in] -> shared: sum over the sets [M, in] (the two sets are two
    K segments of one launch); else [2, M, in]."""
    G8, K = w_ih.shape[1], w_ih.shape[2]
    M = g_gx.shape[1]
    if not (ops.gemm_ok(M, K, G8, 2, True) and g_gx.is_contiguous() and w_ih.is_contiguous()):
        if shared:
            g_inp = torch.mm(g_gx[0], w_ih[0])
            return g_inp.addmm_(g_gx[1], w_ih[1])              # (in place: torch.addmm copies its addend first)
        return torch.bmm(g_gx, w_ih)
    if shared:
        g_inp = torch.empty((M, K), dtype=torch.float32, device=g_gx.device)
        return ops.gemm_f32(g_gx, w_ih, g_inp, M, K, G8, G8, K, K, False, nseg=2, a_seg=M * G8, b_seg=G8 * K)
    g_inp = torch.empty((2, M, K), dtype=torch.float32, device=g_gx.device)
    return ops.gemm_f32(g_gx, w_ih, g_inp, M, K, G8, G8, K, K, False, nbatch=2, a_batch=M * G8, b_batch=G8 * K, c_batch=M * K)


class _ProjectFn(torch.autograd.Function):
    """_project(per_set=True) / _project_bwd under autograd, for LSTM parameters that are not views of a dp.FlatBucket (the
    weight gradient then goes through torch: not the captured step's path)."""

    @staticmethod
    def forward(ctx, inp, w_ih):
        ctx.save_for_backward(inp, w_ih)
        return _project(inp, w_ih, shared=inp.dim() == 2, per_set=True)

    @staticmethod
    def backward(ctx, g):
        inp, w_ih = ctx.saved_tensors
        g = g.contiguous()
        g_inp = _project_bwd(g, w_ih, shared=inp.dim() == 2) if ctx.needs_input_grad[0] else None
        g_w = torch.matmul(g.transpose(1, 2), inp) if ctx.needs_input_grad[1] else None
        return g_inp, g_w


class _LstmLayerFn(torch.autograd.Function):
    """One bidirectional layer of both LSTM parameter sets over STACKED parameters that are views of a
    dp.FlatBucket (st[kind] = (value view, gradient view); weight_ih [2, 8H, in], weight_hh [2, 2, 4H, H],
    biases [2, 8H]): no per-parameter cat / stack in forward, and backward accumulates the four parameter
    gradients straight into the bucket (returns None to autograd) — ~12 launches instead of ~60 per layer."""

    @staticmethod
    def forward(ctx, inp, anchor, st, B2, S):
        w_ih, w_hh = st['weight_ih'][0], st['weight_hh'][0]
        need = inp.requires_grad or anchor.requires_grad
        # the gate biases (b_ih + b_hh, [2 sets, 8H] = [set][dir][4H]) are added inside the recurrence kernel: no add, no
        # bias-broadcast pass over gx
        ctx.planes = 0
        if inp.is_complex():
            # first layer, straight from the complex latent [B, S, in]: rows {re | im} of the stacking x2 of complex_lstm are
            # read in place by the GEMM (and by the weight gradient), the data gradient is written back interleaved
            zr = torch.view_as_real(inp)
            G8, K = w_ih.shape[1], w_ih.shape[2]
            ctx.planes = R0 = zr.shape[0] * zr.shape[1]
            gx = torch.empty((2 * R0, 2 * G8), dtype=torch.float32, device=inp.device)
            ops.gemm_f32(zr, w_ih, gx, 2 * R0, 2 * G8, K, K, K, 2 * G8, True, a_planes=R0)
            inp = zr
            G4 = G8 // 2
            strides = (G8, S * 2 * G8, 2 * G8)
        elif inp.dim() == 2:
            # first layer: both parameter sets read the SAME rows -> one GEMM against the stacked [2*8H, in] weight; the
            # recurrence takes gx by strides (set stride 8H inside a row), so nothing is expanded or copied
            G8 = w_ih.shape[1]
            gx = _project(inp, w_ih, shared=True)                                               # [(n t), (set, dir*4H)]
            G4 = G8 // 2
            strides = (G8, S * 2 * G8, 2 * G8)
        else:
            gx = _project(inp, w_ih, shared=False)                                              # (set, n*t, dir*4H)
            G4 = gx.shape[-1] // 2
            strides = (B2 * S * 2 * G4, S * 2 * G4, 2 * G4)
        out, gates, c, hprev = ops.lstm_layer(gx, w_hh, 2, B2, S, strides, need, True,
                                              bias=(st['bias_ih'][0], st['bias_hh'][0]))
        ctx.st, ctx.dims = st, (B2, S)
        if need:
            ctx.save_for_backward(inp, hprev, gates, c)
        return out

    @staticmethod
    def backward(ctx, g_out):
        global sink_hits
        inp, hprev, gates, c = ctx.saved_tensors
        st, (B2, S) = ctx.st, ctx.dims
        w_ih, w_hh = st['weight_ih'][0], st['weight_hh'][0]
        H = w_hh.shape[-1]
        NT = B2 * S
        g_pre, b_part = ops.lstm_layer_bwd(g_out.contiguous(), gates, c, w_hh, 2, B2, S, True)
        g_gx = g_pre.view(2, NT, 8 * H)
        # (queued for the side stream like the attention blocks' FC weight gradients — PENDING_SIDE — the two MFMA products and the
        # reduction of a layer, ~40 us, run beside the encoder's backward and the step is 0.012 ms SLOWER: profiles/r04_side_stream_ab.txt)
        _LstmLayerFn._param_grads(st, inp, hprev, g_pre, b_part, g_gx, B2, S, H, NT, planes=ctx.planes)
        if ctx.planes:
            g_z = None
            if ctx.needs_input_grad[0]:
                G8, K = w_ih.shape[1], w_ih.shape[2]
                gz = torch.empty(inp.shape, dtype=torch.float32, device=inp.device)               # [B, S, in, 2]
                ops.gemm_f32(g_gx, w_ih, gz, NT, K, G8, G8, K, K, False, nseg=2, a_seg=NT * G8, b_seg=G8 * K, c_planes=ctx.planes)
                g_z = torch.view_as_complex(gz)
            return g_z, None, None, None, None
        g_inp = _project_bwd(g_gx, w_ih, shared=inp.dim() == 2) if ctx.needs_input_grad[0] else None
        return g_inp, None, None, None, None

    @staticmethod
    def _param_grads(st, inp, hprev, g_pre, b_part, g_gx, B2, S, H, NT, planes=0):
        global sink_hits
        # W_hh gradient per direction d: g_pre[set, (n t), d, :]^T h_prev[set, (n t), d, :] — strided views, no copies
        # (n t) is cut into CK chunks that ride the batch axis (rocBLAS runs a [4H x H] output with K = 4096 on 16
        # workgroups otherwise), summed afterwards in a fixed order
        CK = 16 if NT % 32 == 0 else 1
        if H % 64 == 0 and NT % (8 * CK) == 0:
            part = ops.lstm_whh_grad(g_pre, hprev, NT, CK, H)           # one MFMA launch (rocBLAS: four ~26 us batched GEMMs)
        else:
            R = NT // CK
            part = torch.empty((2, 2 * CK, 4 * H, H), dtype=g_pre.dtype, device=g_pre.device)
            for d in range(2):
                a = g_pre.as_strided((2 * CK, R, 4 * H), (R * 8 * H, 8 * H, 1), g_pre.storage_offset() + d * 4 * H)
                h = hprev.as_strided((2 * CK, R, H), (R * 2 * H, 2 * H, 1), hprev.storage_offset() + d * H)
                torch.bmm(a.transpose(1, 2), h, out=part[d])
        b_part = b_part.contiguous()
        g_wih = st['weight_ih'][1]
        sink_hits += 16
        # W_ih gradient g_gx[s]^T inp over the (n t) rows: chunked A^T B on the MFMA pipe (rocBLAS: 25 us per 512 x 2048 x 256);
        # its chunk sum rides the launch that sums the W_hh products and the per-sequence bias sums into their gradient views
        # (autograd's spelling: two reductions and three adds — and one more reduction here)
        ih = None
        if planes:                                                      # inp = the complex latent as float [B, S, in, 2] (checked by the caller)
            ih = ops.atb_chunks_acc_planes(g_gx, inp, g_wih, 2, 8 * H, 8 * H, inp.shape[-2], planes, 8, reduce=False)
        else:
            n_in = inp.shape[-1]
            if (g_wih.is_contiguous() and n_in % 64 == 0 and (8 * H) % 32 == 0 and NT % (8 * CK) == 0 and inp.is_contiguous()):
                ih = ops.atb_chunks_acc(g_gx, inp, g_wih, 2, 8 * H, n_in, 8 * H, n_in, NT, CK, b_shared=inp.dim() == 2, reduce=False)
        ops.lstm_param_grads(part, b_part, st['weight_hh'][1], st['bias_ih'][1], st['bias_hh'][1], CK, B2, H,
                             ih=None if ih is None else (ih[0], ih[1], g_wih))
        if ih is None:                                                  # shapes the MFMA kernel does not take
            if inp.dim() == 2:
                for s_ in range(2):
                    g_wih[s_].addmm_(g_gx[s_].t(), inp)
            else:
                torch.baddbmm(g_wih, g_gx.transpose(1, 2), inp, out=g_wih)  # accumulate in place


class _LstmCombineFn(torch.autograd.Function):
    """real = L_r(x_r) - L_i(x_i), imag = L_r(x_i) + L_i(x_r) (c_network.py:43-46) from the stacked outputs
    o[set, {re rows | im rows}]: three launches forward, three backward (autograd's slice / complex / sub graph
    costs ~20 fill / copy / add launches in backward); one HIP launch each way."""

    @staticmethod
    def forward(ctx, o, B):
        return ops.lstm_combine(o.contiguous(), B)

    @staticmethod
    def backward(ctx, g):
        g2 = torch.view_as_real(g if g.is_contiguous() else g.contiguous())      # [B, S, 2H, 2]
        return ops.lstm_combine_bwd(g2), None


def _stacked_lstm(real_lstm):
    """Stacked parameter views registered by dp.FlatBucket, if they still alias the live parameters and their
    gradients (and autograd is going to need them)."""
    st = getattr(real_lstm, '_dcs_stacked', None)
    if not st:
        return None
    p = real_lstm.weight_ih_l0
    v, g = st[0]['weight_ih']
    if p.data_ptr() != v.data_ptr() or p.grad is None or p.grad.data_ptr() != g.data_ptr():
        return None
    return st


_LSTM_EVAL_OPERANDS = weakref.WeakKeyDictionary()      # real nn.LSTM -> {layer: (key, parameter refs, stacks)}


def _lstm_layer_operands(sets, layer):
    """Stacked (w_ih [2, 8H, in], bias [2, 8H], w_hh [2, 2, 4H, H]) of one layer of both nn.LSTM containers.  Without
    autograd (inference) the stacks are cached per parameter identity + version: re-making them costs ~12 cat / add
    launches per layer and pass."""
    names = [f'_l{layer}', f'_l{layer}_reverse']
    params = [getattr(m, k + n) for m in sets for n in names for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    cacheable = not (torch.is_grad_enabled() and any(p.requires_grad for p in params))
    if cacheable:
        key = (_param_generation, tuple((p.data_ptr(), p._version) for p in params))
        hit = _LSTM_EVAL_OPERANDS.get(sets[0], {}).get(layer)
        if hit is not None and hit[0] == key and all(r() is p for r, p in zip(hit[1], params)):
            return hit[2]
    w_ih = torch.stack([torch.cat([getattr(m, 'weight_ih' + n) for n in names]) for m in sets])      # [2, 8H, in]
    bias = torch.stack([torch.cat([getattr(m, 'bias_ih' + n) + getattr(m, 'bias_hh' + n) for n in names])
                        for m in sets])                                                                # [2, 8H]
    w_hh = torch.stack([torch.stack([getattr(m, 'weight_hh' + n) for n in names]) for m in sets]).contiguous()   # [2,2,4H,H]
    if cacheable:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return w_ih, bias, w_hh             # tensors made during capture live in the graph's pool: do not keep them
        _LSTM_EVAL_OPERANDS.setdefault(sets[0], {})[layer] = (
            key, tuple(weakref.ref(p) for p in params), (w_ih.detach(), bias.detach(), w_hh.detach()))
    return w_ih, bias, w_hh


def complex_lstm(z, real_lstm, imag_lstm):
    """ComplexLSTM.forward (c_network.py:33-47) for two bidirectional batch_first nn.LSTM
    parameter containers.  Per layer: one input-projection GEMM for all time steps
    (dcs_gemm_f32) + one persistent HIP launch for the recurrence of all 4 passes x 2 directions
    (hand-written BPTT)."""
    if not (real_lstm.bidirectional and real_lstm.batch_first and real_lstm.hidden_size == 64):
        raise DcsHipError('complex_lstm: the HIP path implements the reference geometry '
                          '(bidirectional, batch_first, hidden 64: c_network.py:118-123)')
    B, S, I = z.shape
    sets = (real_lstm, imag_lstm)
    # rows 0..B-1: real parts, rows B..2B-1: imaginary parts; both weight sets see the same input
    # (2-D from the start: indexing a [1, rows, I] tensor costs a zero fill and a copy in the backward of the select)
    inp = None
    stacked = _stacked_lstm(real_lstm) if torch.is_grad_enabled() else None
    # the stacking read in place (no x2, no permute in the backward) where the kernels take the shape: whole 8-row chunks of the
    # weight gradient inside each part (dcs_atb_chunks_strided), a launch the in-tree GEMM is meant for
    in_place = (stacked is not None and z.is_contiguous() and (B * S) % 64 == 0 and I % 64 == 0
                and ops.gemm_ok(2 * B * S, 16 * real_lstm.hidden_size, I, 1, True) and ops.gemm_ok(2 * B * S, I, 8 * real_lstm.hidden_size, 2, True)
                and stacked[0]['weight_ih'][0].is_contiguous() and stacked[0]['weight_ih'][1].is_contiguous())
    x2 = None if in_place else torch.view_as_real(z).permute(3, 0, 1, 2).reshape(2 * B * S, I)
    for layer in range(real_lstm.num_layers):
        if stacked is not None:
            out = _LstmLayerFn.apply((z if in_place else x2) if layer == 0 else inp, real_lstm.weight_ih_l0, stacked[layer], 2 * B, S)
            inp = out.view(2, 2 * B * S, -1)
            continue
        w_ih, bias, w_hh = _lstm_layer_operands(sets, layer)
        if not (torch.is_grad_enabled() and ((x2 if layer == 0 else inp).requires_grad or bias.requires_grad or w_ih.requires_grad)):
            # inference: bare projection, the gate biases are added inside the recurrence kernel (baddbmm's bias broadcast was
            # a 27 us pass over gx per layer at [16,256,2000])
            if layer == 0:
                # both sets read the same rows: one launch against the stacked weight, the recurrence takes gx by strides
                G8 = w_ih.shape[1]
                gx = _project(x2, w_ih.contiguous(), shared=True, train=False)
                strides = (G8, S * 2 * G8, 2 * G8)
            else:
                gx = _project(inp, w_ih.contiguous(), shared=False, train=False)
                strides = (2 * B * S * gx.shape[-1], S * gx.shape[-1], gx.shape[-1])
            out = ops.lstm_layer(gx, w_hh.contiguous(), 2, 2 * B, S, strides, False, bias=(bias.contiguous(), None))[0]
            inp = out.view(2, 2 * B * S, -1)
            continue
        gx = _ProjectFn.apply(x2 if layer == 0 else inp, w_ih) + bias.unsqueeze(1)             # (set, n*t, dir*4H)
        out = _LstmRecFn.apply(gx.view(2, 2 * B, S, 2, -1), w_hh)                 # [2*2B, S, 2H]
        inp = out.view(2, 2 * B * S, -1)
    o = inp.view(2, 2 * B, S, -1)
    if torch.is_grad_enabled() and o.requires_grad:
        return _LstmCombineFn.apply(o, B)
    # rows: o[0] = real_lstm(re | im), o[1] = imag_lstm(re | im)
    return ops.lstm_combine(o.contiguous(), B)


class _TapSumFn(torch.autograd.Function):
    """y = tapsum(z) + bias; the two real bias scalars of the Cout = 1 layer enter and leave through the kernels (their
    gradients: one complex sum of the cotangent, written straight into the gradient sinks when there are any)."""

    @staticmethod
    def forward(ctx, z, ksize, up, pad, b_r, b_i):
        ctx.cfg = (tuple(z.shape), tuple(ksize), tuple(up), tuple(pad))
        ctx.bias = b_r is not None
        ctx.sinks = (_sink(b_r), _sink(b_i)) if ctx.bias else (None, None)
        return ops.tapsum(z, ksize, up, pad, bias=(b_r, b_i) if ctx.bias else None)

    @staticmethod
    def backward(ctx, g):
        shape, ksize, up, pad = ctx.cfg
        g = g.contiguous()
        if not ctx.bias:
            return ops.tapsum(shape, ksize, up, pad, backward=True, grad=g), None, None, None, None, None
        new = lambda: torch.empty(1, dtype=torch.float32, device=g.device)
        dst = tuple(s_ if s_ is not None else new() for s_ in ctx.sinks)
        gz = ops.tapsum(shape, ksize, up, pad, backward=True, grad=g, bias_grad=dst)
        gb = tuple(None if s_ is not None else d for d, s_ in zip(dst, ctx.sinks))
        return gz, None, None, None, gb[0], gb[1]


class _TapRowsConvFn(torch.autograd.Function):
    """conv1x1(cat(x1, x2): Cin -> ct tap channels) whose weight rows are the flipped taps of a [Cin,1,kh,kw]
    ConvTranspose2d weight: packed straight from the parameters (dcs_pack_tap_rows: no flip / pad / reshape kernels, and
    the pack is part of the pack plan), weight gradient scattered straight back (dcs_tap_rows_wgrad_scatter)."""

    @staticmethod
    def forward(ctx, x1, x2, w_r, w_i, ct):
        wp, bias = packed_weight(w_r, w_i, None, None, False, (1, 1), tap_rows=ct)
        z = ops.cconv2d(x1, x2, wp, bias, (1, 1), (1, 1), (0, 0), (1, 1), ACT_NONE)
        ctx.ct, ctx.w_shape = ct, tuple(w_r.shape)
        ctx.sinks = (_sink(w_r), _sink(w_i))
        ctx.save_for_backward(x1, x2, wp)
        return z

    @staticmethod
    def backward(ctx, gz):
        x1, x2, wp = ctx.saved_tensors
        gz = gz.contiguous()
        C1 = x1.shape[3]
        Cin = C1 + (x2.shape[3] if x2 is not None else 0)
        gx1 = gx2 = gw_r = gw_i = None
        if ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            gt_r, gt_i, _, _ = ops.cconv2d_bwd_weight(x1, x2, gz, (ctx.ct, Cin, 1, 1), False, (1, 1), (1, 1), (0, 0))
            g = ops.tap_rows_scatter(gt_r, gt_i, ctx.w_shape, ctx.sinks)
            if ctx.sinks[0] is None or ctx.sinks[1] is None:
                gw_r, gw_i = g
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx1, gx2 = ops.cconv2d_bwd_data(gz, ops.pack_conv_weight_bwd(wp, (1, 1)), (x1.shape[1], x1.shape[2], Cin),
                                            (1, 1), (1, 1), (0, 0), (1, 1), C1)
        return gx1, gx2, gw_r, gw_i, None


class _Up2SingleFn(torch.autograd.Function):
    """The last decoder stage (16 -> 1 channels, 3x3, 2x2 upsample) with its forward in ONE kernel (conv_up1.hip: no
    tap-channel intermediate) and the factored backward of _TapSumFn + _TapRowsConvFn (which needs the tap-channel
    cotangent anyway)."""

    @staticmethod
    def forward(ctx, x1, x2, w_r, w_i, b_r, b_i, ct):
        wt, _ = packed_weight(w_r, w_i, None, None, False, (1, 1), tap_rows=ct)
        ctx.ct, ctx.w_shape, ctx.bias = ct, tuple(w_r.shape), b_r is not None
        ctx.sinks = (_sink(w_r), _sink(w_i), _sink(b_r), _sink(b_i))
        ctx.save_for_backward(x1, x2, wt)
        return ops.cconv_up2_single(x1, x2, wt, b_r, b_i)

    @staticmethod
    def backward(ctx, g):
        x1, x2, wt = ctx.saved_tensors
        B, Hs, Ws, C1, _ = x1.shape
        C2 = x2.shape[3] if x2 is not None else 0
        Cin = C1 + C2
        sk = ctx.sinks
        gb = (None, None)
        dst = None
        if ctx.bias:
            new = lambda: torch.empty(1, dtype=torch.float32, device=g.device)
            dst = tuple(s_ if s_ is not None else new() for s_ in sk[2:])
            gb = tuple(None if s_ is not None else d for d, s_ in zip(dst, sk[2:]))
        g = g.contiguous()
        gx1 = gx2 = gw_r = gw_i = None
        want_w = ctx.needs_input_grad[2] or ctx.needs_input_grad[3] or ctx.bias
        want_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        if C1 % 4 == 0 and C2 % 4 == 0:
            # Round 5: both gradients straight from the cotangent (conv_up1.hip: tap sums in registers) — no tap-channel tensor, two
            # kernels + a 5-workgroup reduce instead of six launches.  Train step: the weight / bias gradient goes to the
            # weight-gradient side stream like every other layer's (forked in front of the data gradient, issued behind it)
            side = WGRAD_SIDE if (want_w and ops.WGRAD_DEFER is not None and g.is_cuda and sk[0] is not None and sk[1] is not None) else None
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
            elif want_w:
                gw = ops.cconv_up2_single_bwd_weight(g, x1, x2, ctx.w_shape, sk[:2], dst)
                if sk[0] is None or sk[1] is None:
                    gw_r, gw_i = gw
            if want_x:
                gx1, gx2 = ops.cconv_up2_single_bwd_data(g, wt, C1, C2, x1.dtype)
            if side is not None:
                with torch.cuda.stream(side):
                    ops.cconv_up2_single_bwd_weight(g, x1, x2, ctx.w_shape, sk[:2], dst)
                ops.WGRAD_DEFER.append((g, x1, x2))               # alive until the join
            return gx1, gx2, gw_r, gw_i, gb[0], gb[1], None
        # other channel splits: the factored form (tap sums as a tensor, then the 1x1 tap conv's two gradients)
        gz = ops.tapsum((B, Hs, Ws, ctx.ct, 2), (3, 3), (2, 2), (1, 1), backward=True, grad=g, bias_grad=dst,
                        out_dtype=x1.dtype)                  # (bf16 where the activations are stored in bf16)
        want_w = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        if want_w:
            gt_r, gt_i, _, _ = ops.cconv2d_bwd_weight(x1, x2, gz, (ctx.ct, Cin, 1, 1), False, (1, 1), (1, 1), (0, 0))
            gw = ops.tap_rows_scatter(gt_r, gt_i, ctx.w_shape, sk[:2])
            if sk[0] is None or sk[1] is None:
                gw_r, gw_i = gw
        if want_x:
            gx1, gx2 = ops.cconv2d_bwd_data(gz, ops.pack_conv_weight_bwd(wt, (1, 1)), (Hs, Ws, Cin), (1, 1), (1, 1), (0, 0),
                                            (1, 1), C1)
        return gx1, gx2, gw_r, gw_i, gb[0], gb[1], None


def cconv_single_output(x1, x2, w_r, w_i, b_r, b_i, ksize, pad, up):
    """Stride-1 ComplexConvTranspose2d with ONE output channel over upsample(cat(x1, x2)) (the last decoder
    stage).  Cout = 1 would leave the MFMA tile's N dimension 2 wide, so the stage is factored as
        y = tapsum( conv1x1(cat(x1, x2): Cin -> kh*kw tap channels) ) + bias
    where the 1x1 weight row `tap` is the (flipped) transposed-conv kernel at that tap: a full-lane MFMA GEMM
    at SOURCE resolution plus an HBM-bound 9-load gather (elementwise.hip).  w_*: [Cin, 1, kh, kw]."""
    kh, kw = ksize
    ct = (kh * kw + 7) // 8 * 8                                # tap channels, padded for the MFMA N tile
    c2 = 0 if x2 is None else x2.shape[3]
    if ((kh, kw) == (3, 3) and tuple(up) == (2, 2) and tuple(pad) == (1, 1) and x1.shape[3] + c2 == 16 and
            not (x1.shape[3] & 1) and not (c2 & 1)):           # the configured dec6: forward in one kernel
        return _Up2SingleFn.apply(x1, x2, w_r, w_i, b_r, b_i, ct)
    z = _TapRowsConvFn.apply(x1, x2, w_r, w_i, ct)
    # stride-1 transposed conv == correlation with padding k-1-p (already in `pad`)
    return _TapSumFn.apply(z, (kh, kw), tuple(up), tuple(pad), b_r, b_i)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop_p, seed):
        ctx.cfg = (float(drop_p), int(seed))
        return ops.dropout(x, drop_p, seed)

    @staticmethod
    def backward(ctx, g):
        return ops.dropout(g.contiguous(), *ctx.cfg), None, None


def dropout(x, drop_p, seed):
    return _DropoutFn.apply(x, drop_p, seed)


class _BoundCrmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, M, eps):
        ctx.eps = eps
        ctx.save_for_backward(M)
        return ops.bound_crm(M, eps)

    @staticmethod
    def backward(ctx, g):
        (M,) = ctx.saved_tensors
        return ops.bound_mask_apply_bwd(None, M, g.contiguous(), None, None, ctx.eps), None


def bound_crm(M, eps=10e-7):
    """bound_cRM on an interleaved float view [..., 2]."""
    return _BoundCrmFn.apply(M, eps)


class _BoundMaskApplyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Y, M_in, eps):
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, M_in)
        return ops.bound_mask_apply(Y, M_in, eps)

    @staticmethod
    def backward(ctx, gM, gN, gS):
        Y, M_in = ctx.saved_tensors
        c = lambda t: None if t is None else t.contiguous()
        if gM is None and gN is None and gS is None:
            return None, None, None
        return None, ops.bound_mask_apply_bwd(Y, M_in, c(gM), c(gN), c(gS), ctx.eps), None


class _BoundMaskApplyPairFn(torch.autograd.Function):
    """bound + multiply + subtract with the two estimates stacked in one [2, B, F, T, 2] output (ops.bound_mask_apply_pair)."""

    @staticmethod
    def forward(ctx, Y, M_in, eps):
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, M_in)
        return ops.bound_mask_apply_pair(Y, M_in, eps)

    @staticmethod
    def backward(ctx, gM, gNS):
        Y, M_in = ctx.saved_tensors
        if gM is None and gNS is None:
            return None, None, None
        gM = None if gM is None else gM.contiguous()
        gN = gS = None
        if gNS is not None:
            gNS = gNS.contiguous()
            gN, gS = gNS[0], gNS[1]
        return None, ops.bound_mask_apply_bwd(Y, M_in, gM, gN, gS, ctx.eps), None


class _Bound2MaskApplyPairFn(torch.autograd.Function):
    """Both bound_cRM applications (the network's own, c_network.py:225, and the step function's, network_functions.py:240)
    + multiply + subtract over the network's RAW last-stage output in one kernel each way: returns (M, NS) — the
    twice-bounded mask and the stacked estimates [Y (.) M ; Y - Y (.) M]."""

    @staticmethod
    def forward(ctx, Y, D_raw, eps, drop_p, seed):
        ctx.eps, ctx.drop = eps, (float(drop_p), int(seed))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, D_raw)
        _, M, NS = ops.bound2_mask_apply(Y, D_raw, eps, pair=True, drop_p=drop_p, seed=seed)
        return M, NS

    @staticmethod
    def backward(ctx, gM, gNS):
        Y, D_raw = ctx.saved_tensors
        if gM is None and gNS is None:
            return None, None, None, None, None
        gM = None if gM is None else gM.contiguous()
        gN = gS = None
        if gNS is not None:
            gNS = gNS.contiguous()
            gN, gS = gNS[0], gNS[1]
        return None, ops.bound2_mask_apply_bwd(Y, D_raw, None, gM, gN, gS, ctx.eps, *ctx.drop), None, None, None


def bound2_mask_apply_pair_complex(Y, D_raw, eps=10e-7, drop=(0.0, 0)):
    """(bound_cRM(bound_cRM(D)), [Y (.) M ; Y - Y (.) M]) from the network's UNBOUNDED output (C_NETWORK.forward(x,
    bound=False)); differentiable w.r.t. D_raw.  drop = (p, seed): D = dropout(D_raw) with dcs_dropout_fwd's mask, applied
    inside the same kernels (C_NETWORK.forward(bound=False) in training hands its last dropout over instead of running it)."""
    y = torch.view_as_real(Y.contiguous())
    d = torch.view_as_real(D_raw.contiguous())
    M, NS = _Bound2MaskApplyPairFn.apply(y, d, eps, float(drop[0]), int(drop[1]))
    return torch.view_as_complex(M), torch.view_as_complex(NS)


def bound2_mask_apply_complex(Y, D_raw, eps=10e-7):
    """Inference form: (M, N_hat, S_hat) from the network's unbounded output, one kernel (no autograd)."""
    y = torch.view_as_real(Y.contiguous())
    d = torch.view_as_real(D_raw.contiguous())
    _, M, N, S = ops.bound2_mask_apply(y, d, eps)
    return torch.view_as_complex(M), torch.view_as_complex(N), torch.view_as_complex(S)


class _PolarFramesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, Fp, eps):
        ctx.cfg = (Fp, eps)
        ctx.save_for_backward(z)
        return ops.polar_frames(z, Fp, eps)

    @staticmethod
    def backward(ctx, g):
        (z,) = ctx.saved_tensors
        return ops.polar_frames(z, ctx.cfg[0], ctx.cfg[1], grad=g.contiguous()), None, None


def polar_frames_complex(z, pad_bins=1, eps=10e-7):
    """|z| (cos phi + j sin phi), phi = atan2(z_i, z_r + eps), with `pad_bins` zero bins appended on the frequency
    axis — the spectrum mag_phase_2_wave hands to the iSTFT (network_functions.py:140-145) — returned FRAME-MAJOR:
    complex [B, T, F + pad_bins] for z complex [B, F, T]."""
    zr = torch.view_as_real(z.contiguous())
    return torch.view_as_complex(_PolarFramesFn.apply(zr, z.shape[1] + pad_bins, eps))


class _IstftOlaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, frames, window, inv_env, hop, scale):
        ctx.cfg = (tuple(frames.shape), hop, scale)
        ctx.save_for_backward(window, inv_env)
        return ops.istft_ola(frames.contiguous(), window, inv_env, hop, scale)

    @staticmethod
    def backward(ctx, g):
        window, inv_env = ctx.saved_tensors
        shape, hop, scale = ctx.cfg
        return ops.istft_ola(shape, window, inv_env, hop, scale, grad=g.contiguous()), None, None, None, None


def istft_ola(frames, window, inv_env, hop, scale=1.0):
    """Windowed overlap-add, envelope division and centre trim of torch.istft over irfft frames [B,T,n_fft]."""
    return _IstftOlaFn.apply(frames, window, inv_env, hop, scale)


class _PolarWaveFn(torch.autograd.Function):
    """z -> waveform of mag_phase_2_wave(|z|, atan2(z_i, z_r + eps)) (network_functions.py:140-150, :213-221) as ONE node:
    polar round trip + zero bin + frame-major transpose (HIP), unnormalised inverse real FFT (fft512.hip at n_fft = 512, else
    rocFFT; its 1/n folded into the overlap-add scale), window / overlap-add / envelope / trim (HIP).  Backward: adjoint gather (HIP), one forward real
    FFT, and the polar backward kernel applies the one-sided x2 weighting itself — autograd's irfft node costs a scale
    kernel forward and two element-wise complex kernels backward per signal."""

    @staticmethod
    def forward(ctx, z, window, inv_env, n_fft, hop, scale, eps):
        B, Fb, T, _ = z.shape
        if Fb + 1 != n_fft // 2 + 1:
            raise DcsHipError(f'polar_wave: {Fb} bins + 1 zero bin is not the one-sided spectrum of n_fft = {n_fft}')
        comp = ops.polar_frames(z, Fb + 1, eps)
        ctx.cfg = ((B, T, n_fft), n_fft, hop, scale / n_fft, eps)
        ctx.save_for_backward(z, window, inv_env)
        if n_fft == 512 and ops.irfft512_ola_ok(T, hop):   # hop 64 / 128 / 256: inverse FFT + overlap-add in one kernel (not the configured hop = 32)
            return ops.irfft512_ola(comp, window, inv_env, hop, scale / n_fft)
        if n_fft == 512:                                   # hand-written 512-point pair (fft512.hip)
            frames = ops.irfft512(comp)
        else:
            frames = torch.fft.irfft(torch.view_as_complex(comp), n=n_fft, dim=-1, norm='forward')
        return ops.istft_ola(frames, window, inv_env, hop, scale / n_fft)

    @staticmethod
    def backward(ctx, g):
        z, window, inv_env = ctx.saved_tensors
        shape, n_fft, hop, scale, eps = ctx.cfg
        if n_fft == 512 and not (hop & 1):
            G = ops.rfft512_ola(g.contiguous(), window, inv_env, shape[1], hop, scale)     # [B, T, 257, 2]; the frames are not stored
        else:
            g_frames = ops.istft_ola(shape, window, inv_env, hop, scale, grad=g.contiguous())
            G = ops.rfft512(g_frames) if n_fft == 512 else torch.view_as_real(torch.fft.rfft(g_frames, dim=-1))
        return ops.polar_frames(z, z.shape[1] + 1, eps, grad=G, hermitian=True), None, None, None, None, None, None


def polar_wave(z, window, inv_env, n_fft, hop, scale, eps):
    """Complex z [B, F, T] (F = n_fft/2 bins 1..F of the STFT) -> waveform [B, hop (T-1)]."""
    return _PolarWaveFn.apply(torch.view_as_real(z.contiguous()), window, inv_env, n_fft, hop, scale, eps)


class _Bound2ApplyPolarWaveFn(torch.autograd.Function):
    """(Y, raw network output D) -> (twice-bounded mask or None, waveforms [2B, L] of Y (.) M and Y - Y (.) M): _Bound2MaskApplyPairFn
    and _PolarWaveFn as ONE node whose first / last kernel is the fused mask + polar pass (mask.hip, Round 5) — the estimates never
    exist in HBM; the backward recomputes them from (Y, D)."""

    @staticmethod
    def forward(ctx, Y, D_raw, window, inv_env, n_fft, hop, scale, eps, drop_p, seed, want_mask):
        B, Fb, T, _ = Y.shape
        if Fb + 1 != n_fft // 2 + 1 or n_fft != 512:
            raise DcsHipError(f'bound2_apply_polar_wave: {Fb} bins, n_fft = {n_fft}: the fused form is built for n_fft = 512')
        M, comp = ops.bound2_apply_polar_frames(Y, D_raw, Fb + 1, eps, drop_p, seed, want_mask)
        ctx.cfg = ((2 * B, T, n_fft), hop, scale / n_fft, eps, float(drop_p), int(seed))
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(Y, D_raw, window, inv_env)
        if ops.irfft512_ola_ok(T, hop):
            return M, ops.irfft512_ola(comp, window, inv_env, hop, scale / n_fft)
        return M, ops.istft_ola(ops.irfft512(comp), window, inv_env, hop, scale / n_fft)

    @staticmethod
    def backward(ctx, gM, g):
        Y, D_raw, window, inv_env = ctx.saved_tensors
        shape, hop, scale, eps, drop_p, seed = ctx.cfg
        if g is None and gM is None:
            return (None,) * 11
        if g is None:
            return (None, ops.bound2_mask_apply_bwd(Y, D_raw, None, gM.contiguous(), None, None, eps, drop_p, seed)) + (None,) * 9
        if hop & 1:
            G = ops.rfft512(ops.istft_ola(shape, window, inv_env, hop, scale, grad=g.contiguous()))
        else:
            G = ops.rfft512_ola(g.contiguous(), window, inv_env, shape[1], hop, scale)     # [2B, T, 257, 2]; the frames are not stored
        gD = ops.bound2_apply_polar_frames(Y, D_raw, Y.shape[1] + 1, eps, drop_p, seed, grad=G,
                                           g_M=None if gM is None else gM.contiguous(), hermitian=True)
        return (None, gD) + (None,) * 9


def bound2_apply_polar_wave_pair(Y, D_raw, window, inv_env, n_fft, hop, scale, eps=10e-7, drop=(0.0, 0), want_mask=False):
    """Complex Y, D_raw [B, F, T] -> (M complex [B, F, T] or None, waveforms float [2B, hop (T - 1)]): rows [0, B) the noise
    estimate Y (.) M, rows [B, 2B) the speech estimate Y - Y (.) M, M = bound_cRM(bound_cRM(dropout(D_raw)))
    (c_network.py:221-225, network_functions.py:240-247); differentiable w.r.t. D_raw."""
    y = torch.view_as_real(Y.contiguous())
    d = torch.view_as_real(D_raw.contiguous())
    M, wave = _Bound2ApplyPolarWaveFn.apply(y, d, window, inv_env, n_fft, hop, scale, eps, float(drop[0]), int(drop[1]), bool(want_mask))
    return (None if M is None else torch.view_as_complex(M)), wave


class _SiSNRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clean, est, eps):
        snr, coef = ops.sisnr(clean, est, eps)
        ctx.save_for_backward(clean, est, coef)
        return snr.mean()

    @staticmethod
    def backward(ctx, g):
        clean, est, coef = ctx.saved_tensors
        return None, ops.sisnr_bwd(clean, est, coef, g.contiguous(), 1.0 / est.shape[0]), None


def sisnr_mean(clean, est, eps=1e-8):
    """torch.mean over utterances of the SiSNR (network_functions.py:30-42); gradient to `est` only (the clean signal
    is data).  float [B, L] inputs."""
    return _SiSNRFn.apply(clean.detach().contiguous(), est.contiguous(), eps)


class _SiSNRLossesFn(torch.autograd.Function):
    """(noise_loss, speech_loss, total) of network_functions.py:168-208 for noise_loss_type 6 / speech_loss_type 0:
    two dcs_sisnr_fwd + one combine launch forward, two dcs_sisnr_bwd backward — no scalar glue kernels."""

    @staticmethod
    def forward(ctx, clean, est_clean, noise, est_noise, alpha, eps):
        snr_s, coef_s = ops.sisnr(clean, est_clean, eps)
        snr_n, coef_n = ops.sisnr(noise, est_noise, eps)
        out = ops.sisnr_losses(snr_s, snr_n, alpha)
        ctx.save_for_backward(clean, est_clean, coef_s, noise, est_noise, coef_n)
        ctx.alpha = float(alpha)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_noise, g_speech, g_total):
        clean, est_clean, coef_s, noise, est_noise, coef_n = ctx.saved_tensors
        B = est_clean.shape[0]

        def both(a, b):                                    # upstream gradients that reach one SiSNR
            if a is None:
                return b
            return a if b is None else a + b
        gs, gn = both(g_speech, g_total), both(g_noise, g_total)
        # speech_loss = -alpha mean(snr_s);  noise_loss = 1 + alpha mean(snr_n)
        g_ec = None if gs is None else ops.sisnr_bwd(clean, est_clean, coef_s, gs.contiguous(), -ctx.alpha / B)
        g_en = None if gn is None else ops.sisnr_bwd(noise, est_noise, coef_n, gn.contiguous(), ctx.alpha / B)
        return None, g_ec, None, g_en, None, None


def sisnr_losses(clean, est_clean, noise, est_noise, alpha, eps=1e-8):
    return _SiSNRLossesFn.apply(clean.detach().contiguous(), est_clean.contiguous(), noise.detach().contiguous(),
                                est_noise.contiguous(), float(alpha), eps)


class _SiSNRLossesPairFn(torch.autograd.Function):
    """The same three losses from STACKED signals (rows [0,B) noise, [B,2B) speech): one dcs_sisnr_fwd over 2B utterances,
    the combine launch (which also writes the train step's NaN flag when given one), one dcs_sisnr_pair_bwd."""

    @staticmethod
    def forward(ctx, target, est, alpha, eps, skip):
        B = est.shape[0] // 2
        snr, coef = ops.sisnr(target, est, eps)
        out = ops.sisnr_losses_guard(snr[B:], snr[:B], alpha, skip)
        ctx.save_for_backward(target, est, coef)
        ctx.alpha = float(alpha)
        ctx.set_materialize_grads(False)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_noise, g_speech, g_total):
        target, est, coef = ctx.saved_tensors
        if g_noise is None and g_speech is None and g_total is None:
            return None, None, None, None, None
        c = lambda t: None if t is None else t.contiguous()
        return None, ops.sisnr_pair_bwd(target, est, coef, c(g_noise), c(g_speech), c(g_total), ctx.alpha), None, None, None


def sisnr_losses_pair(target, est, alpha, eps=1e-8, skip=None):
    """(noise_loss, speech_loss, total) from target / est float [2B, L], rows [0,B) = noise, [B,2B) = speech."""
    return _SiSNRLossesPairFn.apply(target.detach().contiguous(), est.contiguous(), float(alpha), eps, skip)


def bound_mask_apply_pair_complex(Y, M_in, eps=10e-7):
    """(bound_cRM(M_in), [Y (.) M ; Y - Y (.) M]) — the estimates stacked on a new leading axis; differentiable w.r.t. M_in."""
    y = torch.view_as_real(Y.contiguous())
    m = torch.view_as_real(M_in.contiguous())
    M, NS = _BoundMaskApplyPairFn.apply(y, m, eps)
    return torch.view_as_complex(M), torch.view_as_complex(NS)


def bound_crm_complex(M, eps=10e-7):
    return torch.view_as_complex(bound_crm(torch.view_as_real(M.contiguous()), eps))


def bound_mask_apply_complex(Y, M_in, eps=10e-7):
    """(bound_cRM(M_in), Y (.) M, Y - Y (.) M) in one kernel; differentiable w.r.t. M_in."""
    y = torch.view_as_real(Y.contiguous())
    m = torch.view_as_real(M_in.contiguous())
    M, N, S = _BoundMaskApplyFn.apply(y, m, eps)
    return torch.view_as_complex(M), torch.view_as_complex(N), torch.view_as_complex(S)


def crm_complex(S, Y, eps=1e-8):
    return torch.view_as_complex(ops.crm(torch.view_as_real(S.contiguous()), torch.view_as_real(Y.contiguous()), eps))
