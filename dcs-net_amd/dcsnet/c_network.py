"""DCS-Net's complex encoder / decoder behind the reference's ``c_network`` surface
(c_network.py:12-416): same class names, constructor ``C_NETWORK(config, hparams, seed)``,
sub-module names and registration order (=> same state_dict keys), same ``forward(x)`` contract
(complex64 ``[B,256,T]`` in, bounded complex mask out, batch dimension squeezed when ``B == 1``),
same trainer hook names — so the reference's ``train.py`` / ``test.py`` drive it unchanged.

What differs is underneath: ``forward`` does not walk the module tree.  It runs the path as a
sequence of fused HIP kernels on channels-last interleaved-complex activations
(libdcsnet_hip.so, include/dcsnet_hip.h):

  encoder stage   conv (LDS-tiled / MFMA)  ->  CBN statistics  ->  CBN apply + CReLU + dropout
  skip attention  channel pool + FC  ->  per-pixel channel pool of ca*x  ->  7x7 conv + sigmoid
  decoder stage   convT whose input gather performs  cat(d, sa*ca*skip)  and the nearest upsample
                  -> CBN + CLReLU -> decoder attention (+ dropout)
  output          bound_cRM

The sub-modules (``ComplexConv2d`` ...) are the parameter containers and remain individually
callable (layer-by-layer drop-in surface, dcsnet/complexLayers.py).
"""
import sys

import os
import torch

from . import functional as F
from ._pl_compat import LightningModule, seed_everything
from .complexLayers import (ComplexConv2d, ComplexConvTranspose2d, ComplexBatchNorm2d,   # noqa: F401
                            ComplexLinear, ComplexReLU)
from .complexFunctions import complex_upsample, complex_relu                               # noqa: F401
from .network_functions import *                                                           # noqa: F401,F403
from .network_functions import (ComplexAdaptiveAvgPool2d, ComplexAdaptiveMaxPool2d, ComplexSigmoid,
                                train_batch_2_loss, val_batch_2_metric_loss, test_batch_2_metric_loss,
                                epoch_end, _mode)


class ComplexLSTM(torch.nn.Module):
    """c_network.py:12-51.  Two real 2-layer bidirectional LSTMs (MIOpen through PyTorch-ROCm:
    latency-bound, 2.8 % of the forward FLOPs — SURVEY.md §7), combined as a complex product."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional, batch_first, projection_dim=None):
        super().__init__()
        self.input_dim, self.rnn_units = input_size, hidden_size
        kw = dict(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers,
                  bidirectional=bidirectional, batch_first=batch_first)
        self.real_lstm = torch.nn.LSTM(**kw)
        self.imag_lstm = torch.nn.LSTM(**kw)
        self.projection_dim = projection_dim
        if projection_dim is not None:
            width = hidden_size * (2 if bidirectional else 1)
            self.r_trans = torch.nn.Linear(width, projection_dim)
            self.i_trans = torch.nn.Linear(width, projection_dim)

    def forward(self, inputs):
        out = F.complex_lstm(inputs, self.real_lstm, self.imag_lstm)
        if self.projection_dim is not None:          # unused by C_NETWORK (c_network.py:118-123)
            out = torch.complex(self.r_trans(out.real), self.i_trans(out.imag))
        return out

    def flatten_parameters(self):
        self.imag_lstm.flatten_parameters()
        self.real_lstm.flatten_parameters()


class ComplexChannelAttention(torch.nn.Module):
    """c_network.py:53-69."""

    def __init__(self, no_channels, reduction_ratio):
        super().__init__()
        hidden = max(no_channels // reduction_ratio, 1)
        self.avg_pool = ComplexAdaptiveAvgPool2d(1)
        self.max_pool = ComplexAdaptiveMaxPool2d(1)
        self.fc = torch.nn.Sequential(ComplexConv2d(no_channels, hidden, kernel_size=1, bias=False),
                                      ComplexReLU(),
                                      ComplexConv2d(hidden, no_channels, kernel_size=1, bias=False))
        self.sigmoid = ComplexSigmoid()

    def hip(self, x):
        """x: float [B,H,W,C,2] -> ca float [B,C,2]."""
        return F.channel_attention(x, self.fc[0].conv_r.weight, self.fc[0].conv_i.weight,
                                   self.fc[2].conv_r.weight, self.fc[2].conv_i.weight)

    def forward(self, x):
        B, C = x.shape[0], x.shape[1]
        return torch.view_as_complex(self.hip(F.to_nhwc(x))).view(B, C, 1, 1)


class ComplexSpatialAttention(torch.nn.Module):
    """c_network.py:71-84."""

    def __init__(self, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.conv1 = ComplexConv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = ComplexSigmoid()

    def hip(self, x, ca=None):
        """sa float [B,H,W,1,2] of z = ca*x (ca=None: of x)."""
        return F.spatial_attention(x, ca, self.conv1.conv_r.weight, self.conv1.conv_i.weight, self.kernel_size)

    def forward(self, x):
        return F.from_nhwc(self.hip(F.to_nhwc(x)))


class C_NETWORK(LightningModule):
    def __init__(self, config, hparams, seed):
        super().__init__()
        seed_everything(seed)
        self.config = config
        self.hparams.update(hparams)
        self.save_hyperparameters(self.hparams)
        hp = self.hparams
        ch, L = hp['channels'], hp['no_of_layers']

        # registration order is part of the checkpoint contract (c_network.py:95-98)
        self.encoder = torch.nn.ModuleList()
        self.decoder = torch.nn.ModuleList()
        self.decoder_attention = torch.nn.ModuleList()
        self.skip_attention = torch.nn.ModuleList()

        self.initial_batchnorm = ComplexBatchNorm2d(max(ch[0] // 2, 1))
        for i in range(L):
            cin = 1 if i == 0 else ch[i] // 2
            cout = ch[i + 1] // 2
            self.encoder.append(torch.nn.Sequential(
                ComplexConv2d(cin, cout, kernel_size=config.kernel_sizeE[i], stride=config.strideE[i],
                              padding=config.paddingE[i]),
                ComplexBatchNorm2d(cout),
                config.CactivationE()))

        self.lstm = ComplexLSTM(input_size=ch[4], hidden_size=ch[4] // 2, num_layers=hp['lstm_layers'],
                                bidirectional=hp['lstm_bidir'], batch_first=True)
        self.fc = ComplexLinear(ch[5] // 2, ch[5] // 2)

        ratio, sk = hp['channel_attention_reduction_ratio'], hp['spatial_attention_kernel_size']
        for i in range(L):
            both = ch[L - i]
            cout = max(ch[L - 1 - i] // 2, 1)
            convt = ComplexConvTranspose2d(both, cout, kernel_size=config.kernel_sizeD[i], stride=config.strideD,
                                           padding=config.paddingD[i])
            if i == L - 1:
                self.decoder.append(convt)
            else:
                self.decoder.append(torch.nn.Sequential(convt, ComplexBatchNorm2d(cout), config.CactivationD()))
            self.skip_attention.append(ComplexChannelAttention(both // 2, ratio))
            self.skip_attention.append(ComplexSpatialAttention(sk))
            self.decoder_attention.append(ComplexChannelAttention(cout, ratio))
            self.decoder_attention.append(ComplexSpatialAttention(sk))

        self.dropout_conv = torch.nn.Dropout(hp['dropout_conv'])
        self.dropout_fc = torch.nn.Dropout(hp['dropout_fc'])
        self._drop_calls = 0
        self._counted = False
        self.weights_init()

    def weights_init(self):
        init = self.hparams['initialisation_distribution']
        for m in self.modules():
            if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d, torch.nn.Linear)):
                init(m.weight)

    # ---- fused HIP forward ------------------------------------------------------------------

    def _drop(self, p):
        """(p, seed) for the next fused dropout site; p = 0 in eval like nn.Dropout."""
        if not self.training or p <= 0.0:
            return 0.0, 0
        self._drop_calls += 1
        # the rank is part of the seed: data-parallel ranks share torch's seed (seed_everything) but must draw
        # independent masks for their shards of the minibatch, as one process dropping a batch of B*world would
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and
                                                torch.distributed.is_initialized()) else 0
        return p, (torch.initial_seed() * 0x9E3779B1 + self._drop_calls * 0x85EBCA77 +
                   rank * 0xC2B2AE3D27D4EB4F) & 0x7FFFFFFFFFFFFFFF

    def _bn(self, bn, x, act, p=0.0, two=False, stat=None):
        dp, seed = self._drop(p)
        return bn._hip_forward(x, act, dp, seed, count=not self._counted, two=two, stat=stat)

    def _count_batches(self):
        """num_batches_tracked += 1 for every CBN of the forward path as ONE launch: the 14 scalar buffers are
        views of one int64 tensor (re-made whenever a buffer was replaced, e.g. by .to(device))."""
        bns = [self.initial_batchnorm] + [st[1] for st in self.encoder] + [st[1] for st in self.decoder
                                                                           if isinstance(st, torch.nn.Sequential)]
        self._counted = False
        if not (self.training and all(b.track_running_stats and b.momentum is not None for b in bns)):
            return
        shared = getattr(self, '_nbt_shared', None)
        dev = bns[0].num_batches_tracked.device
        if (shared is None or shared.device != dev or shared.numel() != len(bns) or
                any(b.num_batches_tracked.data_ptr() != shared[i].data_ptr() for i, b in enumerate(bns))):
            shared = torch.stack([b.num_batches_tracked.detach().to(torch.long) for b in bns])
            for i, b in enumerate(bns):
                b._buffers['num_batches_tracked'] = shared[i]
            self.__dict__['_nbt_shared'] = shared
        if self.__dict__.get('_dcs_defer_nbt', False):
            self.__dict__['_nbt_pending'] = shared       # dp.TrainStep adds the 1 in the step's own counter launch (dcs_step_advance_counters)
        else:
            shared += 1
        self._counted = True

    def forward(self, x, bound=True):
        """bound=False (this build's step functions and bench only): return the last stage's RAW output, i.e. skip the final
        bound_cRM (c_network.py:225) because the caller fuses it with the second bound + mask application
        (F.bound2_mask_apply_*): same numbers, one pass over the [B,256,T] mask less each way.  The last stage's dropout
        (c_network.py:221-222) is then left to those kernels as well: (p, seed) in self._pending_dropout, (0, 0) in eval."""
        hp, cfg = self.hparams, self.config
        L = hp['no_of_layers']
        p_conv, p_fc = self.dropout_conv.p, self.dropout_fc.p
        if x.dim() != 3 or x.dtype != torch.complex64:
            raise F.DcsHipError(f'C_NETWORK.forward expects complex64 [B,F,T], got {x.dtype} {tuple(x.shape)}')
        self._check_conv_precision()
        B, Fbins, T = x.shape
        if Fbins % (1 << L) or T % 8:
            raise F.DcsHipError(f'F must be a multiple of {1 << L} and T of 8 (config.py:99,105); got F={Fbins}, T={T}')

        # [B,F,T] complex IS channels-last with C = 1 (c_network.py:190)
        e = torch.view_as_real(x.contiguous()).view(B, Fbins, T, 1, 2)
        self._count_batches()
        # every encoder stage output has TWO consumers (the next stage — the LSTM for the last one — and a skip attention):
        # `enc` serves the first, `enc_skip` the second; with autograd on they are two tensors over one storage, so the
        # two cotangents reach the CBN backward kernels separately and are summed there (F._CbnTwoFn)
        # dp.TrainStep's weight re-layout for this step, left for this point: on its own stream (behind the event recorded when
        # the step began) beside the initial CBN, joined before the first convolution reads a packed weight
        fork = self.__dict__.pop('_dcs_pack_fork', None)
        if fork is not None:
            cur = torch.cuda.current_stream(x.device)
            ps = self.__dict__.get('_pack_stream')
            if ps is None:
                ps = self.__dict__['_pack_stream'] = torch.cuda.Stream(device=x.device)
            ps.wait_event(fork[0])
            with torch.cuda.stream(ps):
                F.run_pack_plan(fork[1])
        enc = [self._bn(self.initial_batchnorm, e, F.ACT_NONE)]
        if fork is not None:
            cur.wait_stream(ps)
        adt = self.activation_dtype
        if adt != torch.float32:                             # bf16 activation storage (set_activation_dtype): everything between
            enc[0] = enc[0].to(adt)                          # the initial CBN and the last decoder conv lives in bf16 in HBM
        enc_skip = [None]
        infer = not self.training and not torch.is_grad_enabled()
        # Inference (Round 4): the skip attentions of the first `early` encoder outputs — the large maps: most of the batched
        # kernels' HBM traffic — start on the side stream as soon as those outputs exist and stream beside the remaining
        # encoder convs (MFMA-bound); only the small late blocks are left to run beside the LSTM, whose recurrence kernel
        # doubled in length while it shared the card with all seven (367 vs 175 us at S = 500).
        early = self.skip_attention_early if (self.overlap_skip_attention and infer and x.is_cuda and self.batch_skip_attention) else 0
        side = skips_early = None
        for i in range(L):                                   # c_network.py:193-197
            conv, bn = self.encoder[i][0], self.encoder[i][1]
            coef = bn.eval_coef() if infer else None
            if coef is not None:                             # inference: CBN folded into the conv epilogue (constants)
                self._drop(p_conv)
                enc.append(F.cconv2d_cbn_eval(enc[i], None, conv.conv_r.weight, conv.conv_i.weight, conv.conv_r.bias,
                                              conv.conv_i.bias, False, conv.kernel_size, conv.stride, conv.padding, (1, 1),
                                              coef, F.ACT_RELU))
                enc_skip.append(enc[-1])
                if early and i + 1 == early and early < L:
                    cur = torch.cuda.current_stream(x.device)
                    side = self.__dict__.get('_side_stream')
                    if side is None or side.device != x.device:
                        side = self.__dict__['_side_stream'] = torch.cuda.Stream(device=x.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):
                        skips_early = self._skip_attentions(enc_skip, stages=[s_ for s_ in range(L) if L - s_ <= early])
                continue
            stat = None
            if self.training:                                # batch statistics straight from the conv's epilogue
                c, stat = F.cconv2d_with_stats(enc[i], None, conv.conv_r.weight, conv.conv_i.weight, conv.conv_r.bias,
                                               conv.conv_i.bias, False, conv.kernel_size, conv.stride, conv.padding)
            else:
                c = F.cconv2d(enc[i], None, conv.conv_r.weight, conv.conv_i.weight, conv.conv_r.bias, conv.conv_i.bias,
                              False, conv.kernel_size, conv.stride, conv.padding)
            a, b = self._bn(bn, c, F.ACT_RELU, p_conv, two=True, stat=stat)
            enc.append(a)
            enc_skip.append(b)

        # latent (c_network.py:199-205): channels-last [B,F7,T7,C] is already [B, seq, C]
        lat = enc[L]
        _, F7, T7, C7, _ = lat.shape
        # The skip attentions depend on the encoder outputs only, and the latent LSTM is a sequential kernel with one workgroup
        # per CU (350 us at S = 500): at inference they run on a side stream beside it — the fork / join become graph
        # dependencies under capture (inference pass 3.46 -> 3.37 ms).  Not in training: with the backward's mirror-image
        # fork the two cross-stream edges cost more than the overlap returns there (step 4.06 -> 4.12 ms, measured).
        if self.overlap_skip_attention and infer and x.is_cuda:
            cur = torch.cuda.current_stream(x.device)
            if side is None:
                side = self.__dict__.get('_side_stream')
                if side is None or side.device != x.device:
                    side = torch.cuda.Stream(device=x.device)
                    self.__dict__['_side_stream'] = side
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                if skips_early is None:
                    skips = self._skip_attentions(enc_skip)
                else:
                    late = self._skip_attentions(enc_skip, stages=[s_ for s_ in range(L) if L - s_ > early])
                    skips = [None] * L
                    for s_, t_ in list(skips_early.items()) + list(late.items()):
                        skips[s_] = t_
        z = self.lstm(torch.view_as_complex(lat if adt == torch.float32 else lat.float()).view(B, F7 * T7, C7))   # (LSTM + fc: fp32)
        if side is not None:
            # join BEFORE self.fc: the side stream's VALU kernels overlap the LSTM recurrence only (a VALU kernel too), never
            # an MFMA conv kernel — see the packed-fp32 / bf16-MFMA co-residency note in DESIGN.md §3
            cur.wait_stream(side)
            for t_ in skips:
                t_.record_stream(cur)
            for t_ in enc_skip[1:]:
                t_.record_stream(side)
        z = self.fc(z)
        zr = torch.view_as_real(z.contiguous())
        dp, seed = self._drop(p_fc)
        if dp > 0:
            zr = F.dropout(zr, dp, seed)
        d = zr.view(B, F7, T7, C7, 2)
        if adt != torch.float32:
            d = d.to(adt)

        if side is None:
            skips = self._skip_attentions(enc_skip)
        for i in range(L):                                   # c_network.py:207-222
            skip = skips[i]
            stat = None
            stage = self.decoder[i]
            convt = stage if i == L - 1 else stage[0]
            up = tuple(cfg.upsample_scale_factor[i])
            if convt.conv_tran_r.out_channels == 1 and (d.shape[3] + skip.shape[3]) % 8 == 0:
                y = F.cconv_single_output(d, skip, convt.conv_tran_r.weight, convt.conv_tran_i.weight,
                                          convt.conv_tran_r.bias, convt.conv_tran_i.bias, convt.kernel_size,
                                          convt.corr_padding, up)
            else:
                coef = stage[1].eval_coef() if infer and i != L - 1 else None
                if coef is not None:                         # inference: conv + CBN + CLReLU in one kernel, then the attention
                    a = F.cconv2d_cbn_eval(d, skip, convt.conv_tran_r.weight, convt.conv_tran_i.weight,
                                           convt.conv_tran_r.bias, convt.conv_tran_i.bias, True, convt.kernel_size, (1, 1),
                                           convt.corr_padding, up, coef, F.ACT_LRELU)
                    self._drop(p_conv)
                    self._drop(0.0)
                    d = self._attend(self.decoder_attention[2 * i], self.decoder_attention[2 * i + 1], a)
                    continue
                stat = None
                if self.training and i != L - 1:             # batch statistics straight from the conv's epilogue
                    y, stat = F.cconv2d_with_stats(d, skip, convt.conv_tran_r.weight, convt.conv_tran_i.weight,
                                                   convt.conv_tran_r.bias, convt.conv_tran_i.bias, True, convt.kernel_size,
                                                   (1, 1), convt.corr_padding, up)
                else:
                    y = F.cconv2d(d, skip, convt.conv_tran_r.weight, convt.conv_tran_i.weight, convt.conv_tran_r.bias,
                                  convt.conv_tran_i.bias, True, convt.kernel_size, (1, 1), convt.corr_padding, up)
            dp, seed = self._drop(p_conv)
            if i != L - 1:                                   # CBN + CLReLU + decoder attention (+ dropout): one node
                ca_m, sa_m = self.decoder_attention[2 * i], self.decoder_attention[2 * i + 1]
                self._drop(0.0)                              # the CBN's (unused) dropout site keeps the seed sequence
                d = stage[1]._hip_forward(y, F.ACT_LRELU, 0.0, 0, count=not self._counted, attention=(
                    ca_m.fc[0].conv_r.weight, ca_m.fc[0].conv_i.weight, ca_m.fc[2].conv_r.weight, ca_m.fc[2].conv_i.weight,
                    sa_m.conv1.conv_r.weight, sa_m.conv1.conv_i.weight, sa_m.kernel_size, dp, seed), stat=stat)
            elif not bound:                                  # the caller's fused bound + apply kernels run this dropout too
                d = y
                self.__dict__['_pending_dropout'] = (dp, seed)
            else:
                d = F.dropout(y, dp, seed) if dp > 0 else y

        net_out = F.bound_crm(d.view(B, Fbins, T, 2), hp['atan2_eps']) if bound else d.view(B, Fbins, T, 2)
        return torch.squeeze(torch.view_as_complex(net_out))          # c_network.py:224

    def _skip_attentions(self, enc, stages=None):
        """skip_i = sa_i (.) ca_i (.) enc[L - i] for every decoder stage (c_network.py:208-211).  Each depends on one
        encoder output only, so all of them run as one batched set of launches (F.attention_blocks); blocks the batched
        entry does not cover (spatial kernel != 7) fall back to one launch set per block — same HIP kernels.
        stages: a subset of the decoder stages (their encoder outputs must exist) -> {stage: skip}; None: the list of all."""
        L = self.hparams['no_of_layers']
        sel = list(range(L)) if stages is None else list(stages)
        mods = [(self.skip_attention[2 * i], self.skip_attention[2 * i + 1]) for i in sel]
        if not sel:
            return {}
        if L > F.ATTENTION_BATCH_MAX or any(sa_m.kernel_size != 7 for _, sa_m in mods) or not self.batch_skip_attention:
            outs = [self._attend(ca_m, sa_m, enc[L - i]) for i, (ca_m, sa_m) in zip(sel, mods)]
        else:
            params = [(ca_m.fc[0].conv_r.weight, ca_m.fc[0].conv_i.weight, ca_m.fc[2].conv_r.weight, ca_m.fc[2].conv_i.weight,
                       sa_m.conv1.conv_r.weight, sa_m.conv1.conv_i.weight) for ca_m, sa_m in mods]
            outs = F.attention_blocks([enc[L - i] for i in sel], params, 7)
        return list(outs) if stages is None else dict(zip(sel, outs))

    batch_skip_attention = True
    skip_attention_early = int(os.environ.get('DCS_SKIP_EARLY', '4'))     # inference: encoder outputs whose skip attentions start early (0: none)
    supports_unbounded_forward = True      # forward(x, bound=False): see forward
    accepts_pack_fork = True               # forward issues dp.TrainStep's pending weight re-layout on a side stream
    pack_plan_safe = True                  # every packed weight derives from a registered parameter through functional.packed_weight (dp.TrainStep)
    activation_dtype = torch.float32

    def set_activation_dtype(self, dtype):
        """torch.float32 (the reference: precision 32, config.py:70) or torch.bfloat16 — BASELINE configs[4]: the activations
        between the initial CBN and the last decoder conv, and their cotangents, live in bf16 in HBM (the _h entry points of
        include/dcsnet_hip.h); parameters, CBN statistics, accumulators, the attention maps, the LSTM and the mask stay fp32.
        bf16 storage runs the MFMA convs on bf16 operands: the process-wide conv precision is switched with it."""
        dtype = {'f32': torch.float32, 'fp32': torch.float32, 'bf16': torch.bfloat16}.get(dtype, dtype)
        if dtype not in (torch.float32, torch.bfloat16):
            raise F.DcsHipError(f'activation dtype {dtype}: float32 or bfloat16')
        self.activation_dtype = dtype
        if dtype == torch.bfloat16:
            F.ops._IN_SET_ACTIVATION_DTYPE = True
            try:
                F.ops.set_conv_precision('bf16')
            finally:
                F.ops._IN_SET_ACTIVATION_DTYPE = False
        # the conv precision is ONE switch per process (include/dcsnet_hip.h): remember what this network was configured for,
        # so that forward() can refuse to run under another network's setting instead of silently computing with it
        self.__dict__['_conv_precision'] = 'bf16' if dtype == torch.bfloat16 else None
        return self

    def _check_conv_precision(self):
        """bf16 storage needs the process-wide conv precision 'bf16'; fp32 storage must NOT run under it (that is the
        bf16-operand mode, a different arithmetic, entered only through ops.set_conv_precision('bf16') on purpose — never
        because another network in the process switched it).  Raises instead of computing with the wrong mode."""
        want = self.__dict__.get('_conv_precision')
        have = F.ops.conv_precision()
        if want == 'bf16' and have != 'bf16':
            raise F.DcsHipError(f"this network stores activations in bf16 and needs conv precision 'bf16', the process is in "
                                f"'{have}': another network or caller switched it — call ops.set_conv_precision('bf16') "
                                f"(one conv precision per process: include/dcsnet_hip.h)")
        if want is None and have == 'bf16' and not F.ops.BF16_OPERANDS_ON_PURPOSE:
            raise F.DcsHipError("fp32-storage network under conv precision 'bf16': a bf16-storage network in this process "
                                "switched the process-wide mode (set_activation_dtype); call ops.set_conv_precision('bf16x6' | "
                                "'f32') before running this one, or ops.set_conv_precision('bf16') yourself to run "
                                "bf16 operands on purpose")
    import os as _os
    overlap_skip_attention = _os.environ.get('DCS_OVERLAP_SKIP', '1') == '1'      # inference only (see forward); 0 disables

    @staticmethod
    def _attend(ca_m, sa_m, x, drop_p=0.0, seed=0):
        """sa (.) ca (.) x with both attentions computed from x (c_network.py:208-211 / :219-220)."""
        fc0, fc2, c1 = ca_m.fc[0], ca_m.fc[2], sa_m.conv1
        return F.attention_block(x, fc0.conv_r.weight, fc0.conv_i.weight, fc2.conv_r.weight, fc2.conv_i.weight,
                                 c1.conv_r.weight, c1.conv_i.weight, sa_m.kernel_size, drop_p, seed)

    # ---- trainer hooks (c_network.py:229-416) -------------------------------------------------

    def configure_optimizers(self):
        hp = self.hparams
        optimiser = torch.optim.Adam(self.parameters(), lr=hp['lr'], eps=hp['optim_eps'],
                                     weight_decay=hp['optim_weight_decay'], amsgrad=hp['optim_amsgrad'])
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimiser, patience=10)
        return {'optimizer': optimiser, 'lr_scheduler': scheduler,
                'monitor': 'val_loss' if _mode() in ('dcs', 'drs') else 'speech_loss'}

    _step_dtype = 'complex'            # R_NETWORK shares these hooks with 'real' (r_network.py:176-363)

    def training_step(self, train_batch, batch_idx):
        out = train_batch_2_loss(self, train_batch, batch_idx, dtype=self._step_dtype)
        if _mode() in ('dcs', 'drs'):
            noise_loss, speech_loss, loss = out
            metrics = {'train_loss': loss.detach(), 'noise_loss': noise_loss.detach(),
                       'speech_loss': speech_loss.detach()}
        else:
            loss = out
            metrics = {'speech_loss': loss.detach()}
        self.log_dict(metrics, on_epoch=True)
        if torch.any(torch.isnan(loss)):
            print(f"found NaN in {'C' if self._step_dtype == 'complex' else 'R'} train loss!")
            return None
        return loss

    def _eval_step(self, fn, batch, idx, prefix):
        out = fn(self, batch, idx, dtype=self._step_dtype)
        if _mode() in ('dcs', 'drs'):
            noise_loss, speech_loss, loss, pesq_av, stoi_av, n_hat, s_hat, noise, noisy, clean = out[:10]
            metrics = {f'{prefix}_loss': loss.detach(), f'{prefix}_noise_loss': noise_loss.detach(),
                       f'{prefix}_speech_loss': speech_loss.detach(),
                       f'{prefix}_pesq': torch.tensor(pesq_av), f'{prefix}_stoi': torch.tensor(stoi_av)}
            audio = {'clean': clean, 'predict_clean': s_hat, 'noise': noise, 'predict_noise': n_hat, 'noisy': noisy}
        else:
            speech_loss, pesq_av, stoi_av, s_hat, noise, noisy, clean = out[:7]
            loss = speech_loss
            metrics = {f'{prefix}_speech_loss': speech_loss.detach(),
                       f'{prefix}_pesq': torch.tensor(pesq_av), f'{prefix}_stoi': torch.tensor(stoi_av)}
            audio = {'clean': clean, 'predict_clean': s_hat, 'noise': noise, 'noisy': noisy}
        return loss, {k: v.detach().cpu().numpy() for k, v in audio.items()}, metrics

    def validation_step(self, val_batch, val_idx):
        loss, audio, metrics = self._eval_step(val_batch_2_metric_loss, val_batch, val_idx, 'val')
        if torch.any(torch.isnan(loss)):
            print(f"found a NaN in {'C' if self._step_dtype == 'complex' else 'R'} val loss!")
            return None
        return audio, metrics

    def test_step(self, test_batch, test_idx):
        _, audio, metrics = self._eval_step(test_batch_2_metric_loss, test_batch, test_idx, 'test')
        return audio, metrics

    def _epoch_end(self, step_outputs, prefix):
        step_outputs = [o for o in step_outputs if o is not None]
        audio = [o[0] for o in step_outputs]
        keys = step_outputs[0][1].keys() if step_outputs else []
        metrics = {k: torch.stack([torch.as_tensor(o[1][k]).float() for o in step_outputs]).mean() for k in keys}
        metrics['step'] = self.current_epoch
        epoch_end(self, audio, prefix)
        self.log_dict(metrics, on_epoch=True)
        return metrics

    def validation_epoch_end(self, validation_step_outputs):
        return self._epoch_end(validation_step_outputs, 'val')

    def test_epoch_end(self, test_step_outputs):
        return self._epoch_end(test_step_outputs, 'test')

    def on_after_backward(self):
        trainer = getattr(self, 'trainer', None)
        if trainer is None or trainer.global_step % 25 != 0 or getattr(self, 'logger', None) is None:
            return
        grads = [p.grad.flatten() if p.grad is not None else p.new_zeros(1) for p in self.parameters()]
        vals = torch.cat(grads)
        self.logger.experiment.add_scalar('grad val avg', vals.mean(), global_step=trainer.global_step)
        self.logger.experiment.add_scalar('grad norm', torch.linalg.norm(vals), global_step=trainer.global_step)
