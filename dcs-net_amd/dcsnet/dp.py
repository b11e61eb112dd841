"""Utterance-level data parallelism for DCS-Net training — NEW relative to the
reference, whose DDP lines are commented out (train.py:62-65,138-139).

One process per GPU, identical weights, the minibatch sharded along B.  The model is 11.65 MB:
nothing is split, and the only exchange per step is ONE sum-all-reduce of ONE flat fp32 gradient
bucket (RCCL over xGMI with backend "nccl"; gloo in the CPU tests).  At 8 GPUs a ring moves
2*(7/8)*11.65 MB = 20 MB through a ~150 GB/s link (~0.13 ms) — latency-trivial next to the step,
so there is no bucketing or overlap machinery.  BatchNorm statistics stay LOCAL: with 32
utterances per GPU every rank reproduces the reference's batch-32 semantics (config.py:43).

  FlatBucket      re-homes the parameters that are on the forward path into one contiguous buffer
                  (p.data become views; gradients accumulate straight into a second flat buffer),
                  so the all-reduce and the fused optimizer kernel each touch ONE tensor
  TrainStep       zero-grad -> loss.backward() -> all-reduce -> device-side global norm ->
                  dcs_adam_amsgrad_step (averaging + clip + Adam/AMSGrad in one HIP launch)
"""
import os
import torch
import torch.distributed as dist

from . import _lib
from ._lib import check, ptr, cur_stream

# built by the reference but never run (c_network.py:160-163 vs :218): their .grad stays None and
# torch.optim skips them, so they must not be decayed or averaged either
UNUSED_PREFIXES = ('decoder_attention.12.', 'decoder_attention.13.')


def hot_parameters(net):
    return [(n, p) for n, p in net.named_parameters() if p.requires_grad and not n.startswith(UNUSED_PREFIXES)]


def _lstm_stack_groups(net):
    """Parameter groups that functional._LstmLayerFn wants contiguous: for every ComplexLSTM-like module (two
    bidirectional nn.LSTM named real_lstm / imag_lstm, c_network.py:26-31), per layer and kind the four tensors
    (real fwd, real rev, imag fwd, imag rev) — stacked they are [2, 8H, in] / [2, 2, 4H, H] / [2, 8H]."""
    groups = []
    for mod in net.modules():
        rl, il = getattr(mod, 'real_lstm', None), getattr(mod, 'imag_lstm', None)
        if not (isinstance(rl, torch.nn.LSTM) and isinstance(il, torch.nn.LSTM)):
            continue
        if not (rl.bidirectional and il.bidirectional and rl.bias and il.bias and rl.num_layers == il.num_layers and
                rl.hidden_size == il.hidden_size and rl.input_size == il.input_size and rl.proj_size == 0):
            continue
        for layer in range(rl.num_layers):
            for kind in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh'):
                ps = [getattr(m, f'{kind}_l{layer}{suf}') for m in (rl, il) for suf in ('', '_reverse')]
                groups.append((rl, layer, kind, ps))
    return groups


class FlatBucket:
    def __init__(self, net):
        named = hot_parameters(net)
        if not named:
            raise ValueError('no trainable parameters')
        dev, dtype = named[0][1].device, named[0][1].dtype
        # LSTM parameter groups first, each contiguous in stacking order; everything else in registration order
        hot = {id(p) for _, p in named}
        groups = [g for g in _lstm_stack_groups(net)
                  if all(id(p) in hot and p.numel() % 4 == 0 and p.shape == g[3][0].shape for p in g[3])]
        name_of = {id(p): n for n, p in named}
        grouped = [p for g in groups for p in g[3]]
        taken = {id(p) for p in grouped}
        ordered = grouped + [p for _, p in named if id(p) not in taken]
        self.names = [name_of[id(p)] for p in ordered]
        self.params = ordered
        # 4-float alignment of every slice keeps float4 access legal for any parameter shape
        offs, total = [], 0
        for p in self.params:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel = total
        self.flat = torch.zeros(total, dtype=dtype, device=dev)
        # 4 extra floats behind the gradients: [0] = the step's NaN-loss flag.  It rides the gradient all-reduce, so
        # "some rank saw a NaN loss" reaches every rank in the one collective of the step and all ranks skip together.
        self.grad_all = torch.zeros(total + 4, dtype=dtype, device=dev)
        self.grad = self.grad_all[:total]
        self.skip = self.grad_all[total:total + 1]
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                n = p.numel()
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                p._dcs_grad_sink = p.grad        # backward kernels write here directly (functional._sink)
        self.offsets = offs
        off_of = {id(p): o for p, o in zip(self.params, offs)}
        for rl, layer, kind, ps in groups:        # stacked views of value and gradient (functional._LstmLayerFn)
            o, n = off_of[id(ps[0])], sum(p.numel() for p in ps)
            shape = (2, 2, *ps[0].shape) if kind == 'weight_hh' else (2, 2 * ps[0].shape[0], *ps[0].shape[1:])
            st = rl.__dict__.setdefault('_dcs_stacked', {})
            st.setdefault(layer, {})[kind] = (self.flat[o:o + n].view(shape), self.grad[o:o + n].view(shape))

    def zero_grad(self):
        self.grad_all.zero_()
        for p, o in zip(self.params, self.offsets):        # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)
                p._dcs_grad_sink = p.grad

    def allreduce(self, events=None):
        """Sum over ranks, in place; the 1/world factor is folded into the optimizer kernel.  events: a list that
        receives a (start, end) HIP event pair recorded on the current stream around the collective."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            timed = events is not None and self.grad_all.is_cuda
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            dist.all_reduce(self.grad_all, op=dist.ReduceOp.SUM)
            if timed:
                e1.record()
                events.append((e0, e1))
            return dist.get_world_size()
        return 1


class FusedAdam:
    """dcs_adam_amsgrad_step over a FlatBucket (HIP; no fallback).  The update count also lives on the
    device so that a captured hipGraph can be replayed while the bias corrections advance."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0):
        self.b = bucket
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_norm
        self.m = torch.zeros_like(bucket.flat)
        self.v = torch.zeros_like(bucket.flat)
        self.vmax = torch.zeros_like(bucket.flat)
        self.t = 0
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=bucket.flat.device)
        # fp64 partial sums of g^2 (dcs_grad_sumsq_parts); made here, not in step(): a first step under capture would put it in the graph's pool
        self._parts = torch.empty(512, dtype=torch.float64, device=bucket.flat.device) if bucket.flat.is_cuda else None

    def step(self, world=1, seed_state=None, counters=None):
        """One update, unless the bucket's skip flag is set (a NaN loss on some rank: the reference's trainer skips
        the update, c_network.py:257-261) — decided on the device, so the same launches serve a replayed graph.
        `t` counts calls; the update count the bias corrections use is `t_dev` (advances only with a real update).
        seed_state: the dropout seed offset, advanced in the same launch as `t_dev`; counters: an int64 tensor whose elements
        that launch also advances (the network's num_batches_tracked buffers, TrainStep._counted)."""
        b = self.b
        if not b.flat.is_cuda:
            raise _lib.DcsHipError('FusedAdam: expected CUDA (HIP) parameters; the HIP path has no CPU fallback')
        self.t += 1
        lib = _lib.load()
        nc = 0 if counters is None else counters.numel()
        if self.max_norm > 0 and b.numel >= 4:
            # ||g||^2 as fp64 partial sums + the step's device counters in ONE launch (torch.linalg.vector_norm is a memset and
            # a reduction, the counters were a third); every Adam workgroup adds the partials up itself
            check(lib.dcs_grad_sumsq_parts(ptr(b.grad), b.numel, ptr(self._parts), self._parts.numel(), ptr(b.skip), ptr(self.t_dev),
                                           ptr(seed_state), ptr(counters), nc, cur_stream()), 'dcs_grad_sumsq_parts')
            check(lib.dcs_adam_amsgrad_step_sumsq(ptr(b.flat), ptr(b.grad), ptr(self.m), ptr(self.v), ptr(self.vmax),
                                                  ptr(self._parts), self._parts.numel(), float(self.max_norm), 1.0 / world,
                                                  b.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                                  ptr(self.t_dev), ptr(b.skip), cur_stream()), 'dcs_adam_amsgrad_step_sumsq')
        else:
            check(lib.dcs_step_advance_counters(ptr(b.skip), ptr(self.t_dev), ptr(seed_state), ptr(counters), nc, cur_stream()),
                  'dcs_step_advance_counters')
            # a bucket below four elements (dcs_grad_sumsq_parts reads float4s) is still clipped: its norm by torch, handed to the
            # kernel as a device scalar (this branch used to pass no norm at all and dropped the clipping silently)
            gn = torch.linalg.vector_norm(b.grad).reshape(1) if self.max_norm > 0 else None
            check(lib.dcs_adam_amsgrad_step(ptr(b.flat), ptr(b.grad), ptr(self.m), ptr(self.v), ptr(self.vmax),
                                            ptr(gn), float(self.max_norm) if gn is not None else 0.0, 1.0 / world, b.numel, self.lr,
                                            self.betas[0], self.betas[1], self.eps, self.wd, self.t,
                                            ptr(self.t_dev), ptr(b.skip), cur_stream()), 'dcs_adam_amsgrad_step')
        # the kernel rewrote the parameters behind torch's version counters: invalidate packed weights
        from . import functional
        functional.bump_param_generation()


class TorchAdam:
    """Reference optimizer on the same flat bucket (torch.optim.Adam + clip_grad_norm_): what the
    reference's Trainer does.  Used by the CPU (gloo) data-parallel tests and as the parity
    reference of FusedAdam; never on the product path."""

    def __init__(self, bucket, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_norm=0.0):
        self.b, self.max_norm = bucket, max_norm
        self.opt = torch.optim.Adam(bucket.params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                                    amsgrad=True)

    def step(self, world=1, seed_state=None, counters=None):
        if seed_state is not None:
            seed_state += 1
        if counters is not None:
            counters += 1
        if float(self.b.skip) != 0.0:             # a NaN loss on some rank: every rank skips (c_network.py:257-261)
            return
        if world > 1:
            self.b.grad.div_(world)
        if self.max_norm > 0:
            torch.nn.utils.clip_grad_norm_(self.b.params, self.max_norm)
        self.opt.step()


class TrainStep:
    """One optimisation step of the reference's recipe over a (possibly sharded) minibatch.

    use_graph=True captures forward + losses + backward (+ the fused optimizer when single-process) into
    ONE hipGraph after `graph_warmup` eager steps and replays it afterwards: ~1.3 k launches per step
    become one submission.  Per-step state that must advance under replay lives on the device: the
    dropout seed offset (ops.SEED_STATE), Adam's update count and the NaN-loss flag (the reference skips the update on a
    NaN loss, c_network.py:257-261: here the optimizer kernel is a no-op when the flag element of the gradient bucket is
    set).  The batch is copied into static buffers before each replay; a batch whose shape differs from the captured one
    runs eagerly.  Any capture failure falls back to eager execution, loudly."""

    def __init__(self, net, optimizer_cls=FusedAdam, use_graph=False, graph_warmup=3, use_pack_plan=True, pack_fork=True):
        hp = net.hparams
        self.net = net
        self.bucket = FlatBucket(net)
        self.opt = optimizer_cls(self.bucket, lr=hp['lr'], eps=hp['optim_eps'], weight_decay=hp['optim_weight_decay'],
                                 max_norm=hp.get('gradient_clip_val', 0.0) or 0.0)
        self.use_graph = bool(use_graph) and self.bucket.flat.is_cuda
        self.graph_warmup = graph_warmup
        self.seed_state = None
        if self.bucket.flat.is_cuda:
            from . import ops
            self.seed_state = torch.zeros(1, dtype=torch.int64, device=self.bucket.flat.device)
            ops.SEED_STATE = self.seed_state
        self._graph = self._graph_opt = self._static_batch = self._static_loss = None
        # The pack plan replays every weight re-layout recorded during the first step from the recorded SOURCE POINTERS: safe only
        # for a network all of whose packs read registered parameters (stable addresses in the flat bucket) through
        # functional.packed_weight (destinations kept alive by the plan) — C_NETWORK says so (pack_plan_safe).  R_NETWORK derives its
        # packed panels from temporaries (paired filters, flipped / sliced copies: r_network.py): replayed, those jobs read and
        # WRITE freed memory (found in round 5: the second step of the first DR-Net TrainStep of a process differed from later ones).
        self.use_pack_plan, self._plan = bool(use_pack_plan) and bool(getattr(net, 'pack_plan_safe', False)), None
        self.pack_fork = bool(pack_fork)
        self._calls = 0
        # weight-gradient kernels on a side stream beside the data-gradient chain (functional._CConv2dFn.backward): same kernels,
        # same results; DCS_WGRAD_SIDE=0 keeps the step on one stream (same-box A/B: profiles/r04_side_stream_ab.txt)
        self.wgrad_side_stream = os.environ.get('DCS_WGRAD_SIDE', '1') != '0'
        self.comm_events = None        # bench.py sets a list: (start, end) HIP events around every gradient all-reduce

    def _counting(self, on):
        """While the step's forward runs, the network leaves the `+= 1` of its num_batches_tracked buffers to the optimizer
        half's counter launch (c_network.C_NETWORK._count_batches); _counted() hands the buffer over exactly once."""
        if self.bucket.flat.is_cuda:
            self.net.__dict__['_dcs_defer_nbt'] = bool(on)

    def _counted(self):
        c = self.net.__dict__.pop('_nbt_pending', None)
        if c is not None and c.dtype != torch.int64:
            c += 1                                     # (never the case for the buffers _count_batches makes)
            return None
        return c

    def _step_body(self, batch, batch_idx):
        """NaN guard of the reference (c_network.py:257-261: training_step returns None, the trainer skips the update).
        With several ranks the decision must be the same everywhere and every rank must still enter the collective:
        a rank whose loss is NaN raises the flag element of the gradient bucket and joins the all-reduce with zero
        gradients; the summed flag makes every rank's optimizer a no-op."""
        self.bucket.zero_grad()
        self._counting(True)
        try:
            loss = self.net.training_step(batch, batch_idx)
        finally:
            self._counting(False)
        if loss is None:
            self.bucket.skip.fill_(1.0)
        else:
            self._backward(loss)
        world = self.bucket.allreduce(self.comm_events)
        self.opt.step(world, self.seed_state, self._counted())
        if loss is None or (world > 1 and float(self.bucket.skip) != 0.0):
            return None
        return loss.detach()

    def _backward(self, loss):
        """loss.backward() with the ~28 weight-gradient slab reductions deferred and run as one batched launch at
        the end (nothing reads a weight gradient before the all-reduce / optimizer)."""
        if not self.bucket.flat.is_cuda:
            loss.backward()
            return
        from . import ops
        if os.environ.get('DCS_WGRAD_DEFER', '1') == '0':       # profiling aid: every reduction as its own kernel
            loss.backward()
            return
        # the cotangent of the loss is a constant 1: one cached tensor instead of autograd's ones_like fill per step
        one = self.__dict__.get('_loss_seed')
        if one is None or one.shape != loss.shape or one.dtype != loss.dtype or one.device != loss.device:
            if torch.cuda.is_current_stream_capturing():
                one = None                                  # (made outside a capture only: the three eager warm-up steps do)
            else:
                one = self.__dict__['_loss_seed'] = torch.ones_like(loss)
        from . import functional
        side = None
        if self.wgrad_side_stream:
            side = self.__dict__.get('_wgrad_side')
            if side is None or side.device != loss.device:
                side = self.__dict__['_wgrad_side'] = torch.cuda.Stream(device=loss.device)
        ops.wgrad_defer_begin()
        functional._POOL_ADD.clear()                                   # (left over if an earlier backward raised mid-pass: ADVICE r4)
        del functional.PENDING_SIDE[:]
        functional.WGRAD_SIDE = side
        functional.FLUSH_AT_NEXT_FORK = False
        try:
            if one is None:
                loss.backward()
            else:
                loss.backward(gradient=one)
        finally:
            functional.WGRAD_SIDE = None
            if functional.PENDING_SIDE:                            # (queued after the last fork: none in the networks of this repo)
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        functional._run_pending_side()
                else:
                    functional._run_pending_side()
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)      # the one join: every weight-gradient kernel is in
            ops.wgrad_defer_flush()

    def _eager(self, batch, batch_idx):
        """The first GPU step records every weight pack it makes into a pack plan (the packs still run); later
        steps re-derive all packed weights up front with the plan's 4 launches instead of ~200."""
        from . import functional
        if not (self.use_pack_plan and self.bucket.flat.is_cuda):
            return self._step_body(batch, batch_idx)
        if self._plan is None:
            functional.begin_pack_plan()
            loss = None
            try:
                loss = self._step_body(batch, batch_idx)
            finally:
                plan = functional.end_pack_plan()
            if loss is None:                   # skipped step: backward never packed, record again next time
                from . import ops
                ops.pack_plan_drop()
            else:
                self._plan = plan
            return loss
        functional.run_pack_plan(self._plan)
        return self._step_body(batch, batch_idx)

    def _loss_no_sync(self, batch, batch_idx):
        """training_step without its NaN test (a host synchronisation, illegal under capture)."""
        from .network_functions import train_batch_2_loss
        out = train_batch_2_loss(self.net, batch, batch_idx, dtype=getattr(self.net, '_step_dtype', 'complex'))
        return out[2] if isinstance(out, tuple) else out

    def _device_step(self, batch, world=1):
        """The step exactly as the captured graph holds it: no host synchronisation anywhere (the NaN test is the device-side
        guard).  With world > 1 it stops before the all-reduce (the collective and the optimizer half follow outside)."""
        from . import functional
        if self._plan is not None:
            if self.pack_fork and getattr(self.net, 'accepts_pack_fork', False):
                # the weight re-layout (4 launches, ~70 us, reads only what Adam wrote) on its own stream beside the part of the
                # step that needs no weights: the target synthesis and the initial CBN.  C_NETWORK.forward issues it — i.e.
                # AFTER the target kernels in capture order, which is what lets the two branches overlap in a replayed graph
                # (profiles/r04_side_stream_ab.txt) — and joins before the first convolution.  -20 us / step at B = 32.
                e0 = torch.cuda.Event()
                e0.record()
                self.net.__dict__['_dcs_pack_fork'] = (e0, self._plan)
            else:
                functional.run_pack_plan(self._plan)
        self.bucket.zero_grad()
        # the NaN test training_step makes on the host, on the device: flag element of the gradient bucket.  The fused loss
        # assembly writes it in its own launch when handed the flag; any other loss goes through dcs_step_guard.
        self.net._dcs_skip_flag, self.net._dcs_skip_written = self.bucket.skip, False
        self._counting(True)
        unrun = None
        try:
            loss = self._loss_no_sync(batch, 0)
        finally:
            self.net._dcs_skip_flag = None
            self._counting(False)
            unrun = self.net.__dict__.pop('_dcs_pack_fork', None)          # (never left behind for a later forward, whatever was raised)
        if unrun is not None:
            raise RuntimeError('TrainStep: the step function never reached the network forward; the weight re-layout did not run')
        if not self.net._dcs_skip_written:
            check(_lib.load().dcs_step_guard(ptr(loss), ptr(self.bucket.skip), cur_stream()), 'dcs_step_guard')
        self._backward(loss)
        if world == 1:                                 # collectives stay outside the graph
            self.opt.step(1, self.seed_state, self._counted())
        return loss

    def uncaptured_step(self, batch):
        """One step through the same launches as a graph replay, issued eagerly and without any host synchronisation
        (measurement harnesses: bench.py brackets individual launches of it with events)."""
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        loss = self._device_step(batch, world)
        if world > 1:
            world = self.bucket.allreduce(self.comm_events)
            self.opt.step(world, self.seed_state, self._counted())
        from . import functional
        functional.bump_param_generation()
        return loss

    def _capture(self, batch):
        from . import functional
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        functional.bump_param_generation()                 # packed weights must be re-made INSIDE the graph
        # (noise, clean, noisy) in ONE allocation: the step synthesises the two target waveforms as a single batch of 2B
        # signals straight from the adjacent noise | clean buffers (network_functions._stacked: no concatenation)
        noise, noisy, clean = batch[:3]
        if noise.shape == noisy.shape == clean.shape and noise.dtype == noisy.dtype == clean.dtype:
            buf = torch.empty((3,) + tuple(noise.shape), dtype=noise.dtype, device=noise.device)
            buf[0].copy_(noise); buf[1].copy_(clean); buf[2].copy_(noisy)
            self._static_batch = [buf[0], buf[2], buf[1]]
        else:
            self._static_batch = [t.clone() for t in batch[:3]]
        static = (*self._static_batch, *batch[3:])
        g = torch.cuda.CUDAGraph()
        # thread_local: a collective backend's watchdog thread may touch the HIP runtime while this thread captures
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            loss = self._device_step(static, world)
        functional.bump_param_generation()                 # cache entries made during capture live in its pool
        self._graph, self._static_loss, self._graph_world = g, loss.detach(), world
        self._graph_opt = None
        self._graph_counters = self._counted() if world > 1 else None      # (world 1: the captured optimizer half took them)
        if world > 1:
            # the optimizer half (device-side norm, fused clip + Adam, seed advance) as a second graph replayed after the
            # all-reduce: ~6 eager launches per step otherwise sit between the collective and the next forward
            try:
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, capture_error_mode='thread_local'):
                    self.opt.step(world, self.seed_state, self._graph_counters)
                self._graph_opt = g2
            except Exception as e:                          # noqa: BLE001 - the eager optimizer still works
                import warnings
                warnings.warn(f'TrainStep: optimizer graph capture failed ({type(e).__name__}: {e}); optimizer runs eagerly')
                torch.cuda.synchronize()

    def input_buffers(self):
        """(noise, noisy, clean) device tensors the captured step reads — None before capture.  A data pipeline that writes
        the next batch straight into them (e.g. `buf.copy_(pinned_host_tensor, non_blocking=True)`) and passes THEM to
        __call__ spares the step its three device-to-device staging copies (16.8 MB each at [32,256,256])."""
        return None if self._static_batch is None else tuple(self._static_batch)

    def __call__(self, batch, batch_idx=0):
        if not self.use_graph:
            return self._eager(batch, batch_idx)
        self._calls += 1
        if self._graph is None:
            if self._calls <= self.graph_warmup:
                return self._eager(batch, batch_idx)
            try:
                self._capture(batch)
            except Exception as e:                          # noqa: BLE001 - report and keep training eagerly
                import warnings
                warnings.warn(f'TrainStep: hipGraph capture failed ({type(e).__name__}: {e}); running eagerly')
                self.use_graph = False
                torch.cuda.synchronize()
                return self._eager(batch, batch_idx)
        # the graph holds the captured shapes: a batch of another shape (the reference's DataLoader has no drop_last,
        # config.py:66-69, so an epoch ends on a partial batch) runs eagerly — same kernels, same bucket, same optimizer
        if any(tuple(s.shape) != tuple(b.shape) or s.dtype != b.dtype for s, b in zip(self._static_batch, batch[:3])):
            return self._eager(batch, batch_idx)
        for s, b in zip(self._static_batch, batch[:3]):
            if s is not b:
                s.copy_(b)
        self._graph.replay()
        from . import functional
        if self._graph_world > 1:
            world = self.bucket.allreduce(self.comm_events)
            if self._graph_opt is not None:
                self._graph_opt.replay()
            else:
                self.opt.step(world, self.seed_state, self._graph_counters)
        functional.bump_param_generation()     # the replays rewrote parameters and running statistics behind torch's back
        # a NaN loss is returned as such (the caller may test it); the update was skipped on the device
        return self._static_loss.clone()
