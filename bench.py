#!/usr/bin/env python3
"""bench.py — throughput of the DCS-Net hot path on MI355X, one JSON line on rank 0.

  python bench.py --gpus N --steps K --warmup W [--mode infer|train]
  (N > 1: launched by torch.distributed.run, one rank per GPU, RCCL)

A "step" is one pass of the hot path over one synthetic batch already resident in HBM:
  infer  BASELINE.json configs[1]: C_NETWORK.eval() forward + bound/mask-apply/subtract,
         complex64 [16,256,2000] per GPU (4 s of 16 kHz audio per utterance = 2000 STFT frames)
  train  BASELINE.json configs[2]: forward + backward + Adam on [32,256,256] per GPU
Utterances are independent, so ranks shard the batch with no data-path collective in `infer`
(weak scaling: per-GPU work fixed) and one flat-bucket gradient all-reduce in `train`.

Besides the contract fields the line carries
  roofline      the dominant kernel (complex conv) timed live with HIP events inside the timed
                region: algorithmic FLOPs of every launch / summed launch duration
  cpu_baseline  the CPU oracle (oracle/cnet_oracle.py, a port: the reference's Python never
                travels) timed on this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'dcs-net_amd'))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_* = 64 FLOP/clk/SIMD
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA (same guide; AMD's headline figure includes 2:1 sparsity)
PEAK_BF16X6_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0     # fp32 products emulated by six bf16 MFMAs: the ceiling of that path
PEAK_HBM_GBS = 8000.0
SUSTAINED_BF16X6_TFLOPS = 1897.4 / 6.0   # measured: tools/micro/mfma_peak.hip mode 4 (clock 1.845 GHz under bf16 MFMA load)
SUSTAINED_F32_MFMA_TFLOPS = 154.5       # measured: the same tool, mode 0 (2.39 GHz)


def synthetic_stft_batch(B, T, device, seed=0):
    """Audio-like synthetic batch: clean = 0.1 randn, noise = 0.05 randn, STFT
    n_fft 512 / hop 32 / hann / normalized, bins 1..256 (data.py:112-118).  samples = T*hop - hop."""
    g = torch.Generator(device='cpu').manual_seed(seed)
    n = T * 32 - 32
    clean = 0.1 * torch.randn(B, n, generator=g)
    noise = 0.05 * torch.randn(B, n, generator=g)
    win = torch.hann_window(512)
    st = lambda a: torch.stft(a, n_fft=512, hop_length=32, win_length=512, window=win, return_complex=True,
                              normalized=True)[:, 1:257, :]
    out = [st(noise), st(clean + noise), st(clean)]
    assert out[0].shape == (B, 256, T), out[0].shape
    return [o.contiguous().to(device) for o in out]


class ConvTimer:
    """Duration of every conv-family C-ABI call of the instrumented pass from the library's kernel timer
    (dcs_kernel_timer_*: the call's kernels are dispatched with start / stop events stamped by the command processor
    around the dispatch itself — what rocprofv3's kernel trace reports.  Event pairs recorded on the stream around a
    launch, the first form of this class, also timed 6-10 us of marker and dispatch latency per launch)."""

    def __init__(self, stride=1):
        # stride > 1: in instrumented step s only the launches j with (j + s) % stride == 0 carry a timer, so a timed
        # kernel runs between UNtimed neighbours (a timed dispatch is followed by the profiling signal's cache write-back:
        # with every launch timed, each kernel starts on a colder L2 than it does in the replayed step)
        self.slots, self.active, self.stride = [], False, max(1, int(stride))
        self.flops_by_seq = {}
        self._seq, self._step, self._ms = 0, 0, None

    def next_step(self):
        self._seq = 0
        self._step += 1

    def begin(self, flops, tag=None, executed=1.0, emulated=False, nbytes=0.0):
        """flops: ALGORITHMIC flops of the launch (what the reference computes); executed: the share of them the kernel
        actually issues (< 1 for the upsample-folded decoder launches).  tag: sub-family of the launch ('enc_fwd' =
        forward ComplexConv2d of the encoder stack, the layers BASELINE.json's target names), summed separately as well.
        emulated: the launch multiplies on the bf16 MFMA (six bf16 MFMA flops per fp32 flop) instead of the fp32 MFMA."""
        if not self.active:
            return None
        j = self._seq
        self._seq += 1
        self.flops_by_seq[j] = (flops, executed, tag, bool(emulated), float(nbytes))
        if (j + self._step) % self.stride:
            return None
        from dcsnet import _lib
        slot = len(self.slots)
        _lib.check(_lib.load().dcs_kernel_timer_begin(slot), 'dcs_kernel_timer_begin')
        return (slot, j)

    def end(self, ev):
        if ev is None:
            return
        from dcsnet import _lib
        if _lib.load().dcs_kernel_timer_end() == 0:          # 1: the call launched nothing (an empty deferred flush)
            self.slots.append(ev)

    def _read(self):
        """{launch index within a step: [ms, ...]}; call after the stream has been synchronised."""
        if self._ms is None:
            import ctypes
            from dcsnet import _lib
            lib, out = _lib.load(), {}
            for slot, j in self.slots:
                v = ctypes.c_float(0.0)
                _lib.check(lib.dcs_kernel_timer_read(slot, ctypes.byref(v)), 'dcs_kernel_timer_read')
                out.setdefault(j, []).append(v.value)
            self._ms = out
        return self._ms

    def _median_ms(self, pred):
        """Per step: sum over the selected launches of the median of their timed samples; a dict with
        ms        measured time of the launches
        flops     ALGORITHMIC flops (8 per complex MAC of the reference's formulation), exec: the flops actually issued
        ceil_ms   the time the launches' ALGORITHMIC flops need at the ceiling of the pipe each launch runs on: the fp32 MFMA
                  peak for native launches, bf16 peak / 6 for launches that emulate fp32 by six bf16 MFMAs per product
        pipe_ms   the same for the EXECUTED flops (folded decoder launches issue 6/9 or 4/9 of the taps)
        emu_flops algorithmic flops of the emulated launches."""
        r = dict(ms=0.0, n=0, flops=0.0, exec=0.0, ceil_ms=0.0, pipe_ms=0.0, emu_flops=0.0, bytes=0.0)
        for j, d in sorted(self._read().items()):
            f, e, tag, emu, nb = self.flops_by_seq[j]
            if pred(tag):
                r['bytes'] += nb
            if pred(tag):
                peak = PEAK_BF16X6_TFLOPS if emu else PEAK_F32_MFMA_TFLOPS
                r['ms'] += sorted(d)[len(d) // 2]
                r['flops'] += f
                r['exec'] += f * e
                r['ceil_ms'] += f / (peak * 1e12) * 1e3
                r['pipe_ms'] += f * e / (peak * 1e12) * 1e3
                r['emu_flops'] += f if emu else 0.0
                r['n'] += 1
        return r

    def summary(self):
        return self._median_ms(lambda tag: True)

    def tag_summary(self, tag):
        return self._median_ms(lambda t: t == tag)

    def tag_layers(self, tag):
        """Per launch of the tagged sub-family, in issue order within a step: (median duration in ms, flops, emulated)."""
        return [(sorted(d)[len(d) // 2], self.flops_by_seq[j][0], self.flops_by_seq[j][3])
                for j, d in sorted(self._read().items()) if self.flops_by_seq[j][2] == tag]

    def samples_per_launch(self):
        r = self._read()
        return min((len(d) for d in r.values()), default=0)


def host_cores():
    """Cores this process may actually use: the scheduler affinity / cgroup share, not the machine's
    core count (a GPU box exposes hundreds of cores but grants ~16 per GPU; asking torch for all of them
    oversubscribes and stalls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def _timed_protocol(one, what, budget_s, warmup=3, timed=10):
    """BASELINE.md "CPU baseline plan": >= 3 warm-up and >= 10 timed repetitions, median — bounded by a wall-clock budget
    (the default bench.py run has to finish within minutes): when the budget truncates the protocol the note says so.
    Returns (median seconds per repetition, note)."""
    t_begin = time.perf_counter()
    done_w = 0
    for _ in range(warmup):
        one()
        done_w += 1
        if time.perf_counter() - t_begin > 0.3 * budget_s:
            break
    if done_w == 1:
        print(f'[bench] cpu baseline: {what}: first repetition {time.perf_counter() - t_begin:.1f} s', file=sys.stderr, flush=True)
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        one()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_begin > budget_s:
            break
    ts.sort()
    note = f'{len(ts)} timed repetitions after {done_w} warm-up, median'
    if done_w < warmup or len(ts) < timed:
        note += f' (protocol is {warmup} + {timed}: truncated by the {budget_s:.0f} s wall-clock budget)'
        print(f'[bench] cpu baseline: {what}: truncated by the {budget_s:.0f} s budget after {done_w} warm-up + {len(ts)} timed',
              file=sys.stderr, flush=True)
    return ts[len(ts) // 2], note


def _cpu_infer_sample(Bs, T, threads, budget_s):
    from oracle.cnet_oracle import C_NETWORK_Oracle
    from oracle.nf_oracle import mask_apply_subtract
    from oracle.seeded_state import fill_state, seeded_input
    torch.set_num_threads(threads)
    net = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), 0).eval()
    x = seeded_input(Bs, 256, T, seed=0, scale=0.1)
    print(f'[bench] cpu baseline: oracle forward, B={Bs}, T={T}, on {torch.get_num_threads()} thread(s) ...', file=sys.stderr, flush=True)

    def one():
        with torch.no_grad():
            mask_apply_subtract(x, net(x))
    dt, note = _timed_protocol(one, f'forward B={Bs} T={T} x{threads}', budget_s)
    return Bs * T / dt, note


def cpu_baseline(B, T):
    """configs[1] on the host cores: the GPU run's own batch on all the cores this process may use (identical
    inputs and batch sizes), and one utterance on ONE thread."""
    v, note = _cpu_infer_sample(B, T, host_cores(), 75.0)
    cores = torch.get_num_threads()
    T1 = min(T, 400)                      # one thread: a 400-frame utterance keeps the sample within the budget
    v1, note1 = _cpu_infer_sample(1, T1, 1, 25.0)
    return {'value': v, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle C_NETWORK eval forward + mask apply, B={B}, T={T}: {note}',
            'single_thread': {'value': v1, 'unit': 'frames/s', 'cores': 1,
                              'sample': f'the same pass, B=1, T={T1}: {note1}'}}


def _cpu_train_sample(Bs, T, threads, budget_s):
    """Oracle train step (forward, SiSNR losses, backward, clip, Adam/AMSGrad) on `threads` host threads; (frames/s, note)."""
    from oracle.cnet_oracle import C_NETWORK_Oracle
    from oracle.nf_oracle import dcs_train_losses
    from oracle.seeded_state import fill_state, seeded_input
    torch.set_num_threads(threads)
    net = fill_state(C_NETWORK_Oracle(), 0).train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, eps=1e-6, weight_decay=1e-4, amsgrad=True)
    clean, noise = seeded_input(Bs, 256, T, 1, 0.1), seeded_input(Bs, 256, T, 2, 0.05)
    noisy = clean + noise

    def one():
        opt.zero_grad()
        loss = dcs_train_losses(net, noise, noisy, clean)[2]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 100.0)
        opt.step()

    print(f'[bench] cpu baseline: oracle train step, B={Bs}, on {torch.get_num_threads()} thread(s) ...', file=sys.stderr, flush=True)
    dt, note = _timed_protocol(one, f'train step B={Bs} x{threads}', budget_s)
    return Bs * T / dt, note


def cpu_baseline_train(B, T):
    """The CPU port beside the GPU number, by BASELINE.md's plan: the GPU run's batch size, >= 3 warm-up and >= 10 timed steps,
    median, on all the cores this process may use — and on ONE thread (a smaller batch: a B = 32 step on one thread takes
    about a minute)."""
    v, note = _cpu_train_sample(B, T, host_cores(), 90.0)
    cores = torch.get_num_threads()
    v1, note1 = _cpu_train_sample(min(B, 2), T, 1, 30.0)
    return {'value': v, 'unit': 'frames/s', 'cores': cores, 'kind': 'port',
            'sample': f'oracle C_NETWORK train step (fwd + losses + bwd + clip + Adam), B={B}, T={T}: {note}',
            'single_thread': {'value': v1, 'unit': 'frames/s', 'cores': 1,
                              'sample': f'the same step, B={min(B, 2)}, T={T}: {note1}'}}


def pmc_traffic(mode, B, T):
    """HBM bytes per step of the conv kernel family from the committed PMC passes (profiles/), or None when
    the workload differs from the one that was profiled.  PMC counters cannot be read live from here."""
    try:
        d = json.load(open(os.path.join(REPO, 'profiles', 'traffic.json')))[mode]
    except (OSError, KeyError, ValueError):
        return None
    if (mode == 'train' and (B, T) == (32, 256)) or (mode == 'infer' and (B, T) == (16, 2000)):
        return d['conv_family_hbm_bytes_per_step']
    return None


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        return s_.getsockname()[1]


def self_launch(n):
    """Run this script as n ranks of one node through torch.distributed.run (one process per GPU, rendezvous on
    127.0.0.1), stream their output through, and check that the JSON line rank 0 printed really saw n ranks.  Returns the
    exit code.  No GPU call happens in this (parent) process."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    seen = None
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
        if ln.startswith('{'):
            try:
                d = json.loads(ln)
                seen = (d.get('n_gpus'), (d.get('config') or {}).get('world_seen'))
            except ValueError:
                pass
    rc = proc.wait()
    if rc != 0:
        return rc
    if seen != (n, n):
        print(f'bench.py: asked for --gpus {n}, the ranks reported (n_gpus, world_seen) = {seen}', file=sys.stderr)
        return 3
    return 0


def launch_only(world, rank):
    """DCS_BENCH_LAUNCH_ONLY=1 (tests/test_dp_cpu.py): the launch plumbing without the GPU workload — join a gloo group,
    one all-reduce, rank 0 prints a line with what it saw."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world > 1:
        dist.init_process_group('gloo')
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({'metric': 'launch-only', 'n_gpus': world, 'config': {'world_seen': seen}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_mfma_busy(mode, B, T):
    """SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) per kernel family from the committed PMC pass (profiles/), or None
    when the workload differs from the profiled one."""
    try:
        d = json.load(open(os.path.join(REPO, 'profiles', 'traffic.json')))[mode]
    except (OSError, KeyError, ValueError):
        return None
    if (mode == 'train' and (B, T) == (32, 256)) or (mode == 'infer' and (B, T) == (16, 2000)):
        return d.get('mfma_busy')
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--mode', default='train', choices=['infer', 'train'])
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the config\'s)')
    ap.add_argument('--frames', type=int, default=None)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch every kernel eagerly instead of replaying a hipGraph')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='operand precision of the MFMA conv forward / data gradient (bf16 = BASELINE configs[4] mixed '
                         'precision; NOT the headline: the reference computes in fp32)')
    ap.add_argument('--no-native-line', action='store_true',
                    help='skip the second measurement with the native fp32 MFMA (default run, one GPU: a child process '
                         'repeats the timed region with --f32-mfma native and its numbers are attached as "native_f32_mfma")')
    ap.add_argument('--no-sub-lines', action='store_true',
                    help='skip the child-process measurements of the other BASELINE configs (default run, one GPU, train mode: '
                         '"infer" = configs[1] and "bf16_b64" = configs[4]\'s per-GPU share are attached to the line)')
    ap.add_argument('--f32-mfma', default='bf16x6', choices=['native', 'bf16x6'],
                    help='how the fp32 MFMA conv forward / data gradient multiplies (--dtype f32 only): native = '
                         'v_mfma_f32_32x32x2_f32; bf16x6 = fp32 emulated on the bf16 MFMA (exact three-way bf16 splits of both '
                         'operands, six cross products, fp32 accumulate: error vs fp64 no larger than the native instruction\'s, '
                         'profiles/*conv_precision.txt)')
    args = ap.parse_args()
    conv_mode = args.dtype if args.dtype != 'f32' else ('f32' if args.f32_mfma == 'native' else 'bf16x6')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N rank processes ourselves, BEFORE this process makes
        # any GPU call (a parent that has initialised HIP must not fork / exec rank processes), relay rank 0's line
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; refusing to print '
                         f'a line whose n_gpus would not be what was asked for')
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('DCS_BENCH_LAUNCH_ONLY') == '1':
        return launch_only(world, rank)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the HIP path has no CPU fallback')
    # rehearsal aids for a one-GPU box (never set by the driver): DCS_BENCH_DEVICE pins every rank to one card and
    # DCS_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device), so `torch.distributed.run
    # --nproc-per-node 2 bench.py --gpus 2` exercises the multi-rank control flow end to end
    if 'DCS_BENCH_DEVICE' in os.environ:
        local = int(os.environ['DCS_BENCH_DEVICE'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('DCS_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    from dcsnet import _lib, ops
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet import functional as F
    _lib.load()
    ops.set_conv_precision(conv_mode)

    def log(msg):
        if rank == 0:
            print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)

    train = args.mode == 'train'
    bf16 = args.dtype == 'bf16'
    # --dtype bf16: BASELINE configs[4]'s per-GPU share — bf16 activations in HBM, B = 64 per GPU in training
    B = args.batch or ((64 if bf16 else 32) if train else 16)
    T = args.frames or (256 if train else 2000)
    torch.manual_seed(0)
    net = C_NETWORK(config, hparams, 0).to(dev)
    if bf16:
        net.set_activation_dtype('bf16')
    noise, noisy, clean = synthetic_stft_batch(B, T, dev, seed=rank)

    # DCS_BENCH_TIMER_STRIDE=7 times every 7th launch only (a different residue each step): measured identical to timing
    # every launch (tools/micro/timer_stride_probe.py), i.e. the timers do not disturb their neighbours
    timer = ConvTimer(stride=int(os.environ.get('DCS_BENCH_TIMER_STRIDE', '1')))
    ops.CONV_TIMER = timer

    if train:
        from dcsnet.dp import TrainStep
        net.train()                       # reference dropout (0.1 / 0.2) and batch statistics
        # flat bucket + fused HIP Adam/AMSGrad + clip 100 + all-reduce; fwd+bwd(+optimizer) replayed as one hipGraph
        ts = TrainStep(net, use_graph=not args.no_graph)
        batch = (noise, noisy, clean, list(range(B)))

        def step():
            return ts(batch)

        def eager_step():
            return ts._eager(batch, 0)

        def queued_step():                # the captured step's launches, issued eagerly, no host synchronisation
            return ts.uncaptured_step(batch)
        setup_steps = ts.graph_warmup + 1 if ts.use_graph else 0
    else:
        net.eval()

        def eager_step():
            with torch.no_grad():
                # forward(bound=False) + one kernel for both bound_cRM applications, the multiply and the subtract
                # (c_network.py:225 + network_functions.py:240-243): the once-bounded mask makes no round trip through HBM
                d_raw = net(noisy, bound=False)
                return F.bound2_mask_apply_complex(noisy, d_raw, hparams['atan2_eps'])
        step, setup_steps, queued_step = eager_step, 0, eager_step
        if not args.no_graph:
            for _ in range(2):
                eager_step()
            torch.cuda.synchronize()
            try:
                infer_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(infer_graph):
                    static_out = eager_step()
                step = infer_graph.replay
            except Exception as e:                      # noqa: BLE001
                log(f'hipGraph capture failed ({type(e).__name__}: {e}); running eagerly')
                torch.cuda.synchronize()

    for _ in range(setup_steps):          # eager steps + the capture itself: setup, neither warm-up nor timed
        step()
    torch.cuda.synchronize()
    if train and ts.input_buffers() is not None:
        # the synthetic batch is resident in HBM already: place it in the captured step's own input buffers (what a loader
        # with pinned host memory would do with its H2D copies), so no device-to-device staging copy runs per step
        bufs = ts.input_buffers()
        for dst, src in zip(bufs, batch[:3]):
            dst.copy_(src)
        batch = (*bufs, batch[3])
        torch.cuda.synchronize()

    log(f'inputs ready: B={B} T={T}')
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f'warmup {i} done')
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    graphed = (train and ts.use_graph) or (not train and step is not eager_step)
    timer.active = not graphed
    if timer.active:
        timer.stride = 1                  # un-graphed runs: every launch of the timed region carries a timer
    if train:
        ts.comm_events = []               # (start, end) events around every gradient all-reduce of the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timer.next_step()
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timer.active = False
    comm_ms = None
    if train:
        ev, ts.comm_events = ts.comm_events, None
        if ev:
            d = sorted(a.elapsed_time(b) for a, b in ev)
            comm_ms = d[len(d) // 2]
    if graphed and (rank == 0 or (train and world > 1)):
        # kernels inside a hipGraph replay cannot be bracketed by events: time the conv family in an
        # instrumented EAGER pass of the same K steps, right after (and outside) the timed region.  Rank 0 holds the
        # timer; with world > 1 every rank runs the pass, because a train step contains the gradient all-reduce (a
        # collective issued by rank 0 alone would pair with the other ranks' next collective)
        # The pass is issued BEHIND A HELD STREAM (dcs_stream_hold): a one-wave kernel parks the stream on a word of pinned
        # host memory while the host enqueues the whole step — launches and event records — and is released afterwards, so
        # the kernels execute back to back out of the queue, as they do under graph replay, not at the pace of Python's
        # launch calls (an idle card between launches times every kernel cold: +10-15 % on the 30-70 us conv launches).
        timer.active = rank == 0
        side_was = None
        if train:
            # the timed region runs the weight-gradient kernels on a side stream, beside the data-gradient chain; a kernel's
            # roofline is its OWN duration, so the instrumented pass issues the same launches on one stream (co-scheduled, every
            # kernel's duration includes the time it shares the card: `kernel_ms_per_step` would count overlapped time twice)
            side_was, ts.wgrad_side_stream = ts.wgrad_side_stream, False
        else:
            # likewise the inference pass: its skip attentions run on a side stream beside the encoder convs and the LSTM
            side_was, net.overlap_skip_attention = net.overlap_skip_attention, False
        flag = torch.zeros(1, dtype=torch.int32).pin_memory()
        hold = world == 1                  # with a collective inside the step the ranks would hold each other
        for _ in range(max(args.steps, 3 * timer.stride)):      # every launch timed at least three times
            timer.next_step()
            if hold:
                flag[0] = 0
                _lib.check(_lib.load().dcs_stream_hold(flag.data_ptr(), 5000, _lib.cur_stream()), 'dcs_stream_hold')
            queued_step()
            if hold:
                flag[0] = 1
            torch.cuda.synchronize()
        timer.active = False
        if side_was is not None:
            if train:
                ts.wgrad_side_stream = side_was
            else:
                net.overlap_skip_attention = side_was
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    log(f'timed region done: {dt / args.steps * 1e3:.2f} ms/step')
    if rank == 0:
        frames = B * T * world * args.steps
        cs = timer.summary()                                                               # per step
        conv_ms, n_launch, conv_flops, conv_exec, conv_pipe_ms = cs['ms'], cs['n'], cs['flops'], cs['exec'], cs['pipe_ms']
        # `peak` is the ceiling of the pipe the launches actually run on: launches that emulate an fp32
        # product by six bf16 MFMAs are priced at bf16 peak / 6 = 416.7 TFLOP/s, native fp32-MFMA launches at 157.3, combined
        # by the launches' flops (peak = flops / time-at-ceiling) — so `frac` cannot exceed 1 for work that is really done.
        # `frac_vs_f32_mfma_peak` keeps the figure of rounds 1-2 (algorithmic fp32 flops over the fp32 MFMA peak), which is
        # the "relevant per-GPU roofline" BASELINE.json's 70 % target was written against; it may exceed 1 for emulated launches.
        if args.dtype != 'f32':
            peak = PEAK_BF16_MFMA_TFLOPS
        else:
            peak = conv_flops / (cs['ceil_ms'] * 1e-3) / 1e12 if cs['ceil_ms'] > 0 else PEAK_F32_MFMA_TFLOPS
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        assert achieved <= peak * 1.0001, (achieved, peak)
        line = {
            'metric': ('STFT frames/sec (train fwd+bwd+Adam)' if train else
                       'STFT frames/sec (forward-only inference: C_NETWORK forward + bound/mask-apply/subtract)'),
            'value': frames / dt, 'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.dtype == 'f32' else 'bf16 (activations and cotangents stored in bf16 in HBM, bf16 MFMA operands; fp32 accumulators, CBN statistics, attention maps, LSTM, mask, parameters, gradients, Adam)',
            'data': 'synthetic',
            'config': {'workload': (f'BASELINE configs[4], per-GPU share: DCS-Net bf16 mixed-precision train step (fwd + SiSNR losses + '
                                    f'bwd + grad all-reduce + clip 100 + Adam/AMSGrad), complex64 [{B},256,{T}] x (noise, noisy, clean) per '
                                    'GPU, dropout 0.1/0.2, batch-statistics CBN, random-init weights seed 0 — NOT the headline '
                                    '(the reference trains at precision 32)' if train and bf16 else
                                    ('BASELINE configs[2]/[3]' if (B, T) == (32, 256) else 'BASELINE configs[2] at another batch') +
                                    ': DCS-Net full train step (fwd + SiSNR losses + bwd + grad '
                                    f'all-reduce + clip 100 + Adam/AMSGrad), complex64 [{B},256,{T}] x (noise, noisy, clean) '
                                    'per GPU, dropout 0.1/0.2, batch-statistics CBN, random-init weights seed 0'
                                    if train else
                                    f'BASELINE configs[1]: DCS-Net forward-only inference, complex64 [{B},256,{T}] per GPU '
                                    '(4 s / 16 kHz STFT, n_fft 512 hop 32, bins 1..256), random-init weights seed 0'),
                       'world_seen': (dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1),
                       'collective_backend': (dist.get_backend() if dist.is_available() and dist.is_initialized() else None),
                       'allreduce_ms_per_step': comm_ms, 'allreduce_bytes': (4 * (ts.bucket.numel + 4) if train else 0),
                       'f32_mfma': (None if args.dtype != 'f32' else
                                    'native: v_mfma_f32_32x32x2_f32' if conv_mode == 'f32' else
                                    'bf16x6: conv forward / data gradient emulate fp32 on v_mfma_f32_32x32x16_bf16 (exact 3-way '
                                    'bf16 split of both operands, 6 cross products, fp32 accumulate; error vs fp64 below the native '
                                    'MFMA\'s: profiles/*conv_precision.txt); the weight gradients run on the same emulation (cconv_wgrad_x6_kernel), and so does the first encoder conv (v_mfma_f32_16x16x32_bf16); its weight gradient runs on the native fp32 MFMA, the 1-output-channel decoder layer and the 7x7 attention convs on the VALU'),
                       'streams': (('2: the weight-gradient kernels run on a side stream beside the data-gradient chain of the backward '
                                    'pass (one fork per conv layer, one join in front of the slab reductions), all inside the captured graph')
                                   if train and getattr(ts, 'wgrad_side_stream', False) else
                                   ('2: the skip attentions run on a side stream beside the encoder convs and the latent LSTM (captured)'
                                    if not train and getattr(net, 'overlap_skip_attention', False) else 1)),
                       'per_gpu_batch': B, 'frames_per_utterance': T, 'global_batch': B * world,
                       'frames_per_step': B * T * world, 'hip_graph': bool(graphed), 'parallelism': (f'dp{world} (utterance sharding, one flat-bucket gradient all-reduce)' if train
                                       else f'dp{world} (utterance sharding, no collective)')},
            'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak, 'traffic': pmc_traffic(args.mode, B, T),
                         'peak_definition': ('flops-weighted ceiling of the pipes the launches run on: 2500 / 6 = 416.7 TFLOP/s for '
                                             'launches that emulate fp32 on the bf16 MFMA (six MFMAs per product), 157.3 for native '
                                             'fp32-MFMA launches' if args.dtype == 'f32' else 'dense bf16 MFMA peak'),
                         'frac_vs_f32_mfma_peak': achieved / PEAK_F32_MFMA_TFLOPS,
                         # what the same pipes SUSTAIN on this part (tools/micro/mfma_peak.hip, profiles/r03_mfma_peak_bf16.txt: a
                         # bare bf16 MFMA loop runs at 1.845 GHz, 1.90 PFLOP/s -> 316 TFLOP/s of emulated fp32; fp32 MFMA 154.5)
                         'sustained_peak': ({'bf16x6': SUSTAINED_BF16X6_TFLOPS, 'f32_mfma': SUSTAINED_F32_MFMA_TFLOPS,
                                             'frac_of_flops_weighted': achieved / (conv_flops / (
                                                 cs['emu_flops'] / SUSTAINED_BF16X6_TFLOPS +
                                                 (conv_flops - cs['emu_flops']) / SUSTAINED_F32_MFMA_TFLOPS)) if conv_flops > 0 else None,
                                             'source': 'profiles/r03_mfma_peak_bf16.txt'} if args.dtype == 'f32' else None),
                         'emulated_share_of_flops': (cs['emu_flops'] / conv_flops if conv_flops > 0 else 0.0),
                         'mfma_busy_pmc': pmc_mfma_busy(args.mode, B, T),
                         'kernel': 'complex conv / convT (dcs_cconv2d_fwd' + (
                             ', _bwd_data, _bwd_weight; the last decoder stage\'s backward dcs_cconv_up2_single_bwd_data / _bwd_weight '
                             'counted with the flops of the 1x1 tap-conv launches it replaced in round 5' if train else '')
                                   + '), all launches of the timed region',
                         'launches_per_step': n_launch, 'kernel_ms_per_step': conv_ms,
                         'samples_per_launch': timer.samples_per_launch(),
                         'measured': ('HIP start / stop events attached to every kernel dispatch of the family '
                                      '(hipExtLaunchKernelGGL through dcs_kernel_timer_*: the dispatch\'s own duration, as in '
                                      'rocprofv3\'s kernel trace) in an instrumented pass of the same K steps right after the '
                                      'timed region — the timed steps replay a hipGraph, whose nodes take no events; the '
                                      'pass is enqueued behind a held stream and drains back to back; per launch the median over '
                                      'the K steps.  The start stamp is a marker in front of the dispatch, so each launch '
                                      'carries ~5 us of dispatch latency that the replayed graph does not pay: the committed '
                                      'rocprofv3 kernel stats (profiles/) hold the dispatches\' own durations' if graphed else
                                      'HIP start / stop events attached to every kernel dispatch of the family '
                                      '(hipExtLaunchKernelGGL through dcs_kernel_timer_*) inside the timed region'),
                         'algorithmic_gflop_per_step': conv_flops / 1e9,
                         # MACs the kernels actually issue: the upsample-folded decoder launches run 6/9 or 4/9 of the
                         # reference's taps (same result), so their algorithmic rate can exceed the MFMA peak; this one cannot
                         'executed_gflop_per_step': conv_exec / 1e9,
                         'executed_achieved': (conv_exec / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0),
                         # share of the launches' time that the MFMA pipes need at their dense peak for the instructions
                         # actually issued: executed fp32 flops at 157.3 TFLOP/s on the native path, six bf16 MFMA flops per
                         # fp32 flop at 2500 TFLOP/s on the emulated one
                         'executed_frac': (conv_pipe_ms / conv_ms if conv_ms > 0 else 0.0)},
        }
        line['roofline']['algorithmic_gbytes_per_step'] = cs['bytes'] / 1e9
        if bf16:
            # BASELINE.md / SURVEY §8(d): with bf16 storage every conv stage sits below the bf16 ridge (310 flop/B) — the family's
            # roofline is HBM.  achieved = algorithmic bytes (every activation read once and written once at 4 B per complex
            # value, the weights once) / the launches' measured time; the MFMA figures stay beside it.
            mf = dict(line['roofline'])
            gbs = cs['bytes'] / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
            line['roofline'] = {'bound': 'hbm', 'achieved': gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': gbs / PEAK_HBM_GBS,
                                'traffic': None, 'kernel': mf['kernel'], 'launches_per_step': n_launch,
                                'kernel_ms_per_step': conv_ms, 'samples_per_launch': mf['samples_per_launch'],
                                'algorithmic_gbytes_per_step': cs['bytes'] / 1e9, 'measured': mf['measured'],
                                'mfma': {'achieved': mf['achieved'], 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                         'frac': mf['achieved'] / PEAK_BF16_MFMA_TFLOPS,
                                         'algorithmic_gflop_per_step': mf['algorithmic_gflop_per_step']}}
        # BASELINE.json's target names the ComplexConv2d ENCODER stack: its forward launches on their own (same pass)
        es = timer.tag_summary('enc_fwd')
        e_ms, e_n, e_fl = es['ms'], es['n'], es['flops']
        if e_ms > 0:
            e_tf = e_fl / (e_ms * 1e-3) / 1e12
            e_peak = peak if args.dtype != 'f32' else e_fl / (es['ceil_ms'] * 1e-3) / 1e12
            lp = lambda emu: PEAK_BF16_MFMA_TFLOPS if args.dtype != 'f32' else (PEAK_BF16X6_TFLOPS if emu else PEAK_F32_MFMA_TFLOPS)
            line['roofline']['encoder_stack_forward'] = {
                'achieved': e_tf, 'peak': e_peak, 'frac': e_tf / e_peak, 'frac_vs_f32_mfma_peak': e_tf / PEAK_F32_MFMA_TFLOPS,
                'unit': 'TFLOP/s', 'launches_per_step': e_n, 'kernel_ms_per_step': e_ms,
                'per_layer': [{'layer': f'enc{i}', 'us': ms * 1e3, 'tflops': fl / (ms * 1e-3) / 1e12,
                               'pipe': 'bf16x6' if emu else ('f32' if args.dtype == 'f32' else 'bf16'),
                               'frac': fl / (ms * 1e-3) / 1e12 / lp(emu),
                               'frac_vs_f32_mfma_peak': fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
                              for i, (ms, fl, emu) in enumerate(timer.tag_layers('enc_fwd'))]}
        if not args.no_cpu_baseline and world == 1:
            line['cpu_baseline'] = cpu_baseline_train(B, T) if train else cpu_baseline(B, T)
        else:
            line['cpu_baseline'] = None
        line['native_f32_mfma'] = None
        if world == 1 and args.dtype == 'f32' and conv_mode == 'bf16x6' and not args.no_native_line:
            # the same timed region on the native fp32 MFMA, in a child process (its own graph, plans and packed panels)
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), '--mode', args.mode, '--steps', str(args.steps), '--warmup',
                   str(args.warmup), '--f32-mfma', 'native', '--no-cpu-baseline', '--no-native-line', '--no-sub-lines']
            cmd += (['--batch', str(args.batch)] if args.batch else []) + (['--frames', str(args.frames)] if args.frames else [])
            cmd += ['--no-graph'] if args.no_graph else []
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                nat = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
                enc = nat['roofline'].get('encoder_stack_forward') or {}
                line['native_f32_mfma'] = {'value': nat['value'], 'ms_per_step': nat['ms_per_step'],
                                           'roofline_frac': nat['roofline']['frac'],
                                           'executed_frac': nat['roofline']['executed_frac'],
                                           'encoder_stack_forward_frac': enc.get('frac'),
                                           'kernel_ms_per_step': nat['roofline']['kernel_ms_per_step']}
            except Exception as e:                              # the headline line must not depend on the second run
                line['native_f32_mfma'] = {'error': repr(e)[:200]}
        # The other single-GPU BASELINE configs, each measured by a fresh child process in this same run:
        # configs[1] (forward-only inference, [16,256,2000]) and configs[4]'s per-GPU share (bf16 storage, B = 64 train step),
        # each with its own roofline object — so that the driver's one `python bench.py --gpus 1` times all three.
        line['sub_lines'] = None
        if (world == 1 and train and args.dtype == 'f32' and conv_mode == 'bf16x6' and not args.no_sub_lines and not args.no_native_line and
                args.batch is None and args.frames is None):
            import subprocess
            line['sub_lines'] = {}
            common = ['--steps', str(args.steps), '--warmup', str(args.warmup), '--no-cpu-baseline', '--no-native-line',
                      '--no-sub-lines'] + (['--no-graph'] if args.no_graph else [])
            for name, extra in (('infer', ['--mode', 'infer']), ('bf16_b64', ['--mode', 'train', '--dtype', 'bf16', '--batch', '64'])):
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + common, capture_output=True,
                                       text=True, timeout=600)
                    sub = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
                    rf = sub['roofline']
                    keep = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_ms_per_step', 'launches_per_step',
                            'executed_frac', 'algorithmic_gflop_per_step', 'algorithmic_gbytes_per_step', 'mfma',
                            'encoder_stack_forward')
                    line['sub_lines'][name] = {'metric': sub['metric'], 'value': sub['value'], 'unit': sub['unit'],
                                               'ms_per_step': sub['ms_per_step'], 'steps': sub['steps'], 'warmup': sub['warmup'],
                                               'dtype': sub['dtype'].split(' ')[0], 'workload': sub['config']['workload'],
                                               'hip_graph': sub['config']['hip_graph'],
                                               'roofline': {k_: rf[k_] for k_ in keep if k_ in rf}}
                except Exception as e:                          # the headline line must not depend on the extra runs
                    line['sub_lines'][name] = {'error': repr(e)[:200]}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
