"""GPU: the optimizer kernel against torch.optim.Adam(amsgrad) + clip_grad_norm_, and the full
train step (config-3 recipe) against the CPU oracle's step on the same seeded state and batch."""
import os
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.cnet_oracle import C_NETWORK_Oracle      # noqa: E402
from oracle.nf_oracle import dcs_train_losses         # noqa: E402
from oracle.seeded_state import fill_state, seeded_input   # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    from dcsnet import _lib
    _lib.load()
    return torch.device('cuda:0')


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        self.a = torch.nn.Parameter(torch.randn(37, 5, generator=g))
        self.b = torch.nn.Parameter(torch.randn(1001, generator=g))
        self.c = torch.nn.Parameter(torch.randn(3, generator=g))
        self.hparams = {'lr': 1e-2, 'optim_eps': 1e-6, 'optim_weight_decay': 1e-3, 'gradient_clip_val': 0.5}


@pytest.mark.parametrize('world', [1, 4])
def test_fused_adam_matches_torch_adam_amsgrad_with_clipping(dev, world):
    from dcsnet.dp import FlatBucket, FusedAdam, TorchAdam
    m1, m2 = _Toy().to(dev), _Toy().to(dev)
    b1, b2 = FlatBucket(m1), FlatBucket(m2)
    kw = dict(lr=1e-2, eps=1e-6, weight_decay=1e-3, max_norm=0.5)
    o1, o2 = FusedAdam(b1, **kw), TorchAdam(b2, **kw)
    g = torch.Generator(device='cpu').manual_seed(1)
    for it in range(5):
        grad = (torch.randn(b1.numel, generator=g) * (3.0 if it % 2 else 0.01)).to(dev)   # clipped / not clipped
        for b in (b1, b2):
            b.zero_grad()
            for p, o in zip(b.params, b.offsets):
                p.grad.copy_(grad[o:o + p.numel()].view(p.shape))
        o1.step(world)
        o2.step(world)
        for p, q in zip(b1.params, b2.params):
            assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), it
    assert o1.t == 5


def test_train_step_tracks_the_oracle(dev):
    """Three optimisation steps, dropout off: the loss trajectory of the HIP path follows the CPU
    oracle's.  (Parameters are not compared element-wise: Adam's first update is lr * sign(g), so
    weights whose gradient is rounding noise — e.g. biases in front of a batch norm — legitimately
    move in either direction.)"""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
    ref = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), 2).train()
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    noisy = clean + noise
    ts = TrainStep(net)
    opt = torch.optim.Adam(ref.parameters(), lr=hp['lr'], eps=hp['optim_eps'], weight_decay=hp['optim_weight_decay'],
                           amsgrad=True)
    batch = (noise.to(dev), noisy.to(dev), clean.to(dev), [0, 1])
    got, want = [], []
    for _ in range(3):
        got.append(float(ts(batch)))
        opt.zero_grad()
        loss = dcs_train_losses(ref, noise, noisy, clean)[2]
        loss.backward()
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 100.0)
        opt.step()
        want.append(float(loss))
    for a, b in zip(got, want):
        assert abs(a - b) <= 2e-3 * abs(b) + 1e-3, (got, want)
    assert net.decoder_attention[12].fc[0].conv_r.weight.grad is None       # never run, never updated
    sd = net.state_dict()
    assert int(sd['encoder.0.1.num_batches_tracked']) == 3


def test_train_step_with_reference_dropout_runs_and_is_finite(dev):
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    net = C_NETWORK(config, hparams, 0).to(dev).train()
    clean, noise = seeded_input(4, 256, 64, 1, 0.1), seeded_input(4, 256, 64, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1, 2, 3])
    ts = TrainStep(net)
    losses = [float(ts(batch)) for _ in range(4)]
    assert all(l == l and abs(l) < 1e4 for l in losses), losses
    assert torch.isfinite(ts.bucket.flat).all()


def test_graph_replayed_train_step_matches_eager(dev):
    """hipGraph capture of forward + losses + backward + fused optimizer: same loss trajectory as eager
    launches (dropout off), update count and dropout seed advance on the device under replay."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    runs = []
    for use_graph in (False, True):
        net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
        ts = TrainStep(net, use_graph=use_graph, graph_warmup=2)
        losses = [float(ts(batch)) for _ in range(6)]
        runs.append((losses, ts))
    (eager, ts_e), (graph, ts_g) = runs
    assert ts_g._graph is not None, 'capture did not happen (fell back to eager)'
    for a, b in zip(eager, graph):
        assert abs(a - b) <= 1e-3 * abs(a) + 1e-3, (eager, graph)
    assert int(ts_g.opt.t_dev) == 6 and int(ts_e.opt.t_dev) == 6
    assert int(ts_g.seed_state) == 6
    assert torch.allclose(ts_g.bucket.flat, ts_e.bucket.flat, atol=5e-4)
    # num_batches_tracked of all fourteen CBNs rides the step's counter launch (dcs_step_advance_counters): one per step, eager
    # or replayed; a training forward outside the step driver still counts by itself
    for ts in (ts_e, ts_g):
        sd = ts.net.state_dict()
        assert {int(v) for k, v in sd.items() if k.endswith('num_batches_tracked') and
                (k.startswith('encoder') or k.startswith('initial') or k[:9] in ('decoder.0', 'decoder.5'))} == {6}
    ts_g.net(batch[1])
    assert int(ts_g.net.state_dict()['encoder.3.1.num_batches_tracked']) == 7


def test_graph_replay_draws_fresh_dropout_masks(dev):
    from dcsnet import ops
    x = torch.ones(1 << 16, device=dev)
    state = torch.zeros(1, dtype=torch.int64, device=dev)
    old = ops.SEED_STATE
    ops.SEED_STATE = state
    try:
        ops.dropout(x, 0.5, 3)                       # warm-up outside capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = ops.dropout(x, 0.5, 3)
            state += 1
        g.replay()
        a = y.clone()
        g.replay()
        b = y.clone()
        assert not torch.equal(a, b)                 # the device-side seed offset advanced
        assert abs(float((a != 0).float().mean()) - 0.5) < 0.02
    finally:
        ops.SEED_STATE = old


def test_backward_kernels_write_into_the_flat_gradient_bucket(dev):
    """With a FlatBucket attached, conv / CBN / attention backward write parameter gradients straight into
    the bucket (no per-parameter accumulate kernels), and the result equals plain autograd accumulation."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import FlatBucket
    from dcsnet import functional
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    x = seeded_input(2, 256, 32, 3).to(dev)

    def grads(with_bucket):
        net = fill_state(C_NETWORK(config, hp, 0), 4).to(dev).train()
        bucket = FlatBucket(net) if with_bucket else None
        if bucket:
            bucket.zero_grad()
        out = net(x)
        (out.real ** 2 + 0.3 * out.imag).sum().backward()
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    before = functional.sink_hits
    a = grads(True)
    assert functional.sink_hits - before > 150          # ~190 conv / CBN / attention parameter tensors
    b = grads(False)
    assert a.keys() == b.keys()
    for n in a:           # same math, different fp32 summation order in places (LSTM bias: per-sequence partial sums)
        scale = float(b[n].abs().max())
        assert float((a[n] - b[n]).abs().max()) <= 2e-5 * scale + 1e-7, n


def test_inference_constants_follow_training_updates(dev):
    """Eval-mode constants kept between forward passes (CBN coefficients, stacked LSTM operands, packed weights) must not
    survive a train step that rewrites parameters and running statistics behind torch's version counters (fused Adam,
    in-kernel running-statistic updates, a replayed train-step graph): eval -> train -> eval equals a fresh eval."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    import bench
    for use_graph in (False, True):
        net = fill_state(C_NETWORK(config, hparams, 0), 7).to(dev)
        noise, noisy, clean = bench.synthetic_stft_batch(2, 32, dev, seed=5)
        x = seeded_input(2, 256, 32, 9).to(dev)

        def ev():
            net.eval()
            with torch.no_grad():
                return net(x).clone()
        e0, e0b = ev(), ev()
        assert torch.equal(e0, e0b)                      # cached constants: same result
        net.train()
        ts = TrainStep(net, use_graph=use_graph, graph_warmup=1)
        for _ in range(3):
            ts((noise, noisy, clean, [0, 1]))
        e1 = ev()
        assert not torch.allclose(e1, e0, atol=1e-6)     # parameters and running statistics moved
        ref = C_NETWORK(config, hparams, 0).to(dev)      # same state, nothing cached
        ref.load_state_dict(net.state_dict())
        ref.eval()
        with torch.no_grad():
            want = ref(x)
        assert torch.allclose(e1, want, atol=1e-6), float((e1 - want).abs().max())


def test_pack_plan_step_is_bit_identical_to_per_layer_packing(dev):
    """dcs_pack_plan_*: every weight re-layout of a step replayed in one launch per dependency level gives
    exactly the parameters the ~200 per-layer pack launches give (same kernels bodies, same inputs)."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    runs = []
    for use_plan in (False, True):
        net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
        ts = TrainStep(net, use_graph=False, use_pack_plan=use_plan)
        losses = [float(ts(batch)) for _ in range(4)]
        runs.append((losses, ts))
    (l0, ts0), (l1, ts1) = runs
    assert ts0._plan is None and ts1._plan is not None
    jobs, launches = ts1._plan.stats()
    assert jobs > 80 and launches <= 6, (jobs, launches)
    assert len(ts1._plan.fwd) > 40 and len(ts1._plan.bwd) > 20
    assert l0 == l1
    assert torch.equal(ts0.bucket.flat, ts1.bucket.flat)


@pytest.mark.parametrize('graph', [False, True], ids=['uncaptured', 'graph'])
def test_weight_relayout_on_its_own_stream_changes_nothing(dev, graph):
    """dp.TrainStep(pack_fork=True): the step's weight re-layout runs on a side stream beside the target synthesis and the
    initial CBN and is joined before the first convolution — same parameters, same losses as the in-line order, eager and
    replayed (dropout on: the two runs draw the same seeds from the same seed state)."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    runs = []
    for fork in (False, True):
        net = fill_state(C_NETWORK(config, hparams, 0), 2).to(dev).train()
        ts = TrainStep(net, use_graph=graph, graph_warmup=1, pack_fork=fork)
        if graph:
            losses = [float(ts(batch)) for _ in range(5)]
        else:                                  # the graph's launches issued eagerly (the first call records the plan)
            losses = [float(ts(batch))] + [float(ts.uncaptured_step(batch)) for _ in range(4)]
        assert '_dcs_pack_fork' not in net.__dict__
        assert ('_pack_stream' in net.__dict__) == fork
        runs.append((losses, ts))
    (l0, ts0), (l1, ts1) = runs
    assert l0 == l1
    assert torch.equal(ts0.bucket.flat, ts1.bucket.flat)


def test_graph_replay_skips_the_update_on_a_nan_loss(dev):
    """c_network.py:257-261: a NaN loss skips the update.  Under hipGraph replay there is no host test: the loss's NaN
    flag stays on the device and turns the fused optimizer launch into a no-op (parameters, Adam moments and the
    update count unchanged), while the dropout stream still advances."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    net = fill_state(C_NETWORK(config, hparams, 0), 2).to(dev).train()
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    good = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    bad_noisy = (clean + noise).clone()
    bad_noisy[1, 7, 3] = complex(float('nan'), 0.0)
    bad = (good[0], bad_noisy.to(dev), good[2], [0, 1])
    ts = TrainStep(net, use_graph=True, graph_warmup=1)
    for _ in range(3):
        ts(good)
    assert ts._graph is not None
    snap = [t.clone() for t in (ts.bucket.flat, ts.opt.m, ts.opt.v, ts.opt.vmax)]
    t_before, seed_before = int(ts.opt.t_dev), int(ts.seed_state)
    loss = ts(bad)
    assert loss != loss                                                  # the NaN loss is reported, not hidden
    for a, b in zip(snap, (ts.bucket.flat, ts.opt.m, ts.opt.v, ts.opt.vmax)):
        assert torch.equal(a, b)
    assert int(ts.opt.t_dev) == t_before and int(ts.seed_state) == seed_before + 1
    loss = ts(good)
    assert float(loss) == float(loss)
    assert not torch.equal(snap[0], ts.bucket.flat) and torch.isfinite(ts.bucket.flat).all()
    assert int(ts.opt.t_dev) == t_before + 1
    # the eager path takes the same decision on the host
    ts_e = TrainStep(fill_state(C_NETWORK(config, hparams, 0), 2).to(dev).train())
    before = ts_e.bucket.flat.clone()
    assert ts_e(bad) is None and torch.equal(before, ts_e.bucket.flat) and int(ts_e.opt.t_dev) == 0


def test_partial_batch_after_capture_runs_eagerly(dev):
    """The reference's DataLoader has no drop_last (config.py:66-69): an epoch ends on a smaller batch.  A captured graph
    holds the captured shapes, so that batch must not be broadcast into the static buffers — it runs eagerly."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    clean, noise = seeded_input(3, 256, 32, 1, 0.1), seeded_input(3, 256, 32, 2, 0.05)
    full = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1, 2])
    part = tuple(t[:1].contiguous() for t in full[:3]) + ([0],)
    runs = []
    for use_graph in (False, True):
        net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
        ts = TrainStep(net, use_graph=use_graph, graph_warmup=1)
        losses = [float(ts(b)) for b in (full, full, full, part, full)]
        runs.append((losses, ts))
    (eager, ts_e), (graph, ts_g) = runs
    assert ts_g._graph is not None
    for a, b in zip(eager, graph):
        assert abs(a - b) <= 1e-3 * abs(a) + 1e-3, (eager, graph)
    assert abs(eager[3] - eager[2]) > 1e-4          # the partial batch has its own loss (not a broadcast of sample 0 x3)
    assert torch.allclose(ts_g.bucket.flat, ts_e.bucket.flat, atol=5e-4)
    l1, l2 = ts_g(full), ts_g(full)
    assert l1.data_ptr() != l2.data_ptr()           # each call returns its own loss tensor


def test_rnetwork_graph_replayed_train_step_matches_eager(dev):
    """DR-Net through TrainStep's CAPTURED step (_loss_no_sync hard-coded dtype='complex', so capture always failed
    for R_NETWORK and fell back to eager with a warning): the graph must really be captured and follow the eager trajectory."""
    import sys
    import warnings
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from dcsnet.dp import TrainStep
    from oracle.seeded_state import fill_state_stream
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    argv = sys.argv
    sys.argv = ['train.py', 'drs', '0']
    try:
        runs = []
        for use_graph in (False, True):
            net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
            ts = TrainStep(net, use_graph=use_graph, graph_warmup=2)
            with warnings.catch_warnings():
                warnings.simplefilter('error')                      # a capture failure warns: make it fail here
                losses = [float(ts(batch)) for _ in range(5)]
            runs.append((losses, ts))
    finally:
        sys.argv = argv
    (eager, _), (graph, ts_g) = runs
    assert ts_g._graph is not None, 'capture did not happen (fell back to eager)'
    for a, b in zip(eager, graph):
        assert abs(a - b) <= 1e-3 * abs(a) + 1e-3, (eager, graph)


def test_rnetwork_train_steps_do_not_depend_on_what_ran_before(dev):
    """Round 5: dp.TrainStep replayed its pack plan for DR-Net as well — whose packed panels derive from temporaries (paired
    filters, flipped / sliced copies), so the replayed jobs read and wrote freed memory: the second step of the FIRST TrainStep of
    a process differed from later ones by 1e-3 and the graph / eager comparison above failed once in a full run.  The plan is now
    used only by networks that declare pack_plan_safe (C_NETWORK): two DR-Net TrainSteps in one process give the same bits."""
    import sys
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    from oracle.seeded_state import fill_state_stream
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    clean, noise = seeded_input(2, 256, 32, 1, 0.1), seeded_input(2, 256, 32, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    argv = sys.argv
    sys.argv = ['train.py', 'drs', '0']
    try:
        runs = []
        for _ in range(3):
            net = fill_state_stream(R_NETWORK(config, hp, 0), 5).to(dev).train()
            ts = TrainStep(net, use_graph=False)
            runs.append([float(ts(batch)) for _ in range(4)])
            assert ts._plan is None and not ts.use_pack_plan
    finally:
        sys.argv = argv
    assert runs[0] == runs[1] == runs[2], runs
    assert getattr(C_NETWORK, 'pack_plan_safe', False) and not getattr(R_NETWORK, 'pack_plan_safe', False)


@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'graph'])
def test_inference_side_stream_overlap_is_bit_identical(dev, graph):
    """C_NETWORK.overlap_skip_attention (inference: the batched skip attentions on a side stream beside the LSTM, joined
    BEFORE the fc conv — profiles/r03_pk_fma_op_sel_hazard.txt) against the single-stream pass, bit for bit, at a shape where
    the attentions outlast the LSTM (long frequency axis, 8 LSTM steps: the side stream is still busy at the join), eager
    and captured."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    net = fill_state(C_NETWORK(config, hparams, 0), 4).to(dev).eval()
    x = seeded_input(16, 256, 32, seed=9).to(dev)
    outs = {}
    for overlap in (False, True):
        net.overlap_skip_attention = overlap

        def run():
            with torch.no_grad():
                return net(x)
        run(); run()
        torch.cuda.synchronize()
        if graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                static = run()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            outs[overlap] = static.clone()
        else:
            outs[overlap] = run().clone()
            torch.cuda.synchronize()
    assert torch.equal(outs[False], outs[True])
    assert bool(torch.isfinite(torch.view_as_real(outs[True])).all())


def test_train_step_lstm_gemms_stay_in_tree_at_any_size(dev):
    """at the bf16 B = 64 shape the LSTM projections (2.1 GFLOP per launch) crossed DCS_LSTM_GEMM_MAX_GFLOP and went to
    the library GEMM — rocBLAS kernels (and ATen copies) nobody scans for the packed-FMA erratum, co-resident with the side
    stream's weight-gradient MFMA waves.  The cap now applies to inference launches only: a TRAIN launch of the shape family
    dcs_gemm_f32 takes is in-tree whatever its size."""
    from dcsnet import ops
    M, N, K = 2 * 64 * 64, 1024, 128                                   # configs[4]'s per-GPU share: 8192 rows x [128 -> 2 x 512]
    assert 2e-9 * M * N * K > ops.LSTM_GEMM_MAX_GFLOP
    assert not ops.gemm_ok(M, N, K) and ops.gemm_ok(M, N, K, 1, True)
    assert ops.gemm_ok(M, K, 512, 2, True) and not ops.gemm_ok(M, 100, K, 1, True)      # (the shape rules still hold)


@pytest.mark.parametrize('storage', ['f32', 'bf16'])
@pytest.mark.parametrize('graph', [False, True], ids=['eager', 'graph'])
def test_side_stream_weight_gradients_are_bit_identical(dev, graph, storage):
    """TrainStep.wgrad_side_stream (Round 4): the weight-gradient kernels and their slab reductions on a side stream beside
    the data-gradient chain — one fork per conv layer, one join in front of the flush — against the same step on one stream.
    Same kernels on the same operands, so losses, gradients, parameters and moments agree BIT FOR BIT after several updates
    with the reference's dropout on (a race on an operand or a slab recycled too early would show as a difference), eager
    and captured; the batch is large enough for the multi-slab weight-gradient plans."""
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    clean, noise = seeded_input(4, 256, 64, 1, 0.1), seeded_input(4, 256, 64, 2, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1, 2, 3])
    runs = []
    for side in (False, True):
        torch.manual_seed(0)
        net = fill_state(C_NETWORK(config, dict(hparams), 0), 2).to(dev).train()
        from dcsnet import ops
        mode0 = ops.conv_precision()
        if storage == 'bf16':                                          # (the bf16-storage step runs the same two streams)
            net.set_activation_dtype('bf16')
        try:
            ts = TrainStep(net, use_graph=graph, graph_warmup=2)
            ts.wgrad_side_stream = side
            losses = [float(ts(batch)) for _ in range(5)]
            torch.cuda.synchronize()
        finally:
            if storage == 'bf16':
                ops.set_conv_precision(mode0)                          # (one conv precision per process: hand it back)
        runs.append((losses, ts.bucket.flat.clone(), ts.bucket.grad.clone() if hasattr(ts.bucket, 'grad') else None, ts))
    (l0, p0, g0, ts0), (l1, p1, g1, ts1) = runs
    if graph:
        assert ts1._graph is not None, 'capture did not happen (fell back to eager)'
    assert l0 == l1, (l0, l1)
    assert torch.equal(p0, p1)
    if g0 is not None:
        assert torch.equal(g0, g1)
    assert all(torch.isfinite(torch.tensor(l1)))
