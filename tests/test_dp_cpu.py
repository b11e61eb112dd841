"""CPU, world_size 2, gloo: the data-parallel host logic of dcsnet/dp.py — flat bucket, one
gradient all-reduce, identical parameters on every rank after the step, BatchNorm statistics
kept local.  The model run here is the CPU oracle (same parameter names as the HIP C_NETWORK);
the optimizer is the torch reference (TorchAdam): the fused HIP Adam is covered by the GPU tests.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _OracleAdapter(torch.nn.Module):
    """Gives the oracle net the two attributes TrainStep touches: hparams and training_step."""

    def __init__(self, seed):
        super().__init__()
        from oracle.cnet_oracle import C_NETWORK_Oracle
        from oracle.seeded_state import fill_state
        self.net = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed)
        self.hparams = {'lr': 1e-4, 'optim_eps': 1e-6, 'optim_weight_decay': 1e-4, 'gradient_clip_val': 100.0}

    def named_parameters(self, *a, **k):                # same names as C_NETWORK
        return self.net.named_parameters(*a, **k)

    def training_step(self, batch, idx):
        from oracle.nf_oracle import dcs_train_losses
        noise, noisy, clean = batch[:3]
        loss = dcs_train_losses(self.net, noise, noisy, clean)[2]
        if torch.any(torch.isnan(loss)):                 # the reference's guard (c_network.py:257-261)
            return None
        return loss


def _batch(rank, B=2, T=16):
    from oracle.seeded_state import seeded_input
    clean = seeded_input(B, 256, T, 10 + rank, 0.1)
    noise = seeded_input(B, 256, T, 20 + rank, 0.05)
    return noise, clean + noise, clean


def _worker(rank, world, port, out):
    for p in (REPO, os.path.join(REPO, 'dcs-net_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dcsnet.dp import TrainStep, TorchAdam
    model = _OracleAdapter(seed=3)
    ts = TrainStep(model, optimizer_cls=TorchAdam)
    assert ts.bucket.numel >= 2912707 - 3000 and len(ts.bucket.params) == len(ts.bucket.names)
    assert not any(n.startswith('decoder_attention.12') or n.startswith('decoder_attention.13') for n in ts.bucket.names)

    # local gradient (no all-reduce) for the expectation
    ts.bucket.zero_grad()
    model.training_step(_batch(rank), 0).backward()
    local = ts.bucket.grad.clone()
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    mean_grad = sum(gathered) / world
    # TorchAdam then clips the averaged bucket in place (Trainer gradient_clip_val = 100)
    mean_grad = mean_grad * min(1.0, 100.0 / (float(mean_grad.norm()) + 1e-6))

    # reload the same initial state, then one real data-parallel step
    model2 = _OracleAdapter(seed=3)
    ts2 = TrainStep(model2, optimizer_cls=TorchAdam)
    before = ts2.bucket.flat.clone()
    loss = ts2(_batch(rank), 0)
    averaged = ts2.bucket.grad.clone()          # TorchAdam divided the summed bucket by world in place
    after = ts2.bucket.flat.clone()
    bn_mean = model2.net.encoder[0][1].running_mean.clone()

    others = [torch.empty_like(after) for _ in range(world)]
    dist.all_gather(others, after)
    bns = [torch.empty_like(bn_mean) for _ in range(world)]
    dist.all_gather(bns, bn_mean)
    if rank == 0:
        torch.save({'grad_err': float((averaged - mean_grad).abs().max()),
                    'grad_scale': float(mean_grad.abs().max()),
                    'param_spread': float((others[0] - others[1]).abs().max()),
                    'moved': float((after - before).abs().max()),
                    'bn_diff': float((bns[0] - bns[1]).abs().max()),
                    'loss': float(loss),
                    'views_ok': all(p.data_ptr() == ts2.bucket.flat.data_ptr() + 4 * o
                                    for p, o in zip(ts2.bucket.params, ts2.bucket.offsets))}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_data_parallel_step(tmp_path):
    out = str(tmp_path / 'res.pt')
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['views_ok']                                     # parameters live in the flat bucket
    # all-reduce == mean of the local gradients (two separate CPU backward passes: threaded
    # reductions are not bitwise repeatable, hence 1e-3 rather than 1e-6)
    assert r['grad_err'] <= 1e-3 * max(r['grad_scale'], 1.0)
    assert r['param_spread'] == 0.0                          # identical weights on both ranks after the step
    assert 0 < r['moved'] <= 1.5e-4                          # Adam's first step moves each weight by ~lr
    assert r['bn_diff'] > 0                                  # BatchNorm statistics stay local
    assert r['loss'] == r['loss']


def _nan_worker(rank, world, port, out):
    for p in (REPO, os.path.join(REPO, 'dcs-net_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from dcsnet.dp import TrainStep, TorchAdam
    model = _OracleAdapter(seed=3)
    ts = TrainStep(model, optimizer_cls=TorchAdam)
    before = ts.bucket.flat.clone()
    noise, noisy, clean = _batch(rank)
    if rank == 1:                                        # ONE rank's shard produces a NaN loss
        noisy = noisy.clone()
        noisy[0, 3, 5] = complex(float('nan'), 0.0)
    r1 = ts((noise, noisy, clean), 0)                    # must not hang: every rank joins the all-reduce
    after_nan = ts.bucket.flat.clone()
    r2 = ts(_batch(rank), 1)                             # the next (clean) step pairs its collective correctly
    after_ok = ts.bucket.flat.clone()
    others = [torch.empty_like(after_ok) for _ in range(world)]
    dist.all_gather(others, after_ok)
    res = {'skipped': r1 is None, 'unchanged': bool(torch.equal(before, after_nan)), 'stepped': r2 is not None,
           'moved': float((after_ok - after_nan).abs().max()), 'finite': bool(torch.isfinite(after_ok).all()),
           'spread': float((others[0] - others[1]).abs().max())}
    torch.save(res, f'{out}.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_nan_loss_on_one_rank_skips_the_step_on_every_rank(tmp_path):
    """The reference skips an update whose loss is NaN (c_network.py:257-261).  Data-parallel, the decision must be
    global and every rank must still issue the step's collective: the NaN flag rides the gradient all-reduce."""
    out = str(tmp_path / 'nan')
    mp.spawn(_nan_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for rank in range(2):
        r = torch.load(f'{out}.{rank}')
        assert r['skipped'], rank                            # BOTH ranks report the skipped step
        assert r['unchanged'], rank                          # no parameter moved, NaN gradients never reached Adam
        assert r['stepped'] and 0 < r['moved'] <= 1.5e-4 and r['finite'], (rank, r)
        assert r['spread'] == 0.0


def test_flat_bucket_preserves_values_and_gradient_views():
    from dcsnet.dp import FlatBucket, hot_parameters
    m = _OracleAdapter(seed=1)
    ref = {n: p.detach().clone() for n, p in m.named_parameters()}
    b = FlatBucket(m)
    for n, p in hot_parameters(m):
        assert torch.equal(p.detach(), ref[n]), n
    b.zero_grad()
    p0 = b.params[0]
    (p0 * 2).sum().backward()
    assert float(b.grad[:p0.numel()].sum()) == 2.0 * p0.numel()          # autograd wrote into the bucket
    assert all(o % 4 == 0 for o in b.offsets)                # float4-aligned slices for the fused kernel


def test_train_step_side_stream_switch_and_cpu_backward(monkeypatch):
    """TrainStep.wgrad_side_stream (the weight-gradient kernels on a side stream of the captured step) is a host-side switch:
    on by default, off with DCS_WGRAD_SIDE=0, and without a GPU the backward pass neither creates a stream nor opens a
    deferred-reduce scope (the oracle model's step still runs)."""
    from dcsnet import functional, ops
    from dcsnet.dp import TrainStep, TorchAdam
    monkeypatch.delenv('DCS_WGRAD_SIDE', raising=False)
    m = _OracleAdapter(seed=2)
    ts = TrainStep(m, optimizer_cls=TorchAdam)
    assert ts.wgrad_side_stream is True
    monkeypatch.setenv('DCS_WGRAD_SIDE', '0')
    assert TrainStep(_OracleAdapter(seed=2), optimizer_cls=TorchAdam).wgrad_side_stream is False
    p0 = ts.bucket.params[0]
    ts.bucket.zero_grad()
    ts._backward((p0 * p0).sum())
    assert functional.WGRAD_SIDE is None and ops.WGRAD_DEFER is None and '_wgrad_side' not in ts.__dict__
    assert torch.allclose(p0.grad, 2 * p0.detach())


def test_bench_gpus_flag_launches_its_own_ranks_and_refuses_a_mismatch():
    """`python bench.py --gpus N` with no launcher around it must start N rank processes itself (before any GPU call) and
    relay a line that saw N ranks; under a launcher whose WORLD_SIZE differs from --gpus it must fail, not print a line
    labelled with the wrong n_gpus (missing #4).  DCS_BENCH_LAUNCH_ONLY=1 runs the launch plumbing over gloo
    without the GPU workload."""
    import json
    import subprocess
    import sys
    bench = os.path.join(REPO, 'bench.py')
    env = dict(os.environ, DCS_BENCH_LAUNCH_ONLY='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, bench, '--gpus', '2'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['config']['world_seen'] == 2
    r = subprocess.run([sys.executable, bench, '--gpus', '4'], env=dict(env, WORLD_SIZE='2', RANK='0'), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and 'WORLD_SIZE=2' in r.stderr
