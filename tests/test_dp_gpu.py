"""GPU, world_size 2: the data-parallel train step on the HIP path — both ranks on the one card of a test box, gradients
exchanged with gloo (the exchange itself is torch.distributed's; RCCL needs one device per rank).  Covers what the CPU
test cannot: with world > 1 the captured hipGraph holds forward + backward only, the bucket all-reduce and the fused
optimizer run after each replay, and every rank ends every step with identical parameters."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    for p in (REPO, os.path.join(REPO, 'dcs-net_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle.seeded_state import fill_state, seeded_input
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    dev = torch.device('cuda:0')
    hp = dict(hparams)
    hp['dropout_conv'], hp['dropout_fc'] = 0.0, 0.0
    net = fill_state(C_NETWORK(config, hp, 0), 2).to(dev).train()
    ts = TrainStep(net, use_graph=True, graph_warmup=2)
    clean, noise = seeded_input(2, 256, 32, 10 + rank, 0.1), seeded_input(2, 256, 32, 20 + rank, 0.05)
    batch = (noise.to(dev), (clean + noise).to(dev), clean.to(dev), [0, 1])
    losses = [float(ts(batch)) for _ in range(5)]
    flat = ts.bucket.flat.detach().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({'same': bool(torch.equal(gathered[0], gathered[1])), 'graph': ts._graph is not None,
                    'graph_world': getattr(ts, '_graph_world', None), 'losses': losses, 't': int(ts.opt.t_dev),
                    'finite': bool(torch.isfinite(flat).all())}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graph_replayed_train_step(tmp_path):
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    out = str(tmp_path / 'dp_gpu.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['graph'] and r['graph_world'] == 2, r
    assert r['same'], 'ranks diverged after 5 data-parallel steps'
    assert r['finite'] and r['t'] == 5
    assert all(l == l for l in r['losses'])


@pytest.mark.parametrize('mode', ['train', 'infer'])
def test_bench_two_rank_rehearsal(mode):
    """bench.py launched exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), with the
    rehearsal switches that put both ranks on the one card and replace RCCL by gloo: the multi-rank control flow
    (warm-up, barriers, graph A -> all-reduce -> graph B, the instrumented pass on every rank, MAX over ranks, one JSON
    line from rank 0) must run to completion — a collective issued by one rank alone hangs the job."""
    import json
    import subprocess
    env = dict(os.environ, DCS_BENCH_DEVICE='0', DCS_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(REPO, 'bench.py'), '--gpus', '2', '--mode', mode, '--steps', '2',
           '--warmup', '1', '--no-cpu-baseline', '--batch', '4', '--frames', '64']
    r = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-500:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['steps'] == 2 and out['value'] > 0 and out['scaling'] == 'weak'
    assert out['config']['global_batch'] == 8 and out['roofline']['achieved'] > 0
