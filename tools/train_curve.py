"""Loss curves of the captured train step over a few hundred updates on a small fixed set of synthetic batches, fp32 activation
storage against bf16 storage (BASELINE configs[4]) — does the bf16 path TRAIN like the fp32 one, not just match one step?
usage (GPU box): python tools/train_curve.py [steps] [B] [T]  ->  gpurun_out/train_curves.json
Both runs start from the same seed-0 weights, see the same batches in the same order and use the reference's optimizer
settings (Adam/AMSGrad lr 1e-4, clip 100: config.py:31,44-47); dropout as in the reference (0.1 / 0.2, counter-based masks:
the two runs draw the same masks)."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
sys.argv, ARGS = [sys.argv[0]], sys.argv[1:]
import bench
from dcsnet import ops
from dcsnet.config import config, hparams
from dcsnet.c_network import C_NETWORK
from dcsnet.dp import TrainStep

steps = int(ARGS[0]) if len(ARGS) > 0 else 300
B = int(ARGS[1]) if len(ARGS) > 1 else 32
T = int(ARGS[2]) if len(ARGS) > 2 else 256
NB = 4                                      # distinct batches, cycled
dev = torch.device('cuda:0')
batches = [bench.synthetic_stft_batch(B, T, dev, seed=100 + i) for i in range(NB)]
out = {'steps': steps, 'B': B, 'T': T, 'distinct_batches': NB}
default = ops.conv_precision()
for mode in ('f32', 'bf16'):
    torch.manual_seed(0)
    ops.set_conv_precision('bf16' if mode == 'bf16' else default)
    net = C_NETWORK(config, hparams, 0).to(dev).train()
    if mode == 'bf16':
        net.set_activation_dtype('bf16')
    ts = TrainStep(net, use_graph=True)
    losses = []
    for s in range(steps):
        noise, noisy, clean = batches[s % NB]
        loss = ts((noise, noisy, clean, list(range(B))))
        losses.append(None if loss is None else float(loss))
    torch.cuda.synchronize()
    out[mode] = losses
    ok = [l for l in losses if l is not None]
    print(f'{mode}: first 4 {[round(l, 4) for l in ok[:4]]}  last 4 {[round(l, 4) for l in ok[-4:]]}  skipped {len(losses) - len(ok)}', flush=True)
    del ts, net
ops.set_conv_precision(default)
f, h = out['f32'], out['bf16']
pairs = [(a, b) for a, b in zip(f, h) if a is not None and b is not None]
out['max_abs_loss_gap'] = max(abs(a - b) for a, b in pairs)
out['mean_abs_loss_gap_last_50'] = sum(abs(a - b) for a, b in pairs[-50:]) / max(1, len(pairs[-50:]))
print(f"max |loss_f32 - loss_bf16| = {out['max_abs_loss_gap']:.4f}; mean over the last 50 steps = {out['mean_abs_loss_gap_last_50']:.4f}")
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'train_curves.json'), 'w'))
