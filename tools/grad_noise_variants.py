"""NOTE (round 5): the DCS_MFMA_* variants below need a library built with -DDCS_PLAN_KNOBS (tools/exp_build.py; csrc/dcs_common.h: dcs_knob)
exported through DCS_LIB_PATH; the shipped library ignores those variables.
How much of a gradient tensor's distance to the fp64 oracle at the bench size is NOISE?  The same train step
([32,256,256], dropout off) is evaluated by numerically equivalent variants — the HIP path under different kernel plans /
arithmetic modes (environment toggles, one process each) and the CPU fp32 oracle with different thread counts (different
reduction orders) — and every tensor's relative L2 error against the fp64 oracle is listed per variant.
usage (GPU box): python tools/grad_noise_variants.py            -> gpurun_out/grad_noise_variants.json
internal:        python tools/grad_noise_variants.py --hip <out.pt>   (one HIP evaluation with the current environment)"""
import json, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dcs-net_amd'))
from oracle.seeded_state import fill_state, seeded_input
B, T, seed = 32, 256, 3
clean, noise = seeded_input(B, 256, T, 1, 0.1), seeded_input(B, 256, T, 2, 0.05)
noisy = clean + noise

if len(sys.argv) > 2 and sys.argv[1] == '--hip':
    sys.argv = sys.argv[:1]
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet.dp import TrainStep
    dev = torch.device('cuda:0')
    hp = dict(hparams); hp['dropout_conv'] = hp['dropout_fc'] = 0.0
    net = fill_state(C_NETWORK(config, hp, seed), seed).to(dev).train()
    net.hparams['lr'] = 0.0; net.hparams['optim_weight_decay'] = 0.0
    ts = TrainStep(net)
    loss = float(ts((noise.to(dev), noisy.to(dev), clean.to(dev), list(range(B)))))
    torch.save({'loss': loss, 'g': {n: (None if p.grad is None else p.grad.detach().cpu().double())
                                    for n, p in net.named_parameters()}}, os.environ['DCS_NOISE_OUT'])
    sys.exit(0)


def oracle(dtype, threads):
    from oracle.cnet_oracle import C_NETWORK_Oracle
    from oracle.nf_oracle import dcs_train_losses
    from oracle import cpt_oracle, nf_oracle
    torch.set_num_threads(threads)
    cpt_oracle.CDTYPE = nf_oracle.CDTYPE = torch.complex128 if dtype == torch.float64 else torch.complex64
    ref = fill_state(C_NETWORK_Oracle({'dropout_conv': 0.0, 'dropout_fc': 0.0}), seed).train()
    c = lambda z: z.to(torch.complex128) if dtype == torch.float64 else z
    if dtype == torch.float64:
        ref = ref.double()
    loss = dcs_train_losses(ref, c(noise), c(noisy), c(clean))[2]
    loss.backward()
    return {n: (None if p.grad is None else p.grad.detach().double()) for n, p in ref.named_parameters()}


ncpu = min(32, len(os.sched_getaffinity(0)))
g64 = oracle(torch.float64, ncpu)
print('fp64 oracle done', flush=True)
variants = {}
for th in (ncpu, 1):
    variants[f'cpu_fp32_{th}_threads'] = oracle(torch.float32, th)
    print(f'cpu fp32 oracle, {th} threads done', flush=True)
hip_envs = {'hip_default': {}, 'hip_native_f32_mfma': {'DCS_CONV_PRECISION': '0'}, 'hip_enc1_two_waves': {'DCS_MFMA_ENC1_WK4': '0'},
            'hip_no_wk64': {'DCS_MFMA_WK64': '0'}, 'hip_no_wk': {'DCS_MFMA_WK': '0'}, 'hip_stats_kernels': {'DCS_STATS_EPILOGUE': '0'},
            'hip_stats_kernels_native': {'DCS_STATS_EPILOGUE': '0', 'DCS_CONV_PRECISION': '0'},
            'hip_stats_kernels_enc1_two_waves': {'DCS_STATS_EPILOGUE': '0', 'DCS_MFMA_ENC1_WK4': '0'},
            'hip_stats_kernels_no_wk': {'DCS_STATS_EPILOGUE': '0', 'DCS_MFMA_WK': '0'}}
for name, env in hip_envs.items():
    out = f'/tmp/noise_{name}.pt'
    e = dict(os.environ, DCS_NOISE_OUT=out, **env)
    subprocess.run([sys.executable, os.path.abspath(__file__), '--hip', out], env=e, check=True, stderr=subprocess.DEVNULL)
    variants[name] = torch.load(out)['g']
    print(f'{name} done', flush=True)
skip = lambda n: n.endswith('.0.conv_r.bias') or n.endswith('.0.conv_i.bias') or 'conv_tran_r.bias' in n or 'conv_tran_i.bias' in n
res = {}
for n, w in g64.items():
    if w is None or float(w.norm()) == 0.0:
        continue
    res[n] = {v: float((g[n] - w).norm() / w.norm()) for v, g in variants.items() if g.get(n) is not None}
worst = sorted(res.items(), key=lambda kv: -max(kv[1].values()))[:12]
names = list(variants)
print('rel-L2 error vs the fp64 oracle, twelve worst tensors (max over the variants):')
print(' | '.join(f'{v[:18]:>18}' for v in names))
for n, d in worst:
    print(' | '.join(f'{d.get(v, float("nan")):18.2e}' for v in names), n)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump({'variants': names, 'rel_l2_vs_fp64': res}, open(os.path.join(ROOT, 'gpurun_out', 'grad_noise_variants.json'), 'w'))
