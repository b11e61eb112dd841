"""Every documented switch that changes which kernels run gets one smoke-level run: three
updates of the train step at [4,256,64] with dropout on plus one inference pass, in a process of its own (C statics and
module-level environment reads are taken once per process), against the same run under the defaults.  The alternatives are
numerically equivalent formulations (other kernels, other summation orders, other streams), so losses, parameters and the mask
agree to fp32 accumulation noise; the stream / scheduling switches must agree exactly.  Tile-plan thresholds are NOT switches of
the shipped library any more (csrc/dcs_common.h: dcs_knob)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, 'tests', '_switch_probe.py')


def _run(env=None, attrs=(), graph=False):
    e = dict(os.environ, **(env or {}), PROBE_GRAPH='1' if graph else '0')
    r = subprocess.run([sys.executable, PROBE, *attrs], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (env, attrs, r.stdout[-2000:], r.stderr[-4000:])
    return json.loads(r.stdout.strip().split('\n')[-1]), r.stderr


@pytest.fixture(scope='module')
def base():
    return _run()[0]


def _same(got, want, rel):
    assert got['finite']
    for a, b in zip(got['losses'], want['losses']):
        assert abs(a - b) <= rel * max(1.0, abs(b)), (got['losses'], want['losses'])
    assert abs(got['pnorm'] - want['pnorm']) <= rel * want['pnorm'], (got['pnorm'], want['pnorm'])
    assert abs(got['mask'] - want['mask']) <= 10 * rel * want['mask'], (got['mask'], want['mask'])


# (what, environment, module attributes, exact?)
SWITCHES = [
    ('native fp32 MFMA everywhere', {'DCS_CONV_PRECISION': '0'}, (), False),
    ('native weight-gradient kernels', {'DCS_WGRAD_X6': '0'}, (), False),
    ('first encoder conv on the fp32 MFMA', {'DCS_ENC0_F32': '1'}, (), False),
    ('cotangent split inside the weight-gradient kernel', {'DCS_WGRAD_PA_MIN_CHUNKS': '999'}, (), False),
    ('producer / consumer conv kernel', {'DCS_CONV_RING': '1', 'DCS_RING_MIN_WG': '1'}, (), False),
    ('16-column kernel: one workgroup per class', {'DCS_CLASS_FUSE': '0'}, (), False),
    ('one stream', {'DCS_WGRAD_SIDE': '0'}, (), True),
    ('every slab reduction as its own launch', {'DCS_WGRAD_DEFER': '0'}, (), False),
    ('CBN statistics by their own kernels', {'DCS_STATS_EPILOGUE': '0'}, (), False),
    ('CBN apply and channel pool as two launches', {'DCS_FUSE_APPLY_POOL': '0'}, (), False),
    ('skip attentions add their pool term themselves', {'DCS_SPLIT_SKIP_POOL': '0'}, (), False),
    ('attention FC weight gradients on the main chain', {'DCS_DEFER_FC': '0'}, (), True),
    ('LSTM projections through the library GEMM', {'DCS_LSTM_GEMM': '0'}, (), False),
    ('inference: skip attentions on the main stream', {'DCS_OVERLAP_SKIP': '0', 'DCS_SKIP_EARLY': '0'}, (), True),
    ('plan trace', {'DCS_MFMA_TRACE': '1'}, (), True),
]


@pytest.mark.parametrize('what,env,attrs,exact', SWITCHES, ids=[s[0].replace(' ', '_') for s in SWITCHES])
def test_documented_switch_gives_the_same_training_run(base, what, env, attrs, exact):
    got, err = _run(env, attrs)
    _same(got, base, 0.0 if exact else 2e-4)
    if 'DCS_MFMA_TRACE' in env:
        assert '[mfma]' in err



def test_no_kernel_reads_memory_nobody_wrote():
    """tools/nanfill_probe.py: C_NETWORK train steps (bf16 storage, fp32, B = 1; captured) and inference with every torch.empty
    filled with NaN (torch.utils.deterministic.fill_uninitialized_memory) print exactly what they print without the fill — a kernel
    that read a buffer element before anything wrote it would turn a loss or a checksum into NaN (round 5: the probe that cleared
    the kernels when DR-Net's replayed pack plan turned out to read freed memory)."""
    probe = os.path.join(ROOT, 'tools', 'nanfill_probe.py')
    outs = []
    for arg in ([], ['nanfill']):
        r = subprocess.run([sys.executable, probe, *arg], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (arg, r.stdout[-2000:], r.stderr[-3000:])
        outs.append([ln for ln in r.stdout.split('\n') if ln.startswith('train') or ln.strip().startswith('eval')])
    assert len(outs[0]) == 6 and outs[0] == outs[1], outs
    assert 'nan' not in ' '.join(outs[1]).lower()
