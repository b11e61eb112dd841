"""CPU-only: the C-ABI library loads and exports every symbol include/dcsnet_hip.h declares, the
ctypes signatures cover exactly those symbols, and the host-side mirror of the reference's
surface (module names, state_dict keys, config literals, argument checks) is intact.
No compute call is made here — there is no GPU."""
import os
import sys
import re
import ctypes
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, 'include', 'dcsnet_hip.h')


def _declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dcs_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from dcsnet import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build the library first: python dcs-net_amd/build.py'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/dcsnet_hip.h but not exported'


def test_ctypes_signatures_cover_the_header():
    from dcsnet import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.dcs_abi_version() >= 1
    assert lib.dcs_error_string(0) == b'ok'
    assert b'workspace' in lib.dcs_error_string(-3)


def test_host_side_geometry_queries():
    """The two workspace-size entry points are pure host arithmetic: callable without a GPU."""
    from dcsnet import _lib
    lib = _lib.load()
    assert lib.dcs_cbn_workspace_bytes(32 * 128 * 128, 8) > 0
    assert lib.dcs_cbn_workspace_bytes(100, 1) > 0
    assert lib.dcs_cbn_workspace_bytes(100, 3) < 0          # odd channel counts are not on the path
    assert lib.dcs_ca_workspace_bytes(4, 1000, 128) > 0
    assert lib.dcs_ca_workspace_bytes(4, 1000, 1) < 0


def test_null_and_bad_arguments_are_rejected_before_any_launch():
    from dcsnet import _lib
    lib = _lib.load()
    assert lib.dcs_bound_crm_fwd(None, None, 10, 1e-6, None) == -1
    assert lib.dcs_cconv2d_fwd(None, None, None, None, None, None, 0, 1, 1, 1, 1, 0, 1, 1, 1, 3, 3, 1, 1, 1, 1, 0, None) == -1
    assert lib.dcs_dropout_fwd(None, None, 0, 0.1, 1, None, None) == -1


def test_ops_refuse_cpu_tensors():
    from dcsnet import ops, DcsHipError
    with pytest.raises(DcsHipError, match='no CPU fallback'):
        ops.bound_crm(torch.zeros(8, 2))
    with pytest.raises(DcsHipError):
        ops.cbn(torch.zeros(1, 2, 2, 8, 2), None, None, None, None)


def test_c_network_surface_matches_reference_contract():
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK, ComplexLSTM, ComplexChannelAttention, ComplexSpatialAttention  # noqa
    from oracle.cnet_oracle import C_NETWORK_Oracle
    net = C_NETWORK(config, hparams, 0)
    assert list(net.state_dict().keys()) == list(C_NETWORK_Oracle().state_dict().keys())
    assert sum(p.numel() for p in net.parameters()) == 2912707
    kids = [n for n, _ in net.named_children()]
    assert kids[:4] == ['encoder', 'decoder', 'decoder_attention', 'skip_attention']   # c_network.py:95-98
    for hook in ('forward', 'configure_optimizers', 'training_step', 'validation_step', 'validation_epoch_end',
                 'test_step', 'test_epoch_end', 'on_after_backward', 'weights_init'):
        assert callable(getattr(net, hook))
    sd = net.state_dict()
    assert sd['encoder.0.1.running_mean'].dtype == torch.complex64
    assert sd['encoder.0.1.running_covar'].shape == (8, 3)
    assert sd['decoder.0.0.conv_tran_r.weight'].shape == (256, 128, 3, 3)
    assert sd['skip_attention.0.fc.0.conv_r.weight'].shape == (8, 128, 1, 1)
    assert torch.allclose(sd['encoder.2.1.weight'][:, :2], torch.full((32, 2), 2 ** 0.5))
    opt = net.configure_optimizers()
    adam = opt['optimizer']
    g = adam.param_groups[0]
    assert (g['lr'], g['eps'], g['weight_decay'], g['amsgrad']) == (1e-4, 1e-6, 1e-4, True)   # config.py:31,44-47


def test_state_dict_round_trip_with_oracle():
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from oracle.cnet_oracle import C_NETWORK_Oracle
    from oracle.seeded_state import fill_state
    o = fill_state(C_NETWORK_Oracle(), 3)
    net = C_NETWORK(config, hparams, 0)
    net.load_state_dict(o.state_dict())                 # the reference's checkpoint layout loads as is
    for k, v in net.state_dict().items():
        assert torch.equal(v, o.state_dict()[k]), k


def test_rnetwork_state_dict_round_trip_with_oracle():
    """DR-Net: the oracle (pinned against the reference's r_network.py) and the HIP module share every state_dict key,
    shape and dtype, so a reference checkpoint loads; the HIP forward refuses CPU tensors (no fallback)."""
    from dcsnet.config import config, hparams
    from dcsnet.r_network import R_NETWORK
    from dcsnet import DcsHipError
    from oracle.rnet_oracle import R_NETWORK_Oracle
    from oracle.seeded_state import fill_state_stream
    o = fill_state_stream(R_NETWORK_Oracle(), 5)
    net = R_NETWORK(config, hparams, 0)
    assert list(net.state_dict().keys()) == list(o.state_dict().keys())
    net.load_state_dict(o.state_dict())
    for k, v in net.state_dict().items():
        assert torch.equal(v, o.state_dict()[k]), k
    with pytest.raises(DcsHipError):
        with torch.no_grad():
            net.eval()(torch.zeros(1, 256, 32))


def test_product_path_never_touches_the_oracle_and_has_no_cpu_fallback():
    """The oracle is test infrastructure: no file of the product package (or of the drop-in shims, bench's product half
    excepted by contract) may import it, and importing the package must not pull it in; every op refuses CPU tensors."""
    import re, subprocess
    pkg = os.path.join(REPO, 'dcs-net_amd')
    pat = re.compile(r'^\s*(from|import)\s+oracle\b', re.M)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                assert not pat.search(open(os.path.join(root, f)).read()), os.path.join(root, f)
    code = ('import sys; sys.path.insert(0, %r); import dcsnet, dcsnet.c_network, dcsnet.r_network, dcsnet.dp, dcsnet.frontend; '
            'assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), "oracle imported"' % pkg)
    subprocess.run([sys.executable, '-c', code], check=True, cwd=REPO)
    from dcsnet import ops, DcsHipError
    with pytest.raises(DcsHipError):
        ops.bound_crm(torch.zeros(4, 2))


def test_forward_validates_input():
    from dcsnet.config import config, hparams
    from dcsnet.c_network import C_NETWORK
    from dcsnet import DcsHipError
    net = C_NETWORK(config, hparams, 0)
    with pytest.raises(DcsHipError):
        net(torch.zeros(2, 256, 12, dtype=torch.complex64))      # T % 8 != 0
    with pytest.raises(DcsHipError):
        net(torch.zeros(2, 256, 16))                             # not complex


def test_dropin_names_resolve():
    import sys
    d = os.path.join(REPO, 'dcs-net_amd', 'dropin')
    sys.path.insert(0, d)
    try:
        for m in ('c_network', 'network_functions', 'config', 'complexPyTorch.complexLayers',
                  'complexPyTorch.complexFunctions'):
            sys.modules.pop(m, None)
        import c_network as cn
        import config as cfg
        from complexPyTorch.complexLayers import ComplexConv2d, ComplexBatchNorm2d    # noqa: F401
        from complexPyTorch.complexFunctions import complex_upsample, complex_relu      # noqa: F401
        assert cfg.hparams['channels'] == [1, 16, 32, 64, 128, 256, 256, 256]
        assert cfg.config.strideE[3] == (2, 1) and cfg.config.upsample_scale_factor[4] == (2, 2)
        for name in ('C_NETWORK', 'ComplexLSTM', 'bound_cRM', 'cRM', 'complex_mat_mult', 'SiSNR',
                     'train_batch_2_loss', 'ComplexConv2d', 'complex_upsample', 'torch'):
            assert hasattr(cn, name), name
    finally:
        sys.path.remove(d)
        for m in ('c_network', 'network_functions', 'config', 'complexPyTorch', 'complexPyTorch.complexLayers',
                  'complexPyTorch.complexFunctions'):
            sys.modules.pop(m, None)


def test_crop_batch_follows_the_dataset_rule():
    """dcsnet.frontend.crop_batch = data.py:87-104 per item: equal lengths enforced, short items zero-padded, otherwise
    one random window shared by the clean and the noisy signal."""
    from dcsnet.frontend import crop_batch
    g = torch.Generator().manual_seed(0)
    clean = [torch.arange(100.), torch.arange(40.), torch.arange(64.)]
    noisy = [c + 1000 for c in clean]
    c, n = crop_batch(clean, noisy, 64, generator=g)
    assert c.shape == n.shape == (3, 64)
    assert torch.equal(n[0] - c[0], torch.full((64,), 1000.)) and float(c[0][1] - c[0][0]) == 1.0
    assert 0 <= float(c[0][0]) < 36
    assert torch.equal(c[1][:40], torch.arange(40.)) and float(c[1][40:].abs().sum()) == 0 and float(n[1][40:].abs().sum()) == 0
    assert torch.equal(c[2], torch.arange(64.))
    with pytest.raises(Exception):
        crop_batch([torch.zeros(10)], [torch.zeros(11)], 8)


def test_tile_split_float_reciprocal_is_exact_below_2_22():
    """conv_enc0.hip's persistent kernels split a linear tile index into (image, tile row, tile column) with float
    reciprocals instead of integer divisions; the launch refuses >= 2**22 tiles.  Same arithmetic in float32 here."""
    import numpy as np
    rng = np.random.default_rng(0)
    for per in (1, 3, 68, 69, 544, 2176, 100003, (1 << 22) - 1):
        for tw in (1, 3, 4, 32, 33):
            if tw > per:
                continue
            tiles = np.unique(np.concatenate([rng.integers(0, 1 << 22, 50000), np.arange(0, min(1 << 22, per * 20)),
                                              (1 << 22) - 1 - np.arange(500)])).astype(np.int64)
            inv_per, inv_w = np.float32(1.0) / np.float32(per), np.float32(1.0) / np.float32(tw)
            b = ((tiles.astype(np.float32) + np.float32(0.5)) * inv_per).astype(np.int32)
            tl = tiles - b.astype(np.int64) * per
            ty = ((tl.astype(np.float32) + np.float32(0.5)) * inv_w).astype(np.int32)
            tx = tl - ty * tw
            assert np.array_equal(b, tiles // per) and np.array_equal(ty, (tiles % per) // tw)
            assert np.array_equal(tx, (tiles % per) % tw)


def test_calc_loss_every_noise_loss_type_against_the_reference(golden_dir):
    """network_functions.py:168-208: noise_loss_type 0-6, values recorded from the reference's own calc_loss
    (oracle/make_golden.py::loss_vectors).  Type 0 is the L1 of COMPLEX masks = mean |a - b|."""
    import types
    import numpy as np
    from dcsnet import network_functions as nf
    d = np.load(os.path.join(golden_dir, 'loss_vectors.npz'))
    kw = {k: torch.from_numpy(d[k]) for k in ('noisy_audio', 'noise_audio', 'clean_audio', 'predict_noise_audio',
                                              'predict_clean_audio', 'target_noise_mask', 'predict_noise_mask')}
    cfg = types.SimpleNamespace(L1=torch.nn.L1Loss(), mse=torch.nn.MSELoss(), SiSNR=nf.SiSNR(), wSDR=nf.wSDR())
    argv = sys.argv
    sys.argv = ['train.py', 'dcs', '0']
    try:
        for t in range(7):
            me = types.SimpleNamespace(hparams={'noise_loss_type': t, 'speech_loss_type': 0, 'speech_alpha': 0.7}, config=cfg)
            got = [float(v) for v in nf.calc_loss(me, **kw)]
            want = d[f'type{t}']
            for g, w in zip(got, want):
                assert abs(g - w) <= 1e-5 * max(1.0, abs(w)), (t, got, want)
        with pytest.raises(ValueError):
            nf.calc_loss(types.SimpleNamespace(hparams={'noise_loss_type': 9, 'speech_loss_type': 0, 'speech_alpha': 0.7},
                                               config=cfg), **kw)
    finally:
        sys.argv = argv


def test_stoi_restatement_properties():
    """dcsnet/metrics.py::stoi (the published algorithm with pystoi 0.3.3's constants; PARITY UNPINNED: pystoi is not in
    the image and the reference holds no STOI vectors): identity scores 1, the score falls monotonically with the SNR,
    is invariant to the level of the processed signal, keeps pystoi's too-short-signal convention, and the band matrix
    has the published layout."""
    import numpy as np
    from dcsnet.metrics import stoi, thirdoct, resample_oct
    fs = 16000
    t = np.arange(3 * fs) / fs
    x = sum(np.sin(2 * np.pi * f * t) * (0.5 + 0.5 * np.sin(2 * np.pi * (3 + i) * t))
            for i, f in enumerate((180, 360, 720, 1500, 2800, 4200)))
    x = x * (np.sin(2 * np.pi * 1.5 * t) > -0.3)                      # pauses: exercises the silent-frame removal
    assert abs(stoi(x, x, fs) - 1.0) < 1e-6
    rng = np.random.default_rng(0)
    noise = rng.standard_normal(len(x))
    noise *= np.linalg.norm(x) / np.linalg.norm(noise)
    scores = [stoi(x, x + noise / 10 ** (snr / 20), fs) for snr in (20, 10, 0, -10)]
    assert all(a > b for a, b in zip(scores, scores[1:])) and 0.3 < scores[-1] < scores[0] < 1.0, scores
    assert abs(stoi(x, 3.7 * (x + noise), fs) - stoi(x, x + noise, fs)) < 1e-9
    assert stoi(x[:2000], x[:2000], fs) == 1e-5                       # fewer than 30 frames
    obm, cf = thirdoct(10000, 512, 15, 150)
    assert obm.shape == (15, 257) and abs(cf[0] - 150) < 1e-9 and abs(cf[-1] - 150 * 2 ** (14 / 3)) < 1e-6
    assert (obm.sum(0) <= 1).all() and obm.sum() > 0                 # bands do not overlap
    y = resample_oct(np.sin(2 * np.pi * 440 * t), 10000, fs)          # 16 kHz -> 10 kHz keeps a 440 Hz tone
    k = int(np.argmax(np.abs(np.fft.rfft(y))))
    assert len(y) == 3 * 10000 and abs(k * 10000 / len(y) - 440) < 1.0
    with pytest.raises(ValueError):
        stoi(x, x[:-1], fs)


# Kernels allowed to carry a packed-f32 op with an op_sel bit set.  EMPTY by construction since round 4: every source but
# two is built without the SLP vectoriser (build.py), the hand-written packed complex MACs (conv_k7.hip, conv_up1.hip,
# conv_wgrad_small.hip) keep the value they broadcast in the LOW half of a register pair or move it (dcs_bcast2), and the two
# sources that keep the vectoriser (conv_wgrad_mfma.hip, lstm.hip: it pays there) form no selecting pair.  An entry here needs
# its reason written next to it.
PACKED_F32_OP_SEL_ALLOWED = ()


def test_no_cross_half_packed_f32_in_any_kernel(tmp_path):
    """Static guard for the gfx950 behaviour pinned in profiles/r03_pk_fma_op_sel_hazard.txt: a v_pk_fma_f32 whose LOW lane
    takes the HIGH dword of a source pair (an op_sel bit set) can lose that lane's product while co-resident waves issue
    bf16 MFMAs.  Round 3 kept such ops out of the kernels that issue MFMAs themselves and relied on stream discipline for the
    97 VALU kernels that carried them; since round 4 NO kernel of the built library may contain one, so
    co-residency — two ranks on one card, a second stream — cannot meet the condition at all."""
    import re
    import struct
    import subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not available')
    from dcsnet import _lib
    data = open(_lib.LIB_PATH, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    n_obj = n_kern = n_mfma = 0
    bad = []
    for m in re.finditer(magic, data):
        p = m.start()
        q = p + len(magic)
        cnt = struct.unpack_from('<Q', data, q)[0]
        q += 8
        for _ in range(cnt):
            off, size, tl = struct.unpack_from('<QQQ', data, q)
            q += 24
            triple = data[q:q + tl].decode()
            q += tl
            if 'gfx950' not in triple or size == 0:
                continue
            blob = data[p + off:p + off + size]
            f = tmp_path / f'co{n_obj}.o'
            f.write_bytes(blob)
            n_obj += 1
            asm = subprocess.run([objdump, '-d', '--mcpu=gfx950', str(f)], capture_output=True, text=True, check=True).stdout
            parts = re.split(r'\n[0-9a-f]+ <([^>]+)>:\n', asm)
            for name, body in zip(parts[1::2], parts[2::2]):
                n_kern += 1
                n_mfma += 'v_mfma' in body
                hits = [ln for ln in body.split('\n')
                        if re.search(r'v_pk_(fma|mul|add)_f32', ln) and re.search(r'op_sel:\[[^\]]*1', ln)]
                if hits and not any(a in name for a in PACKED_F32_OP_SEL_ALLOWED):
                    bad.append((name, len(hits), hits[0].strip()))
    assert n_obj > 0 and n_kern > 300 and n_mfma > 20, (n_obj, n_kern, n_mfma)
    assert not bad, f'{len(bad)} kernels carry a packed f32 op with a cross-half operand selection: {bad[:8]}'


def test_conv_kernel_fragment_loads_stay_whole(tmp_path):
    """Static guard (Round 4): the emulated forward / data-gradient kernel reads an A fragment with ONE ds_read_b128.  When the
    tap loop was restructured, the optimiser lost sight of the fragments' 16-byte alignment behind a loop phi and emitted a
    third of them as pairs of ds_read2_b32 (two instructions, two-way bank conflicts: profiles/r04_conv_probes.txt) — silently,
    results unchanged.  The instances the two networks launch without a statistics epilogue must hold no ds_read2_b32 at all;
    the emulated weight-gradient kernel must keep its transposed reads and, in the pre-split form, whole 16-byte g_Y loads."""
    import re
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'kernel_isa.py')
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        pytest.skip('llvm-objdump not available')
    from dcsnet import _lib

    def isa(pattern):
        out = tmp_path / 'k.isa'
        subprocess.run([sys.executable, tool, _lib.LIB_PATH, pattern, str(out)], check=True)
        return {b.split('\n', 1)[0]: b for b in re.split(r'^=== ', out.read_text(), flags=re.M)[1:]}

    fwd = isa('cconv_mfma_kernel<')
    checked = 0
    for inst in ('<2, 2, 1, 32, 2, 1, false, 2>', '<2, 2, 1, 16, 2, 1, false, 2>', '<2, 1, 1, 32, 2, 1, false, 1>',
                 '<2, 1, 1, 32, 2, 1, false, 2>', '<2, 1, 1, 16, 2, 1, false, 1>', '<2, 2, 1, 8, 2, 1, false, 2>'):
        body = [b for n, b in fwd.items() if 'cconv_mfma_kernel' + inst in n]
        assert len(body) == 1, inst
        assert 'v_mfma_f32_32x32x16_bf16' in body[0] and body[0].count('ds_read_b128') >= 20, inst
        assert body[0].count('ds_read2_b32') == 0, (inst, body[0].count('ds_read2_b32'))
        checked += 1
    wg = isa('cconv_wgrad_x6_kernel<2, 3, 2, 1, false, true>')
    assert len(wg) == 1
    body = next(iter(wg.values()))
    assert body.count('ds_read_b64_tr_b16') >= 200 and body.count('global_load_dwordx4') >= 48
    assert 'global_load_dword ' not in body.replace('global_load_dwordx', 'X')
    assert checked == 6


def test_ring_kernel_load_destinations_are_untouched_until_their_wait(tmp_path):
    """Static guard for csrc/conv_ring.hip (Round 5): its producer waves issue their patch loads as inline asm (hipcc would
    otherwise wait vmcnt(0) for an ordinary load beside the LDS-DMA requests in flight and drain the B ring every step) and count
    them by hand.  hipcc does not know those destination registers are pending: a loop-carried definition or a wait under a
    branch made it COPY them (v_mov_b64) in front of the hand-written s_waitcnt — garbage on some launches, silently.  Between
    every such load and the next `s_waitcnt vmcnt` no instruction may name its destination registers."""
    import re
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'kernel_isa.py')
    if not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'):
        pytest.skip('llvm-objdump not available')
    from dcsnet import _lib
    out = tmp_path / 'ring.isa'
    subprocess.run([sys.executable, tool, _lib.LIB_PATH, 'cconv_ring_kernel<', str(out)], check=True)
    kernels = re.split(r'^=== ', out.read_text(), flags=re.M)[1:]
    assert len(kernels) >= 8, len(kernels)
    checked = 0
    for body in kernels:
        lines = body.split('\n')
        name = lines[0]
        assert 'global_load_lds_dwordx4' in body and 'v_mfma_f32_32x32x16_bf16' in body, name
        # the steady-state loads: `global_load_dwordx4 v[a:b], vN, s[c:d]` with NO offset / modifiers behind a scalar base and
        # followed (before any other wait) by the hand-written counted wait
        for i, ln in enumerate(lines):
            m = re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]\s*(//.*)?$', ln)
            if not m:
                continue
            lo, hi = int(m.group(1)), int(m.group(2))
            regs = set(range(lo, hi + 1))
            for ln2 in lines[i + 1:]:
                if 's_waitcnt' in ln2 and 'vmcnt' in ln2:
                    break
                if 'global_load_dwordx4' in ln2 and re.search(r'global_load_dwordx4 v\[(\d+):(\d+)\]', ln2):
                    continue                                   # (the next load of the same burst defines its own registers)
                used = set()
                for a, b in re.findall(r'v\[(\d+):(\d+)\]', ln2):
                    used |= set(range(int(a), int(b) + 1))
                used |= {int(a) for a in re.findall(r'\bv(\d+)\b', ln2)}
                assert not (used & regs), (name, ln.strip(), ln2.strip())
            checked += 1
    assert checked >= 16, checked


def test_bf16_operand_flag_follows_explicit_choices_and_the_process_preset():
    """ADVICE r4 (dcsnet/c_network.py:_check_conv_precision): conv precision 'bf16' counts as chosen ON PURPOSE when the caller set
    it (ops.set_conv_precision) or the process was preset with DCS_CONV_PRECISION=1 — and a later set_activation_dtype('bf16')
    (which sets the mode as a side effect) must not take that choice back; only a switch away from 'bf16' does."""
    import subprocess
    import sys
    from dcsnet import ops
    mode0, flag0 = ops.conv_precision(), ops.BF16_OPERANDS_ON_PURPOSE
    try:
        ops.set_conv_precision('bf16x6')
        assert ops.BF16_OPERANDS_ON_PURPOSE is False
        ops._IN_SET_ACTIVATION_DTYPE = True                    # as C_NETWORK.set_activation_dtype('bf16') does
        ops.set_conv_precision('bf16')
        ops._IN_SET_ACTIVATION_DTYPE = False
        assert ops.BF16_OPERANDS_ON_PURPOSE is False           # a side effect is not a choice
        ops.set_conv_precision('bf16')
        assert ops.BF16_OPERANDS_ON_PURPOSE is True
        ops._IN_SET_ACTIVATION_DTYPE = True
        ops.set_conv_precision('bf16')
        ops._IN_SET_ACTIVATION_DTYPE = False
        assert ops.BF16_OPERANDS_ON_PURPOSE is True            # ... and does not undo one
        ops.set_conv_precision('f32')
        assert ops.BF16_OPERANDS_ON_PURPOSE is False
    finally:
        ops._IN_SET_ACTIVATION_DTYPE = False
        ops.set_conv_precision(mode0)
        ops.BF16_OPERANDS_ON_PURPOSE = flag0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); from dcsnet import ops; "
            "print(ops.BF16_OPERANDS_ON_PURPOSE, ops.conv_precision())" % (root, os.path.join(root, 'dcs-net_amd')))
    for preset, want in (('1', 'True bf16'), ('2', 'False bf16x6')):
        r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, DCS_CONV_PRECISION=preset), capture_output=True,
                           text=True, check=True)
        assert r.stdout.strip().endswith(want), (preset, r.stdout, r.stderr)
