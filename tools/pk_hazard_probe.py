#!/usr/bin/env python3
"""Controlled experiment on the wrong-result of round 2 (conv_mfma.hip epilogue, folded eval-mode CBN): which instruction
sequence loses the q1 * im product beside co-resident bf16-MFMA workgroups?

  python tools/pk_hazard_probe.py --build      (CPU: hipcc) variant libraries -> dcs-net_amd/lib/exp/libdcsnet_hip_epiN.so
  python tools/pk_hazard_probe.py [--iters K]  (GPU) each variant in its own child process, K launches of the failing
                                               geometry (enc5 forward at the inference bench shape), compared element by
                                               element with the native-fp32-MFMA result of the shipped library

Variants (DCS_EXP_EPI in conv_mfma.hip):
  0  shipped: scalar v_fma_f32 kept apart by empty asm statements
  1  compiler-paired: v_pk_add_f32 (bias) directly followed by v_pk_fma_f32 ... op_sel:[0,1,0] (the form that failed)
  3  that instruction sequence written by hand in one asm block (exact spacing)
  4  the same with `s_nop 1` between the v_pk_add_f32 and the op_sel v_pk_fma_f32
  5  the same v_pk_add_f32 followed by scalar v_fma_f32 (control)
  6  packed FMAs WITHOUT any cross-half selection (operands broadcast into register pairs by v_mov first)
  7  first FMA pair scalar, second pair = v_pk_fma_f32 ... op_sel_hi:[1,0,1] (the HIGH lane reading the LOW half)
  8  only the op_sel:[0,1,0] v_pk_fma_f32, 16 wait states away from the producer of its operands and from its consumer
"""
import argparse
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'dcs-net_amd')
EXP = os.path.join(PKG, 'lib', 'exp')
VARIANTS = (0, 1, 3, 4, 5, 6, 7, 8)


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location('dcsnet_build', os.path.join(PKG, 'build.py'))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()                                                    # the shipped objects
    os.makedirs(EXP, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = [os.path.join(b.OBJ, s[:-4] + '.o') for s in b._sources() if s != 'conv_mfma.hip']
    for v in VARIANTS:
        if v == 0:
            continue
        obj = os.path.join(EXP, f'conv_mfma_epi{v}.o')
        subprocess.run([hipcc] + b.FLAGS + [f'-DDCS_EXP_EPI={v}', '-c', os.path.join(b.CSRC, 'conv_mfma.hip'), '-o', obj],
                       check=True, capture_output=True)
        lib = os.path.join(EXP, f'libdcsnet_hip_epi{v}.so')
        subprocess.run([hipcc, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', '-o', lib, obj] + objs, check=True,
                       capture_output=True)
        print('built', lib)


def child(iters):
    sys.path.insert(0, PKG)
    import torch
    from dcsnet import ops, functional as F
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(9)
    B, H, W, C, k, st = 16, 8, 250, 128, 3, (2, 1)
    x = torch.randn(B, H, W, C, 2, generator=g).to(dev)
    w_r, w_i = (torch.randn(C, C, k, k, generator=g) * 0.05).to(dev), (torch.randn(C, C, k, k, generator=g) * 0.05).to(dev)
    b_r, b_i = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    coef = torch.randn(C, 6, generator=g).to(dev)
    ops.set_conv_precision('f32')
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
    ref = ops.cconv2d(x, None, wp, bias, (k, k), st, (1, 1), (1, 1), F.ACT_NONE, coef=coef)
    raw = ops.cconv2d(x, None, wp, bias, (k, k), st, (1, 1), (1, 1), F.ACT_NONE)          # conv + bias, no affine map
    ops.set_conv_precision('bf16x6')
    wp, bias = ops.pack_conv_weight(w_r, w_i, b_r, b_i, False, (1, 1))
    scale = float(ref.abs().max())
    bad_total, bad_launches, lost_q1 = 0, 0, 0
    part = [0, 0]
    lanes = {}
    for _ in range(iters):
        y = ops.cconv2d(x, None, wp, bias, (k, k), st, (1, 1), (1, 1), F.ACT_NONE, coef=coef)
        bad = (y - ref).abs() > 1e-4 * scale
        n = int(bad.sum())
        if n:
            bad_launches += 1
            bad_total += n
            idx = bad.nonzero()
            # does the wrong value equal q0 * re + q4 (real part) i.e. the q1 * im product missing?
            for b_, h_, w_, c_, p_ in idx[:2000].tolist():
                q = coef[c_]
                re, im = float(raw[b_, h_, w_, c_, 0]), float(raw[b_, h_, w_, c_, 1])
                got = float(y[b_, h_, w_, c_, p_])
                part[p_] += 1
                want_missing = (q[0] * re + q[4]) if p_ == 0 else (q[2] * re + q[5])
                if abs(got - float(want_missing)) <= 1e-4 * scale:
                    lost_q1 += 1
                lanes[(c_ // 2) % 8 + 8 * ((h_ * W + w_) % 8)] = lanes.get((c_ // 2) % 8 + 8 * ((h_ * W + w_) % 8), 0) + 1
    torch.cuda.synchronize()
    print(json.dumps({'iters': iters, 'outputs_per_launch': ref.numel(), 'launches_with_wrong_values': bad_launches,
                      'wrong_values': bad_total, 'of_which_equal_to_the_map_without_q1_im': lost_q1,
                      'wrong_real_parts_low_lane': part[0], 'wrong_imag_parts_high_lane': part[1]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--build', action='store_true')
    ap.add_argument('--iters', type=int, default=60)
    ap.add_argument('--child', action='store_true')
    a = ap.parse_args()
    if a.build:
        return build()
    if a.child:
        return child(a.iters)
    for v in VARIANTS:
        env = dict(os.environ)
        if v:
            env['DCS_LIB_PATH'] = os.path.join(EXP, f'libdcsnet_hip_epi{v}.so')
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', '--iters', str(a.iters)], env=env,
                           capture_output=True, text=True, timeout=600)
        out = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        print(f'variant {v}:', out[-1] if out else f'FAILED rc={r.returncode} {r.stderr[-500:]}', flush=True)


if __name__ == '__main__':
    main()
