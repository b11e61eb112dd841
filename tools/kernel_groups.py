"""Group a rocprofv3 kernel_stats.csv into families: python tools/kernel_groups.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1
fam = [('conv mfma fwd/dgrad', ('cconv_mfma_kernel', 'cconv_mfma16_kernel', 'splitk_reduce')),
       ('small-channel convs', ('cconv_k7', 'cconv_wgrad_small', 'cconv_small_dgrad', 'cconv_up1', 'cconv_enc0', 'tap_rows', 'csum_')), ('conv direct fwd/dgrad', ('cconv_direct_kernel',)),
       ('wgrad mfma', ('cconv_wgrad_mfma', 'cconv_wgrad_x6')), ('wgrad direct', ('cconv_wgrad_kernel',)), ('wgrad reduce', ('wgrad_reduce',)),
       ('weight packs', ('pack_', 'fold_taps')), ('cbn', ('cbn_',)), ('attention', ('att_', 'attention_', 'ca_', 'spatial_pool', 'sa_')),
       ('lstm', ('lstm_',)), ('mask/loss/synthesis hip', ('bound_', 'mask_', 'polar_', 'crm_', 'tapsum', 'upsample_cat', 'sisnr', 'istft', 'stft_')),
       ('adam', ('adam',)), ('copyBuffer/fill', ('copyBuffer', 'fillBuffer', 'FillFunctor')), ('rocBLAS/gemm', ('Cijk', 'gemm', 'rocblas')),
       ('fft', ('fft', 'real2complex', 'complex2real', 'transpose_kernel')), ('torch other', ('at::', 'at_cuda'))]
acc = {f: [0.0, 0] for f, _ in fam}
acc['other'] = [0.0, 0]
other = []
for r in rows:
    if 'stream_hold_kernel' in r['Name']:        # bench.py's queue-ahead gate: spins on a host flag, not step work
        continue
    for f, keys in fam:
        if any(k in r['Name'] for k in keys):
            break
    else:
        f = 'other'; other.append(r['Name'][:70])
    acc[f][0] += float(r['TotalDurationNs']) / 1e6 / steps
    acc[f][1] += int(r['Calls']) / steps
tot = sum(v[0] for v in acc.values())
for f, (ms, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print(f'{f:24s} {ms:7.3f} ms/step {n:7.1f} launches/step {100*ms/tot:5.1f}%')
print(f'total {tot:.3f} ms/step; {sum(v[1] for v in acc.values()):.0f} launches/step')
if other:
    print('other:', other[:12])
