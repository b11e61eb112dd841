"""HBM bytes per step per kernel family from the four --pmc passes of tools/collect_profiles.sh:
    python tools/pmc_family_traffic.py gpurun_out/<tag> <tag> "<commit subject>"
writes profiles/<tag>_pmc_hbm_traffic.txt and profiles/traffic.json (bench.py reads the conv-family totals from it).
Counters are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream (x2); WRITE_SIZE is exact
(MI355X_MICROARCH.md, HBM section)."""
import collections, csv, json, os, sys

src, tag, commit = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else '')
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAM = ['cconv_mfma', 'cconv_wgrad_mfma', 'cconv_wgrad_x6', 'cconv_wgrad_sa', 'cconv_wgrad_small', 'cconv_small_dgrad', 'cconv_direct', 'cconv_k7',
       'wgrad_reduce', 'splitk_reduce', 'tapsum', 'tap_rows', 'pack_', 'cbn_', 'att_', 'ca_', 'spatial_pool', 'attention_apply',
       'lstm', 'adam', 'polar_frames', 'istft_ola', 'sisnr', 'bound_']
CONV = ('cconv_', 'wgrad_reduce', 'splitk_reduce', 'tapsum', 'tap_rows')
STEPS = 3.0                                   # --steps 2 --warmup 1, eager


def load(path, counter):
    d, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name']] += float(r['Counter_Value']); n[r['Kernel_Name']] += 1
    return d, n


out, res = [], {}
for mode, label in (('train', 'train (B=32, T=256)'), ('infer', 'infer (B=16, T=2000)')):
    f, nf = load(os.path.join(src, f'pmc_{mode}_FETCH_SIZE.csv'), 'FETCH_SIZE')
    w, nw = load(os.path.join(src, f'pmc_{mode}_WRITE_SIZE.csv'), 'WRITE_SIZE')
    out.append(f'== {label}, 3 eager steps per pass')
    for k in FAM:
        lf = sum(v for n_, v in nf.items() if k in n_)
        if not lf:
            continue
        fb = sum(v for n_, v in f.items() if k in n_) * 2048 / STEPS
        wb = sum(v for n_, v in w.items() if k in n_) * 1024 / STEPS
        lw = sum(v for n_, v in nw.items() if k in n_)
        out.append(f'{k}: launches {lf}/{lw}; per step: read {fb / 1e6:.1f} MB (FETCH_SIZE x2), write {wb / 1e6:.1f} MB, '
                   f'total {(fb + wb) / 1e6:.1f} MB')
    conv = (sum(v for n_, v in f.items() if any(k in n_ for k in CONV)) * 2048 +
            sum(v for n_, v in w.items() if any(k in n_ for k in CONV)) * 1024) / STEPS
    tot = (sum(f.values()) * 2048 + sum(w.values()) * 1024) / STEPS
    out.append(f'conv family (cconv_* + split-K / wgrad reduces + tapsum + tap_rows) = {conv / 1e6:.0f} MB/step; '
               f'all kernels {tot / 1e6:.0f} MB/step')
    res[mode] = {'conv_family_hbm_bytes_per_step': round(conv, -5), 'all_kernels_hbm_bytes_per_step': round(tot, -5),
                 'source': f'profiles/{tag}_pmc_hbm_traffic.txt',
                 'config': ('B=32,T=256' if mode == 'train' else 'B=16,T=2000') + (f', commit "{commit}"' if commit else '')}
out += ['', '# how: tools/collect_profiles.sh + tools/pmc_family_traffic.py — two separate passes per mode (TCC slots: FETCH_SIZE,',
        '# WRITE_SIZE), each with --kernel-trace only:',
        '#   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --mode <m> --no-graph --steps 2 --warmup 1 --no-cpu-baseline',
        '#   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -- (same)',
        '# 3 steps per pass; KiB -> bytes x1024; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B; MI355X_MICROARCH.md, HBM section).',
        '# calibration: adam must read 105 MB/step (36 B x 2.91 M parameters).',
        '# train step 0 of each pass records the pack plan (per-layer pack launches), so pack_ is above its steady-state value.']
# MFMA busy fraction of the conv families from the same collection's SQ pass (tools/pmc_mfma_util.py wrote the summary)
import re
for mode in ('train', 'infer'):
    try:
        txt = open(os.path.join(REPO, 'profiles', f'{tag}_pmc_mfma_util_{mode}.txt')).read()
    except OSError:
        continue
    busy = {}
    for key, pat in (('fwd_dgrad', r'forward / data gradient\): util ([0-9.]+)'), ('wgrad', r'weight gradient\): util ([0-9.]+)'),
                     ('all_mfma_kernels', r'all MFMA kernels: util ([0-9.]+)')):
        m_ = re.search(pat, txt)
        if m_:
            busy[key] = float(m_.group(1))
    if busy:
        busy['source'] = f'profiles/{tag}_pmc_mfma_util_{mode}.txt (SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES))'
        res[mode]['mfma_busy'] = busy
open(os.path.join(REPO, 'profiles', f'{tag}_pmc_hbm_traffic.txt'), 'w').write('\n'.join(out) + '\n')
json.dump(res, open(os.path.join(REPO, 'profiles', 'traffic.json'), 'w'), indent=1)
print('\n'.join(out))
