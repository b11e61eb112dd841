"""Copy one tools/collect_profiles.sh result set into profiles/ under its tag and print the numbers the docs quote:
    python tools/install_profiles.py <tag> [old_tag_to_remove]"""
import csv, json, os, shutil, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(REPO, 'gpurun_out', tag)
dst = os.path.join(REPO, 'profiles')
if len(sys.argv) > 2:
    for f in os.listdir(dst):
        if f.startswith(sys.argv[2] + '_'):
            os.remove(os.path.join(dst, f))
for a, b in (('bench_train_default.json', 'bench_train_default.json'), ('bench_infer.json', 'bench_infer.json'),
             ('train_kernel_stats.csv', 'train_b32_t256_kernel_stats.csv'), ('infer_kernel_stats.csv', 'infer_b16_t2000_kernel_stats.csv')):
    shutil.copy(os.path.join(src, a), os.path.join(dst, f'{tag}_{b}'))
for a in ('mfma_peak.txt', 'conv_layers_train.txt', 'conv_layers_infer.txt', 'lstm_bench.txt',
          'step_sequence_train.txt', 'step_sequence_infer.txt', 'conv_layers_train_native.txt', 'conv_layers_infer_native.txt',
          'conv_precision.txt', 'train_native_kernel_stats.csv', 'infer_native_kernel_stats.csv', 'bench_train_bf16_b64.json',
          'bench_train_f32_b64.json', 'step_sequence_train_bf16_b64.txt', 'train_side_stream_kernel_stats.csv',
          'step_timeline_train.txt'):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, f'{tag}_{a}'))
for mode, label in (('train', 'train B=32 T=256'), ('infer', 'infer B=16 T=2000')):
    c = os.path.join(src, f'pmc_{mode}_mfma.csv')
    if os.path.exists(c):
        out = subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pmc_mfma_util.py'), c,
                              f'{label}: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -- python3 '
                              f'bench.py --mode {mode} --no-graph --steps 2 --warmup 1 --no-cpu-baseline'],
                             capture_output=True, text=True).stdout
        open(os.path.join(dst, f'{tag}_pmc_mfma_util_{mode}.txt'), 'w').write(out)
commit = subprocess.run(['git', 'log', '-1', '--format=%s'], cwd=REPO, capture_output=True, text=True).stdout.strip()[:80]
subprocess.run([sys.executable, os.path.join(REPO, 'tools', 'pmc_family_traffic.py'), src, tag, commit], stdout=subprocess.DEVNULL)
for m in ('train_default', 'infer'):
    d = json.load(open(os.path.join(src, f'bench_{m}.json')))
    r = d['roofline']
    e = r.get('encoder_stack_forward', {})
    print(f"{m}: {d['ms_per_step']:.3f} ms, {d['value']:.0f} frames/s, conv {r['achieved']:.1f} TF frac {r['frac']:.3f} "
          f"(executed {r['executed_frac']:.3f}), conv ms/step {r['kernel_ms_per_step']:.3f} over {r['launches_per_step']} calls, "
          f"encoder stack {e.get('achieved', 0):.1f} TF frac {e.get('frac', 0):.3f} ({e.get('kernel_ms_per_step', 0) * 1e3:.1f} us), "
          f"traffic {(r['traffic'] or 0) / 1e9:.2f} GB, cpu {d['cpu_baseline'] and d['cpu_baseline']['value']:.0f}")
keys = ('cconv_', 'splitk_reduce', 'wgrad_reduce', 'tapsum', 'tap_rows_scatter', 'csum_')
for mode, per in (('train', 27.0), ('infer', None)):
    rows = [r for r in csv.DictReader(open(os.path.join(src, f'{mode}_kernel_stats.csv')))
            if 'stream_hold_kernel' not in r['Name']]      # the measuring pass's parking kernel spins while the host enqueues
    if per is None:
        per = float(sum(int(r['Calls']) for r in rows if 'bound_mask_apply_kernel' in r['Name']))      # one per pass
    t = sum(float(r['TotalDurationNs']) for r in rows if any(k in r['Name'] for k in keys)) / per / 1e6
    n = sum(int(r['Calls']) for r in rows if any(k in r['Name'] for k in keys)) / per
    tot = sum(float(r['TotalDurationNs']) for r in rows) / per / 1e6
    nk = sum(int(r['Calls']) for r in rows) / per
    print(f'{mode}: rocprof conv family {t:.3f} ms over {n:.0f} kernels per step ({per:.0f} steps in file); all kernels {tot:.3f} ms, {nk:.0f} per step')
print(open(os.path.join(dst, f'{tag}_pmc_hbm_traffic.txt')).read().split('\n== ')[0].split('\n')[-2])
for line in open(os.path.join(dst, f'{tag}_pmc_hbm_traffic.txt')):
    if line.startswith('conv family') or line.startswith('adam'):
        print(line.strip())
