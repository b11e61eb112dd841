"""bench.py's conv timings with every launch timed (stride 1) vs every 7th (the default): python tools/micro/timer_stride_probe.py"""
import json, sys, subprocess, os
for st in (1, 7):
    env = dict(os.environ, DCS_BENCH_TIMER_STRIDE=str(st))
    out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline'], env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(out.stderr[-2000:]); raise
    r = d['roofline']
    e = r['encoder_stack_forward']
    print('stride', st, 'ms/step', round(d['ms_per_step'], 3), 'launches', r['launches_per_step'], 'samples', r['samples_per_launch'],
          'conv ms', round(r['kernel_ms_per_step'], 3), 'frac', round(r['frac'], 3), 'exec', round(r['executed_frac'], 3),
          'enc', round(e['kernel_ms_per_step'], 4), round(e['frac'], 3), [round(l['us'], 1) for l in e['per_layer']])
