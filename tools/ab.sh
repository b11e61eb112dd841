# usage: ab.sh "<ENV for B>"  — runs default (A) and variant (B), train + infer
set -e
V="$1"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_A_train.json 2>gpurun_out/ab.err
env $V python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_B_train.json 2>gpurun_out/ab.err
python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_A_infer.json 2>gpurun_out/ab.err
env $V python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_B_infer.json 2>gpurun_out/ab.err
python - <<PY
import json
for m in ("train","infer"):
    for n in ("A","B"):
        j=json.loads(open("gpurun_out/ab_%s_%s.json"%(n,m)).read().strip().splitlines()[-1])
        e=j["roofline"].get("encoder_stack_forward") or {}
        print(m, n, round(j["ms_per_step"],4), e.get("frac_vs_f32_mfma_peak"), [round(l["us"],1) for l in e.get("per_layer",[])])
PY
