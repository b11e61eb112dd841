# usage: ab_env.sh <outdir-tag> "<ENV assignments for B>"   — per-layer conv bench, default (A) vs the environment variant (B), same box
set -e
out=gpurun_out/$1; mkdir -p $out
python tools/conv_layers_bench.py 32 256 > $out/layers_A.txt 2>&1
env $2 python tools/conv_layers_bench.py 32 256 > $out/layers_B.txt 2>&1
for v in A B; do echo "== $v"; grep -E "^(enc|dec|total)" $out/layers_$v.txt | awk '{print $1, $6, $11, $16}' | tr '\n' ';'; echo; done
