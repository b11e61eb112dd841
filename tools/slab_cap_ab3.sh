export DCS_LIB_PATH=$GRAFT_REPO_ROOT/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
run() { env DCS_WGRAD_SLAB_MB=$1 timeout -k 10 300 python bench.py ${@:2} --no-cpu-baseline --no-sub-lines --no-native-line --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))"; }
for i in 1 2; do
  echo "f32 B64:  96: $(run 96 --batch 64)  32: $(run 32 --batch 64)  16: $(run 16 --batch 64)"
  echo "bf16 B64: 96: $(run 96 --dtype bf16 --batch 64)  32: $(run 32 --dtype bf16 --batch 64)  24: $(run 24 --dtype bf16 --batch 64)"
  echo "bf16 B32: 96: $(run 96 --dtype bf16 --batch 32)  16: $(run 16 --dtype bf16 --batch 32)"
  echo "f32 B16:  96: $(run 96 --batch 16)  16: $(run 16 --batch 16)  8: $(run 8 --batch 16)"
done
