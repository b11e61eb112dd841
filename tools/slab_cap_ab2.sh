export DCS_LIB_PATH=$GRAFT_REPO_ROOT/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
for i in 1 2 3; do
  a=$(DCS_WGRAD_SLAB_MB=96 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(DCS_WGRAD_SLAB_MB=16 timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  c=$(DCS_WGRAD_SLAB_MB=16 timeout -k 10 300 python bench.py --dtype bf16 --batch 64 --no-cpu-baseline --no-sub-lines --no-native-line --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  d=$(DCS_WGRAD_SLAB_MB=96 timeout -k 10 300 python bench.py --dtype bf16 --batch 64 --no-cpu-baseline --no-sub-lines --no-native-line --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "f32 B32: 96MB $a  16MB $b   | bf16 B64: 96MB $d  16MB $c"
done
