# usage: ab_lib.sh <lib-variant> [bench args...] — bench.py under the shipped library (A) and under dcs-net_amd/lib/exp/libdcsnet_hip_<variant>.so (B),
# alternating A B A B on the same box; prints ms/step of each run
for i in 1 2 3; do
  unset DCS_LIB_PATH
  a=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line "${@:2}" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$1.so
  b=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line "${@:2}" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "A(shipped) $a   B($1) $b"
done
