# usage: ab_env.sh <VAR=value> [bench args...] — bench.py without (A) and with (B) one environment setting, alternating A B A B on the
# same box; prints ms/step of each run
for i in 1 2 3; do
  a=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line "${@:2}" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env "$1" timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line "${@:2}" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "A $a   B($1) $b"
done
