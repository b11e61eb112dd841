# same-box A/B of weight-gradient plan knobs (needs the -DDCS_PLAN_KNOBS build: tools/exp_build.py knobs conv_wgrad_mfma.hip -DDCS_PLAN_KNOBS)
#   usage: slab_cap_ab.sh "<bench flags>" "<env A>" "<env B>" ...
export DCS_LIB_PATH=$GRAFT_REPO_ROOT/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
F="--steps 100 --warmup 10 $1"; shift
for rep in 1 2; do
  for e in "$@"; do
    v=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-sub-lines --no-native-line $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms_per_step'],3))")
    echo "rep$rep [$e] $v"
  done
done
