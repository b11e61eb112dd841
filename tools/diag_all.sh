# phase stamps of one layer under several diagnostic builds: diag_all.sh <tag> <layer> <variant> [<variant> ...]
tag=$1; layer=$2; shift; shift
for v in "$@"; do
  DCS_FDIAG_DUMP=$PWD/gpurun_out/$tag/$v DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_$v.so python tools/fwd_diag.py 32 256 $layer fwd 2>&1 | grep "$layer fwd"
done
