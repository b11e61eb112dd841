# NOTE (round 5): the DCS_MFMA_* thresholds are compile-time constants in the shipped library (csrc/dcs_common.h: dcs_knob).  Sweep with a
# diagnostic build:  python tools/exp_build.py knobs conv_mfma.hip -DDCS_PLAN_KNOBS ; export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
for mb in 1024 512 256; do for sb in 512 256 128 0; do
echo "== MIN_BLOCKS=$mb SPLIT_BELOW=$sb"; DCS_CONV_PIPE=0 DCS_MFMA_MIN_BLOCKS=$mb DCS_MFMA_SPLIT_BELOW=$sb timeout -k 5 120 python tools/conv_layers_bench.py 32 256 2>/dev/null | grep -E "enc4|enc5|enc6|dec0|total"
done; done
