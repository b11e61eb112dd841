# NOTE (round 5): the DCS_MFMA_* thresholds are compile-time constants in the shipped library (csrc/dcs_common.h: dcs_knob).  Sweep with a
# diagnostic build:  python tools/exp_build.py knobs conv_mfma.hip -DDCS_PLAN_KNOBS ; export DCS_LIB_PATH=$PWD/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
# plan-threshold sweep for the bf16-storage train step at B = 64 (BASELINE configs[4] per-GPU share): one bench run per setting
out=gpurun_out/$1; mkdir -p $out
run() { name=$1; shift; env "$@" python bench.py --dtype bf16 --batch 64 --no-cpu-baseline --no-native-line --steps 20 > $out/$name.json 2>/dev/null; python -c "
import json,sys
j=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_per_step'],3))"; }
run base DCS_X=0
run minblk512 DCS_MFMA_MIN_BLOCKS=512
run minblk1024 DCS_MFMA_MIN_BLOCKS=1024
run minblk1536 DCS_MFMA_MIN_BLOCKS=1536
run cap16_32k DCS_MFMA_LDS_CAP=32768
run cap16_96k DCS_MFMA_LDS_CAP=98304
run cap32_56k DCS_MFMA_LDS_CAP32=57344
run cap32_72k DCS_MFMA_LDS_CAP32=73728
run wk0 DCS_MFMA_WK=0
run wkbelow1100 DCS_MFMA_WK_BELOW=1100
run split256 DCS_MFMA_SPLIT_BELOW=256
run wk64_0 DCS_MFMA_WK64=0
run wk64_2 DCS_MFMA_WK64=2
run base2 DCS_X=0
