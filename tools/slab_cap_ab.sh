# same-box A/B of the weight-gradient slab budget (needs the -DDCS_PLAN_KNOBS build: tools/exp_build.py knobs conv_wgrad_mfma.hip -DDCS_PLAN_KNOBS)
export DCS_LIB_PATH=$GRAFT_REPO_ROOT/dcs-net_amd/lib/exp/libdcsnet_hip_knobs.so
REPS=2 bash tools/ab_envs.sh slabcap "" "DCS_WGRAD_SLAB_MB=96" "DCS_WGRAD_SLAB_MB=24" "DCS_WGRAD_SLAB_MB=20" "DCS_WGRAD_SLAB_MB=16" "DCS_WGRAD_SLAB_MB=12"
