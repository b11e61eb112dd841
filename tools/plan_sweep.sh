for mb in 1024 512 256; do for sb in 512 256 128 0; do
echo "== MIN_BLOCKS=$mb SPLIT_BELOW=$sb"; DCS_CONV_PIPE=0 DCS_MFMA_MIN_BLOCKS=$mb DCS_MFMA_SPLIT_BELOW=$sb timeout -k 5 120 python tools/conv_layers_bench.py 32 256 2>/dev/null | grep -E "enc4|enc5|enc6|dec0|total"
done; done
