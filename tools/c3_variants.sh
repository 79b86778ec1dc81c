#!/bin/bash
# Band height / register cap variants of k_c3 (conv3x3 forward and input gradient of the training passes): builds one library
# per variant (tools/libbridges_c3_<name>.so) here, then -- on the box, through gpurun -- times one optimiser step of the U-Net
# policy and of ConvNet with each (tools/policy_step_kernels.py).   tools/c3_variants.sh build | run <out dir>
set -e
cd "$(dirname "$0")/.."
variants="r8w1:-DC3_BAND_ROWS_WIDE=8:-DC3_MIN_WAVES=1 r8w2:-DC3_BAND_ROWS_WIDE=8:-DC3_MIN_WAVES=2 r4w1:-DC3_BAND_ROWS_WIDE=4:-DC3_MIN_WAVES=1 r4w2:-DC3_BAND_ROWS_WIDE=4:-DC3_MIN_WAVES=2"
if [ "$1" = build ]; then
  HASH=$(PYTHONPATH=bridges-with-reinforcement-learning_amd python3 -c "from bridges_hip import abi; print(abi.source_hash())")
  for v in $variants; do
    name=${v%%:*}; flags=$(echo ${v#*:} | tr ':' ' ')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "-DBRIDGES_SRC_HASH=\"$HASH\"" $flags \
        bridges-with-reinforcement-learning_amd/csrc/api.hip -o tools/libbridges_c3_$name.so -Rpass-analysis=kernel-resource-usage 2> /tmp/c3_$name.txt
    echo "== $name ($flags)"
    grep -E "Function Name: |VGPRs:|AGPRs:|Occupancy|ScratchSize" /tmp/c3_$name.txt | grep -A4 "k_c3I" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | paste - - - - - | sed 's/  */ /g' | grep "Li16ELi1E" | sed 's/_ZN7bridges4k_c3I//; s/EEEvPKfS2_S2_S2_S2_Pfiiiii//'
  done
else
  out=$2; mkdir -p $out
  for v in $variants; do
    name=${v%%:*}
    for model in UNet ConvNet; do
      BRIDGES_LIB=$(pwd)/tools/libbridges_c3_$name.so python tools/policy_step_kernels.py --model $model 2>/dev/null | grep -E "GPU time|k_c3<" > $out/c3_${name}_$model.txt
      echo "$name $model: $(head -1 $out/c3_${name}_$model.txt | cut -c1-90)  k_c3 total $(grep 'k_c3<' $out/c3_${name}_$model.txt | awk '{s+=$(NF-5)} END {print s}') us/step"
    done
  done
fi
