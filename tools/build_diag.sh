#!/bin/bash
# Diagnostic build of the library: the bridges_task.debug switches (bit0 skip the LPs, bit1 / bit2 skip the half-plane
# runs / the f32 stores of the rasteriser, bit3 per-env phase stamps in k_step, bit4 empty candidate-stability grid) exist
# only here.  Use it through BRIDGES_LIB=tools/libbridges_hip_diag.so (bench.py --debug N, tools/ablate.sh,
# tools/kstep_phases.py); the product library refuses a non-zero debug word.  Add -DLP_PROFILE for the per-phase shader
# cycle counters of the simplex (tools/lp_microbench.py --profile).
set -e
cd "$(dirname "$0")/.."
HASH=$(PYTHONPATH=bridges-with-reinforcement-learning_amd python3 -c "from bridges_hip import abi; print(abi.source_hash())")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DBRIDGES_DIAG "-DBRIDGES_SRC_HASH=\"$HASH\"" "$@" \
    bridges-with-reinforcement-learning_amd/csrc/api.hip -o tools/libbridges_hip_diag.so
echo tools/libbridges_hip_diag.so
