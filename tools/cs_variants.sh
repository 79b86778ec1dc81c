#!/bin/bash
# Occupancy experiment of the candidate-stability kernel's first pass (k_candidate_stability<CS_TAB_SMALL, CS_COLS_SMALL, false>):
# builds variants of the library with other register caps (CS_WAVES waves per SIMD), LDS tableau sizes (CS_TAB_SMALL) and
# new-contact capacities (CS_NEW_IF), prints each variant's register / spill / LDS figures and runs bench.py's
# candidate-stability mode (1 and 3 env groups) through BRIDGES_LIB.   tools/cs_variants.sh <out dir> [variant ...]
# A variant is "WAVES:TAB:NEWIF:COLS", e.g. 4:768:16:92 (the product's values).
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/cs_variants}; shift || true
VARIANTS=${@:-"4:768:16:92 5:768:16:92 5:640:8:92 6:512:8:60"}
mkdir -p "$OUT"
HASH=$(PYTHONPATH=bridges-with-reinforcement-learning_amd python3 -c "from bridges_hip import abi; print(abi.source_hash())")
for v in $VARIANTS; do
  IFS=: read W T N C <<< "$v"
  tag="w${W}_t${T}_n${N}_c${C}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "-DBRIDGES_SRC_HASH=\"$HASH\"" \
      -DCS_WAVES=$W -DCS_TAB_SMALL=$T -DCS_NEW_IF=$N -DCS_COLS_SMALL=$C -Rpass-analysis=kernel-resource-usage \
      bridges-with-reinforcement-learning_amd/csrc/api.hip -o tools/libbridges_hip_cs_$tag.so 2> "$OUT/$tag.usage.txt"
  grep -A10 "Function Name: _ZN7bridges21k_candidate_stabilityILi${T}ELi${C}ELb0" "$OUT/$tag.usage.txt" | grep -o "VGPRs: [0-9]*\|VGPRs Spill: [0-9]*\|SGPRs Spill: [0-9]*\|ScratchSize \[bytes/lane\]: [0-9]*\|Occupancy \[waves/SIMD\]: [0-9]*\|LDS Size \[bytes/block\]: [0-9]*" | tr '\n' ' ' > "$OUT/$tag.regs.txt"
  echo "$tag: $(cat $OUT/$tag.regs.txt)"
  if [ -z "$CS_BUILD_ONLY" ]; then
    for g in 1 3; do
      BRIDGES_LIB=tools/libbridges_hip_cs_$tag.so python3 bench.py --mode candidate-stability --groups $g --no-cpu-baseline --no-other-modes --seeds 0 \
          --steps 100 --warmup 20 > "$OUT/$tag.g$g.json" 2> "$OUT/$tag.g$g.err" || echo "bench failed for $tag g$g"
      python3 - "$OUT/$tag.g$g.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = j["candidate_stability"]
    print("   groups", c["groups"], "env-steps/s %.3g" % j["value"], "LPs/s %.3g" % c["decisions_per_s"], "wall %.3g" % c["decisions_per_s_wall"],
          "ms/lockstep %.3f" % c["ms_per_lockstep"], "errors", c["last_lockstep"]["errors"], "queued", c["last_lockstep"]["queued_large_tableaux"])
except Exception as e:
    print("   no result:", e)
PY
    done
  fi
done
