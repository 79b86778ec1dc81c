#!/bin/bash
# Matrix-pipe counters of the f32-MFMA kernels of the training loop (acting head, middle-layer kernels): separate --pmc passes
# with --kernel-trace only (MI355X_MICROARCH.md; dispatches are serialised under --pmc, so these are the kernels ALONE).
#   tools/pmc_mfma_kernels.sh <tag>   ->  gpurun_out/pmc_<tag>_mfma.json          (run through gpurun from the repo root)
set -o pipefail
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  name=$(echo $ctrs | tr ' ' '+')
  for prog in head step; do
    out=$root/gpurun_out/pmc_${tag}_mfma_${prog}_$name
    if [ $prog = head ]; then cmd="$root/tools/head_fused_bench.py --rows 45056 --reps 3"; else cmd="$root/tools/mlp_step_bench.py --replays 40"; fi
    (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $cmd) > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
    echo "pass $prog $name done"
  done
done
python3 - "$root/gpurun_out" "$tag" <<'PY'
import csv, glob, json, sys
base, tag = sys.argv[1], sys.argv[2]
out = {}
keys = ("k_head_sigmoid_dot", "k_mid_fwd", "k_mid_bwd", "k_lin_fwd(", "k_lin_bwd<true>", "k_lin_bwd<false>", "k_successor_loss")
for f in glob.glob(f"{base}/pmc_{tag}_mfma_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for key in keys:
            if key in r["Kernel_Name"]:
                out.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, d in out.items():
    res[k] = {c: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()}
    g = lambda c: res[k].get(c, {}).get("per_launch_mean")
    der = {}
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CYCLES"):
        der["mfma_busy_share_of_sq_busy"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CYCLES")
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        der["mfma_busy_cycles_per_gpu_active_cycle"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / g("GRBM_GUI_ACTIVE")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY"):
        der["active_inst_share_of_wave_cycles"] = g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")
        der["wait_any_share"] = (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES")
    res[k]["derived"] = der
json.dump(res, open(f"{base}/pmc_{tag}_mfma.json", "w"), indent=1)
for k, v in res.items():
    print(k, v["derived"], {c: round(x["per_launch_mean"]) for c, x in v.items() if c != "derived"})
PY
