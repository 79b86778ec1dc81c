#!/bin/bash
# Matrix-core, LDS and wait counters of the conv training kernels (k_c3, k_c3_wgrad, k_c3_reduce, k_up2_*, k_pw1_*) in one
# optimiser step of the U-Net policy at batch 32: separate --pmc passes with --kernel-trace only, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes (dispatches are serialised under --pmc: the kernels ALONE on the chip).
#   tools/pmc_conv_kernels.sh <tag>   ->  gpurun_out/pmc_<tag>_conv.json      (run on the box through gpurun, from the repo root)
set -o pipefail
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $ctrs | tr ' ' '+')
  out=$root/gpurun_out/pmc_${tag}_conv_$name
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $root/tools/policy_step_kernels.py --model UNet --steps 6 --plain) > $out.log 2>&1 || { echo "pass $name failed:"; tail -5 $out.log; }
done
python3 - "$root/gpurun_out" "$tag" <<'PY'
import csv, glob, json, re, sys
base, tag = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(f"{base}/pmc_{tag}_conv_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        m = re.search(r"bridges::(k_[a-z0-9_]+(?:<[0-9, ]+>)?)", name)
        if m:
            out.setdefault(m.group(1), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(out.items()):
    res[k] = {c: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()}
    g = lambda c: res[k].get(c, {}).get("per_launch_mean")
    der = {}
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs: the share of SIMD-cycles of the
    # launch in which a matrix instruction executes = busy / (GRBM / 8 * 1024)
    if g("GRBM_GUI_ACTIVE") and g("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
        der["mfma_busy_share_of_simd_cycles"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") * 128.0)
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY") is not None:
        der["wait_any_share_of_wave_cycles"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
        der["active_inst_share_of_wave_cycles"] = (g("SQ_ACTIVE_INST_ANY") or 0) / g("SQ_WAVE_CYCLES")
    if g("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_share_of_lds_cycles"] = (g("SQ_LDS_BANK_CONFLICT") or 0) / g("SQ_LDS_IDX_ACTIVE")
    res[k]["derived"] = der
json.dump(res, open(f"{base}/pmc_{tag}_conv.json", "w"), indent=1)
for k, v in res.items():
    print(k, {a: round(b, 3) for a, b in v["derived"].items()})
PY
