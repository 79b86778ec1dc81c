#!/bin/bash
# Matrix-core and LDS counters of the hand-written convolution (k_conv3x3_o16, 16 -> 16 channels, 2048 images of 64x64):
# separate --pmc passes with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Run on the box
# through gpurun from the repo root:  tools/pmc_conv.sh <tag>   ->  gpurun_out/pmc_<tag>_conv.json
set -o pipefail
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for ctrs in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  name=$(echo $ctrs | tr ' ' '+')
  out=$root/gpurun_out/pmc_${tag}_conv_$name
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $root/tools/conv_probe.py) > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
done
python3 - "$root/gpurun_out" "$tag" <<'PY'
import csv, glob, json, sys
base, tag = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(f"{base}/pmc_{tag}_conv_*/**/*counter_collection.csv", recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "k_conv3x3_o16" in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out[k] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
json.dump(out, open(f"{base}/pmc_{tag}_conv.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
