#!/bin/bash
# Issue-rate counters of the two LP kernels (k_step in the headline mode, k_candidate_stability in the candidate-stability
# mode): separate --pmc passes with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (dispatches are
# serialised under --pmc, so these are the kernels ALONE on the chip).  Run on the box through gpurun from the repo root:
#   tools/pmc_lp_kernels.sh <tag>   ->  gpurun_out/pmc_<tag>_lp.json
set -o pipefail
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for ctrs in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $ctrs | tr ' ' '+')
  out=$root/gpurun_out/pmc_${tag}_lp_$name
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $root/bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-modes --seeds 0 --mode candidate-stability) > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
done
python3 - "$root/gpurun_out" "$tag" <<'PY'
import csv, glob, json, sys
base, tag = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(f"{base}/pmc_{tag}_lp_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        # the first pass is whatever instantiation is not the queue pass's <4096, ...> (its LDS tableau size is a tunable)
        key = next((k for k in ("k_step", "k_enumerate", "k_raster") if k in name), None)
        if "k_candidate_stability<" in name:
            key = "k_candidate_stability<4096 (queue pass)" if "k_candidate_stability<4096" in name else \
                  "k_candidate_stability<%s (first pass)" % name.split("k_candidate_stability<")[1].split(",")[0]
        if key:
            out.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, d in out.items():
    res[k] = {c: {"per_launch_mean": sum(v) / len(v), "launches": len(v)} for c, v in d.items()}
    g = lambda c: res[k].get(c, {}).get("per_launch_mean")
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_ANY"):
        res[k]["derived"] = {"active_inst_share_of_wave_cycles": g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES"),
                             "wait_any_share": (g("SQ_WAIT_ANY") or 0) / g("SQ_WAVE_CYCLES"),
                             "wait_inst_share": (g("SQ_WAIT_INST_ANY") or 0) / g("SQ_WAVE_CYCLES")}
json.dump(res, open(f"{base}/pmc_{tag}_lp.json", "w"), indent=1)
print(json.dumps({k: v.get("derived") for k, v in res.items()}, indent=1))
for k, v in res.items():
    print(k, {c: round(x["per_launch_mean"]) for c, x in v.items() if c != "derived"})
PY
