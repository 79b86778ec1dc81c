"""Per-launch HBM bytes of k_raster from two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE) of the same bench command.
rocprofv3 reports the counters in KiB per dispatch; FETCH_SIZE is doubled on gfx950 (128-B requests tallied at 64 B,
MI355X_MICROARCH.md)."""
import csv, glob, json, sys


def per_launch(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if "k_raster" in r["Kernel_Name"] and "generic" not in r["Kernel_Name"] and r["Counter_Name"] == counter]
    vals = vals[len(vals) // 4:]                      # drop the warm-up launches
    return sum(vals) / len(vals) * 1024.0, len(vals)  # KiB -> bytes


w, nw = per_launch(sys.argv[1], "WRITE_SIZE")
f, nf = per_launch(sys.argv[2], "FETCH_SIZE")
bench = json.load(open(sys.argv[1] + ".json"))
alg = bench["roofline"]["algorithmic_bytes_per_launch"]
print(json.dumps({
    "kernel": "k_raster", "round": 4,
    "config": "bench.py --steps 20 --warmup 5 --groups 1 (4096 envs, tower_height=4), separate --pmc passes with --kernel-trace only",
    "launches_averaged": [nw, nf],
    "WRITE_SIZE_bytes_per_launch": w,
    "FETCH_SIZE_bytes_per_launch_raw": f,
    "FETCH_SIZE_bytes_per_launch_corrected_x2": 2 * f,
    "hbm_bytes_per_launch": w + 2 * f,
    "algorithmic_bytes_per_launch": alg,
    "ratio": (w + 2 * f) / alg,
    "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); WRITE_SIZE exact for 16-B-per-lane stores",
}, indent=1))
