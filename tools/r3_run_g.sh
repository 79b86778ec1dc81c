#!/bin/bash
# round-3 record run: PMC traffic of k_raster, issue counters of the LP kernels, kernel stats of the headline mode
set -o pipefail
bash tools/pmc_k_raster.sh r03 || exit 1
bash tools/pmc_lp_kernels.sh r03 || exit 1
bash tools/profile_modes.sh r03 sim cand hex
