#!/bin/bash
R=$(pwd)
run() { echo "== $*"; env "$@" python $R/tools/scratch/prof_e2e.py 2>&1 | grep -E "^update [12]|check_finite \(ms\)" | sed -E 's/.*epochs_s.: ([0-9.]+).*/  epochs_s \1/' | cut -c1-200; }
run NOSNAP=1
run NOSNAP=1 NOVAL=1
run RIFT_PREFETCH=0
run RIFT_PIPELINE=0
