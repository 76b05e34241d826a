#!/bin/bash
# A/B of library builds with tools/kbench.py (default: the K = 256 image configuration; KBENCH_ARGS="--kind uniform --bits 8"
# for another): tools/ab_dense.sh ab/libA.so ab/libB.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}
PK=$R/exploring-discrete-denoising-diffusion-models-for-protein-design_amd
cp $PK/libd3pm_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for lib in "$@"; do
    cp $R/$lib $PK/libd3pm_hip.so
    echo "== $lib (round $round)"
    timeout -k 10 150 python3 -u $R/tools/kbench.py ${KBENCH_ARGS:---kind gaussian --bits 8 --B 128 --L 1024} --reps 30 2>&1 | grep -v amdgpu.ids | cut -c1-150 || exit 1
  done
done
cp /tmp/lib_orig.so $PK/libd3pm_hip.so
