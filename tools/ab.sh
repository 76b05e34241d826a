#!/bin/bash
# A/B of two builds of the library on one box: tools/ab.sh <kprofile --only list> ab/libA.so ab/libB.so ...
# (each library is copied over the package's in turn, twice, and tools/kprofile.py is run on it)
R=${GRAFT_REPO_ROOT:-/root/repo}
PK=$R/exploring-discrete-denoising-diffusion-models-for-protein-design_amd
only=$1; shift
cp $PK/libd3pm_hip.so /tmp/lib_orig.so
for round in 1 2; do
  for lib in "$@"; do
    cp $R/$lib $PK/libd3pm_hip.so
    echo "== $lib (round $round)"
    timeout -k 10 120 python3 -u $R/tools/kprofile.py --only "$only" 2>&1 | grep -v '^\[{' | grep -v amdgpu.ids | cut -c1-70 || exit 1
  done
done
cp /tmp/lib_orig.so $PK/libd3pm_hip.so
