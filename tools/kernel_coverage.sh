#!/bin/bash
# Which kernels of the library does `pytest -m gpu` launch?  Kernel trace of the GPU tests (the multi-process tests
# left out: their children are processes of their own), kernel names with launch counts into
# gpurun_out/kernel_coverage.txt.  Run on the GPU box from the repository root.
set -e
R=$(pwd)
O=$R/gpurun_out/prof_coverage
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o cov -- python3 -u -m pytest $R/tests -m gpu -x -q \
  --timeout 300 --timeout-method thread --ignore=$R/tests/test_gpu_ddp.py -p no:cacheprovider > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 - "$O" <<'P' > $R/gpurun_out/kernel_coverage.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: r["Name"]):
    print(r["Calls"], r["Name"])
P
wc -l $R/gpurun_out/kernel_coverage.txt
rm -rf "$O"
