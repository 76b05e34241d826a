#!/bin/bash
# GPU tests, then the kernel benchmarks; nothing further runs after a step that timed out or was killed.
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/tests.log 2>&1
rc=$?
tail -25 gpurun_out/tests.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest ended with status $rc: stopping"; exit $rc; fi
timeout -k 10 120 python -u tools/kprofile.py --big > gpurun_out/kprofile.txt 2>&1 || { tail -5 gpurun_out/kprofile.txt; exit 1; }
cat gpurun_out/kprofile.txt | grep -v "^\["
timeout -k 10 120 python -u tools/kbench.py --kind gaussian --bits 8 --B 128 --L 1024 --reps 50 > gpurun_out/kb_dense.txt 2>&1 || { tail -5 gpurun_out/kb_dense.txt; exit 1; }
cat gpurun_out/kb_dense.txt
exit $rc
