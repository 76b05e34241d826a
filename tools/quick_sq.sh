#!/bin/bash
# Quick look at one kernel family while iterating on it (run on the GPU box):
#   tools/quick_sq.sh hybrid_uniform,lvb     timing without the profiler, then one SQ counter pass of the same launches
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/quick
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 python3 -u $R/tools/kprofile.py --only "$1" 2>&1 | grep -v '^\[{' | grep -v amdgpu.ids
rm -rf $O/sq*
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O -o sq -- python3 -u $R/tools/kprofile.py --reps 30 --only "$1" > $O/sq.log 2>&1
cd $R && python3 tools/pmc_summary.py $O/pmc_sq.json $O/sq_counter_collection.csv | grep -E "INSTS_VALU|k_|ACTIVE_INST_VALU" | head -40
