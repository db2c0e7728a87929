#!/bin/bash
# usage: tools/gpu_prof.sh <tag>   (run on the GPU box via gpurun); writes gpurun_out/<tag>_*
set -o pipefail
tag=$1; export TMPDIR=/tmp
timeout -k 10 300 python -u bench.py --no-cpu > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 tools/prof_run.py 50 10 > gpurun_out/${tag}_trace.log 2>&1 || { tail -5 gpurun_out/${tag}_trace.log; exit 1; }
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/${tag}_pmc$i -- python3 tools/prof_run.py 5 2 > gpurun_out/${tag}_pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 gpurun_out/${tag}_pmc$i.log; exit 1; }
done
echo all-ok
