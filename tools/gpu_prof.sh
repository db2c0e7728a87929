#!/bin/bash
# usage: tools/gpu_prof.sh <tag> [online|large|dense|build|buildlarge ...]   (run on the GPU box via gpurun); writes gpurun_out/<tag>_<mode>_*
# One rocprofv3 run per counter group (no pass mixes --pmc with a trace), each under its own time limit.
set -o pipefail
tag=$1; shift
modes=${@:-online large dense build buildlarge}
export TMPDIR=/tmp
for mode in $modes; do
  case $mode in
    online) args="online 50 10"; pargs="online 5 2";;
    large)  args="large 10";     pargs="large 3";;
    dense)  args="dense 10";     pargs="dense 3";;
    build)  args="build 5";      pargs="build 2";;
    buildlarge) args="buildlarge 4"; pargs="buildlarge 2";;
  esac
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${mode}_trace -- python3 tools/prof_run.py $args > gpurun_out/${tag}_${mode}_trace.log 2>&1 || { echo "trace $mode failed"; tail -5 gpurun_out/${tag}_${mode}_trace.log; exit 1; }
  i=0
  if [ $mode = online ]; then
    groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM"
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE")
  elif [ $mode = large ] || [ $mode = dense ]; then
    groups=("SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE"
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE")
  else
    groups=("FETCH_SIZE" "WRITE_SIZE")
  fi
  for pmc in "${groups[@]}"; do
    i=$((i+1))
    timeout -s KILL 400 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/${tag}_${mode}_pmc$i -- python3 tools/prof_run.py $pargs > gpurun_out/${tag}_${mode}_pmc$i.log 2>&1 || { echo "pmc $mode pass $i failed"; tail -3 gpurun_out/${tag}_${mode}_pmc$i.log; exit 1; }
  done
done
echo all-ok
