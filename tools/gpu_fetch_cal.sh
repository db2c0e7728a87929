#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_fetch_cal.sh <tag>   -> gpurun_out/<tag>_known.json, gpurun_out/<tag>_pmc*/
# One rocprofv3 --pmc pass per counter group, the program itself after `--`, each pass under its own time limit.
set -o pipefail
tag=${1:-r3cal}
export TMPDIR=/tmp
mkdir -p gpurun_out
B=tools/ubench/fetch_calibration
[ -x $B ] || { echo "build $B first (hipcc --offload-arch=gfx950 -O3 -o $B $B.hip)"; exit 1; }
timeout -k 10 120 $B > gpurun_out/${tag}_known.json || { echo "plain run failed"; exit 1; }
i=0
for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout -s KILL 180 rocprofv3 --pmc $pmc --output-format csv -d gpurun_out/${tag}_pmc$i -- $B > gpurun_out/${tag}_pmc$i.log 2>&1 || { echo "pmc pass $i ($pmc) failed"; tail -3 gpurun_out/${tag}_pmc$i.log; exit 1; }
done
# (last, and allowed to fail: the counter may not be exposed on this ROCm; nothing runs after it)
timeout -s KILL 180 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum --output-format csv -d gpurun_out/${tag}_pmc4 -- $B > gpurun_out/${tag}_pmc4.log 2>&1 || echo "(TCC_EA0_RDREQ_32B_sum not collected)"
echo cal-ok
