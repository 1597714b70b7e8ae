#!/bin/bash
# The GPU tests once per data-dependent specialisation switch: every scene of the suite through the
# general path logic, the generic triangle records, the literal box test, 64-bit record addresses, launches without streams of
# their own (no overlap, no rendering ahead) and NaN rays that walk the tree instead of taking the counts of the upload.  Run from the repo
# root ON THE GPU BOX (~40 s per pass).
set -o pipefail
mkdir -p gpurun_out
rc=0
for sw in "" PTMI_GENERIC_SHADING PTMI_GENERIC_TRIANGLES PTMI_GENERIC_BOXES PTMI_WIDE_RECORDS PTMI_SERIAL_LAUNCHES PTMI_WALK_NAN_RAYS; do
  log=gpurun_out/specialisation_${sw:-default}.log
  if [ -z "$sw" ]; then timeout -k 10 900 python -m pytest tests -q -m gpu > $log 2>&1; else env $sw=1 timeout -k 10 900 python -m pytest tests -q -m gpu > $log 2>&1; fi
  r=$?; [ $r -ne 0 ] && rc=$r
  echo "${sw:-default}: exit $r, $(tail -1 $log)"
done
exit $rc
