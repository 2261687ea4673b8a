#!/bin/bash
# Kernel trace + FETCH_SIZE / WRITE_SIZE passes over the objective kernels at the two MissSIWAE shapes (hepmass, miniboone).
# Usage: bash tools/objprof_strat.sh <tag>      results under gpurun_out/objprof_{hep,mb}_<tag>
set -u
TAG=${1:-x}
for W in hep mb; do
  bash tools/objprof.sh $W || exit 1
  rm -rf gpurun_out/objprof_${W}_$TAG && mv gpurun_out/objprof_$W gpurun_out/objprof_${W}_$TAG
done
