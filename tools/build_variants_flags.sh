#!/bin/bash
# compiler-flag variants, built in parallel (fourth batch: pre-RA scheduler direction / off, on top of the adopted per-family flags)
cd "$(dirname "$0")/.."
rm -f deepmimic_amd/csrc/libdm_hip_[a-z][0-9]*.so
S="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
N="-mllvm -disable-machine-licm"
tools/build_variant.sh w1 0 -mllvm -enable-misched=0 &
tools/build_variant.sh w2 0 -mllvm -misched-topdown &
tools/build_variant.sh w3 0 -mllvm -misched-bottomup &
tools/build_variant.sh w4 0 $S -mllvm -amdgpu-schedule-metric-bias=0 &
tools/build_variant.sh w5 0 $S -mllvm -amdgpu-schedule-relaxed-occupancy=true &
tools/build_variant.sh d1 12 $N -mllvm -enable-misched=0 &
tools/build_variant.sh d2 12 $N -mllvm -misched-topdown &
tools/build_variant.sh d3 12 $N -mllvm -misched-bottomup &
tools/build_variant.sh d4 12 $S $N -mllvm -amdgpu-schedule-relaxed-occupancy=true &
wait
ls deepmimic_amd/csrc/*.so
