#!/bin/bash
# compiler-flag / occupancy variants, built in parallel (third batch: biped + ball at three waves per SIMD)
cd "$(dirname "$0")/.."
rm -f deepmimic_amd/csrc/libdm_hip_[a-z][0-9]*.so
S="-mllvm -amdgpu-sched-strategy=iterative-maxocc"
N="-mllvm -disable-machine-licm"
tools/build_variant.sh o1 9 $S &
tools/build_variant.sh o3 9 $S $N -DDM_OBJ_WAVES=3 &
tools/build_variant.sh o4 9 $S -DDM_OBJ_WAVES=3 &
tools/build_variant.sh o5 9 $N -DDM_OBJ_WAVES=3 &
tools/build_variant.sh o6 9 -DDM_OBJ_WAVES=3 &
wait
ls deepmimic_amd/csrc/*.so
