#!/bin/bash
# build one kernel family with extra compiler flags and link it with the default objects into a variant library (for same-box A/B):
#   tools/build_variant.sh <tag> <family id> <extra hipcc flags...>   ->  deepmimic_amd/csrc/libdm_hip_<tag>.so
set -e
cd "$(dirname "$0")/../deepmimic_amd/csrc"
TAG=$1; ID=$2; shift 2
mkdir -p build_v
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$@" -DDM_TU_F64=0 -DDM_TU_ID=$ID -x hip -c -o build_v/k_$TAG.o dm_kernels.cpp -Rpass-analysis=kernel-resource-usage 2> build_v/k_$TAG.res
OBJS=$(ls build/*.o | grep -v "k_f32_$ID.o" | tr '\n' ' ')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdm_hip_$TAG.so $OBJS build_v/k_$TAG.o
grep -hE "VGPRs:|VGPRs Spill|ScratchSize" build_v/k_$TAG.res | sed 's/^.*remark: [^ ]* //; s/\[-Rpass.*//' | paste - - - | sed "s/^/$TAG: /"
