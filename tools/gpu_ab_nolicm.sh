#!/bin/bash
# same-box A/B of -mllvm -disable-machine-licm (deepmimic_amd/csrc/Makefile NOLICM_IDS) on the compiled-topology families:
# libdm_hip.so = the default set, libdm_hip_t.so = the default set + families 12 13 14 20
OUT=gpurun_out/ab_nolicm2; mkdir -p $OUT
B=deepmimic_amd/csrc/libdm_hip.so; A=deepmimic_amd/csrc/libdm_hip_t.so
python tools/gpu_ab_libs.py dog3d_pace 4096 0 $B $A > $OUT/dog.json 2> $OUT/err.txt
DM_AB_AMP=1 python tools/gpu_ab_libs.py dog3d_pace 4096 0 $B $A > $OUT/dog_amp.json 2>> $OUT/err.txt
DM_AB_AMP=1 python tools/gpu_ab_libs.py humanoid3d_walk 4096 0 $B $A > $OUT/walk_amp.json 2>> $OUT/err.txt
python bench.py --steps 100 --warmup 10 --physics 2 --no-cpu-baseline > $OUT/bench_physics2.json 2>> $OUT/err.txt
python bench.py --steps 100 --warmup 10 --physics 2 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_physics2_dog.json 2>> $OUT/err.txt
DM_HIP_LIB=$A python bench.py --steps 100 --warmup 10 --physics 2 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_physics2_dog_t.json 2>> $OUT/err.txt
python bench.py --steps 100 --warmup 10 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_dog.json 2>> $OUT/err.txt
cat $OUT/*.json | cut -c1-330
