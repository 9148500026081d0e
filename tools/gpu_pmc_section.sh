#!/bin/bash
# The counter passes + phase split + GPU test suite of tools/gpu_round_profile.sh alone (after a change to the device headers that leaves the
# production ISA as it was: re-stamps the committed counter files).  Usage: gpurun -- bash tools/gpu_pmc_section.sh
OUT=gpurun_out/r04
mkdir -p $OUT
export TMPDIR=/tmp
for SC in humanoid3d_walk humanoid3d_spinkick dog3d_pace; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  python tools/gpu_profile_phases.py $SC > $OUT/phases$S.json 2>&1
done
PACK=1 python tools/gpu_profile_phases.py humanoid3d_walk > $OUT/phases_walk_pack1.json 2>&1
for SC in humanoid3d_walk humanoid3d_spinkick dog3d_pace; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  for d in pmc_sq$S pmc_sq2$S pmc_fetch$S pmc_write$S; do rm -rf $OUT/$d; done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --no-closed-loop --groups 1 --sustain-seconds 0 > $OUT/pmc_sq$S.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --no-closed-loop --groups 1 --sustain-seconds 0 > $OUT/pmc_sq2$S.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --no-closed-loop --groups 1 --sustain-seconds 0 > $OUT/pmc_fetch$S.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --no-closed-loop --groups 1 --sustain-seconds 0 > $OUT/pmc_write$S.log 2>&1
done
rm -rf $OUT/pmc_flops
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $OUT/pmc_flops -o pmc -- python bench.py --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --no-closed-loop --groups 1 --sustain-seconds 0 > $OUT/pmc_flops.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/bench_after_prof_change.json 2>> $OUT/bench.err
python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu_final.log 2>&1
tail -3 $OUT/pytest_gpu_final.log
