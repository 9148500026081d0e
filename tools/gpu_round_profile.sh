#!/bin/bash
# Round profile on the GPU box: headline bench, rocprofv3 kernel stats, PMC passes (each in its own run: --pmc is never combined with a
# sys / runtime / hip / hsa trace), per-scene benches, facade workers, parity report.  The GPU test suite is run separately
# (python -m pytest tests -m gpu).  Usage (from the repo root, through gpurun):  bash tools/gpu_round_profile.sh r04
TAG=${1:-r04}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
python bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_style.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --groups 1 --no-cpu-baseline > $OUT/bench_groups1.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --physics 2 --no-cpu-baseline > $OUT/bench_physics2.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --envs 8192 --groups 2 --no-cpu-baseline > $OUT/bench_8192.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 10 --scene humanoid3d_spinkick --no-cpu-baseline > $OUT/bench_spinkick.json 2>> $OUT/bench.err
python bench.py --steps 100 --warmup 10 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_dog.json 2>> $OUT/bench.err
DM_TREE=0 python bench.py --steps 100 --warmup 10 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_dog_dense.json 2>> $OUT/bench.err
python bench.py --steps 300 --warmup 30 --wave-packing 1 --no-cpu-baseline > $OUT/bench_pack1.json 2>> $OUT/bench.err
for SC in amp_heading_zombie amp_dribble_zombie; do
  python bench.py --steps 100 --warmup 10 --scene $SC --no-cpu-baseline > $OUT/bench_$SC.json 2>> $OUT/bench.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_g1 -o stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/stats_g1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dog -o stats -- python bench.py --scene dog3d_pace --steps 40 --warmup 10 --no-cpu-baseline --sustain-seconds 0 > $OUT/stats_dog.log 2>&1
for SC in humanoid3d_walk humanoid3d_spinkick dog3d_pace; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/pmc_sq$S.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/pmc_sq2$S.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/pmc_fetch$S.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$S -o pmc -- python bench.py --scene $SC --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/pmc_write$S.log 2>&1
done
for SC in humanoid3d_walk humanoid3d_spinkick dog3d_pace; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  python tools/gpu_profile_phases.py $SC > $OUT/phases$S.json 2>&1
done
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_INSTS_VALU[A-Z0-9_]*" | sort -u > $OUT/avail_valu_counters.txt
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --kernel-trace --output-format csv -d $OUT/pmc_flops -o pmc -- python bench.py --steps 6 --warmup 2 --min-warmup 40 --no-cpu-baseline --groups 1 --sustain-seconds 0 > $OUT/pmc_flops.log 2>&1
python tools/gpu_scene_bench.py > $OUT/bench_scenes.json 2> $OUT/bench_scenes.err
python tools/gpu_policy_bench.py > $OUT/policy_bench.json 2> $OUT/policy_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_policy -o stats -- python tools/gpu_policy_bench.py > $OUT/stats_policy.log 2>&1
python tools/gpu_tail_probe.py > $OUT/tail_probe.txt 2>&1
python tools/gpu_parity_report.py --steps 300 --envs 8 > $OUT/parity_report.json 2> $OUT/parity_report.err
python bench.py --steps 200 --warmup 30 --force-gather --gather cabi --no-cpu-baseline > $OUT/bench_record_exchange_cabi_1rank.json 2>> $OUT/bench.err
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 200 --warmup 30 --force-gather --no-cpu-baseline > $OUT/bench_record_exchange_rccl_groups_1rank.json 2>> $OUT/bench.err
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/smoke.log | tail -1; cut -c1-260 $OUT/bench.json; ls $OUT
