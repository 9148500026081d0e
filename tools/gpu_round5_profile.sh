#!/bin/bash
# Round-5 evidence run on the GPU box (every --pmc pass is its own run with --kernel-trace only).  Usage, from the repo root through gpurun:
#   bash tools/gpu_round5_profile.sh [part ...]      parts: bench stats pmc classes latency policy scenes (default: all)
OUT=gpurun_out/r05p
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench stats pmc classes latency policy scenes}
B="--no-cpu-baseline --no-closed-loop --no-parity-check --sustain-seconds 0"
P="--steps 6 --warmup 2 --min-warmup 40 --groups 1 $B"
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  python bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
  python bench.py --steps 300 --warmup 30 --groups 1 --no-cpu-baseline > $OUT/bench_groups1.json 2>> $OUT/bench.err
  python bench.py --steps 100 --warmup 10 --scene humanoid3d_spinkick --no-cpu-baseline > $OUT/bench_spinkick.json 2>> $OUT/bench.err
  python bench.py --steps 100 --warmup 10 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_dog.json 2>> $OUT/bench.err
  python bench.py --steps 300 --warmup 30 --physics 2 --no-cpu-baseline > $OUT/bench_physics2.json 2>> $OUT/bench.err
  python bench.py --steps 200 --warmup 30 --force-gather --gather cabi --no-cpu-baseline > $OUT/bench_record_exchange_cabi_1rank.json 2>> $OUT/bench.err
fi
if has stats; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check --sustain-seconds 0 > $OUT/stats.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dog -o stats -- python bench.py --scene dog3d_pace --steps 40 --warmup 10 $B > $OUT/stats_dog.log 2>&1
fi
for SC in humanoid3d_walk dog3d_pace; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  if has pmc; then
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_sq$S.log 2>&1
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_sq2$S.log 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_fetch$S.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_write$S.log 2>&1
  fi
  if has classes; then          # instruction classes of the step kernel, at 10 and at 5 Gauss-Seidel iterations (the sweep's mix = the difference / 5 per iteration)
    for IT in 10 5; do
      rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc_cls_a_it$IT$S -o pmc -- python bench.py --scene $SC $P --solver-iters $IT > $OUT/pmc_cls_a_it$IT$S.log 2>&1
      rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/pmc_cls_b_it$IT$S -o pmc -- python bench.py --scene $SC $P --solver-iters $IT > $OUT/pmc_cls_b_it$IT$S.log 2>&1
    done
  fi
  if has latency; then          # what the s_waitcnt cycles wait FOR: in-flight levels (average latency x count) of LDS / SMEM / VMEM instructions
    rocprofv3 --pmc LdsLatency SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lat_lds$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_lat_lds$S.log 2>&1
    rocprofv3 --pmc SmemLatency SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_lat_smem$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_lat_smem$S.log 2>&1
    rocprofv3 --pmc VmemLatency SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/pmc_lat_vmem$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_lat_vmem$S.log 2>&1
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds$S -o pmc -- python bench.py --scene $SC $P > $OUT/pmc_lds$S.log 2>&1
    python tools/gpu_profile_phases.py $SC > $OUT/phases$S.json 2>&1
  fi
done
if has policy; then
  python tools/gpu_policy_bench.py > $OUT/policy_bench.json 2> $OUT/policy_bench.err
  ENVS=2048 python tools/gpu_policy_bench.py > $OUT/policy_bench_2048.json 2>> $OUT/policy_bench.err
  ENVS=16384 python tools/gpu_policy_bench.py > $OUT/policy_bench_16384.json 2>> $OUT/policy_bench.err
  DM_POLICY_LAYERED=1 python tools/gpu_policy_bench.py > $OUT/policy_bench_layered.json 2>> $OUT/policy_bench.err
  DM_POLICY_PROBE=2 python tools/gpu_policy_bench.py 2>&1 | grep "k_policy_fused phases" > $OUT/policy_phases.txt
  DM_POLICY_PROBE=1 python tools/gpu_policy_bench.py > $OUT/policy_bench_stream_from_l1.json 2>> $OUT/policy_bench.err
  python tools/gpu_policy_fused_check.py > $OUT/policy_fused_vs_layered.json 2>> $OUT/policy_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_policy -o stats -- python tools/gpu_policy_bench.py > $OUT/stats_policy.log 2>&1
fi
if has scenes; then
  python tools/gpu_scene_bench.py > $OUT/bench_scenes.json 2> $OUT/bench_scenes.err
  python tools/gpu_parity_report.py --steps 300 --envs 8 > $OUT/parity_report.json 2> $OUT/parity_report.err
  python tools/gpu_tail_probe.py > $OUT/tail_probe.txt 2>&1
fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls $OUT | head -80
