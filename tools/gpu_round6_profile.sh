#!/bin/bash
# Round-6 evidence run on the GPU box (every --pmc pass is its own run with --kernel-trace only).  Usage, from the repo root through gpurun:
#   bash tools/gpu_round6_profile.sh [part ...]      parts: bench trace pmc classes latency closed calib policy scenes (default: all)
OUT=gpurun_out/r06p
mkdir -p $OUT
export TMPDIR=/tmp
PARTS=${@:-bench trace pmc classes latency closed calib policy scenes}
B="--no-cpu-baseline --no-closed-loop --no-parity-check --sustain-seconds 0"
P="--steps 6 --warmup 2 --min-warmup 40 --groups 1 $B"
has() { [[ " $PARTS " == *" $1 "* ]]; }
rp() { d=$1; shift; rocprofv3 "$@" > $OUT/$d.log 2>&1; }
if has bench; then
  python bench.py --steps 300 --warmup 30 > $OUT/bench.json 2> $OUT/bench.err
  python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2>> $OUT/bench.err
  python bench.py --steps 300 --warmup 30 --groups 1 --no-cpu-baseline > $OUT/bench_groups1.json 2>> $OUT/bench.err
  python bench.py --steps 300 --warmup 30 --scene humanoid3d_spinkick --no-cpu-baseline > $OUT/bench_spinkick.json 2>> $OUT/bench.err
  python bench.py --steps 100 --warmup 10 --scene dog3d_pace --no-cpu-baseline > $OUT/bench_dog.json 2>> $OUT/bench.err
  python bench.py --steps 100 --warmup 10 --scene dog3d_pace --groups 1 --no-cpu-baseline > $OUT/bench_dog_groups1.json 2>> $OUT/bench.err
  python bench.py --steps 300 --warmup 30 --physics 2 --no-cpu-baseline > $OUT/bench_physics2.json 2>> $OUT/bench.err
  python bench.py --steps 200 --warmup 30 --force-gather --gather cabi --no-cpu-baseline > $OUT/bench_record_exchange_cabi_1rank.json 2>> $OUT/bench.err
  MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 200 --warmup 30 --force-gather --gather torch --no-cpu-baseline > $OUT/bench_record_exchange_rccl_groups_1rank.json 2>> $OUT/bench.err
fi
if has trace; then      # kernel traces kept whole and reduced per (kernel, grid): tools/kernel_stats_by_grid.py
  rp trace_walk --kernel-trace --stats --output-format csv -d $OUT/trace_walk -o t -- python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity-check --no-closed-loop --sustain-seconds 0
  rp trace_dog_groups1 --kernel-trace --stats --output-format csv -d $OUT/trace_dog_groups1 -o t -- python bench.py --scene dog3d_pace --groups 1 --steps 40 --warmup 10 $B
  rp trace_dog_groups2 --kernel-trace --stats --output-format csv -d $OUT/trace_dog_groups2 -o t -- python bench.py --scene dog3d_pace --groups 2 --steps 40 --warmup 10 $B
  rp trace_closed --kernel-trace --stats --output-format csv -d $OUT/trace_closed -o t -- python bench.py --scene humanoid3d_spinkick --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check --sustain-seconds 0
  for t in trace_walk trace_dog_groups1 trace_dog_groups2 trace_closed; do python tools/kernel_stats_by_grid.py $OUT/$t 3 > $OUT/$t.by_grid.csv; done
fi
for SC in humanoid3d_walk dog3d_pace humanoid3d_spinkick; do
  S=""; [ $SC != humanoid3d_walk ] && S="_$SC"
  if has pmc; then
    rp pmc_sq$S --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_sq2$S --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq2$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_fetch$S --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_write$S --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_tcc$S --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_tcc$S -o pmc -- python bench.py --scene $SC $P
  fi
  [ $SC = humanoid3d_spinkick ] && continue
  if has classes; then
    for IT in 10 5; do
      rp pmc_cls_a_it$IT$S --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc_cls_a_it$IT$S -o pmc -- python bench.py --scene $SC $P --solver-iters $IT
      rp pmc_cls_b_it$IT$S --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/pmc_cls_b_it$IT$S -o pmc -- python bench.py --scene $SC $P --solver-iters $IT
    done
    rp pmc_cls_c$S --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 --kernel-trace --output-format csv -d $OUT/pmc_cls_c$S -o pmc -- python bench.py --scene $SC $P
  fi
  if has latency; then
    rp pmc_lat_lds$S --pmc LdsLatency SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lat_lds$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_lat_smem$S --pmc SmemLatency SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/pmc_lat_smem$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_lat_vmem$S --pmc VmemLatency SQ_INSTS_VMEM --kernel-trace --output-format csv -d $OUT/pmc_lat_vmem$S -o pmc -- python bench.py --scene $SC $P
    rp pmc_lds$S --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/pmc_lds$S -o pmc -- python bench.py --scene $SC $P
    python tools/gpu_profile_phases.py $SC > $OUT/phases$S.json 2>&1
  fi
done
if has closed; then
  for SC in humanoid3d_walk humanoid3d_spinkick dog3d_pace; do SCENE=$SC python tools/gpu_closed_loop_diag.py > $OUT/closed_loop_$SC.json 2> $OUT/closed_loop_$SC.err; done
fi
if has calib; then
  hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/fetch_calib.hip 2>/dev/null
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do t=$(echo $c | tr ' ' '_'); rp calib_$t --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$t -o p -- /tmp/fetch_calib; python tools/pmc_sum.py $OUT/calib_$t k_calib > $OUT/calib_$t.jsonl; done
fi
if has policy; then
  python tools/gpu_policy_bench.py > $OUT/policy_bench.json 2> $OUT/policy_bench.err
fi
if has scenes; then
  python tools/gpu_scene_bench.py > $OUT/bench_scenes.json 2> $OUT/bench_scenes.err
  python tools/gpu_parity_report.py --steps 300 --envs 8 > $OUT/parity_report.json 2> $OUT/parity_report.err
  python tools/gpu_tail_probe.py > $OUT/tail_probe.txt 2>&1
fi
find $OUT -name "*kernel_trace.csv" -size +6M -delete
find $OUT -name "*agent_info.csv" -delete
ls $OUT | head -120
