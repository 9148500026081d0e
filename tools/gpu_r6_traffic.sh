OUT=gpurun_out/r6p; mkdir -p $OUT; export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib tools/fetch_calib.hip 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do t=$(echo $c | tr ' ' '_'); rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/calib_$t -o p -- /tmp/fetch_calib > $OUT/calib_$t.log 2>&1; python tools/pmc_sum.py $OUT/calib_$t k_calib > $OUT/calib_$t.jsonl; done
P="--steps 6 --warmup 2 --min-warmup 40 --groups 1 --no-cpu-baseline --no-closed-loop --no-parity-check --sustain-seconds 0"
for v in xcd noxcd; do L=deepmimic_amd/csrc/libdm_hip.so; [ $v = noxcd ] && L=deepmimic_amd/csrc/libdm_hip_noxcd.so
 for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do t=$(echo $c | tr ' ' '_'); DM_HIP_LIB=$PWD/$L rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/step_${v}_$t -o p -- python bench.py $P > $OUT/step_${v}_$t.log 2>&1; python tools/pmc_sum.py $OUT/step_${v}_$t k_env_step > $OUT/step_${v}_$t.jsonl; done; done
python tools/gpu_ab_bench.py deepmimic_amd/csrc/libdm_hip_noxcd.so deepmimic_amd/csrc/libdm_hip.so > $OUT/ab_xcd.json 2>&1
find $OUT -name "*.csv" -size +2M -delete
