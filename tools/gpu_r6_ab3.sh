# same-box A/B of family-0 variants against the round-5 kernel (libdm_hip_orig.so): usage  bash tools/gpu_r6_ab3.sh <tag> ...
OUT=gpurun_out/r6q; mkdir -p $OUT
for r in 1 2; do for t in "$@"; do
python tools/gpu_ab_bench.py deepmimic_amd/csrc/libdm_hip_orig.so deepmimic_amd/csrc/libdm_hip_$t.so > $OUT/ab_orig_${t}_$r.json 2>&1
done; done
