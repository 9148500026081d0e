# same-box A/B of variant libraries against the shipped libdm_hip.so: usage  bash tools/gpu_ab_pair.sh <outdir> <scene> <tag> ...   (libdm_hip_<tag>.so)
OUT=gpurun_out/$1; SC=$2; shift 2; mkdir -p $OUT
for r in 1 2; do for t in "$@"; do
python tools/gpu_ab_bench.py deepmimic_amd/csrc/libdm_hip.so deepmimic_amd/csrc/libdm_hip_$t.so $SC > $OUT/ab_${SC}_${t}_$r.json 2>&1
done; done
