#!/bin/bash
# same-box A/B of variant libraries (tools/build_variant.sh): usage  bash tools/gpu_ab_sched.sh <scene> <out tag> <variant tags...>
SCENE=$1; TAG=$2; shift 2
OUT=gpurun_out/ab_sched; mkdir -p $OUT
L=deepmimic_amd/csrc
LIBS="$L/libdm_hip.so"
for v in "$@"; do LIBS="$LIBS $L/libdm_hip_$v.so"; done
python tools/gpu_ab_libs.py $SCENE 4096 0 $LIBS > $OUT/$TAG.json 2> $OUT/$TAG.err
python - <<EOF
import json
d = json.load(open("$OUT/$TAG.json"))
for k, v in d.items():
    if isinstance(v, dict): print("%-44s %.4f ms  %.0f" % (k, v["kernel_ms_median"], v["env_steps_per_s"]))
EOF
python - <<EOF2
import json
d = json.load(open("$OUT/$TAG.json"))
for k, v in d.get("checksums", {}).items(): print("%-44s reward_sum %.9g  state_abs_sum %.9g" % (k, v["reward_sum"], v["state_abs_sum"]))
EOF2
