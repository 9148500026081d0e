#!/bin/sh
# TEST INFRASTRUCTURE -- builds oracle/_ref/libdm_ref.so from the reference's own, unmodified sources where they lie
# (argument 1: path of the reference's DeepMimicCore directory; default /root/reference/DeepMimicCore), against
# oracle/eigen_shim (Eigen is not installed) and oracle/gl_stub (GL type names).  Outputs only into oracle/_ref/
# (git-ignored; it travels to the GPU box with the snapshot).  Nothing from the reference is copied into the repo.
set -e
cd "$(dirname "$0")"
REF="${1:-/root/reference/DeepMimicCore}"
CXX="${CXX:-g++}"
FLAGS="-O2 -std=c++14 -fPIC -w -Ieigen_shim -Igl_stub -I$REF"
SRCS="util/MathUtil util/Rand util/JsonUtil util/FileUtil util/Timer util/Annealer util/DynamicTimeWarper
      util/json/json_reader util/json/json_value util/json/json_writer
      sim/SpAlg sim/RBDUtil sim/RBDModel sim/CtCtrlUtil
      anim/KinTree anim/Shape anim/Motion anim/Character anim/KinCharacter anim/KinController
      anim/MotionController anim/ClipsController"
if [ ! -d "$REF" ]; then
    if [ -f _ref/libdm_ref.so ]; then echo "build_ref: $REF absent, keeping the prebuilt _ref/libdm_ref.so"; exit 0; fi
    echo "build_ref: $REF absent and no prebuilt library" >&2; exit 1
fi
mkdir -p _ref/obj
OBJS=""
for s in $SRCS; do
    o="_ref/obj/$(echo "$s" | tr '/' '.').o"
    if [ ! -f "$o" ] || [ "$REF/$s.cpp" -nt "$o" ] || [ eigen_shim/Eigen/Core -nt "$o" ] || [ eigen_shim/Eigen/Geometry -nt "$o" ] || [ gl_stub/GL/glew.h -nt "$o" ]; then
        $CXX $FLAGS -c "$REF/$s.cpp" -o "$o"
    fi
    OBJS="$OBJS $o"
done
$CXX $FLAGS -Wl,-z,defs -shared -o _ref/libdm_ref.so ref_glue.cpp $OBJS
echo "build_ref: built oracle/_ref/libdm_ref.so from $REF"
