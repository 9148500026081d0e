#!/bin/sh
# TEST INFRASTRUCTURE -- builds oracle/_ref/libdm_ref.so from the reference's own, unmodified sources where they lie
# (argument 1: path of the reference's DeepMimicCore directory; default /root/reference/DeepMimicCore), against
# oracle/eigen_shim (Eigen is not installed), oracle/gl_stub (GL type names) and oracle/bullet_stub (Bullet TYPE NAMES: the
# controller and scene translation units include Bullet headers through sim/World.h, but the routines taken from them --
# stable PD, state features, action mapping, imitation reward, AMP observation, task rewards -- call no Bullet function).
# Outputs only into oracle/_ref/ (git-ignored; it travels to the GPU box with the snapshot).  Nothing from the reference is copied
# into the repo.
#
# Link in two passes: what the compiled reference objects still reference after ref_glue.cpp / ref_standins.cpp (methods of cWorld,
# cSimCharacter, cSimBodyLink, ... that live in Bullet translation units; scene-builder plumbing) is listed from a first link and
# aliased to ref_unreachable() in the second, so the library loads with every symbol bound and a call into such a method aborts
# with a message instead of returning something invented.
set -e
cd "$(dirname "$0")"
REF="${1:-/root/reference/DeepMimicCore}"
CXX="${CXX:-g++}"
FLAGS="-O2 -std=c++14 -fPIC -w -Ieigen_shim -Igl_stub -Ibullet_stub -I$REF"
SRCS="util/MathUtil util/Rand util/JsonUtil util/FileUtil util/Timer util/Annealer util/DynamicTimeWarper util/ArgParser
      util/json/json_reader util/json/json_value util/json/json_writer
      sim/SpAlg sim/RBDUtil sim/RBDModel sim/CtCtrlUtil
      anim/KinTree anim/Shape anim/Motion anim/Character anim/KinCharacter anim/KinController
      anim/MotionController anim/ClipsController anim/KinCtrlBuilder
      sim/Perturb sim/Controller sim/CharController sim/DeepMimicCharController sim/CtController sim/CtPDController
      sim/PDController sim/ExpPDController sim/ImpPDController sim/AgentRegistry util/IndexManager
      scenes/Scene scenes/RLScene scenes/SceneSimChar scenes/RLSceneSimChar scenes/SceneImitate scenes/SceneImitateAMP
      scenes/SceneHeadingAMP scenes/SceneHeadingAMPGetup scenes/SceneTargetAMP scenes/SceneStrikeAMP scenes/SceneDribbleAMP"
if [ ! -d "$REF" ]; then
    if [ -f _ref/libdm_ref.so ]; then echo "build_ref: $REF absent, keeping the prebuilt _ref/libdm_ref.so"; exit 0; fi
    echo "build_ref: $REF absent and no prebuilt library" >&2; exit 1
fi
mkdir -p _ref/obj
# one builder at a time (pytest-xdist workers all call this), and nothing is rebuilt or relinked when it is up to date
exec 9> _ref/.lock
flock 9
OBJS=""
RELINK=0
[ -f _ref/libdm_ref.so ] || RELINK=1
STUBS="eigen_shim/Eigen/Core eigen_shim/Eigen/Geometry gl_stub/GL/glew.h bullet_stub/btBulletDynamicsCommon.h"
for s in $SRCS; do
    o="_ref/obj/$(echo "$s" | tr '/' '.').o"
    stale=0
    if [ ! -f "$o" ] || [ "$REF/$s.cpp" -nt "$o" ]; then stale=1; fi
    for h in $STUBS; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
    if [ $stale = 1 ]; then $CXX $FLAGS -c "$REF/$s.cpp" -o "$o"; RELINK=1; fi
    OBJS="$OBJS $o"
done
for g in ref_glue ref_standins; do
    o="_ref/obj/$g.o"
    stale=0
    if [ ! -f "$o" ] || [ "$g.cpp" -nt "$o" ]; then stale=1; fi
    for h in $STUBS; do if [ "$h" -nt "$o" ]; then stale=1; fi; done
    if [ $stale = 1 ]; then $CXX $FLAGS -c $g.cpp -o "$o"; RELINK=1; fi
    OBJS="$OBJS $o"
done
if [ $RELINK = 0 ]; then echo "build_ref: oracle/_ref/libdm_ref.so is up to date"; exit 0; fi
# pass 1: what is still undefined (reference classes only: names c[A-Z]... / t[A-Z]...)
$CXX -shared -o _ref/libdm_ref.pass1.so $OBJS -Wl,--unresolved-symbols=ignore-all
nm -u _ref/libdm_ref.pass1.so | awk '$1 == "U" {print $2}' | while read sym; do
    d="$(echo "$sym" | c++filt)"
    case "$d" in
        c[A-Z]*|t[A-Z]*|"vtable for c"[A-Z]*|"typeinfo for c"[A-Z]*|"VTT for c"[A-Z]*|"non-virtual thunk to c"[A-Z]*|"virtual thunk to c"[A-Z]*) echo "-Wl,--defsym,$sym=ref_unreachable" ;;
    esac
done > _ref/unreachable.args
rm -f _ref/libdm_ref.pass1.so
# pass 2: everything bound
$CXX -Wl,-z,defs -shared -o _ref/libdm_ref.so.tmp $OBJS @_ref/unreachable.args
mv -f _ref/libdm_ref.so.tmp _ref/libdm_ref.so
echo "build_ref: built oracle/_ref/libdm_ref.so from $REF ($(wc -l < _ref/unreachable.args) Bullet-side methods aliased to ref_unreachable)"
