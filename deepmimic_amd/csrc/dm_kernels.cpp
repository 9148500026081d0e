// One kernel family of libdm_hip.so per object file.  Compiled as
//     hipcc ... -DDM_TU_F64=<0|1> -DDM_TU_ID=<family> -c dm_kernels.cpp -o k_<prec>_<family>.o
// (deepmimic_amd/csrc/Makefile); the emulator build (tests/emu) compiles it once with -DDM_TU_ALL.
// The kernels themselves live in dm_device.h / dm_device_duo.h; this file only holds their launchers (dm_launch.h).
#include "dm_launch.h"
#include "dm_device_duo.h"

namespace dmk {

template <typename Real, int V>
void launch_step_duo(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg) {
    RT_LAUNCH((k_env_step_duo<Real, V == SV_TAPS, V == SV_AMP || V == SV_V2, V == SV_V2>), grid, s, m, st, io, dbg);
}
template <typename Real, typename C, int V>
void launch_step_duo_c(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg) {
    RT_LAUNCH((k_env_step_duo<Real, V == SV_TAPS, V == SV_AMP || V == SV_V2, V == SV_V2, C>), grid, s, m, st, io, dbg);
}
template <typename Real, typename C, int V>
void launch_step(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg) {
    RT_LAUNCH((k_env_step<Real, C, V == SV_TAPS, V == SV_AMP || V == SV_V2, V == SV_V2>), grid, s, m, st, io, dbg);
}
template <typename Real, typename C>
void launch_reset(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const int* env_ids, const double* kin_times, const double* max_times) {
    RT_LAUNCH((k_env_reset<Real, C>), grid, s, m, st, env_ids, kin_times, max_times);
}
template <typename Real, typename C>
void launch_query(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg) {
    RT_LAUNCH((k_env_query<Real, C>), grid, s, m, st, io, dbg);
}
template <typename Real, typename C>
void launch_probe(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const DebugTaps<Real>& dbg, int what, double dt) {
    RT_LAUNCH((k_env_probe<Real, C>), grid, s, m, st, dbg, what, dt);
}
template <typename Real, typename C>
void launch_amp_expert(unsigned grid, rt_stream s, const ModelDev<Real>& m, const double* times, const double* ground_h, float* out, const int* clips) {
    RT_LAUNCH((k_amp_expert<Real, C>), grid, s, m, times, ground_h, out, clips);
}

#define DM_STEP_ARGS(Real) unsigned, rt_stream, const ModelDev<Real>&, const EnvState<Real>&, const StepIO<Real>&, const DebugTaps<Real>&
#define DM_INST_DUO(Real, V) template void launch_step_duo<Real, V>(DM_STEP_ARGS(Real));
#define DM_INST_STEP(Real, C, V) template void launch_step<Real, C, V>(DM_STEP_ARGS(Real));
#define DM_INST_DUOC(Real, C, V) template void launch_step_duo_c<Real, C, V>(DM_STEP_ARGS(Real));
#define DM_INST_MISC(Real, C)                                                                                                                   \
    template void launch_reset<Real, C>(unsigned, rt_stream, const ModelDev<Real>&, const EnvState<Real>&, const int*, const double*, const double*); \
    template void launch_query<Real, C>(DM_STEP_ARGS(Real));                                                                                    \
    template void launch_probe<Real, C>(unsigned, rt_stream, const ModelDev<Real>&, const EnvState<Real>&, const DebugTaps<Real>&, int, double);
#define DM_INST_EXPERT(Real, C) template void launch_amp_expert<Real, C>(unsigned, rt_stream, const ModelDev<Real>&, const double*, const double*, float*, const int*);

// family ids (keep in step with KIDS in the Makefile and tests/emu/Makefile)
#define DM_FAMILY(Real, ID)                                                     \
    DM_FAMILY_##ID(Real)
#define DM_FAMILY_0(Real) DM_INST_DUO(Real, SV_PLAIN)
#define DM_FAMILY_1(Real) DM_INST_DUO(Real, SV_AMP)
#define DM_FAMILY_2(Real) DM_INST_DUO(Real, SV_TAPS)
#define DM_FAMILY_3(Real) DM_INST_STEP(Real, ClsBiped, SV_PLAIN)
#define DM_FAMILY_4(Real) DM_INST_STEP(Real, ClsBiped, SV_AMP)
#define DM_FAMILY_5(Real) DM_INST_STEP(Real, ClsBiped, SV_TAPS)
#define DM_FAMILY_6(Real) DM_INST_STEP(Real, ClsLarge, SV_PLAIN)
#define DM_FAMILY_7(Real) DM_INST_STEP(Real, ClsLarge, SV_AMP)
#define DM_FAMILY_8(Real) DM_INST_STEP(Real, ClsLarge, SV_TAPS)
#define DM_FAMILY_9(Real) DM_INST_STEP(Real, ClsBipedObj, SV_AMP)
#define DM_FAMILY_10(Real) DM_INST_STEP(Real, ClsBipedObj, SV_TAPS)
#define DM_FAMILY_11(Real) DM_INST_MISC(Real, ClsBiped) DM_INST_MISC(Real, ClsBipedObj) DM_INST_MISC(Real, ClsLarge) DM_INST_MISC(Real, ClsLargeTree) DM_INST_MISC(Real, ClsBipedTree) DM_INST_EXPERT(Real, ClsBiped) DM_INST_EXPERT(Real, ClsLarge)
#define DM_FAMILY_12(Real) DM_INST_STEP(Real, ClsLargeTree, SV_PLAIN)
#define DM_FAMILY_13(Real) DM_INST_STEP(Real, ClsLargeTree, SV_AMP)
#define DM_FAMILY_14(Real) DM_INST_STEP(Real, ClsLargeTree, SV_TAPS)
#define DM_FAMILY_15(Real) DM_INST_STEP(Real, ClsBipedTree, SV_PLAIN)
#define DM_FAMILY_16(Real) DM_INST_STEP(Real, ClsBipedTree, SV_AMP)
#define DM_FAMILY_17(Real) DM_INST_STEP(Real, ClsBipedTree, SV_TAPS)
#define DM_FAMILY_18(Real) DM_INST_STEP(Real, ClsBiped, SV_V2)
#define DM_FAMILY_19(Real) DM_INST_STEP(Real, ClsLarge, SV_V2)
#define DM_FAMILY_20(Real) DM_INST_STEP(Real, ClsLargeTree, SV_V2)
#define DM_FAMILY_21(Real) DM_INST_STEP(Real, ClsBipedTree, SV_V2)
#define DM_FAMILY_22(Real) DM_INST_DUO(Real, SV_V2)
#define DM_FAMILY_23(Real) DM_INST_STEP(Real, ClsBipedObj, SV_V2)
#define DM_FAMILY_24(Real) DM_INST_DUOC(Real, ClsBipedObj, SV_AMP)

#ifdef DM_TU_ALL
#define DM_ALL(Real) DM_FAMILY_0(Real) DM_FAMILY_1(Real) DM_FAMILY_2(Real) DM_FAMILY_3(Real) DM_FAMILY_4(Real) DM_FAMILY_5(Real) \
    DM_FAMILY_6(Real) DM_FAMILY_7(Real) DM_FAMILY_8(Real) DM_FAMILY_9(Real) DM_FAMILY_10(Real) DM_FAMILY_11(Real) DM_FAMILY_12(Real) DM_FAMILY_13(Real) DM_FAMILY_14(Real) DM_FAMILY_15(Real) DM_FAMILY_16(Real) DM_FAMILY_17(Real) DM_FAMILY_18(Real) DM_FAMILY_19(Real) DM_FAMILY_20(Real) DM_FAMILY_21(Real) DM_FAMILY_22(Real) DM_FAMILY_23(Real) DM_FAMILY_24(Real)
DM_ALL(float)
DM_ALL(double)
#else
#if DM_TU_F64
typedef double TuReal;
#else
typedef float TuReal;
#endif
#define DM_FAMILY_X(Real, ID) DM_FAMILY(Real, ID)
DM_FAMILY_X(TuReal, DM_TU_ID)
#endif

}  // namespace dmk
