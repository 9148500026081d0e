// Launchers of the env kernels.  libdm_hip.so is built from several translation units: dm_host.cpp (tables, C-ABI, policy
// kernels) only sees these declarations; dm_kernels.cpp is compiled once per (precision, kernel family) and instantiates
// exactly one family per object file (Makefile: k_<prec>_<id>.o), so the 30+ step-kernel instantiations compile in parallel.
#pragma once
#ifdef DM_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include "dm_types.h"

#ifdef DM_EMU
typedef void* rt_stream;
#define RT_LAUNCH(kern, grid, stream, ...) emu::launch((unsigned)(grid), 64, [&]() { kern(__VA_ARGS__); })
#define RT_LAUNCH4(kern, grid, stream, ...) emu::launch((unsigned)(grid), 256, [&]() { kern(__VA_ARGS__); })
#else
typedef hipStream_t rt_stream;
#define RT_LAUNCH(kern, grid, stream, ...) hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(64), 0, stream, __VA_ARGS__)
#define RT_LAUNCH4(kern, grid, stream, ...) hipLaunchKernelGGL(kern, dim3((unsigned)(grid)), dim3(256), 0, stream, __VA_ARGS__)   // four wavefronts per workgroup
#endif

namespace dmk {

// step-kernel variants: the plain production instantiation, the AMP / goal / perturbation instantiation, the tap build
enum { SV_PLAIN = 0, SV_AMP = 1, SV_TAPS = 2, SV_V2 = 3 };      // SV_V2: the AMP instantiation + DM-physics v2 (one character per wavefront only)

template <typename Real, int V>
void launch_step_duo(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg);
template <typename Real, typename C, int V>
void launch_step_duo_c(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg);      // two characters per wavefront of class C (round 6: ClsBipedObj)
template <typename Real, typename C, int V>
void launch_step(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg);
template <typename Real, typename C>
void launch_reset(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const int* env_ids, const double* kin_times, const double* max_times);
template <typename Real, typename C>
void launch_query(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const StepIO<Real>& io, const DebugTaps<Real>& dbg);
template <typename Real, typename C>
void launch_probe(unsigned grid, rt_stream s, const ModelDev<Real>& m, const EnvState<Real>& st, const DebugTaps<Real>& dbg, int what, double dt);
template <typename Real, typename C>
void launch_amp_expert(unsigned grid, rt_stream s, const ModelDev<Real>& m, const double* times, const double* ground_h, float* out, const int* clips);

}  // namespace dmk
