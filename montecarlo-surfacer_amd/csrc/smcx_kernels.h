// smcx_kernels.h -- host-visible launchers of the gfx950 kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "smcx_device.hpp"

namespace smcx {

// HIP events around every launch of the sweep kernel proper (not the helpers beside it, e.g. the z sort of
// sweep_kernel_mb64): what smcx_last_kernel_ms reports and rocprofv3's per-kernel average must agree with
struct SweepTimer {
    std::vector<hipEvent_t> evs; // start, stop, start, stop, ...
    int n = 0;                   // events recorded in the current run
    hipError_t mark(hipStream_t st)
    {
        if ((int)evs.size() <= n) {
            hipEvent_t e;
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return rc;
            evs.push_back(e);
        }
        return hipEventRecord(evs[n++], st);
    }
};

bool geometry_supported(int S, int WPR);
bool fp64_supported(int S, int WPR);
// fp32-screened sweep kernels (smcx_sweep_mx.hip)
bool mx_supported(int S, int WPR);
bool mx_lds_z(int S, int WPR, double Lz); // the variant with z as fp16 in LDS is the one launched
const char *mx_kernel_name(int S, int WPR, double Lz);
void mx_bound_values(double L, double Lz, double cutoff2, bool lds_z, double *thr, double *u2, double *toFix,
                     double *zsafe);
const char *fp64_kernel_name(int S, int WPR);
hipError_t launch_sweeps_mx(const SweepArgs &a, const DevCtx &c, int S, int WPR, int nsweeps, double A,
                            hipStream_t st);

// integer-screen sweep kernel for one wavefront per replica (smcx_sweep_mi.hip)
bool mi_supported(int S, int WPR, double L, double Lz, double cutoff2);
const char *mi_kernel_name(int S, int N, double L, double Lz, double cutoff2);
void mi_bound_values(double L, double Lz, double cutoff2, double *thr, double *u2, double *toFix, double *zsafe,
                     double *uz, int *negC, int *zshift);
hipError_t launch_sweeps_mi(const SweepArgs &a, const DevCtx &c, int S, int nsweeps, double A, hipStream_t st,
                            SweepTimer *tm);

// hand-scheduled form of the same kernel for 64 particles per lane (smcx_sweep_ma.hip)
bool ma_supported(int S, int WPR, int N, int M2);
const char *ma_kernel_name(int S, int N);
bool mb_supported(int S, int WPR, int N, int M2);
bool mc_supported(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2);
bool mc_box_supported(double L, double Lz, double cutoff2);
bool zordered_supported(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2);
bool mcw_supported(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2);
hipError_t launch_sweeps_mcw(const SweepArgs &s, const DevCtx &c, int WPR, int nsweeps, double A, hipStream_t st,
                             SweepTimer *tm);
void mc_bound_values(double L, double cutoff2, double *toFix, double *zsafe, int *negT, int *RZ);
hipError_t launch_sweeps_ma(const SweepArgs &s, const DevCtx &c, int S, const double *wtab, int nsweeps, double A,
                            double toFix, double zFix, double zsafe, int negC, hipStream_t st, SweepTimer *tm);

hipError_t launch_rng_prepass(const DevCtx &c, int nsweeps, double A, hipStream_t st);
// kernel: 0 = auto, 1 = fp64 kernels, 2 = screened kernel (smcx_sweep_mx.hip)
hipError_t launch_sweeps(const DevCtx &c, int S, int WPR, int nsweeps, double A, int kernel, hipStream_t st,
                         SweepTimer *tm = nullptr);
bool sweep_uses_mx(int S, int WPR, int kernel);
hipError_t launch_finalize(const DevCtx &c, int nsweeps, int production, int sweep_base,
                           int first_production, hipStream_t st);
hipError_t launch_hist(const DevCtx &c, hipStream_t st);
hipError_t launch_pressure(const DevCtx &c, int gather, hipStream_t st);
// clusterAnalysis of replicas a.rep0 .. a.rep0+nbatch-1, counters accumulated (smcx_lca.hip)
hipError_t launch_lca(const LcaArgs &a, int nbatch, hipStream_t st);
hipError_t launch_total_energy(const DevCtx &c, double *out, hipStream_t st);
hipError_t launch_eval_moves(const DevCtx &c, const int *nsel, const double *prop, double *out,
                             hipStream_t st);
hipError_t launch_pack_obs(const DevCtx &c, double *dst, hipStream_t st);
// op 0: zero the accumulators of every replica and remember Ecur in save[];
// op 1: put save[] back into Ecur; op 2: Ecur <- save[] (set from host values)
hipError_t launch_obs_op(const DevCtx &c, double *save, int op, hipStream_t st);

} // namespace smcx
