// smcx_kernels.h -- host-visible launchers of the gfx950 kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "smcx_device.hpp"

namespace smcx {

// HIP events around every launch of the sweep kernel proper (not the helpers beside it, e.g. the z sort of
// sweep_kernel_mb64): what smcx_last_kernel_ms reports and rocprofv3's per-kernel average must agree with
struct SweepTimer {
    // A production run launches the z-ordered kernels once per sweep, 1e6-1e7 times: the events are a ring of RING
    // (512 launches may be in flight); when a slot comes round again its pair is folded into sum_ms first, waiting
    // for it if the device is that far behind (it never is: the host then merely stops running ahead).
    static constexpr long RING = 1024;
    std::vector<hipEvent_t> evs; // start, stop, start, stop, ...
    long n = 0;                  // events recorded in the current run (2 per launch)
    long folded = 0;             // events whose pair is already in sum_ms
    double sum_ms = 0.0;
    void reset() { n = 0; folded = 0; sum_ms = 0.0; }
    hipError_t fold_until(long upto) // pairs below event index `upto`
    {
        for (; folded + 1 < upto; folded += 2) {
            const long i = folded % RING;
            hipError_t rc = hipEventSynchronize(evs[i + 1]);
            float t = 0.f;
            if (rc == hipSuccess) rc = hipEventElapsedTime(&t, evs[i], evs[i + 1]);
            if (rc != hipSuccess) return rc;
            sum_ms += t;
        }
        return hipSuccess;
    }
    hipError_t mark(hipStream_t st)
    {
        const long slot = n % RING;
        if ((long)evs.size() <= slot) {
            hipEvent_t e;
            hipError_t rc = hipEventCreate(&e);
            if (rc != hipSuccess) return rc;
            evs.push_back(e);
        } else if (n >= RING && (n & 1) == 0) {
            hipError_t rc = fold_until(n - RING + 2);
            if (rc != hipSuccess) return rc;
        }
        n++;
        return hipEventRecord(evs[slot], st);
    }
    hipError_t finish() { return fold_until(n); } // after the stream has been synchronised
    int launches() const { return (int)(n / 2); }
};

// Which sweep kernel a handle runs is decided ONCE, in smcx_create, from smcx_params (tune_kernel, tune_resort) and --
// only when SMCX_ALLOW_ENV_TUNING=1 -- from the SMCX_* measurement switches; allocation, every launcher and
// smcx_kernel_form read this plan, nothing re-derives it.  The forms are a ladder: a cap stops the choice below it.
enum SweepForm { FORM_NONE = 0, FORM_FP64 = 1, FORM_MX = 3, FORM_MI = 4, FORM_MA = 5, FORM_MB = 6, FORM_MC = 7,
                 FORM_MT = 8 }; // MT: the mc kernel with two teams of wavefronts per replica (latency-bound configurations)
struct Tune {
    int kernel = 0;      // smcx_params.tune_kernel: 0 auto, 1 fp64, 2 screened (auto), 3..7 highest SweepForm allowed
    int lead = -1;       // fp64 kernels with several wavefronts, leader/follower form: -1 auto, 0 never, 1 always
    int mz = -1;         // sweep_kernel_mx with z as fp16 in LDS: -1 auto, 0 never, 1 always
    int resort = 1;      // sweeps per z sort of the z-ordered kernels (smcx_params.tune_resort)
    int zsort_tpb = 512; // threads of zsort_kernel for 4096 cells (128, 256, 512, 1024)
    int check_mb = 0;    // diagnostic build only: 1 = sweep_kernel_mb64, 2 = sweep_kernel_mc64 with the fp64 test beside
    int windows = 1;     // launch groups as windows of units when nrep is no multiple of the resident count (VARIANT: SMCX_NO_WINDOWS)
};
struct KernelPlan {
    int form = FORM_NONE, S = 0, WPR = 0;
    int zs = 0;          // z unit shift of the integer screen (FORM_MI and above)
    bool lead = false, mz = false;
    bool lpos = false;   // sweep_kernel_ml16: mc16 with the cells' fp64 positions in LDS (few replicas of N <= 1024)
    Tune tune;
    const char *name = ""; // the launched instantiation as rocprofv3 prints it
    bool zordered() const { return form >= FORM_MB; } // needs Rs, loc and the z sort
};
// M2 = wall sites (0 without walls); false if no kernel is built for (S, WPR)
bool plan_kernel(int N, int M2, double L, double Lz, double cutoff2, int S, int WPR, const Tune &t, KernelPlan *out);

bool geometry_supported(int S, int WPR);
bool fp64_supported(int S, int WPR);
// fp32-screened sweep kernels (smcx_sweep_mx.hip)
bool mx_supported(int S, int WPR);
bool mx_lds_z(int S, int WPR, double Lz, int force); // the variant with z as fp16 in LDS is the one launched
const char *mx_kernel_name(int S, int WPR, bool mz);
void mx_bound_values(double L, double Lz, double cutoff2, bool lds_z, double *thr, double *u2, double *toFix,
                     double *zsafe);
const char *fp64_kernel_name(int S, int WPR, bool lead);
hipError_t launch_sweeps_mx(const SweepArgs &a, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A,
                            hipStream_t st);

// integer-screen sweep kernel for one wavefront per replica (smcx_sweep_mi.hip)
int mi_built(int S, double L, double Lz, double cutoff2); // z unit shift of the built kernel serving this box, 0 = none
const char *mi_kernel_name(int S, int zs);
void mi_bound_values(double L, double Lz, double cutoff2, double *thr, double *u2, double *toFix, double *zsafe,
                     double *uz, int *negC, int *zshift);
hipError_t launch_sweeps_mi(const SweepArgs &a, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A,
                            hipStream_t st, SweepTimer *tm);

// hand-scheduled form of the same kernel for 64 particles per lane (smcx_sweep_ma.hip)
bool ma_built(int S, int N, int M2);           // one wavefront per replica, 32 S < N <= 64 S
int ma_cap(const Tune &t, int S);              // highest form the build offers for S (the diagnostic build: by check_mb)
const char *ma_kernel_name(int form, int S, int WPR);
bool mc_box_supported(double L, double Lz, double cutoff2);
bool mcw_built(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2);
bool mt_built(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2);
hipError_t launch_sweeps_mt(const SweepArgs &s, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A, hipStream_t st,
                            SweepTimer *tm);
hipError_t launch_sweeps_mcw(const SweepArgs &s, const DevCtx &c, int WPR, int nsweeps, double A, hipStream_t st,
                             SweepTimer *tm);
void mc_bound_values(double L, double cutoff2, double *toFix, double *zsafe, int *negT, int *RZ);
int ma_resident_replicas(const KernelPlan &pl, int device); // replicas the device runs at once with this kernel (0: unknown)
hipError_t launch_sweeps_ma(const SweepArgs &s, const DevCtx &c, const KernelPlan &pl, const double *wtab, int nsweeps,
                            double A, double toFix, double zFix, double zsafe, int negC, hipStream_t st, SweepTimer *tm);

hipError_t launch_rng_prepass(const DevCtx &c, int nsweeps, double A, hipStream_t st);
hipError_t launch_sweeps(const DevCtx &c, const KernelPlan &pl, int nsweeps, double A, hipStream_t st,
                         SweepTimer *tm = nullptr);
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
