/*
 * smcx.h -- C ABI of the MI355X-native Smart-Monte-Carlo engine.
 *
 * This is the drop-in boundary for the hot path of Kryohi/MonteCarlo-Surfacer:
 * the force-biased single-particle sweep `oneParticleMoves` and the kernels it
 * calls, the chain bookkeeping of `sMC`, and the density histogram -- batched
 * over many independent replica chains on one GPU.  The reference has no
 * plugin/FFI layer; what this header replaces are the plain C prototypes of
 * SMC.h:92-114.  Each entry point below names the reference interface it
 * stands in for.  See INTEGRATION.md for the binding a reference maintainer
 * would add.
 *
 * Conventions
 *  - plain C types only; every call returns an int status (SMCX_OK == 0);
 *    nothing aborts or exits (the reference returns void and prints,
 *    SMC.c:99-100, 428, 463).
 *  - positions are the reference layout: AoS double r[3N] = x0,y0,z0,x1,...
 *    (SMC.c:44, 567-571), one block of 3N per replica.
 *  - wall strengths W[2*M*M] = a0,b0,a1,b1,... (SMC.c:495-496).
 *  - N, M, cutoff, a0, b0, Ncx, Ncz are run-time fields here; the reference
 *    bakes them in as macros (SMC.h:26-58).
 *  - RNG: each replica owns a glibc-compatible rand() stream (TYPE_3 additive
 *    feedback, what srand()/rand() are on glibc 2.35); it is explicit state in
 *    the handle, not libc's hidden global (SMC.c:40, 290, 335).
 *  - a handle is confined to one host thread and one GPU.
 */
#ifndef SMCX_H
#define SMCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMCX_VERSION 100 /* 0.1.0 */

/* status codes */
#define SMCX_OK 0
#define SMCX_ERR_PARAM 1       /* invalid argument / parameter combination */
#define SMCX_ERR_HIP 2         /* a HIP runtime call failed (smcx_last_error_string) */
#define SMCX_ERR_STATE 3       /* call order violated (e.g. run before upload) */
#define SMCX_ERR_NOMEM 4
#define SMCX_ERR_UNSUPPORTED 5 /* valid request this build cannot serve */
#define SMCX_ERR_NODEVICE 6    /* no HIP device visible */
#define SMCX_ERR_RCCL 7        /* the RCCL observable gather across the node's GPUs failed (smcx_host_sMC_multi) */

/* flags */
#define SMCX_FLAG_WALLS 0x1u      /* K3/K4 wall terms on (SMC.c:300-304); off = particles only */
#define SMCX_FLAG_E0_RESTART 0x2u /* production energy series restarts from the pre-
                                     thermalisation energy, as SMC.c:116-117 vs 194 does */
#define SMCX_FLAG_SERIES 0x4u     /* keep per-sweep E and accepted-count series (data_*.csv
                                     columns, SMC.c:214-215) for the last smcx_run */
#define SMCX_FLAG_FULL_HIST 0x8u   /* keep the full Ncx x Ncx x Ncz density and mobility counters of
                                     localDensityAndMobility (local_*.csv, SMC.c:218-225) */
#define SMCX_FLAG_CLUSTERS 0x20u   /* clusterAnalysis every lca_time-th gather (SMC.c:143-155) */
#define SMCX_FLAG_PRESSURE 0x10u   /* evaluate pressure + wallsPressure at every gather (SMC.c:140) */
#define SMCX_FLAGS_REFERENCE (SMCX_FLAG_WALLS | SMCX_FLAG_E0_RESTART)

/* smcx_params.tune_kernel: measurement switch, not needed in normal use.  The screened sweep kernels are a ladder,
 * each rung a faster form of the same algorithm with identical results to rounding; a value >= 3 stops the choice at
 * that rung (where the geometry and box allow it at all).  SMCX_KERNEL_MA is the last rung whose screen visits EVERY
 * cell for every probe, as the reference's loops do (SMC.c:563-578): the like-for-like "all pairs" kernel. */
#define SMCX_KERNEL_AUTO 0
#define SMCX_KERNEL_FP64 1     /* sweep_kernel / sweep_kernel_lead: positions and tests in fp64 */
#define SMCX_KERNEL_SCREENED 2
#define SMCX_KERNEL_MX 3       /* sweep_kernel_mx: int16 x,y + fp32/fp16 z screen, any number of wavefronts */
#define SMCX_KERNEL_MI 4       /* sweep_kernel_mi: all-integer screen, one wavefront per replica */
#define SMCX_KERNEL_MA 5       /* sweep_kernel_ma16/32/64: the same, hand-scheduled */
#define SMCX_KERNEL_MB 6       /* sweep_kernel_mb64: cells in z order, only groups in reach are screened */
#define SMCX_KERNEL_MC 7       /* sweep_kernel_mc16/32/64, mc32x4, mc64x4, mc32x8: one word per cell (default where built) */
#define SMCX_KERNEL_MT 8       /* sweep_kernel_mt64x8: the same with TWO TEAMS of wavefronts per replica, probe A and probe B
                                  evaluated side by side; built for 8192 < N <= 16384 ONLY (the plan's choice there up to 256
                                  replicas per GPU): asked for at any other N, smcx_create answers SMCX_ERR_UNSUPPORTED */

typedef struct smcx_params {
    int32_t N;       /* particles per replica            (SMC.h:29)  even, >= 2 */
    int32_t M;       /* wall sites per side, M*M total   (SMC.h:26)  M*M+1 <= 30 */
    int32_t nrep;    /* replica chains held by this handle (the reference: one per MPI rank) */
    int32_t device;  /* HIP device ordinal */
    double L, Lz;    /* box: L x L x Lz, periodic in x,y only (main.c:35-44) */
    double T;        /* temperature (main.c:18) */
    double A;        /* SMC step parameter A = gamma*T (main.c:48-51) */
    double cutoff;   /* LJ_CUTOFF (SMC.h:38) */
    double a0, b0;   /* featureless-plane 12-6 coefficients (SMC.h:32-33) */
    int32_t Ncx, Ncz; /* histogram cells (SMC.h:53-55); Ncx,Ncz <= 255 */
    uint32_t flags;  /* SMCX_FLAG_* */
    uint32_t base_seed;     /* replica r is seeded srand(base_seed + first_replica + r) */
    uint32_t first_replica; /* global index of this handle's first replica (multi-GPU shard) */
    int32_t tune_slots;     /* 0 = auto; else particles per lane (1,2,4,...,64) */
    int32_t tune_waves;     /* 0 = auto; else wavefronts per replica (1,2,4,8,16) */
    int32_t lca_time;       /* LCA_TIME: cluster analysis every lca_time-th gather (SMC.h:48) */
    int32_t tune_kernel;    /* SMCX_KERNEL_*: 0 = auto; 1 = fp64 sweep kernels; 2 = screened (compact-copy) sweep
                               kernels, best form; 3..7 = the best screened form not above that rung */
    int32_t tune_resort;    /* sweeps per z sort of the one-wavefront z-ordered kernels: 0 = default (2), 1 = before every sweep, k */
    double lca_cutoff;      /* LCA_cutoff (SMC.h:50) */
} smcx_params;

/* fills *p with the reference's defaults (SMC.h macros, main.c:35-51: L=33,
 * Lz=240, T=A=1.1, cutoff 3, M=3, 33x33x33 cells, seed 12345) for given N,nrep */
void smcx_default_params(smcx_params *p, int32_t N, int32_t nrep);

typedef struct smcx_handle smcx_handle;

int smcx_device_count(int *count);
int smcx_create(const smcx_params *p, smcx_handle **out);
int smcx_destroy(smcx_handle *h);
const char *smcx_strerror(int status);
/* text of the last HIP error seen by this handle (h may be NULL: process-wide) */
const char *smcx_last_error_string(const smcx_handle *h);

/* Load initial state: R0 (shared [3N] if r0_per_replica==0, else [nrep][3N]; x,y must lie
 * in [-L/2, L/2] as initializeBox and the sweep itself leave them (SMC.c:315-316, 461), and
 * |z| <= 4 Lz: SMCX_ERR_PARAM otherwise),
 * W[2*M*M] (may be NULL when walls are off), seeds[nrep] (NULL = base_seed
 * rule).  Seeds the RNG streams, zeroes observables and evaluates
 * E[0] = energy + wallsEnergy on the device (SMC.c:44-48). */
int smcx_upload(smcx_handle *h, const double *R0, int r0_per_replica, const double *W,
                const uint32_t *seeds);

/* The loop of sMC (SMC.c:108-126, 134-196) for every replica:
 * `eqsteps` thermalisation sweeps at 2A, then `maxsteps` production sweeps at
 * A with the density histogram taken when (n+1) % gather_lapse == 0, before
 * that sweep's moves.  Observables are reset at entry and describe this call. */
int smcx_run(smcx_handle *h, int eqsteps, int maxsteps, int gather_lapse);

/* Results of the last smcx_run; any pointer may be NULL.
 *  acceptance_ratio[nrep] = intmean(jj)/N                      (SMC.c:248)
 *  meanE[nrep]            = mean(E[0..maxsteps] + 3NT/2)       (SMC.c:210-211, 244)
 *  dE[nrep]               = sqrt(variance(E))                  (SMC.c:245)
 *  zhist[nrep][Ncz]       = sum over i,j of the cumulative D[i][j][k] of
 *                           localDensityAndMobility (SMC.c:912-927; plotting.jl:134-166)
 *  accepted[nrep]         = total accepted moves in production
 *  E_last[nrep]           = last entry of the energy series (without 3NT/2) */
int smcx_observables(smcx_handle *h, double *acceptance_ratio, double *meanE, double *dE,
                     uint64_t *zhist, uint64_t *accepted, double *E_last);
/* thermalisation acceptance of the last run, intmean(jt)/N (SMC.c:124) */
int smcx_therm_acceptance(smcx_handle *h, double *ratio);
/* number of histogram calls and of out-of-range cells (undefined behaviour in
 * the reference: counted, not written) per replica; either may be NULL */
int smcx_hist_info(smcx_handle *h, uint64_t *gathers, uint64_t *oob);

/* per-sweep series of the last run (needs SMCX_FLAG_SERIES):
 * E_series[nrep][maxsteps+1] without 3NT/2, jj[nrep][maxsteps] */
int smcx_series(smcx_handle *h, double *E_series, int32_t *jj);

/* full cell counters of the last run (needs SMCX_FLAG_FULL_HIST): D and Mu of
 * localDensityAndMobility (SMC.c:912-927), [nrep][Ncx*Ncx*Ncz] each, cell index
 * i*Ncx*Ncz + j*Ncz + k; either may be NULL */
int smcx_density(smcx_handle *h, uint64_t *D, uint64_t *Mu);
/* pressure(R) + wallsPressure(R) of every gather of the last run, in gather order (needs
 * SMCX_FLAG_PRESSURE): P[nrep][ngathers]; without the ideal-gas term rho*T the reference adds
 * at SMC.c:207-208, and with wallsPressure's geometry as the reference has it (SMC.c:880) */
int smcx_pressure_series(smcx_handle *h, double *P, int *ngathers);

/* ---- common-neighbour cluster analysis (clusterAnalysis, SMC.h:116, SMC.c:971-1045) ----
 * With SMCX_FLAG_CLUSTERS smcx_run analyses every replica at the gathers k with
 * k % lca_time == 0 (k = 1, 2, ... counts gathers, SMC.c:138, 143) and accumulates, per replica,
 * what the loop at SMC.c:146-155 walks over: n1 = pair entries with num1 != 0, h2[v] / h3[v] =
 * those of them with num2 == v / num3 == v (v > 15 lands in bin 15).  These are plain counts:
 * the reference adds the weight 1/(gather_steps/LCA_TIME) -- an int division, 0 unless the
 * ratio is 1 -- into l1, l2[num2], l3[num3]; a caller applies whatever weight it wants.
 * The reference's pair index (l*l-3*l+2)/2+i, under which (l,l-1) and (l+1,0) share an entry,
 * its (i,i2) look-up for i > i2 and its consecutive-only bond test are reproduced exactly;
 * stores past its common_nn[8] (undefined behaviour there) are dropped and counted in
 * `overflow`.  Any output pointer may be NULL.  Counters restart at every smcx_run. */
int smcx_cluster_counts(smcx_handle *h, uint64_t *n1 /*[nrep]*/, uint64_t *h2 /*[nrep][16]*/,
                        uint64_t *h3 /*[nrep][16]*/, uint64_t *overflow /*[nrep]*/,
                        int *analyses);
/* analyse the positions as they are now and add to the counters (any flags) */
int smcx_cluster_update(smcx_handle *h);
/* clusterAnalysis(r, N, L, LCA) for one replica's current positions: LCA[3*idx + {0,1,2}] =
 * num1, num2, num3, N(N-1)/2 entries (SMC.c:1038-1044); counters are not touched */
int smcx_cluster_analysis(smcx_handle *h, int replica, int32_t *LCA, uint64_t *overflow);

/* Autocorrelation of the production energy series of the last run, as the reference's
 * fft_acf computes it (SMC.c:1051-1089, called at SMC.c:234 with KMAX = 2500000), for
 * every replica (needs SMCX_FLAG_SERIES): acf[nrep][*k_eff] with k_eff = k_max, or
 * (maxsteps+1)/2 - 2 when the series is shorter than 2 k_max + 1 (SMC.c:1054-1057);
 * tau[nrep] = sum(acf) (SMC.c:235); cv[nrep] = variance(E)/T^2 (SMC.c:250).  Pass acf = NULL
 * to ask for k_eff only... all output pointers may be NULL. */
int smcx_acf(smcx_handle *h, int k_max, double *acf, int *k_eff, double *tau, double *cv);

/* current positions, [nrep][3N] (struct Sim.Rfinal, SMC.h:84) */
int smcx_download_positions(smcx_handle *h, double *R);

/* energy + wallsEnergy of the current positions, recomputed from scratch
 * (SMC.c:626-646, 822-859), E[nrep] */
int smcx_total_energy(smcx_handle *h, double *E);

/* RNG checkpoint: 32 words per replica (31 most recent outputs' state words,
 * oldest first, then the count of generated-but-unconsumed words) */
int smcx_rng_export(smcx_handle *h, uint32_t *state);
int smcx_rng_import(smcx_handle *h, const uint32_t *state);

/* Copy packed per-replica observables of the last run into DEVICE memory the
 * caller owns (for an RCCL gather across ranks): nrep records of
 * SMCX_OBS_RECORD_DOUBLES doubles {accepted, nsamples, sumE, sumE2, E_last,
 * therm_accepted, gathers, oob} followed by nrep*Ncz doubles of zhist.
 * bytes must be smcx_obs_device_bytes(h). */
#define SMCX_OBS_RECORD_DOUBLES 8
size_t smcx_obs_device_bytes(const smcx_handle *h);
int smcx_export_observables_device(smcx_handle *h, void *dst_device, size_t bytes);

/* timing of the sweep kernel launches of the last smcx_run (HIP events around each
 * launch, on the launch stream): their summed milliseconds and their number */
int smcx_last_kernel_ms(smcx_handle *h, double *ms, int *launches);
/* the shader clock the LAST sweep kernel launch ran at, measured in the kernel (s_memtime over
 * s_memrealtime, median over the replicas' wavefronts), and a wavefront's lifetime in shader cycles */
int smcx_last_clock(smcx_handle *h, double *ghz, double *wave_cycles);
/* diagnostics: lifetimes of the wavefronts of the LAST sweep kernel launch in microseconds (100 MHz counter read
 * by every wavefront at its first and last instruction): out4 = min, median, max over the replicas, and the span
 * from the first start to the last end.  With four wavefronts per SIMD the kernel lasts as long as its slowest
 * wavefront; sweep_kernel_mb64 / mc* steer their issue priorities so that all finish together. */
int smcx_debug_wave_spread(smcx_handle *h, double *out4);
/* diagnostics: the raw stamps behind smcx_last_clock / smcx_debug_wave_spread, out[nrep][4] */
int smcx_debug_clk_rows(smcx_handle *h, uint64_t *out);
/* diagnostics (host only, no GPU needed): the launches a group of `nsweeps` sweeps of `nrep` replicas is cut into when the device
 * holds `granule` of them at once and a z sort covers `every` sweeps -- plain launches (one per block over all replicas) when
 * nrep <= granule or nrep is a multiple of it, otherwise windows of `granule` consecutive (replica, block) units
 * (smcx_replica_granule).  out[k] = {workgroups, first unit u0, nrep or 0, block of the first unit, 1 if a window, first sweep
 * and sweep count of that block, first sweep and sweep count of the next block}; returns the number of launches (at most max
 * are written) or a negative status.  What the tests check the schedule's invariants on. */
int smcx_debug_window_schedule(int nrep, int granule, int nsweeps, int every, int32_t *out /*[max][9]*/, int max);
/* device time of the whole last smcx_run (RNG pre-pass and bookkeeping kernels included) */
int smcx_last_run_ms(smcx_handle *h, double *ms);
/* the launch geometry chosen for this handle */
int smcx_geometry(const smcx_handle *h, int *slots, int *waves_per_replica, int *lds_bytes);
/* which sweep kernel this handle launches: 1 = fp64 (positions as fp64 in registers), 2 = screened
 * (compact integer / fp16 / fp32 copies of the positions pre-select pairs, fp64 evaluation of the
 * candidates: sweep_kernel_mi with one wavefront per replica, sweep_kernel_mx with several); name,
 * if not NULL, receives the kernel's name (at most len bytes) */
int smcx_kernel_form(const smcx_handle *h, int *form, char *name, int len);
/* how many replicas the device runs at once with this handle's sweep kernel (workgroups resident per CU x CUs; 0 if unknown
 * for this kernel form).  A sweep is sequential inside a replica, so a launch of more replicas than that runs in rounds.  Since
 * round 5 a replica count that is no multiple of the granule no longer costs a nearly empty last round per launch: the sweeps
 * between two gathers are cut into blocks (one z sort each) and launched as WINDOWS of `granule` (replica, block) units -- the
 * replicas of a launch sit in different blocks, legal because the chains are independent (the reference's MPI ranks,
 * SMC.c:40, 66-95) -- so 4097 replicas of N = 4096 cost 1.16 x the time of 4096 (was 1.55 x), 6144 cost 1.63 x (was 1.95 x) at a
 * gather every 10 sweeps, less with rarer gathers (profiles/r05_replica_cliff.txt).  Results are bit-identical to plain
 * launches.  note (optional, len bytes) receives a one-paragraph advisory when nrep is not a multiple of the granule, ""
 * otherwise.  Multiples of the granule remain the most efficient sizes. */
int smcx_replica_granule(smcx_handle *h, int *granule, char *note, int len);
/* source identity of a sweep kernel of THIS library (host only, no GPU needed): 16 hex digits, the sha256 of the generated
 * body (hand-scheduled kernels) or of the source files (compiled ones) it was built from.  kernel = a name as
 * smcx_kernel_form returns it.  Measurement plumbing: profiles/kernel_counters.json records the id of the library whose
 * instruction counts it holds, and bench.py reports no roofline fraction from counts of another build.
 * SMCX_ERR_PARAM for a name this library does not know. */
int smcx_kernel_source_id(const char *kernel, char *id, int len);
/* the numbers of the screened kernel's conservative cutoff test for a box (host only, no GPU needed):
 * thr = cutoff^2 + margin (fp32), u2 = (L/65536)^2, to_fixed = 65536/L, zsafe = |z| up to which the
 * margin holds; lds_z selects the variant with z as fp16.  tests/test_cabi_host.py checks on the CPU,
 * with the device's arithmetic emulated, that no pair inside the cutoff escapes the test. */
int smcx_screen_bound(const smcx_params *p, int lds_z, double *thr, double *u2, double *to_fixed, double *zsafe);
/* the same for the integer screen of sweep_kernel_mi (one wavefront per replica): thr = threshold in
 * length^2, u2 = (L/65536)^2, uz = the z unit 2^zshift L/65536, neg_c = the accumulator start
 * -(T << 2 zshift) of the x,y dot product; SMCX_ERR_UNSUPPORTED when no built z unit covers the box */
int smcx_screen_bound_int(const smcx_params *p, double *thr, double *u2, double *to_fixed, double *zsafe,
                          double *uz, int32_t *neg_c, int32_t *zshift);
/* the same for the byte screen of sweep_kernel_mc64 (one 32-bit word per particle: z as int16, x and y as
 * int8, all in units of L/256; v_sub_u32 + v_dot4_i32_i8): to_fixed = 256/L, zsafe = |z| the int16 holds,
 * neg_t = the accumulator start -T of the four squared bytes, reach_z = |dz| in units a pair inside the
 * cutoff can have (what a group's z range is widened by); SMCX_ERR_UNSUPPORTED when the unit does not
 * resolve the cutoff (fewer than 16 units) or the box is taller than the int16 */
int smcx_screen_bound_byte(const smcx_params *p, double *to_fixed, double *zsafe, int32_t *neg_t, int32_t *reach_z);

/* Teacher-forced evaluator (stateless; tests and debugging): for each of nrep
 * replicas evaluates what SMC.c:300-304 and 319-321 evaluate for particle
 * n[r]: Um,Fm at R[r][n] and Un,Fn with particle n placed at prop[r].
 * R [nrep][3N], n [nrep], prop [nrep][3], out [nrep][8] = Um,Fmx,Fmy,Fmz,Un,Fnx,Fny,Fnz */
int smcx_eval_moves(const smcx_params *p, const double *R, const double *W, const int32_t *n,
                    const double *prop, double *out);

/* Single-chain shim with the contract of
 *   void oneParticleMoves(double *R, double *Rn, const double *W, double L,
 *                         double Lz, double A, double T, int *j, double *U)   (SMC.h:102)
 * plus what its macros and libc's hidden rand() state carried implicitly.
 * R and Rn are updated in place, *j and *U are accumulated into, rng (32
 * words, smcx_rng_export layout) is advanced by the 4N+1 draws of one sweep. */
void smcx_rng_seed(uint32_t *rng, uint32_t seed); /* srand(seed) */
int smcx_one_particle_moves(const smcx_params *p, uint32_t *rng, double *R, double *Rn,
                            const double *W, double A, double T, int *j, double *U);

#ifdef __cplusplus
}
#endif
#endif /* SMCX_H */
