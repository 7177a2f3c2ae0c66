/*
 * smc_oracle.h -- CPU restatement of the Smart-Monte-Carlo hot path of
 * Kryohi/MonteCarlo-Surfacer.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (montecarlo-surfacer_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * Every function cites the reference file:line it restates.  Everything is
 * fp64, AoS positions r[3N] = x0,y0,z0,x1,... exactly as the reference, with
 * N and M as run-time values (the reference bakes them in as macros,
 * SMC.h:26,29).  Build with -O2 -ffp-contract=off so the arithmetic is the
 * reference's own (no FMA contraction).
 *
 * PARITY PIN STATUS (see oracle/README.md and DESIGN.md section 2): PINNED on the real reference.
 *   oracle/build_ref.sh compiles the hot-path line ranges of the real SMC.c and
 *   SMC_noMPI_noWall.c (and the whole matematicose.c) from /root/reference where they lie into
 *   oracle/_ref/, one library per compile-time N; tests/test_ref_pin.py demands bit-identical
 *   outputs of every orc_* function on the committed fixtures tests/golden/ref_smc.json
 *   (generated from those libraries) and live where the libraries are present.  Not pinned on
 *   reference code: the FFT autocorrelation (fft_acf needs FFTW, absent: numpy restatement in
 *   tests/oracle_lib.py), and sMC itself (needs FFTW and misccose.c; its loop is replayed around
 *   the real functions).
 */
#ifndef SMC_ORACLE_H
#define SMC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- glibc TYPE_3 additive-feedback generator ------ */
typedef struct orc_rng {
    uint32_t s[31];
    int32_t f, r; /* front / rear indices */
} orc_rng;

#define ORC_RAND_MAX 2147483647

void orc_srand(orc_rng *g, unsigned int seed);
int orc_rand(orc_rng *g);

/* matematicose.c:183-193 (walls variant) */
void orc_vec_box_muller(orc_rng *g, double sigma, size_t length, double *A);
/* SMC_noMPI_noWall.c:707-717 (older variant: /RAND_MAX, sigma inside sqrt) */
void orc_vec_box_muller_nw(orc_rng *g, double sigma, size_t length, double *A);

/* ---- system description (the reference's macros as run-time values) ----- */
typedef struct orc_sys {
    int32_t N;      /* SMC.h:29 */
    int32_t M;      /* SMC.h:26 */
    double L, Lz;   /* main.c:35-44 */
    double cutoff;  /* SMC.h:38 LJ_CUTOFF */
    double a0, b0;  /* SMC.h:32-33 */
    int32_t Ncx, Ncz; /* SMC.h:53-55 */
} orc_sys;

/* one record per trial move, filled by orc_one_particle_moves when asked */
typedef struct orc_move_trace {
    int32_t n;        /* particle moved */
    int32_t accepted;
    double Um, Fm[3];
    double delta[3];
    double prop[3];   /* proposed position after the x,y wrap */
    double Un, Fn[3];
    double ap, u;
} orc_move_trace;

/* K1  SMC.c:557-583 */
double orc_energy_single(const orc_sys *s, const double *r, int i);
/* K2  SMC.c:589-618 */
void orc_force_single(const orc_sys *s, const double *r, int i, double F[3]);
/* K3  SMC.c:729-763 */
double orc_walls_energy_single(const orc_sys *s, double rx, double ry, double rz,
                               const double *W);
/* K4  SMC.c:773-813 (adds into F) */
void orc_walls_force(const orc_sys *s, double rx, double ry, double rz,
                     const double *W, double F[3]);
/* K5  SMC.c:626-646, 822-859 */
double orc_energy(const orc_sys *s, const double *r);
double orc_walls_energy(const orc_sys *s, const double *r, const double *W);

/* virial pressure, SMC.c:696-720, and its wall part, SMC.c:862-895 -- restated with
 * the reference's quirks: the wall distance uses L/2 where Lz/2 is meant (:880), is not
 * clamped, and the plane term is added once per wall SITE inside that site's cutoff */
double orc_pressure(const orc_sys *s, const double *r);
double orc_walls_pressure(const orc_sys *s, const double *r, const double *W);

/* S1  SMC.c:278-351; trace may be NULL, else N records */
void orc_one_particle_moves(const orc_sys *s, orc_rng *g, double *R, double *Rn,
                            const double *W, double A0, double T, int *j,
                            double *E, orc_move_trace *trace);

/* H   SMC.c:912-927.  D, Mu: Ncx*Ncx*Ncz counters; Rbin: N ints.
 * Out-of-range cell numbers (undefined behaviour in the reference) are
 * counted in *oob instead of being written. */
void orc_local_density(const orc_sys *s, const double *r, uint64_t *D,
                       int32_t *Rbin, uint64_t *Mu, uint64_t *oob);

/* ---- chain driver: the loop structure of sMC, SMC.c:110-118,134-141,194-195,
 *      207-211, 244-250, without its file I/O ----------------------------- */
typedef struct orc_chain_result {
    double E0;               /* energy + wallsEnergy of R0 (SMC.c:48) */
    double meanE;            /* mean(E[0..maxsteps]) with 3NT/2 added (SMC.c:210-211,244) */
    double dE;               /* sqrt(variance(E)) (SMC.c:245) */
    double acceptance_ratio; /* intmean(jj)/N (SMC.c:248) */
    double therm_acceptance; /* intmean(jt)/N (SMC.c:124) */
    double Efinal;           /* last entry of the energy series, without 3NT/2 */
    uint64_t accepted;       /* sum of jj */
    uint64_t gathers;        /* number of histogram calls */
    uint64_t oob;            /* histogram cells out of range */
    double cv;               /* variance(E)/T^2 (SMC.c:250) */
} orc_chain_result;

#define ORC_FLAG_E0_RESTART 1u /* production energy series restarts from E[0] (SMC.c:194) */

/* R: in = R0, out = final positions.  E_series: maxsteps+1 (may be NULL),
 * jj: maxsteps (may be NULL), zhist: Ncz (may be NULL), D/Mu: full
 * histograms (may be NULL), Rbin handled internally. */
int orc_chain(const orc_sys *s, unsigned int seed, double *R, const double *W,
              double T, double A, int eqsteps, int maxsteps, int gather_lapse,
              unsigned int flags, double *E_series, int32_t *jj, uint64_t *zhist,
              uint64_t *D, uint64_t *Mu, orc_chain_result *res);
/* the same, also returning pressure + wallsPressure of every gather (SMC.c:140), in
 * gather order; P_gathers holds maxsteps/gather_lapse entries (may be NULL) */
int orc_chain_p(const orc_sys *s, unsigned int seed, double *R, const double *W,
                double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                unsigned int flags, double *E_series, int32_t *jj, uint64_t *zhist,
                uint64_t *D, uint64_t *Mu, double *P_gathers, orc_chain_result *res);

/* orc_chain_p that also returns the accepted counts of the thermalisation sweeps (jt of
 * SMC.c:51,117; eqsteps entries, may be NULL) */
int orc_chain_jt(const orc_sys *s, unsigned int seed, double *R, const double *W,
                 double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                 unsigned int flags, double *E_series, int32_t *jj, int32_t *jt,
                 uint64_t *zhist, uint64_t *D, uint64_t *Mu, double *P_gathers,
                 orc_chain_result *res);

/* ---- common-neighbour cluster analysis -------------------- */
/* clusterAnalysis, SMC.c:971-1045: LCA[3*idx+{0,1,2}] = num1,num2,num3 of pair entry idx,
 * N(N-1)/2 entries, the reference's (overlapping) pair index kept; cutoff = LCA_cutoff
 * (SMC.h:50).  Stores/reads past the reference's common_nn[8] are skipped and counted. */
void orc_cluster_analysis(int N, const double *r, double L, double cutoff, int32_t *LCA,
                          uint64_t *overflow);
typedef struct orc_lca_counts {
    uint64_t n1;       /* entries with num1 != 0 (the l1 counter, SMC.c:149) */
    uint64_t h2[16];   /* ... of those, by num2 (l2[], SMC.c:152) */
    uint64_t h3[16];   /* ... by num3 (l3[], SMC.c:153) */
    uint64_t analyses; /* clusterAnalysis calls */
    uint64_t overflow;
} orc_lca_counts;
void orc_cluster_counts(int N, const int32_t *LCA, orc_lca_counts *c);
/* orc_chain_p plus the analysis every lca_time-th gather (k % LCA_TIME == 0, SMC.c:143),
 * counts accumulated over the run; lca_time = 0 or lca = NULL disables it */
int orc_chain_lca(const orc_sys *s, unsigned int seed, double *R, const double *W,
                  double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                  unsigned int flags, double *E_series, int32_t *jj, uint64_t *zhist,
                  uint64_t *D, uint64_t *Mu, double *P_gathers, int lca_time,
                  double lca_cutoff, orc_lca_counts *lca, orc_chain_result *res);

/* ---- inputs to the path (not on it) ------------- */
/* build-defined fcc(Na,Nz) start, SURVEY.md 8d (cell order and +a/4 of
 * SMC.c:432-461); N must be 4*Na*Na*Nz */
int orc_fcc_init(int Na, int Nz, double L, double Lz, double *X);
/* the reference's own rule for Na,Nz (SMC.c:416-431); returns number of
 * particles actually placed (N when the lattice is complete) */
int orc_initialize_box_ref(double L, double Lz, int N, double *X);
/* SMC.c:475-501; `uninit` stands for the uninitialised malloc word the
 * reference reads when M*M is odd (0.0 observed) */
void orc_initialize_walls(double x0m, double x0sigma, double ymm, double ymsigma,
                          int M, double uninit, double *W);

/* ---- older single-file variant, SMC_noMPI_noWall.c (BASELINE config 1) --- */
double orc_nw_energy_single(int N, const double *r, double L, int i);            /* :599-619 */
void orc_nw_force(int N, const double *r, double L, int i, double F[3]);         /* :501-529 */
double orc_nw_energy(int N, const double *r, double L);                          /* :573-591 */
double orc_nw_pressure(int N, const double *r, double L);                        /* :664-684 */
void orc_nw_one_particle_moves(int N, orc_rng *g, double *R, double *Rn, double L,
                               double A, double T, int *j, orc_move_trace *trace); /* :266-316 */
int orc_nw_fcc_init(int N, double L, double *X);                                 /* :359-394 */

/* ---- timing leg for bench.py's cpu_baseline ------------------------------ */
/* runs `sweeps` production sweeps of one chain, returns wall seconds spent in
 * the sweep loop only (clock_gettime), positions updated in place */
double orc_time_sweeps(const orc_sys *s, unsigned int seed, double *R, const double *W,
                       double T, double A, int sweeps, uint64_t *accepted);

#ifdef __cplusplus
}
#endif
#endif
