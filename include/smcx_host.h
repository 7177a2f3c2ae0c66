/*
 * smcx_host.h -- the C host side above the C ABI of smcx.h.
 *
 * Mirrors, for many replica chains at once, what the reference's host code
 * does around the hot path: system preparation (initializeBox SMC.c:413-465,
 * initializeWalls SMC.c:475-501), the simulation driver sMC (SMC.c:21-267) and
 * the run set-up of main (main.c:35-51, 74-87, 98-122).  Built as
 * libsmcx_host.so (plain C, links libsmcx.so) and as the `smcx_main` program.
 */
#ifndef SMCX_HOST_H
#define SMCX_HOST_H

#include "smcx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* libc-compatible rand()/srand() with explicit state (glibc TYPE_3), used by
 * the initialisers below exactly where the reference calls srand(42)/rand() */
typedef struct smcx_host_rng {
    uint32_t s[31];
    int f, r;
} smcx_host_rng;
void smcx_host_srand(smcx_host_rng *g, unsigned int seed);
int smcx_host_rand(smcx_host_rng *g);
/* vecBoxMuller, matematicose.c:183-193 (x1/x2 swap and odd tail kept) */
void smcx_host_vec_box_muller(smcx_host_rng *g, double sigma, size_t length, double *A);

/* fcc(Na,Nz) slab start: N = 4*Na*Na*Nz particles, lattice constant L/Na, cell order
 * i,j,k with k fastest, +a/4 on every coordinate, wrapped into the box with the z
 * period 0.95*Lz (SMC.c:432-461).  Returns N or -1. */
int smcx_host_fcc_init(int Na, int Nz, double L, double Lz, double *X);
/* the reference's own choice of Na,Nz from N (SMC.c:416-431); returns the number of
 * particles it places (< N means the reference leaves the rest at the origin) */
int smcx_host_initialize_box(double L, double Lz, int N, double *X);
/* W[2*M*M] from srand(42) and two vecBoxMuller draws (SMC.c:475-501).  For odd M*M
 * the reference reads one uninitialised word; `uninit` supplies it (0.0 observed). */
void smcx_host_initialize_walls(double x0m, double x0sigma, double ymm, double ymsigma, int M,
                                double uninit, double *W);
/* box table of main.c:35-44 */
void smcx_host_box_for_N(int N, double *L, double *Lz);

/* ensemble results of one smcx_host_sMC call (struct Sim, SMC.h:76-88, per replica
 * and averaged over replicas) */
typedef struct smcx_sim {
    int nrep, N, Ncz;
    double E, dE;              /* ensemble mean of the replicas' mean energy / of their dE */
    double acceptance_ratio;   /* ensemble mean */
    double therm_acceptance;
    double *rep_E;             /* [nrep] owned by the struct: smcx_host_sim_free */
    double *rep_dE;            /* [nrep] */
    double *rep_acceptance;    /* [nrep] */
    double *zprofile;          /* [Ncz] particles per z cell per gather, ensemble mean */
    double *Rfinal;            /* [nrep][3N] */
    /* cluster analysis (SMCX_FLAG_CLUSTERS): bonded pair entries, and those with num2 == v /
     * num3 == v, per analysis and replica (ensemble mean).  This is what the weight at
     * SMC.c:149-153 is after; the reference's own `1 / (gather_steps/LCA_TIME)` is an int
     * division (0 unless the ratio is 1) added into uninitialised l2[7], l3[7]. */
    double l1, l2[16], l3[16];
    int lca_analyses;
    /* SMCX_FLAG_PRESSURE: mean and deviation of the total pressure P[k] + rho*T over the reference's
     * gather_steps entries (P[0] never written, the last gather out of bounds: SMC.c:138-140, 207-208,
     * 246-247), ensemble mean.  SMCX_FLAG_SERIES: tau = sum(acf) and cv = variance(E)/T^2
     * (SMC.c:234-235, 249-250), ensemble mean; the ACF itself: smcx_acf. */
    double P, dP, tau, cv;
    double kernel_ms;          /* device time of the sweep kernels */
    double pair_evals_per_s;   /* nrep*maxsteps*2N(N-1) / kernel time */
} smcx_sim;

/* sMC (SMC.h:92) for nrep replica chains sharing R0 and W, seeds base_seed+r:
 * thermalisation (eqsteps at 2A), production (maxsteps), results.  p supplies N, M,
 * box, T, A, flags, seeds; returns an smcx status. */
int smcx_host_sMC(const smcx_params *p, const double *W, const double *R0, int maxsteps,
                  int gather_lapse, int eqsteps, smcx_sim *out);
void smcx_host_sim_free(smcx_sim *s);

/* The same for the GPUs of one node -- what replaces the reference's one-chain-per-MPI-rank fan-out (SMC.c:40, 43,
 * 66-95: ranks share R0 and W, differ in their seed and never communicate).  p->nrep replica chains are dealt in
 * contiguous blocks to `ndev` devices (devices[0..ndev-1], or 0..ndev-1 when NULL), one handle and one host thread per
 * device; seeds follow the global replica index, so the results do not depend on ndev.  No exchange while sampling;
 * the final observable gather -- (8 + Ncz) doubles per replica -- is ONE RCCL all-gather over xGMI (ncclCommInitAll,
 * one communicator rank per device, ncclAllGather inside a group call), also with ndev == 1 (a one-rank
 * communicator); SMCX_HOST_GATHER=host concatenates through host memory instead.  Final positions travel over PCIe
 * (smcx_download_positions per device).  Returns an smcx status; SMCX_ERR_RCCL if the collective failed
 * (smcx_host_multi_error() has the text).  `out` as for smcx_host_sMC, over all replicas in global order;
 * kernel_ms = the slowest device's. */
int smcx_host_sMC_multi(const smcx_params *p, int ndev, const int *devices, const double *W, const double *R0,
                        int maxsteps, int gather_lapse, int eqsteps, smcx_sim *out);
const char *smcx_host_multi_error(void);

/* ---- BASELINE config 1: the older single-file variant SMC_noMPI_noWall.c, "N=256 LJ particles, 1 chain, CPU reference
 * (plumbing, no GPU)".  One chain on the HOST, by definition of that configuration -- not a fallback of the GPU engine,
 * which serves the walls variant only.  Cubic box L = cbrt(N/rho) periodic in x, y and z, cutoff L/2, neighbour loops
 * from particle 1, fixed visiting order, its own Box-Muller; every quirk of the file kept (host/smcx_host_nowall.c).
 * energySingle :599-619, force :501-529, energy :573-591, pressure :664-684, initializeBox :359-394,
 * oneParticleMoves :266-316, the loop of sMC :196-219. */
double smcx_host_nowall_energy_single(int N, const double *r, double L, int i);
void smcx_host_nowall_force(int N, const double *r, double L, int i, double F[3]);
double smcx_host_nowall_energy(int N, const double *r, double L);
double smcx_host_nowall_pressure(int N, const double *r, double L);
int smcx_host_nowall_fcc(int N, double L, double *X);
/* R, Rn [3N] updated in place, *j += accepted moves; returns an smcx status */
int smcx_host_nowall_sweep(int N, smcx_host_rng *g, double *R, double *Rn, double L, double A, double T, int *j);
/* the box of a run: L = cbrt(N / rho) (:165) */
double smcx_host_nowall_box(int N, double rho);
/* one chain seeded srand(seed) in the box L: E[k], P[k] = energy / pressure BEFORE sweep n when n % gather_lapse == 0 (k = n /
 * gather_lapse; ceil(maxsteps / gather_lapse) entries, either may be NULL), jj[n] accepted moves of sweep n (may be NULL);
 * R [3N]: in = start, out = final positions.  The reference's main uses rho = 0.1, T = 0.4 (:80-81), sMC A = 4e-8 (:192). */
int smcx_host_nowall_sMC(int N, double L, double T, double A, unsigned int seed, int maxsteps, int gather_lapse,
                         double *R, double *E, double *P, int *jj);

/* The reference's result files for the last smcx_run of `h`, one set per replica with
 * the replica's global index as the `_rank` suffix (SMC.c:66-95 names them per MPI rank):
 *   data_N%d_M%d_r%0.4f_T%0.2f_rank%d.csv    "E, P, jj"  one row per gather   (SMC.c:75-77, 214-215)
 *   local_N..._rank%d.csv                     "nx, ny, nz, n, mu" all cells     (SMC.c:80-82, 218-225)
 *   last_state_N%d_M%d_r%0.4f_T%0.2f[_rank%d].csv  3N positions, %0.12f,        (main.c:162-170)
 * Rows follow the reference's indexing: E[k*gather_lapse] + 3NT/2, P[k] + rho*T with P[0] = 0
 * and P[k] the k-th gather (SMC.c:138-140), jj[k] (sic).  The handle must have been created
 * with SMCX_FLAG_SERIES | SMCX_FLAG_FULL_HIST (| SMCX_FLAG_PRESSURE, else the P column holds
 * rho*T only).  Returns an smcx status; `dir` must exist. */
int smcx_host_write_csv(smcx_handle *h, const smcx_params *p, int maxsteps, int gather_lapse,
                        const char *dir);
/* reads a last_state file written by the reference or by smcx_host_write_csv (main.c:98-108);
 * returns the number of coordinates read */
int smcx_host_read_last_state(const char *path, int N, double *R0);

#ifdef __cplusplus
}
#endif
#endif
