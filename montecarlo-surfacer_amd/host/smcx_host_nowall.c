/*
 * smcx_host_nowall.c -- BASELINE config 1: "SMC_noMPI_noWall path: N=256 LJ particles, 1 chain, CPU reference
 * (plumbing, no GPU)".  The older single-file variant of the reference has its own contract for the same sweep
 * (SURVEY 8a row NW) and is defined by BASELINE as a one-chain HOST computation; this file is that path in the C host
 * library: plain C, one chain, explicit glibc-compatible RNG state.  It is NOT a fallback for anything the GPU
 * engine does (smcx_run, smcx_host_sMC* have no CPU path), and the GPU engine does not serve this contract.
 *
 * What differs from the walls variant, kept as the reference has it (SMC_noMPI_noWall.c):
 *   - cubic box L = cbrt(N/rho), minimum image in x, y AND z (:512-517, :606-610), cutoff r^2 < L^2/4 (:519, :612);
 *   - every neighbour loop starts at l = 1: particle 0 is never anybody's neighbour (:508, :576, :603, :667);
 *   - force() accumulates -dV*d with d = r[l] - r[i], dV = 24/r^8 - 48/r^14 (:512-525);
 *   - the sweep visits 0..N-1 in order, no offset draw, no energy bookkeeping; after Un, Fn are taken from the
 *     unwrapped Rn ALL 3N coordinates of Rn are wrapped (:291-294); uniforms are rand()/RAND_MAX (:710-711, :302);
 *     normals sqrt(-2 sigma ln(1 - x)) with sigma INSIDE the root (:712-715);
 *   - sMC (:196-219): A = 4e-8 fixed (:192), E[k] = energy(R), P[k] = pressure(R) when n % gather_lapse == 0,
 *     before that sweep's moves.
 * Pinned bit for bit on outputs of the real file (tests/golden/ref_smc.json, cases "nw", generated from the reference
 * compiled where it lies): tests/test_cabi_host.py::test_nowall_host_path_equals_the_real_reference.
 */
#include "../../include/smcx_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NW_RAND_MAX 2147483647.0

/* minimum-image separation of particles l and i in all three directions; returns r^2 */
static inline double nw_sep(const double *r, double L, int l, int i, double d[3])
{
    for (int c = 0; c < 3; c++) {
        d[c] = r[3 * l + c] - r[3 * i + c];
        d[c] = d[c] - L * rint(d[c] / L);
    }
    return d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
}

double smcx_host_nowall_energy_single(int N, const double *r, double L, int i)
{
    double V = 0.0, d[3];
    for (int l = 1; l < N; l++) {
        if (l == i) continue;
        const double dr2 = nw_sep(r, L, l, i, d);
        if (dr2 < L * L / 4) {
            const double dr6 = dr2 * dr2 * dr2;
            V += 1.0 / (dr6 * dr6) - 1.0 / dr6;
        }
    }
    return V * 4;
}

void smcx_host_nowall_force(int N, const double *r, double L, int i, double F[3])
{
    double d[3];
    F[0] = F[1] = F[2] = 0.0;
    for (int l = 1; l < N; l++) {
        if (l == i) continue;
        const double dr2 = nw_sep(r, L, l, i, d);
        if (dr2 < L * L / 4) {
            const double dr8 = dr2 * dr2 * dr2 * dr2;
            const double dV = 24.0 / dr8 - 48.0 / (dr8 * dr2 * dr2 * dr2);
            F[0] -= dV * d[0];
            F[1] -= dV * d[1];
            F[2] -= dV * d[2];
        }
    }
}

double smcx_host_nowall_energy(int N, const double *r, double L)
{
    double V = 0.0, d[3];
    for (int l = 1; l < N; l++)
        for (int i = 0; i < l; i++) {
            const double dr2 = nw_sep(r, L, l, i, d);
            if (dr2 < L * L / 4)
                V += 1.0 / (dr2 * dr2 * dr2 * dr2 * dr2 * dr2) - 1.0 / (dr2 * dr2 * dr2);
        }
    return V * 4;
}

double smcx_host_nowall_pressure(int N, const double *r, double L)
{
    double P = 0.0, d[3];
    for (int l = 1; l < N; l++)
        for (int i = 0; i < l; i++) {
            const double dr2 = nw_sep(r, L, l, i, d);
            if (dr2 < L * L / 4)
                P += 24.0 / (dr2 * dr2 * dr2) - 48.0 / (dr2 * dr2 * dr2 * dr2 * dr2 * dr2);
        }
    return -P / (3 * L * L * L);
}

/* initializeBox (:359-394): cubic fcc of Na = (int)cbrt(N/4) cells, +a/4, wrapped; returns the particles placed */
int smcx_host_nowall_fcc(int N, double L, double *X)
{
    const int Na = (int)(cbrt(N / 4));
    const double a = L / Na;
    static const double basis[4][3] = {{0, 0, 0}, {.5, .5, 0}, {.5, 0, .5}, {0, .5, .5}};
    int placed = 0;
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Na; j++)
            for (int k = 0; k < Na; k++) {
                const int n = i * Na * Na + j * Na + k;
                if (4 * (n + 1) > N) continue;
                for (int b = 0; b < 4; b++) {
                    X[n * 12 + 3 * b + 0] = a * i + (basis[b][0] != 0 ? a / 2 : 0.0);
                    X[n * 12 + 3 * b + 1] = a * j + (basis[b][1] != 0 ? a / 2 : 0.0);
                    X[n * 12 + 3 * b + 2] = a * k + (basis[b][2] != 0 ? a / 2 : 0.0);
                }
                placed += 4;
            }
    for (int n = 0; n < 3 * N; n++) X[n] += a / 4;
    for (int n = 0; n < 3 * N; n++) X[n] = X[n] - L * rint(X[n] / L);
    return placed;
}

/* vecBoxMuller of this variant (:707-717) */
static void nw_box_muller(smcx_host_rng *g, double sigma, size_t length, double *A)
{
    for (size_t i = 0; i < length / 2; i++) {
        const double x1 = (double)smcx_host_rand(g) / NW_RAND_MAX;
        const double x2 = (double)smcx_host_rand(g) / NW_RAND_MAX;
        A[2 * i] = sqrt(-2 * sigma * log(1 - x1)) * cos(2 * M_PI * x2);
        A[2 * i + 1] = sqrt(-2 * sigma * log(1 - x2)) * sin(2 * M_PI * x1);
    }
}

/* oneParticleMoves (:266-316): one trial move per particle in index order; *j += accepted moves */
int smcx_host_nowall_sweep(int N, smcx_host_rng *g, double *R, double *Rn, double L, double A, double T, int *j)
{
    double *displ = (double *)malloc(3 * (size_t)N * sizeof(double));
    if (!displ) return SMCX_ERR_NOMEM;
    nw_box_muller(g, sqrt(2 * A), 3 * (size_t)N, displ);
    memcpy(Rn, R, 3 * (size_t)N * sizeof(double));
    for (int n = 0; n < N; n++) {
        double Fm[3], Fn[3], dl[3];
        const double Um = smcx_host_nowall_energy_single(N, R, L, n);
        smcx_host_nowall_force(N, R, L, n, Fm);
        for (int c = 0; c < 3; c++) {
            dl[c] = Fm[c] * (A / T) + displ[3 * n + c];
            Rn[3 * n + c] = R[3 * n + c] + dl[c];
        }
        const double Un = smcx_host_nowall_energy_single(N, Rn, L, n);
        smcx_host_nowall_force(N, Rn, L, n, Fn);
        for (int q = 0; q < 3 * N; q++) Rn[q] = Rn[q] - L * rint(Rn[q] / L); /* shiftSystem(Rn, L), :294 */
        const double gx = Fn[0] - Fm[0], gy = Fn[1] - Fm[1], gz = Fn[2] - Fm[2];
        const double deltaW = (gx * gx + gy * gy + gz * gz + 2 * (gx * Fm[0] + gy * Fm[1] + gz * Fm[2])) * A / (4 * T);
        const double ap = exp(-(Un - Um + (dl[0] * (Fn[0] + Fm[0]) + dl[1] * (Fn[1] + Fm[1]) + dl[2] * (Fn[2] + Fm[2])) / 2 + deltaW) / T);
        if ((double)smcx_host_rand(g) / NW_RAND_MAX < ap) {
            for (int c = 0; c < 3; c++) R[3 * n + c] = Rn[3 * n + c];
            *j += 1;
        } else {
            for (int c = 0; c < 3; c++) Rn[3 * n + c] = R[3 * n + c];
        }
    }
    free(displ);
    return SMCX_OK;
}

/* the loop of this variant's sMC (:196-219) for one chain seeded srand(seed): E[k], P[k] (k = n / gather_lapse) before
 * the moves of sweep n when n % gather_lapse == 0; jj[n] = accepted moves of sweep n; R: in = start, out = final */
double smcx_host_nowall_box(int N, double rho) { return cbrt(N / rho); } /* :165 */

int smcx_host_nowall_sMC(int N, double L, double T, double A, unsigned int seed, int maxsteps, int gather_lapse,
                         double *R, double *E, double *P, int *jj)
{
    if (N < 2 || !(L > 0) || !(T > 0) || !(A > 0) || maxsteps < 0 || gather_lapse < 1 || !R) return SMCX_ERR_PARAM;
    double *Rn = (double *)calloc(3 * (size_t)N, sizeof(double));
    if (!Rn) return SMCX_ERR_NOMEM;
    smcx_host_rng g;
    smcx_host_srand(&g, seed);
    int rc = SMCX_OK;
    for (int n = 0; n < maxsteps && rc == SMCX_OK; n++) {
        if (n % gather_lapse == 0) {
            const int k = n / gather_lapse;
            if (E) E[k] = smcx_host_nowall_energy(N, R, L);
            if (P) P[k] = smcx_host_nowall_pressure(N, R, L);
        }
        int j = 0;
        rc = smcx_host_nowall_sweep(N, &g, R, Rn, L, A, T, &j);
        if (jj) jj[n] = j;
    }
    free(Rn);
    return rc;
}
