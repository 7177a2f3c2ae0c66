/*
 * smcx_main.c -- command-line driver in the spirit of the reference's main.c:
 *   smcx_main [--gpus G] eqsteps maxsteps numdata T [N [nrep [Na Nz]]]
 * (main.c:13-19 takes the first four; N is a macro there, SMC.h:29).  Prepares
 * the walls and the lattice (main.c:74-113), runs nrep replica chains on GPU 0 --
 * or, with --gpus G, dealt over GPUs 0..G-1 with the observables gathered by RCCL
 * (smcx_host_sMC_multi: the reference's MPI ranks, SMC.c:40, 66-95) -- and prints
 * the ensemble results (main.c:126-131).
 *   smcx_main --nowall maxsteps gather_lapse [N [seed]]
 * is BASELINE config 1: the older variant SMC_noMPI_noWall.c (its main: rho = 0.1, T = 0.4, :80-81; sMC: A = 4e-8,
 * :192), one chain on the host CPU.
 */
#include "../../include/smcx_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    int gpus = 0;
    if (argc > 3 && strcmp(argv[1], "--nowall") == 0) { /* SMC_noMPI_noWall.c main :74-143 */
        const int maxsteps = (int)strtol(argv[2], NULL, 10), gl = (int)strtol(argv[3], NULL, 10);
        const int N = argc > 4 ? (int)strtol(argv[4], NULL, 10) : 256;
        const unsigned seed = argc > 5 ? (unsigned)strtoul(argv[5], NULL, 10) : 12345u;
        if (maxsteps < 1 || gl < 1 || N < 8) { fprintf(stderr, "usage: smcx_main --nowall maxsteps gather_lapse [N [seed]]\n"); return 2; }
        const double rho = 0.1, T = 0.4, A = 4e-8, L = smcx_host_nowall_box(N, rho);
        double *R = (double *)calloc(3 * (size_t)N, sizeof(double));
        const int ng = (maxsteps + gl - 1) / gl;
        double *E = (double *)calloc(ng, sizeof(double)), *P = (double *)calloc(ng, sizeof(double));
        int *jj = (int *)calloc(maxsteps, sizeof(int));
        if (!R || !E || !P || !jj || smcx_host_nowall_fcc(N, L, R) != N) { fprintf(stderr, "Can't make a cubic FCC crystal with this N\n"); return 2; }
        int rc = smcx_host_nowall_sMC(N, L, T, A, seed, maxsteps, gl, R, E, P, jj);
        if (rc != SMCX_OK) { fprintf(stderr, "smcx_host_nowall_sMC: %s\n", smcx_strerror(rc)); return 1; }
        double mE = 0, mP = 0, acc = 0;
        for (int k = 0; k < ng; k++) { mE += E[k] / ng; mP += P[k] / ng; }
        for (int n = 0; n < maxsteps; n++) acc += (double)jj[n] / maxsteps;
        printf("noWall variant, host CPU: N=%d L=%0.6f T=%0.2f rho=%0.2f A=%g, %d sweeps\n", N, L, T, rho, A, maxsteps);
        printf("E[0] = %0.12f, mean E = %0.12f, mean P = %0.12f, acceptance ratio %f\n", E[0], mE, mP, acc / N);
        free(R); free(E); free(P); free(jj);
        return 0;
    }
    if (argc > 2 && strcmp(argv[1], "--gpus") == 0) {
        gpus = (int)strtol(argv[2], NULL, 10);
        if (gpus < 1) { fprintf(stderr, "--gpus needs a positive count\n"); return 2; }
        argv += 2; argc -= 2;
    }
    if (argc < 5) {
        fprintf(stderr, "usage: smcx_main [--gpus G] eqsteps maxsteps numdata T [N [nrep [Na Nz]]]\n");
        return 2;
    }
    const int eqsteps = (int)strtol(argv[1], NULL, 10);
    const int maxsteps = (int)strtol(argv[2], NULL, 10);
    const int numdata = (int)strtol(argv[3], NULL, 10);
    const double T = strtod(argv[4], NULL);
    const int N = argc > 5 ? (int)strtol(argv[5], NULL, 10) : 108;
    const int nrep = argc > 6 ? (int)strtol(argv[6], NULL, 10) : 1;
    if (numdata < 1 || maxsteps < numdata) {
        fprintf(stderr, "need 1 <= numdata <= maxsteps\n");
        return 2;
    }
    const int gather_lapse = maxsteps / numdata; /* main.c:32 */

    smcx_params p;
    smcx_default_params(&p, N, nrep);
    smcx_host_box_for_N(N, &p.L, &p.Lz);
    p.T = T;
    p.A = 1.0 * T; /* gamma = 1, main.c:48-51 */
    /* sMC always evaluates the pressure, the energy autocorrelation and the cluster analysis (SMC.c:140-155, 234) */
    p.flags |= SMCX_FLAG_CLUSTERS | SMCX_FLAG_PRESSURE | SMCX_FLAG_SERIES;

    double W[2 * 3 * 3];
    smcx_host_initialize_walls(1.6, 0.0, 3.0, 0.5, p.M, 0.0, W); /* main.c:74-87 */

    double *R0 = (double *)calloc(3 * (size_t)N, sizeof(double));
    int placed;
    if (argc > 8) {
        placed = smcx_host_fcc_init((int)strtol(argv[7], NULL, 10), (int)strtol(argv[8], NULL, 10), p.L, p.Lz, R0);
        if (placed != N) { fprintf(stderr, "4*Na*Na*Nz must equal N\n"); return 2; }
    } else {
        placed = smcx_host_initialize_box(p.L, p.Lz, N, R0);
        if (placed != N) {
            fprintf(stderr, "Can't make the reference's crystal with N=%d (%d particles placed); "
                            "give Na Nz explicitly\n", N, placed);
            return 2;
        }
    }
    printf("Starting %d replica(s) of %d particles in %0.1fx%0.1fx%0.1f box, T=%0.2f, A=%0.3f, "
           "%d+%d sweeps...\n", nrep, N, p.L, p.L, p.Lz, T, p.A, eqsteps, maxsteps);

    smcx_sim sim;
    int rc = gpus ? smcx_host_sMC_multi(&p, gpus, NULL, W, R0, maxsteps, gather_lapse, eqsteps, &sim)
                  : smcx_host_sMC(&p, W, R0, maxsteps, gather_lapse, eqsteps, &sim);
    if (gpus && rc == SMCX_OK) printf("(%d GPU(s), observables gathered by RCCL)\n", gpus);
    if (rc != SMCX_OK) {
        fprintf(stderr, "%s: %s (%s)\n", gpus ? "smcx_host_sMC_multi" : "smcx_host_sMC", smcx_strerror(rc),
                gpus && smcx_host_multi_error()[0] ? smcx_host_multi_error() : smcx_last_error_string(NULL));
        free(R0);
        return 1;
    }
    printf("\n###  Final results  ###");
    printf("\nMean energy: %f +- %f", sim.E, sim.dE);
    printf("\nMean pressure: %f +- %f", sim.P, sim.dP);                       /* main.c:128-131 style */
    printf("\nAverage autocorrelation time: %f, cv: %f", sim.tau, sim.cv);
    printf("\nAverage acceptance ratio: %f (thermalisation %f)", sim.acceptance_ratio, sim.therm_acceptance);
    printf("\nDevice time %0.1f ms, %0.3e pair-evals/s", sim.kernel_ms, sim.pair_evals_per_s);
    if (sim.lca_analyses > 0) { /* SMC.c:227-231 */
        printf("\nl1[1] = %0.9f (%d analyses)", sim.l1, sim.lca_analyses);
        printf("\nl2[0..5] =");
        for (int v = 0; v < 6; v++) printf(" %0.9f", sim.l2[v]);
        printf("\nl3[0..5] =");
        for (int v = 0; v < 6; v++) printf(" %0.9f", sim.l3[v]);
    }
    printf("\nz profile (particles per cell per gather):");
    for (int k = 0; k < sim.Ncz; k++) printf(" %0.3f", sim.zprofile[k]);
    printf("\n");
    smcx_host_sim_free(&sim);
    free(R0);
    return 0;
}
