/*
 * smcx_host.c -- C host side of the engine: system preparation and the sMC
 * driver, calling the C ABI of libsmcx.so (HIP kernels).  No physics of the hot
 * path is evaluated here; without a GPU smcx_host_sMC returns the ABI's error.
 */
#include "../../include/smcx_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- libc-compatible generator (what srand/rand are on glibc) ------------------ */
void smcx_host_srand(smcx_host_rng *g, unsigned int seed)
{
    int32_t word;
    if (seed == 0) seed = 1;
    g->s[0] = seed;
    word = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        word = (int32_t)(16807 * lo - 2836 * hi);
        if (word < 0) word += 2147483647;
        g->s[i] = (uint32_t)word;
    }
    g->f = 3;
    g->r = 0;
    for (int i = 0; i < 310; i++) (void)smcx_host_rand(g);
}

int smcx_host_rand(smcx_host_rng *g)
{
    uint32_t v = (g->s[g->f] += g->s[g->r]);
    if (++g->f == 31) g->f = 0;
    if (++g->r == 31) g->r = 0;
    return (int)(v >> 1);
}

void smcx_host_vec_box_muller(smcx_host_rng *g, double sigma, size_t length, double *A)
{
    const double scale = 1.0 / 2147483648.0; /* rand()/(RAND_MAX+1.0), exact */
    for (size_t k = 0; k + 1 < length; k += 2) {
        const double x1 = smcx_host_rand(g) * scale;
        const double x2 = smcx_host_rand(g) * scale;
        A[k] = sigma * sqrt(-2 * log(1 - x1)) * cos(2 * M_PI * x2);
        A[k + 1] = sigma * sqrt(-2 * log(1 - x2)) * sin(2 * M_PI * x1); /* x1,x2 swapped: as the reference */
    }
}

/* ---- lattices ------------------------------------------------------------------- */
static void place_fcc(int Na, int Nz, double a, int limit, double *X)
{
    const double h = a / 2;
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Na; j++)
            for (int k = 0; k < Nz; k++) {
                const int c = (i * Na + j) * Nz + k; /* k fastest */
                const double x = a * i, y = a * j, z = a * k;
                const double site[4][3] = {{x, y, z}, {x + h, y + h, z}, {x + h, y, z + h}, {x, y + h, z + h}};
                for (int b = 0; b < 4; b++) {
                    const int p = 4 * c + b;
                    if (p < limit) memcpy(X + 3 * p, site[b], 3 * sizeof(double));
                }
            }
}

static void finish_lattice(double *X, int N, double a, double L, double Lz)
{
    const double Lzs = Lz - Lz / 20.0; /* SMC.c:461 */
    for (int p = 0; p < N; p++) {
        double *q = X + 3 * p;
        for (int c = 0; c < 3; c++) q[c] += a / 4; /* off the cell edges, SMC.c:455-459 */
        q[0] -= L * rint(q[0] / L);
        q[1] -= L * rint(q[1] / L);
        q[2] -= Lzs * rint(q[2] / Lzs);
    }
}

int smcx_host_fcc_init(int Na, int Nz, double L, double Lz, double *X)
{
    if (Na < 1 || Nz < 1 || !X) return -1;
    const int N = 4 * Na * Na * Nz;
    const double a = L / Na;
    place_fcc(Na, Nz, a, N, X);
    finish_lattice(X, N, a, L, Lz);
    return N;
}

int smcx_host_initialize_box(double L, double Lz, int N, double *X)
{
    if (N < 4 || !X) return -1;
    const int cells = N / 4;
    int Na = 1;
    while ((Na + 1) * (Na + 1) * (Na + 1) <= cells) Na++; /* largest cube not above N/4 */
    const int Nz = cells / (Na * Na);
    const double a = L / Na;
    memset(X, 0, 3 * (size_t)N * sizeof(double));
    place_fcc(Na, Nz, a, N, X);
    finish_lattice(X, N, a, L, Lz);
    const int placed = 4 * Na * Na * Nz;
    return placed < N ? placed : N;
}

void smcx_host_initialize_walls(double x0m, double x0sigma, double ymm, double ymsigma, int M,
                                double uninit, double *W)
{
    const size_t n = (size_t)M * M;
    double *X0 = (double *)malloc(2 * n * sizeof(double));
    double *YM = X0 + n;
    smcx_host_rng g;
    smcx_host_srand(&g, 42); /* SMC.c:477 */
    for (size_t m = 0; m < 2 * n; m++) X0[m] = uninit;
    smcx_host_vec_box_muller(&g, x0sigma, n, X0);
    smcx_host_vec_box_muller(&g, ymsigma, n, YM);
    for (size_t m = 0; m < n; m++) {
        const double x0 = X0[m] + x0m, depth = YM[m] + ymm;
        W[2 * m] = pow(x0, 12.0) * depth;
        W[2 * m + 1] = pow(x0, 6.) * depth;
    }
    free(X0);
}

void smcx_host_box_for_N(int N, double *L, double *Lz)
{
    if (N == 32) { *L = 20; *Lz = 120; }
    else if (N < 150) { *L = 33; *Lz = 200; }
    else { *L = 33; *Lz = 240; }
}

/* ---- sMC ---------------------------------------------------------------------- */
void smcx_host_sim_free(smcx_sim *s)
{
    if (!s) return;
    free(s->rep_E); free(s->rep_dE); free(s->rep_acceptance); free(s->zprofile); free(s->Rfinal);
    memset(s, 0, sizeof(*s));
}

int smcx_host_sMC(const smcx_params *p, const double *W, const double *R0, int maxsteps,
                  int gather_lapse, int eqsteps, smcx_sim *out)
{
    if (!p || !R0 || !out) return SMCX_ERR_PARAM;
    memset(out, 0, sizeof(*out));
    smcx_handle *h = NULL;
    int rc = smcx_create(p, &h);
    if (rc != SMCX_OK) return rc;
    const int nrep = p->nrep, N = p->N, Ncz = p->Ncz;
    uint64_t *zh = NULL, *gath = NULL;
    double *therm = NULL;
    do {
        rc = smcx_upload(h, R0, 0, W, NULL);
        if (rc != SMCX_OK) break;
        rc = smcx_run(h, eqsteps, maxsteps, gather_lapse);
        if (rc != SMCX_OK) break;
        out->nrep = nrep; out->N = N; out->Ncz = Ncz;
        out->rep_E = (double *)calloc(nrep, sizeof(double));
        out->rep_dE = (double *)calloc(nrep, sizeof(double));
        out->rep_acceptance = (double *)calloc(nrep, sizeof(double));
        out->zprofile = (double *)calloc(Ncz, sizeof(double));
        out->Rfinal = (double *)malloc((size_t)nrep * 3 * N * sizeof(double));
        zh = (uint64_t *)calloc((size_t)nrep * Ncz, sizeof(uint64_t));
        gath = (uint64_t *)calloc(nrep, sizeof(uint64_t));
        therm = (double *)calloc(nrep, sizeof(double));
        if (!out->rep_E || !out->rep_dE || !out->rep_acceptance || !out->zprofile || !out->Rfinal ||
            !zh || !gath || !therm) { rc = SMCX_ERR_NOMEM; break; }
        rc = smcx_observables(h, out->rep_acceptance, out->rep_E, out->rep_dE, zh, NULL, NULL);
        if (rc != SMCX_OK) break;
        rc = smcx_hist_info(h, gath, NULL);
        if (rc != SMCX_OK) break;
        rc = smcx_therm_acceptance(h, therm);
        if (rc != SMCX_OK) break;
        rc = smcx_download_positions(h, out->Rfinal);
        if (rc != SMCX_OK) break;
        double gsum = 0;
        for (int r = 0; r < nrep; r++) {
            out->E += out->rep_E[r] / nrep;
            out->dE += out->rep_dE[r] / nrep;
            out->acceptance_ratio += out->rep_acceptance[r] / nrep;
            out->therm_acceptance += therm[r] / nrep;
            gsum += (double)gath[r];
            for (int k = 0; k < Ncz; k++) out->zprofile[k] += (double)zh[(size_t)r * Ncz + k];
        }
        if (gsum > 0)
            for (int k = 0; k < Ncz; k++) out->zprofile[k] /= gsum;
        if (p->flags & SMCX_FLAG_CLUSTERS) {
            uint64_t *c = (uint64_t *)calloc((size_t)nrep * 33, sizeof(uint64_t));
            if (!c) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_cluster_counts(h, c, c + nrep, c + (size_t)nrep * 17, NULL, &out->lca_analyses);
            if (rc == SMCX_OK && out->lca_analyses > 0) {
                const double w = 1.0 / ((double)nrep * out->lca_analyses);
                for (int r = 0; r < nrep; r++) {
                    out->l1 += w * (double)c[r];
                    for (int v = 0; v < 16; v++) {
                        out->l2[v] += w * (double)c[nrep + (size_t)r * 16 + v];
                        out->l3[v] += w * (double)c[(size_t)nrep * 17 + (size_t)r * 16 + v];
                    }
                }
            }
            free(c);
            if (rc != SMCX_OK) break;
        }
        if (p->flags & SMCX_FLAG_PRESSURE) { /* SMC.c:207-208, 246-247 with the reference's indexing */
            const int gather_steps = maxsteps / gather_lapse;
            int ng = 0;
            double *Ps = (double *)calloc((size_t)nrep * (gather_steps + 1), sizeof(double));
            if (!Ps) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_pressure_series(h, Ps, &ng);
            if (rc == SMCX_OK && gather_steps > 0) {
                const double rho = N / (p->L * p->L * p->Lz);
                for (int r = 0; r < nrep; r++) {
                    double sum = 0, sum2 = 0;
                    for (int k = 0; k < gather_steps; k++) { /* P[0] = 0, P[k] = k-th gather */
                        const double v = (k >= 1 && k - 1 < ng ? Ps[(size_t)r * ng + (k - 1)] : 0.0) + rho * p->T;
                        sum += v; sum2 += v * v;
                    }
                    const double mean = sum / gather_steps, var = sum2 / gather_steps - mean * mean;
                    out->P += mean / nrep;
                    out->dP += sqrt(var > 0 ? var : 0) / nrep;
                }
            }
            free(Ps);
            if (rc != SMCX_OK) break;
        }
        if (p->flags & SMCX_FLAG_SERIES) { /* SMC.c:234-235, 249-250 */
            double *tc = (double *)calloc(2 * (size_t)nrep, sizeof(double));
            int keff = 0;
            if (!tc) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_acf(h, 2500000 /* KMAX, SMC.h:61 */, NULL, &keff, tc, tc + nrep);
            for (int r = 0; rc == SMCX_OK && r < nrep; r++) { out->tau += tc[r] / nrep; out->cv += tc[nrep + r] / nrep; }
            free(tc);
            if (rc != SMCX_OK) break;
        }
        int launches = 0;
        smcx_last_kernel_ms(h, &out->kernel_ms, &launches);
        if (out->kernel_ms > 0)
            out->pair_evals_per_s = (double)nrep * (eqsteps + maxsteps) * 2.0 * N * (N - 1.0) /
                                    (out->kernel_ms * 1e-3);
    } while (0);
    free(zh); free(gath); free(therm);
    smcx_destroy(h);
    if (rc != SMCX_OK) smcx_host_sim_free(out);
    return rc;
}

/* ---- the reference's CSV outputs ------------------------------------------------- */
int smcx_host_write_csv(smcx_handle *h, const smcx_params *p, int maxsteps, int gather_lapse,
                        const char *dir)
{
    if (!h || !p || !dir || maxsteps < 0 || gather_lapse < 1) return SMCX_ERR_PARAM;
    const int nrep = p->nrep, N = p->N, Ncx = p->Ncx, Ncz = p->Ncz;
    const size_t Nc = (size_t)Ncx * Ncx * Ncz;
    const int gather_steps = maxsteps / gather_lapse;        /* SMC.c:27 */
    const double rho = N / (p->L * p->L * p->Lz);             /* SMC.c:24 */
    double *E = (double *)malloc((size_t)nrep * (maxsteps + 1) * sizeof(double));
    int32_t *jj = (int32_t *)malloc((size_t)nrep * (maxsteps > 0 ? maxsteps : 1) * sizeof(int32_t));
    uint64_t *D = (uint64_t *)malloc((size_t)nrep * Nc * sizeof(uint64_t));
    uint64_t *Mu = (uint64_t *)malloc((size_t)nrep * Nc * sizeof(uint64_t));
    double *R = (double *)malloc((size_t)nrep * 3 * N * sizeof(double));
    double *P = NULL;
    int ng = 0, rc = SMCX_OK;
    char path[1024];
    do {
        if (!E || !jj || !D || !Mu || !R) { rc = SMCX_ERR_NOMEM; break; }
        if ((rc = smcx_series(h, E, jj)) != SMCX_OK) break;
        if ((rc = smcx_density(h, D, Mu)) != SMCX_OK) break;
        if ((rc = smcx_download_positions(h, R)) != SMCX_OK) break;
        if (p->flags & SMCX_FLAG_PRESSURE) {
            if ((rc = smcx_pressure_series(h, NULL, &ng)) != SMCX_OK) break;
            P = (double *)calloc((size_t)nrep * (ng > 0 ? ng : 1), sizeof(double));
            if (!P) { rc = SMCX_ERR_NOMEM; break; }
            if ((rc = smcx_pressure_series(h, P, &ng)) != SMCX_OK) break;
        }
        for (int r = 0; r < nrep && rc == SMCX_OK; r++) {
            const int rank = (int)p->first_replica + r;
            snprintf(path, sizeof path, "%s/data_N%d_M%d_r%0.4f_T%0.2f_rank%d.csv", dir, N, p->M, rho, p->T, rank);
            FILE *f = fopen(path, "w");
            if (!f) { rc = SMCX_ERR_STATE; break; }
            fprintf(f, "E, P, jj\n");
            for (int k = 0; k < gather_steps; k++) { /* SMC.c:207-215 */
                const double e = E[(size_t)r * (maxsteps + 1) + (size_t)k * gather_lapse] + 3 * N * p->T / 2;
                const double pk = ((k >= 1 && P && k - 1 < ng) ? P[(size_t)r * ng + (k - 1)] : 0.0) + rho * p->T;
                fprintf(f, "%0.9lf, %0.9lf, %d\n", e, pk, jj[(size_t)r * maxsteps + k]);
            }
            fclose(f);
            snprintf(path, sizeof path, "%s/local_N%d_M%d_r%0.4f_T%0.2f_rank%d.csv", dir, N, p->M, rho, p->T, rank);
            f = fopen(path, "w");
            if (!f) { rc = SMCX_ERR_STATE; break; }
            fprintf(f, "nx, ny, nz, n, mu\n");
            for (int i = 0; i < Ncx; i++)
                for (int j = 0; j < Ncx; j++)
                    for (int k = 0; k < Ncz; k++) { /* SMC.c:218-225 */
                        const size_t v = (size_t)i * Ncx * Ncz + (size_t)j * Ncz + k;
                        fprintf(f, "%d, %d, %d, %lu, %lu\n", i, j, k, (unsigned long)D[r * Nc + v],
                                (unsigned long)Mu[r * Nc + v]);
                    }
            fclose(f);
            if (nrep == 1 && p->first_replica == 0)
                snprintf(path, sizeof path, "%s/last_state_N%d_M%d_r%0.4f_T%0.2f.csv", dir, N, p->M, rho, p->T);
            else
                snprintf(path, sizeof path, "%s/last_state_N%d_M%d_r%0.4f_T%0.2f_rank%d.csv", dir, N, p->M, rho,
                         p->T, rank);
            f = fopen(path, "w");
            if (!f) { rc = SMCX_ERR_STATE; break; }
            for (int i = 0; i < 3 * N; i++) fprintf(f, "%0.12f,", R[(size_t)r * 3 * N + i]); /* main.c:169-170 */
            fclose(f);
        }
    } while (0);
    free(E); free(jj); free(D); free(Mu); free(R); free(P);
    return rc;
}

int smcx_host_read_last_state(const char *path, int N, double *R0)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int n = 0;
    while (n < 3 * N && fscanf(f, "%lf,", &R0[n]) == 1) n++; /* main.c:104-105 */
    fclose(f);
    return n;
}
