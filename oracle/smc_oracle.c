/*
 * smc_oracle.c -- CPU restatement of the reference hot path.  TEST
 * INFRASTRUCTURE ONLY (see smc_oracle.h for who may use it and for the
 * parity pin status).
 *
 * Written from the reference's behaviour, not from its text: run-time N/M,
 * one shared minimum-image helper, explicit RNG handle instead of libc's
 * hidden global.  The floating-point expression trees follow the cited
 * reference lines operation by operation (association included) so that,
 * built with -O2 -ffp-contract=off, results are the reference's own.
 */
#include "smc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------ */
/* glibc random_r TYPE_3 (x^31 + x^3 + 1), as used by srand()/rand().       */
/* SURVEY.md 8a row R; checked against this image's libc in tests.          */
/* ------------------------------------------------------------------------ */
void orc_srand(orc_rng *g, unsigned int seed)
{
    if (seed == 0)
        seed = 1;
    int32_t w = (int32_t)seed; /* glibc keeps the running word in int32_t: seeds >= 2^31 go negative */
    g->s[0] = (uint32_t)seed;
    for (int i = 1; i < 31; i++) {
        /* 16807 * w mod (2^31 - 1) by Schrage's split, signed */
        int64_t hi = w / 127773, lo = w % 127773;
        w = (int32_t)(16807 * lo - 2836 * hi);
        if (w < 0)
            w += 2147483647;
        g->s[i] = (uint32_t)w;
    }
    g->f = 3;
    g->r = 0;
    for (int i = 0; i < 310; i++)
        (void)orc_rand(g);
}

int orc_rand(orc_rng *g)
{
    g->s[g->f] += g->s[g->r];
    int out = (int)(g->s[g->f] >> 1);
    g->f = (g->f + 1 == 31) ? 0 : g->f + 1;
    g->r = (g->r + 1 == 31) ? 0 : g->r + 1;
    return out;
}

/* matematicose.c:183-193.  Pairs (x1,x2); the second output swaps the roles
 * of x1 and x2 (not a true Box-Muller pair) and an odd `length` leaves the
 * last element untouched -- both kept. */
void orc_vec_box_muller(orc_rng *g, double sigma, size_t length, double *A)
{
    size_t pairs = length / 2;
    for (size_t p = 0; p < pairs; p++) {
        double x1 = (double)orc_rand(g) / (ORC_RAND_MAX + 1.0);
        double x2 = (double)orc_rand(g) / (ORC_RAND_MAX + 1.0);
        A[2 * p] = sigma * sqrt(-2 * log(1 - x1)) * cos(2 * M_PI * x2);
        A[2 * p + 1] = sigma * sqrt(-2 * log(1 - x2)) * sin(2 * M_PI * x1);
    }
}

/* SMC_noMPI_noWall.c:707-717: uniforms over RAND_MAX (can be exactly 1),
 * sigma inside the square root. */
void orc_vec_box_muller_nw(orc_rng *g, double sigma, size_t length, double *A)
{
    size_t pairs = length / 2;
    for (size_t p = 0; p < pairs; p++) {
        double x1 = (double)orc_rand(g) / (double)ORC_RAND_MAX;
        double x2 = (double)orc_rand(g) / (double)ORC_RAND_MAX;
        A[2 * p] = sqrt(-2 * sigma * log(1 - x1)) * cos(2 * M_PI * x2);
        A[2 * p + 1] = sqrt(-2 * sigma * log(1 - x2)) * sin(2 * M_PI * x1);
    }
}

/* ------------------------------------------------------------------------ */
/* geometry helpers                                                         */
/* ------------------------------------------------------------------------ */
static inline double min_image(double d, double box)
{
    return d - box * rint(d / box);
}

/* signed distance to the nearer wall with the reference's clamp,
 * SMC.c:736-739 == 783-786 == 831-834 */
static inline double wall_dz(double rz, double Lz)
{
    double dz = rz + Lz / 2;
    dz = dz - Lz * rint(dz / Lz);
    if (rz <= -Lz / 2.0)
        dz = 0.0001;
    else if (rz >= Lz / 2)
        dz = -0.0001;
    return dz;
}

/* ------------------------------------------------------------------------ */
/* K1 energySingle  SMC.c:557-583                                           */
/* ------------------------------------------------------------------------ */
double orc_energy_single(const orc_sys *s, const double *r, int i)
{
    const int N = s->N;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    const double xi = r[3 * i], yi = r[3 * i + 1], zi = r[3 * i + 2];
    double V = 0.0;
    for (int l = 0; l < N; l++) {
        if (l == i)
            continue;
        double dx = min_image(r[3 * l] - xi, L);
        double dy = min_image(r[3 * l + 1] - yi, L);
        double dz = r[3 * l + 2] - zi; /* z is not periodic, SMC.c:571-572 */
        double dr2 = dx * dx + dy * dy + dz * dz;
        if (dr2 < c2) {
            double dr6 = dr2 * dr2 * dr2;
            V += 1.0 / (dr6 * dr6) - 1.0 / dr6;
        }
    }
    return V * 4;
}

/* ------------------------------------------------------------------------ */
/* K2 forceSingle  SMC.c:589-618 (d = r[i]-r[l]: opposite sign to K1)       */
/* ------------------------------------------------------------------------ */
void orc_force_single(const orc_sys *s, const double *r, int i, double F[3])
{
    const int N = s->N;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    const double xi = r[3 * i], yi = r[3 * i + 1], zi = r[3 * i + 2];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    for (int l = 0; l < N; l++) {
        if (l == i)
            continue;
        double dx = min_image(xi - r[3 * l], L);
        double dy = min_image(yi - r[3 * l + 1], L);
        double dz = zi - r[3 * l + 2];
        double dr2 = dx * dx + dy * dy + dz * dz;
        if (dr2 < c2) {
            double dr8 = dr2 * dr2 * dr2 * dr2;
            double dV = 48.0 / (dr8 * dr2 * dr2 * dr2) - 24.0 / dr8;
            fx += dV * dx;
            fy += dV * dy;
            fz += dV * dz;
        }
    }
    F[0] = fx;
    F[1] = fy;
    F[2] = fz;
}

/* ------------------------------------------------------------------------ */
/* K3 wallsEnergySingle  SMC.c:729-763                                      */
/* ------------------------------------------------------------------------ */
static double wall_energy_terms(const orc_sys *s, double rx, double ry, double rz,
                                const double *W, double V)
{
    const int M = s->M;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    const double dw = L / M;
    double dz = wall_dz(rz, s->Lz);
    double dz6 = dz * dz * dz * dz * dz * dz;
    V += s->a0 / (dz6 * dz6) - s->b0 / dz6; /* featureless plane, no cutoff */
    for (int i = 0; i < M; i++) {
        for (int j = 0; j < M; j++) {
            int m = j + i * M;
            double dx = min_image(rx - i * dw, L);
            double dy = min_image(ry - j * dw, L);
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c2) {
                double dr6 = dr2 * dr2 * dr2;
                V += W[2 * m] / (dr6 * dr6) - W[2 * m + 1] / dr6;
            }
        }
    }
    return V;
}

double orc_walls_energy_single(const orc_sys *s, double rx, double ry, double rz,
                               const double *W)
{
    return wall_energy_terms(s, rx, ry, rz, W, 0.0) * 4;
}

/* ------------------------------------------------------------------------ */
/* K4 wallsForce  SMC.c:773-813 -- accumulates into F, never zeroes it      */
/* ------------------------------------------------------------------------ */
void orc_walls_force(const orc_sys *s, double rx, double ry, double rz,
                     const double *W, double F[3])
{
    const int M = s->M;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    const double dw = L / M;
    double dz = wall_dz(rz, s->Lz);
    double dz8 = dz * dz * dz * dz * dz * dz * dz * dz;
    double dV = 48.0 * s->a0 / (dz8 * dz * dz * dz * dz * dz * dz) - 24.0 * s->b0 / dz8;
    F[2] += dV * dz;
    for (int i = 0; i < M; i++) {
        for (int j = 0; j < M; j++) {
            int m = j + i * M;
            double dx = min_image(rx - i * dw, L);
            double dy = min_image(ry - j * dw, L);
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c2) {
                double dr8 = dr2 * dr2 * dr2 * dr2;
                dV = 48.0 * W[2 * m] / (dr8 * dr2 * dr2 * dr2) - 24.0 * W[2 * m + 1] / dr8;
                F[0] += dV * dx;
                F[1] += dV * dy;
                F[2] += dV * dz;
            }
        }
    }
}

/* ------------------------------------------------------------------------ */
/* K5 energy SMC.c:626-646, wallsEnergy SMC.c:822-859                       */
/* ------------------------------------------------------------------------ */
double orc_energy(const orc_sys *s, const double *r)
{
    const int N = s->N;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    double V = 0.0;
    for (int l = 1; l < N; l++) {
        for (int i = 0; i < l; i++) {
            double dx = min_image(r[3 * l] - r[3 * i], L);
            double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
            double dz = r[3 * l + 2] - r[3 * i + 2];
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c2)
                V += 1.0 / (dr2 * dr2 * dr2 * dr2 * dr2 * dr2) - 1.0 / (dr2 * dr2 * dr2);
        }
    }
    return V * 4;
}

double orc_walls_energy(const orc_sys *s, const double *r, const double *W)
{
    double V = 0.0; /* one running sum over all particles, SMC.c:824-857 */
    for (int n = 0; n < s->N; n++)
        V = wall_energy_terms(s, r[3 * n], r[3 * n + 1], r[3 * n + 2], W, V);
    return V * 4;
}

/* ------------------------------------------------------------------------ */
/* pressure SMC.c:696-720, wallsPressure SMC.c:862-895                      */
/* ------------------------------------------------------------------------ */
double orc_pressure(const orc_sys *s, const double *r)
{
    const int N = s->N;
    const double L = s->L, c2 = s->cutoff * s->cutoff;
    double P = 0.0;
    for (int l = 1; l < N; l++) {
        for (int i = 0; i < l; i++) {
            double dx = min_image(r[3 * l] - r[3 * i], L);
            double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
            double dz = r[3 * l + 2] - r[3 * i + 2];
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c2) {
                double dr6 = dr2 * dr2 * dr2;
                P += 24.0 / dr6 - 48.0 / (dr6 * dr6);
            }
        }
    }
    return -P / (3 * L * L * s->Lz);
}

double orc_walls_pressure(const orc_sys *s, const double *r, const double *W)
{
    const int N = s->N, M = s->M;
    const double L = s->L, Lz = s->Lz, c2 = s->cutoff * s->cutoff;
    const double dw = L / M;
    double P = 0.0;
    for (int i = 0; i < M; i++) {
        for (int j = 0; j < M; j++) {
            int m = j + i * M;
            for (int n = 0; n < N; n++) {
                double dx = min_image(r[3 * n] - i * dw, L);
                double dy = min_image(r[3 * n + 1] - j * dw, L);
                double dz = r[3 * n + 2] + L / 2; /* sic: L/2, SMC.c:880 */
                dz = dz - Lz * rint(dz / Lz);
                double dr2 = dx * dx + dy * dy + dz * dz;
                if (dr2 < c2) {
                    double dr6 = dr2 * dr2 * dr2;
                    P += 24.0 * W[2 * m + 1] / dr6 - 48.0 * W[2 * m] / (dr6 * dr6);
                    double dz6 = dz * dz * dz * dz * dz * dz;
                    P += 24.0 * s->b0 / dz6 - 48.0 * s->a0 / (dz6 * dz6);
                }
            }
        }
    }
    return -P / (3 * L * L * Lz);
}

/* ------------------------------------------------------------------------ */
/* S1 oneParticleMoves  SMC.c:278-351                                       */
/* ------------------------------------------------------------------------ */
void orc_one_particle_moves(const orc_sys *s, orc_rng *g, double *R, double *Rn,
                            const double *W, double A0, double T, int *j,
                            double *E, orc_move_trace *trace)
{
    const int N = s->N;
    const double L = s->L;
    const double A = A0;
    double *displ = (double *)malloc(3 * (size_t)N * sizeof(double));

    orc_vec_box_muller(g, sqrt(2.0 * A), 3 * (size_t)N, displ); /* SMC.c:284 */
    memcpy(Rn, R, 3 * (size_t)N * sizeof(double));              /* SMC.c:286-287 */

    /* SMC.c:290-294 computes (nn+offset)%N in int, which overflows when
     * offset > INT_MAX-N; fenced here with 64-bit arithmetic. */
    const int64_t offset = orc_rand(g);

    for (int nn = 0; nn < N; nn++) {
        const int n = (int)(((int64_t)nn + offset) % N);
        double *p = R + 3 * n, *q = Rn + 3 * n;
        double Fm[3], Fn[3];

        double Um = orc_energy_single(s, R, n) + orc_walls_energy_single(s, p[0], p[1], p[2], W);
        orc_force_single(s, R, n, Fm);
        orc_walls_force(s, p[0], p[1], p[2], W, Fm);

        double dX = Fm[0] * A / T + displ[3 * n];
        double dY = Fm[1] * A / T + displ[3 * n + 1];
        double dZ = Fm[2] * A / T + displ[3 * n + 2];

        q[0] = p[0] + dX;
        q[1] = p[1] + dY;
        q[2] = p[2] + dZ;
        q[0] = q[0] - L * rint(q[0] / L); /* x,y wrapped, z not: SMC.c:315-316 */
        q[1] = q[1] - L * rint(q[1] / L);

        double Un = orc_energy_single(s, Rn, n) + orc_walls_energy_single(s, q[0], q[1], q[2], W);
        orc_force_single(s, Rn, n, Fn);
        orc_walls_force(s, q[0], q[1], q[2], W, Fn);

        double gx = Fn[0] - Fm[0], gy = Fn[1] - Fm[1], gz = Fn[2] - Fm[2];
        double deltaW = (gx * gx + gy * gy + gz * gz +
                         2.0 * (gx * Fm[0] + gy * Fm[1] + gz * Fm[2])) * A / (4.0 * T);
        double ap = exp(-(Un - Um +
                          (dX * (Fn[0] + Fm[0]) + dY * (Fn[1] + Fm[1]) + dZ * (Fn[2] + Fm[2])) / 2.0 +
                          deltaW) / T);
        double u = (double)orc_rand(g) / (double)ORC_RAND_MAX; /* SMC.c:335 */
        int acc = (u < ap);

        if (trace) {
            orc_move_trace *t = &trace[nn];
            t->n = n;
            t->accepted = acc;
            t->Um = Um;
            t->Un = Un;
            t->ap = ap;
            t->u = u;
            t->delta[0] = dX; t->delta[1] = dY; t->delta[2] = dZ;
            for (int c = 0; c < 3; c++) {
                t->Fm[c] = Fm[c];
                t->Fn[c] = Fn[c];
                t->prop[c] = q[c];
            }
        }
        if (acc) {
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            *j += 1;
            *E += Un - Um;
        } else {
            q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
        }
    }
    free(displ);
}

/* ------------------------------------------------------------------------ */
/* H localDensityAndMobility  SMC.c:912-927                                 */
/* The reference stores floor() results in uint8_t (wraps mod 256; negative */
/* values are formally undefined, gcc/x86-64 truncates the int conversion). */
/* ------------------------------------------------------------------------ */
static inline int cell8(double v)
{
    return (int)(uint8_t)(int32_t)floor(v);
}

void orc_local_density(const orc_sys *s, const double *r, uint64_t *D,
                       int32_t *Rbin, uint64_t *Mu, uint64_t *oob)
{
    const int Ncx = s->Ncx, Ncz = s->Ncz;
    const int64_t Nc = (int64_t)Ncx * Ncx * Ncz;
    for (int n = 0; n < s->N; n++) {
        int i = cell8((r[3 * n] / s->L + .5) * Ncx);
        int j = cell8((r[3 * n + 1] / s->L + .5) * Ncx);
        int k = cell8((r[3 * n + 2] / s->Lz + .5) * Ncz);
        int v = i * Ncx * Ncz + j * Ncz + k;
        if (v >= Nc) { /* the reference writes D[v], Mu[v] out of bounds here: not stored, counted; */
            if (oob)   /* Rbin[n] = v is in bounds and kept, so the next in-range visit bumps Mu */
                (*oob)++;
            Rbin[n] = v;
            continue;
        }
        D[v]++;
        if (Rbin[n] != v) {
            Mu[v]++;
            Rbin[n] = v;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* C chain bookkeeping of sMC (SMC.c:44-56, 110-118, 134-141, 194-195,      */
/* 207-211, 244-250), without the CSV writers and without pressure/LCA.     */
/* ------------------------------------------------------------------------ */
int orc_chain(const orc_sys *s, unsigned int seed, double *R, const double *W,
              double T, double A, int eqsteps, int maxsteps, int gather_lapse,
              unsigned int flags, double *E_series, int32_t *jj_out, uint64_t *zhist,
              uint64_t *D_out, uint64_t *Mu_out, orc_chain_result *res)
{
    return orc_chain_p(s, seed, R, W, T, A, eqsteps, maxsteps, gather_lapse, flags, E_series,
                       jj_out, zhist, D_out, Mu_out, NULL, res);
}

int orc_chain_p(const orc_sys *s, unsigned int seed, double *R, const double *W,
                double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                unsigned int flags, double *E_series, int32_t *jj_out, uint64_t *zhist,
                uint64_t *D_out, uint64_t *Mu_out, double *P_gathers, orc_chain_result *res)
{
    return orc_chain_lca(s, seed, R, W, T, A, eqsteps, maxsteps, gather_lapse, flags, E_series,
                         jj_out, zhist, D_out, Mu_out, P_gathers, 0, 0.0, NULL, res);
}

static int chain_impl(const orc_sys *s, unsigned int seed, double *R, const double *W,
                  double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                  unsigned int flags, double *E_series, int32_t *jj_out, uint64_t *zhist,
                  uint64_t *D_out, uint64_t *Mu_out, double *P_gathers, int lca_time,
                  double lca_cutoff, orc_lca_counts *lca, orc_chain_result *res, int32_t *jt_out);

int orc_chain_lca(const orc_sys *s, unsigned int seed, double *R, const double *W,
                  double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                  unsigned int flags, double *E_series, int32_t *jj_out, uint64_t *zhist,
                  uint64_t *D_out, uint64_t *Mu_out, double *P_gathers, int lca_time,
                  double lca_cutoff, orc_lca_counts *lca, orc_chain_result *res)
{
    return chain_impl(s, seed, R, W, T, A, eqsteps, maxsteps, gather_lapse, flags, E_series, jj_out,
                      zhist, D_out, Mu_out, P_gathers, lca_time, lca_cutoff, lca, res, NULL);
}

int orc_chain_jt(const orc_sys *s, unsigned int seed, double *R, const double *W,
                 double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                 unsigned int flags, double *E_series, int32_t *jj_out, int32_t *jt_out,
                 uint64_t *zhist, uint64_t *D_out, uint64_t *Mu_out, double *P_gathers,
                 orc_chain_result *res)
{
    return chain_impl(s, seed, R, W, T, A, eqsteps, maxsteps, gather_lapse, flags, E_series, jj_out,
                      zhist, D_out, Mu_out, P_gathers, 0, 0.0, NULL, res, jt_out);
}

static int chain_impl(const orc_sys *s, unsigned int seed, double *R, const double *W,
                  double T, double A, int eqsteps, int maxsteps, int gather_lapse,
                  unsigned int flags, double *E_series, int32_t *jj_out, uint64_t *zhist,
                  uint64_t *D_out, uint64_t *Mu_out, double *P_gathers, int lca_time,
                  double lca_cutoff, orc_lca_counts *lca, orc_chain_result *res, int32_t *jt_out)
{
    const int N = s->N;
    int32_t *LCA = NULL;
    if (lca && lca_time > 0) {
        memset(lca, 0, sizeof(*lca));
        LCA = (int32_t *)calloc(3 * ((size_t)N * (N - 1) / 2) + 3, sizeof(int32_t));
    }
    const size_t Nc = (size_t)s->Ncx * s->Ncx * s->Ncz;
    if (N < 1 || maxsteps < 0 || eqsteps < 0 || gather_lapse < 1)
        return -1;
    size_t elen = (size_t)(maxsteps > eqsteps ? maxsteps : eqsteps) + 1;
    double *E = (double *)calloc(elen, sizeof(double));
    double *Rn = (double *)calloc(3 * (size_t)N, sizeof(double));
    int32_t *jj = (int32_t *)calloc((size_t)maxsteps + 1, sizeof(int32_t));
    int32_t *jt = (int32_t *)calloc((size_t)eqsteps + 1, sizeof(int32_t));
    int32_t *Rbin = (int32_t *)calloc((size_t)N, sizeof(int32_t));
    uint64_t *D = (uint64_t *)calloc(Nc, sizeof(uint64_t));
    uint64_t *Mu = (uint64_t *)calloc(Nc, sizeof(uint64_t));
    uint64_t oob = 0, gathers = 0;
    orc_rng g;
    orc_srand(&g, seed);

    E[0] = orc_energy(s, R) + orc_walls_energy(s, R, W); /* SMC.c:48 */
    const double E0 = E[0];

    double Ath = A * 2; /* SMC.c:110 */
    for (int n = 0; n < eqsteps; n++) {
        E[n + 1] = E[n];
        int jtmp = 0;
        orc_one_particle_moves(s, &g, R, Rn, W, Ath, T, &jtmp, &E[n + 1], NULL);
        jt[n] = jtmp;
    }
    double Etherm_end = E[eqsteps];
    if (!(flags & ORC_FLAG_E0_RESTART))
        E[0] = Etherm_end; /* non-reference mode: carry the energy across */

    for (int n = 0; n < maxsteps; n++) {
        if ((n + 1) % gather_lapse == 0) { /* SMC.c:137-141: before this sweep's moves */
            if (P_gathers)
                P_gathers[gathers] = orc_pressure(s, R) + orc_walls_pressure(s, R, W);
            orc_local_density(s, R, D, Rbin, Mu, &oob);
            gathers++;
            const int k = (n + 1) / gather_lapse; /* SMC.c:138 */
            if (LCA && k % lca_time == 0) {       /* SMC.c:143-155 */
                orc_cluster_analysis(N, R, s->L, lca_cutoff, LCA, &lca->overflow);
                orc_cluster_counts(N, LCA, lca);
                lca->analyses++;
            }
        }
        E[n + 1] = E[n]; /* SMC.c:194: production restarts from E[0] */
        int jtmp = 0;
        orc_one_particle_moves(s, &g, R, Rn, W, A, T, &jtmp, &E[n + 1], NULL);
        jj[n] = jtmp;
    }

    if (res) {
        memset(res, 0, sizeof(*res));
        res->E0 = E0;
        res->Efinal = E[maxsteps];
        /* SMC.c:210-211 then mean/variance over maxsteps+1 entries */
        double sum = 0., sum2 = 0.;
        for (int n = 0; n < maxsteps + 1; n++) {
            double e = E[n] + 3 * N * T / 2;
            sum += e;
        }
        for (int n = 0; n < maxsteps + 1; n++) {
            double e = E[n] + 3 * N * T / 2;
            sum2 += e * e;
        }
        double len = (double)(maxsteps + 1);
        res->meanE = sum / len;
        double var = sum2 / len - (sum / len) * (sum / len);
        res->dE = sqrt(var);
        res->cv = var / (T * T); /* SMC.c:250 */
        int64_t acc = 0; /* intmean sums in int (overflow past 2^31), fenced */
        for (int n = 0; n < maxsteps; n++)
            acc += jj[n];
        res->accepted = (uint64_t)acc;
        res->acceptance_ratio = maxsteps ? ((double)acc / maxsteps) / N : 0.0;
        int64_t acct = 0;
        for (int n = 0; n < eqsteps; n++)
            acct += jt[n];
        res->therm_acceptance = eqsteps ? ((double)acct / eqsteps) / N : 0.0;
        res->gathers = gathers;
        res->oob = oob;
    }
    if (E_series) {
        for (int n = 0; n < maxsteps + 1; n++)
            E_series[n] = E[n];
    }
    if (jj_out)
        memcpy(jj_out, jj, (size_t)maxsteps * sizeof(int32_t));
    if (jt_out)
        memcpy(jt_out, jt, (size_t)eqsteps * sizeof(int32_t));
    if (zhist) { /* wall-normal profile: sum over i,j of D[i][j][k] (plotting.jl:134-166) */
        memset(zhist, 0, (size_t)s->Ncz * sizeof(uint64_t));
        for (size_t v = 0; v < Nc; v++)
            zhist[v % (size_t)s->Ncz] += D[v];
    }
    if (D_out)
        memcpy(D_out, D, Nc * sizeof(uint64_t));
    if (Mu_out)
        memcpy(Mu_out, Mu, Nc * sizeof(uint64_t));
    free(E); free(Rn); free(jj); free(jt); free(Rbin); free(D); free(Mu); free(LCA);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* LCA  common-neighbour cluster analysis, SMC.c:971-1045, with the          */
/* reference's arithmetic kept as written:                                   */
/*  - the pair index (l*l-3*l+2)/2 + i advances by l-1 per row although row  */
/*    l has l entries, so (l,l-1) and (l+1,0) share an entry: num1 is the OR */
/*    of the two, num2/num3 accumulate over both (in that order, which is    */
/*    also consecutive in the loop, so common_nn simply grows);              */
/*  - the i-i2 entry is taken at (i2*i2-3*i2+2)/2 + i also when i > i2;      */
/*  - only consecutive common neighbours (m-1, m) are tested for a bond.     */
/* common_nn has 8 slots in the reference and is written without a bound     */
/* (undefined behaviour past 8): here such stores and the reads of slots     */
/* >= 8 are skipped and counted in *overflow.                                */
/* ------------------------------------------------------------------------ */
void orc_cluster_analysis(int N_, const double *r, double L, double cutoff, int32_t *LCA,
                          uint64_t *overflow)
{
    const size_t np = (size_t)N_ * (N_ - 1) / 2;
    unsigned char *num1 = (unsigned char *)calloc(np + 1, 1);
    int32_t *num2 = (int32_t *)calloc(np + 1, sizeof(int32_t));
    int32_t *num3 = (int32_t *)calloc(np + 1, sizeof(int32_t));
    int common_nn[8];
    int idx, idx2, idx3;
    double dx, dy, dz, dist2;

    for (int l = 1; l < N_; l++) { /* :984-1000 */
        for (int i = 0; i < l; i++) {
            idx = (l * l - 3 * l + 2) / 2 + i;
            dx = r[3 * l] - r[3 * i];
            dx = dx - L * rint(dx / L);
            dy = r[3 * l + 1] - r[3 * i + 1];
            dy = dy - L * rint(dy / L);
            dz = r[3 * l + 2] - r[3 * i + 2];
            dist2 = dx * dx + dy * dy + dz * dz;
            if (dist2 < cutoff * cutoff)
                num1[idx] = 1;
        }
    }
    for (int l = 1; l < N_; l++) { /* :1003-1035 */
        for (int i = 0; i < l; i++) {
            idx = (l * l - 3 * l + 2) / 2 + i;
            if (!num1[idx])
                continue;
            for (int i2 = 0; i2 < l; i2++) {
                if (i2 == i)
                    continue;
                idx2 = idx - i + i2;
                idx3 = (i2 * i2 - 3 * i2 + 2) / 2 + i;
                if (num1[idx2] & num1[idx3]) {
                    if (num2[idx] < 8)
                        common_nn[num2[idx]] = i2;
                    else if (overflow)
                        (*overflow)++;
                    num2[idx]++;
                }
            }
            if (num2[idx] > 1) {
                for (int m = 1; m < num2[idx]; m++) {
                    if (m >= 8)
                        break; /* slots the reference never owned */
                    idx2 = (common_nn[m] * common_nn[m] - 3 * common_nn[m] + 2) / 2 + common_nn[m - 1];
                    if (num1[idx2])
                        num3[idx]++;
                }
            }
        }
    }
    for (size_t n = 0; n < np; n++) { /* :1038-1044 */
        LCA[3 * n + 0] = (int32_t)num1[n];
        LCA[3 * n + 1] = num2[n];
        LCA[3 * n + 2] = num3[n];
    }
    free(num1); free(num2); free(num3);
}

/* the accumulation loop of sMC, SMC.c:146-155, as plain counts: the reference adds the
 * weight 1/(gather_steps/LCA_TIME) -- an int division, 0 unless the ratio is 1 -- into
 * l1, l2[num2], l3[num3] (l2, l3 uninitialised, 7 entries); the counts are what a caller
 * needs to apply any weight.  Values above 15 land in bin 15. */
void orc_cluster_counts(int N_, const int32_t *LCA, orc_lca_counts *c)
{
    const size_t np = (size_t)N_ * (N_ - 1) / 2;
    for (size_t i = 0; i < np; i++) {
        if (LCA[3 * i] != 0) {
            c->n1++;
            int a = LCA[3 * i + 1], b = LCA[3 * i + 2];
            c->h2[a > 15 ? 15 : a]++;
            c->h3[b > 15 ? 15 : b]++;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* inputs: lattices and wall strengths                                      */
/* ------------------------------------------------------------------------ */
static void fcc_fill(int Na, int Nz, double a, double *X, int capacity_particles)
{
    /* cell order i,j,k with k fastest, four-atom basis, SMC.c:432-453 */
    static const double basis[4][3] = {
        {0.0, 0.0, 0.0}, {0.5, 0.5, 0.0}, {0.5, 0.0, 0.5}, {0.0, 0.5, 0.5}};
    for (int i = 0; i < Na; i++)
        for (int j = 0; j < Na; j++)
            for (int k = 0; k < Nz; k++) {
                int cell = i * Na * Nz + j * Nz + k;
                for (int b = 0; b < 4; b++) {
                    int p = 4 * cell + b;
                    if (p >= capacity_particles)
                        continue;
                    X[3 * p + 0] = basis[b][0] != 0.0 ? a * i + a / 2 : a * i;
                    X[3 * p + 1] = basis[b][1] != 0.0 ? a * j + a / 2 : a * j;
                    X[3 * p + 2] = basis[b][2] != 0.0 ? a * k + a / 2 : a * k;
                }
            }
}

static void shift3d(double *X, int N, double L, double Lzs)
{
    for (int n = 0; n < N; n++) { /* SMC.c:512-519 */
        X[3 * n] = X[3 * n] - L * rint(X[3 * n] / L);
        X[3 * n + 1] = X[3 * n + 1] - L * rint(X[3 * n + 1] / L);
        X[3 * n + 2] = X[3 * n + 2] - Lzs * rint(X[3 * n + 2] / Lzs);
    }
}

int orc_fcc_init(int Na, int Nz, double L, double Lz, double *X)
{
    if (Na < 1 || Nz < 1)
        return -1;
    int N = 4 * Na * Na * Nz;
    double a = L / Na;
    fcc_fill(Na, Nz, a, X, N);
    for (int n = 0; n < 3 * N; n++)
        X[n] += a / 4; /* SMC.c:456-458 (the rand()/RAND_MAX jitter is integer 0) */
    shift3d(X, N, L, Lz - Lz / 20.0); /* SMC.c:461 */
    return N;
}

int orc_initialize_box_ref(double L, double Lz, int N, double *X)
{
    /* SMC.c:416-431: Na = largest nc with nc^3 <= N/4, Nz = (N/4)/(Na*Na) */
    int Nc = N / 4;
    int Na = 1;
    for (int nc = 1; nc < N; nc++) {
        if (nc * nc * nc > Nc) {
            Na = nc - 1;
            break;
        }
    }
    int Nz = (int)rint((double)((N / 4) / (Na * Na)));
    double a = L / Na;
    memset(X, 0, 3 * (size_t)N * sizeof(double)); /* caller's calloc, main.c:96 */
    fcc_fill(Na, Nz, a, X, N);
    for (int n = 0; n < 3 * N; n++)
        X[n] += a / 4;
    shift3d(X, N, L, Lz - Lz / 20.0);
    int placed = 4 * Na * Na * Nz;
    return placed < N ? placed : N;
}

void orc_initialize_walls(double x0m, double x0sigma, double ymm, double ymsigma,
                          int M, double uninit, double *W)
{
    orc_rng g;
    orc_srand(&g, 42); /* SMC.c:477 */
    size_t n = (size_t)M * M;
    double *X0 = (double *)malloc(n * sizeof(double));
    double *YM = (double *)malloc(n * sizeof(double));
    for (size_t i = 0; i < n; i++)
        X0[i] = YM[i] = uninit;
    orc_vec_box_muller(&g, x0sigma, n, X0);
    orc_vec_box_muller(&g, ymsigma, n, YM);
    for (size_t m = 0; m < n; m++) {
        double x0 = X0[m] + x0m;
        W[2 * m] = pow(x0, 12.0) * (YM[m] + ymm);
        W[2 * m + 1] = pow(x0, 6.) * (YM[m] + ymm);
    }
    free(X0);
    free(YM);
}

/* ------------------------------------------------------------------------ */
/* NW: SMC_noMPI_noWall.c hot path (BASELINE config 1)                      */
/* cubic fully periodic box, cutoff L/2, neighbour loops start at l=1       */
/* ------------------------------------------------------------------------ */
double orc_nw_energy_single(int N, const double *r, double L, int i)
{
    double V = 0.0;
    for (int l = 1; l < N; l++) { /* particle 0 is never a neighbour, :603 */
        if (l == i)
            continue;
        double dx = min_image(r[3 * l] - r[3 * i], L);
        double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
        double dz = min_image(r[3 * l + 2] - r[3 * i + 2], L);
        double dr2 = dx * dx + dy * dy + dz * dz;
        if (dr2 < L * L / 4) {
            double dr6 = dr2 * dr2 * dr2;
            V += 1.0 / (dr6 * dr6) - 1.0 / dr6;
        }
    }
    return V * 4;
}

void orc_nw_force(int N, const double *r, double L, int i, double F[3])
{
    double fx = 0.0, fy = 0.0, fz = 0.0;
    for (int l = 1; l < N; l++) { /* :508 */
        if (l == i)
            continue;
        double dx = min_image(r[3 * l] - r[3 * i], L);
        double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
        double dz = min_image(r[3 * l + 2] - r[3 * i + 2], L);
        double dr2 = dx * dx + dy * dy + dz * dz;
        if (dr2 < L * L / 4) {
            double dr8 = dr2 * dr2 * dr2 * dr2;
            double dV = 24.0 / dr8 - 48.0 / (dr8 * dr2 * dr2 * dr2);
            fx -= dV * dx; /* with d = r[l]-r[i]: the force ON l, :512-525 */
            fy -= dV * dy;
            fz -= dV * dz;
        }
    }
    F[0] = fx; F[1] = fy; F[2] = fz;
}

double orc_nw_energy(int N, const double *r, double L)
{
    double V = 0.0;
    for (int l = 1; l < N; l++)
        for (int i = 0; i < l; i++) {
            double dx = min_image(r[3 * l] - r[3 * i], L);
            double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
            double dz = min_image(r[3 * l + 2] - r[3 * i + 2], L);
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < L * L / 4)
                V += 1.0 / (dr2 * dr2 * dr2 * dr2 * dr2 * dr2) - 1.0 / (dr2 * dr2 * dr2);
        }
    return V * 4;
}

double orc_nw_pressure(int N, const double *r, double L)
{
    double P = 0.0;
    for (int l = 1; l < N; l++)
        for (int i = 0; i < l; i++) {
            double dx = min_image(r[3 * l] - r[3 * i], L);
            double dy = min_image(r[3 * l + 1] - r[3 * i + 1], L);
            double dz = min_image(r[3 * l + 2] - r[3 * i + 2], L);
            double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < L * L / 4)
                P += 24.0 / (dr2 * dr2 * dr2) - 48.0 / (dr2 * dr2 * dr2 * dr2 * dr2 * dr2);
        }
    return -P / (3 * L * L * L);
}

void orc_nw_one_particle_moves(int N, orc_rng *g, double *R, double *Rn, double L,
                               double A, double T, int *j, orc_move_trace *trace)
{
    double *displ = (double *)malloc(3 * (size_t)N * sizeof(double));
    orc_vec_box_muller_nw(g, sqrt(2 * A), 3 * (size_t)N, displ);
    memcpy(Rn, R, 3 * (size_t)N * sizeof(double));

    for (int n = 0; n < N; n++) { /* fixed visiting order, :278 */
        double *p = R + 3 * n, *q = Rn + 3 * n;
        double Fm[3], Fn[3];
        double Um = orc_nw_energy_single(N, R, L, n);
        orc_nw_force(N, R, L, n, Fm);

        double dX = Fm[0] * (A / T) + displ[3 * n];
        double dY = Fm[1] * (A / T) + displ[3 * n + 1];
        double dZ = Fm[2] * (A / T) + displ[3 * n + 2];
        q[0] = p[0] + dX; q[1] = p[1] + dY; q[2] = p[2] + dZ;

        double Un = orc_nw_energy_single(N, Rn, L, n); /* from the unwrapped proposal */
        orc_nw_force(N, Rn, L, n, Fn);

        for (int c = 0; c < 3 * N; c++) /* shiftSystem(Rn,L) on all 3N coords, :291-294 */
            Rn[c] = Rn[c] - L * rint(Rn[c] / L);

        double gx = Fn[0] - Fm[0], gy = Fn[1] - Fm[1], gz = Fn[2] - Fm[2];
        double deltaW = (gx * gx + gy * gy + gz * gz +
                         2 * (gx * Fm[0] + gy * Fm[1] + gz * Fm[2])) * A / (4 * T);
        double ap = exp(-(Un - Um +
                          (dX * (Fn[0] + Fm[0]) + dY * (Fn[1] + Fm[1]) + dZ * (Fn[2] + Fm[2])) / 2 +
                          deltaW) / T);
        double u = (double)orc_rand(g) / (double)ORC_RAND_MAX;
        int acc = (u < ap);
        if (trace) {
            orc_move_trace *t = &trace[n];
            t->n = n; t->accepted = acc; t->Um = Um; t->Un = Un; t->ap = ap; t->u = u;
            t->delta[0] = dX; t->delta[1] = dY; t->delta[2] = dZ;
            for (int c = 0; c < 3; c++) { t->Fm[c] = Fm[c]; t->Fn[c] = Fn[c]; t->prop[c] = q[c]; }
        }
        if (acc) {
            p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
            *j += 1;
        } else {
            q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
        }
    }
    free(displ);
}

int orc_nw_fcc_init(int N, double L, double *X)
{
    int Na = (int)cbrt((double)(N / 4)); /* :361 */
    if (4 * Na * Na * Na != N)
        return -1;
    double a = L / Na;
    fcc_fill(Na, Na, a, X, N);
    for (int n = 0; n < 3 * N; n++)
        X[n] += a / 4;
    for (int n = 0; n < 3 * N; n++)
        X[n] = X[n] - L * rint(X[n] / L);
    return N;
}

/* ------------------------------------------------------------------------ */
/* cpu_baseline timing leg                                                  */
/* ------------------------------------------------------------------------ */
double orc_time_sweeps(const orc_sys *s, unsigned int seed, double *R, const double *W,
                       double T, double A, int sweeps, uint64_t *accepted)
{
    orc_rng g;
    orc_srand(&g, seed);
    double *Rn = (double *)calloc(3 * (size_t)s->N, sizeof(double));
    double E = 0.0;
    uint64_t acc = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int n = 0; n < sweeps; n++) {
        int j = 0;
        orc_one_particle_moves(s, &g, R, Rn, W, A, T, &j, &E, NULL);
        acc += (uint64_t)j;
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(Rn);
    if (accepted)
        *accepted = acc;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
