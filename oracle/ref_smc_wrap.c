/*
 * oracle/ref_smc_wrap.c -- tail of the translation unit of oracle/_ref/libref_smc_N<n>.so
 * (see ref_smc_prelude.c for the whole layout).  TEST INFRASTRUCTURE: entry points with
 * plain C types around the REAL reference functions compiled just above this text.
 * Nothing here computes physics: every number returned comes out of the reference's own
 * energySingle / forceSingle / wallsEnergySingle / wallsForce / energy / wallsEnergy /
 * pressure / wallsPressure / oneParticleMoves / localDensityAndMobility / clusterAnalysis /
 * initializeBox / initializeWalls, driven with libc's own srand()/rand().
 */

int refw_N(void) { return N; }
int refw_M(void) { return M; }
int refw_ncx(void) { return Ncx; }
int refw_ncz(void) { return Ncz; }
double refw_cutoff(void) { double L = 0.0; (void)L; return LJ_CUTOFF; }
double refw_a0(void) { return a0; }
double refw_b0(void) { return b0; }
double refw_lca_cutoff(void) { return LCA_cutoff; }
int refw_lca_time(void) { return LCA_TIME; }

void refw_srand(unsigned int seed) { srand(seed); }
int refw_rand(void) { return rand(); }

/* K1..K4 for particle i of state r, called the way oneParticleMoves calls them
 * (SMC.c:300-304): out = { energySingle, wallsEnergySingle, F of forceSingle alone [3],
 * F after wallsForce has added into it [3] } */
void refw_single(const double *r, const double *W, double L, double Lz, int i, double *out)
{
    double Fx, Fy, Fz;
    out[0] = energySingle(r, L, i);
    out[1] = wallsEnergySingle(r[3*i], r[3*i+1], r[3*i+2], W, L, Lz);
    forceSingle(r, L, i, &Fx, &Fy, &Fz);
    out[2] = Fx; out[3] = Fy; out[4] = Fz;
    wallsForce(r[3*i], r[3*i+1], r[3*i+2], W, L, Lz, &Fx, &Fy, &Fz);
    out[5] = Fx; out[6] = Fy; out[7] = Fz;
}

/* K3/K4 at an arbitrary point (clamp branch, beyond-the-wall points); F starts from Fin */
void refw_wall_point(double rx, double ry, double rz, const double *W, double L, double Lz,
                     const double *Fin, double *out)
{
    double Fx = Fin[0], Fy = Fin[1], Fz = Fin[2];
    out[0] = wallsEnergySingle(rx, ry, rz, W, L, Lz);
    wallsForce(rx, ry, rz, W, L, Lz, &Fx, &Fy, &Fz);
    out[1] = Fx; out[2] = Fy; out[3] = Fz;
}

double refw_energy(const double *r, double L) { return energy(r, L); }
double refw_walls_energy(const double *r, const double *W, double L, double Lz)
{ return wallsEnergy(r, W, L, Lz); }
double refw_pressure(const double *r, double L, double Lz) { return pressure(r, L, Lz); }
double refw_walls_pressure(const double *r, const double *W, double L, double Lz)
{ return wallsPressure(r, W, L, Lz); }

/* nsweeps calls of the real oneParticleMoves with sMC's bookkeeping (SMC.c:116-117 /
 * 194-195): E has nsweeps+1 entries, E[0] preloaded by the caller; jj nsweeps entries,
 * zeroed here as sMC's calloc does.  seed_or_neg < 0 keeps libc's current rand() state. */
void refw_sweeps(long seed_or_neg, double *R, double *Rn, const double *W, double L, double Lz,
                 double A, double T, int nsweeps, int *jj, double *E)
{
    if (seed_or_neg >= 0) srand((unsigned int)seed_or_neg);
    for (int n = 0; n < nsweeps; n++) {
        jj[n] = 0;
        E[n+1] = E[n];
        oneParticleMoves(R, Rn, W, L, Lz, A, T, &jj[n], &E[n+1]);
    }
}

void refw_local_density(const double *r, double L, double Lz, unsigned long *D, int *Rbin,
                        unsigned long *Mu)
{ localDensityAndMobility(r, L, Lz, D, Rbin, Mu); }

/* number of histogram counters a caller must provide so that the uint8_t cell numbers of
 * SMC.c:914-920 (0..255 each) can never index past the arrays */
long refw_hist_len(void) { return 255L*Ncx*Ncz + 255L*Ncz + 255L + 1; }

/*
 * The loop structure of sMC (SMC.c:44-48, 110-118, 125, 134-141, 194-195) around the real
 * functions, with srand(seed) in place of srand(time(NULL)) (SMC.c:40) and without the file
 * output, clusterAnalysis and fft_acf.  R: in = R0, out = final positions.
 * E: maxsteps+1 (and >= eqsteps+1) entries; jj: maxsteps; jt: eqsteps; D, Mu: refw_hist_len()
 * counters (zeroed by the caller); P: maxsteps/gather_lapse + 1 entries (the reference
 * writes P[k] for k = 1..gather_steps, one past its own array, SMC.c:49,140).
 */
int refw_chain(unsigned int seed, double *R, const double *W, double L, double Lz, double T,
               double A, int eqsteps, int maxsteps, int gather_lapse,
               double *E, int *jj, int *jt, unsigned long *D, unsigned long *Mu, double *P)
{
    srand(seed);
    double *Rn = calloc(3*N, sizeof(double));
    int *Rbin = calloc(N, sizeof(int));
    if (!Rn || !Rbin) return -1;
    E[0] = energy(R, L) + wallsEnergy(R, W, L, Lz);
    A = A*2;
    for (int n = 0; n < eqsteps; n++) {
        jt[n] = 0;
        E[n+1] = E[n];
        oneParticleMoves(R, Rn, W, L, Lz, A, T, &jt[n], &E[n+1]);
    }
    A = A/2;
    for (int n = 0; n < maxsteps; n++) {
        if ((n+1) % gather_lapse == 0) {
            int k = (int)((n+1)/gather_lapse);
            if (P) P[k] = pressure(R, L, Lz) + wallsPressure(R, W, L, Lz);
            localDensityAndMobility(R, L, Lz, D, Rbin, Mu);
        }
        jj[n] = 0;
        E[n+1] = E[n];
        oneParticleMoves(R, Rn, W, L, Lz, A, T, &jj[n], &E[n+1]);
    }
    free(Rn); free(Rbin);
    return 0;
}

/* final reductions of sMC (SMC.c:207-211, 244-250) with the reference's own mean / variance /
 * intmean: out = { results.E, results.dE, results.acceptance_ratio, results.cv } */
void refw_results(double *E, const int *jj, int maxsteps, double T, double *out)
{
    for (int n = 0; n < maxsteps+1; n++)
        E[n] += 3*N*T/2;
    out[0] = mean(E, maxsteps+1);
    out[1] = sqrt(variance(E, maxsteps+1));
    out[2] = intmean(jj, maxsteps)/N;
    out[3] = variance(E, maxsteps+1) / (T*T);
}

void refw_initialize_box(double L, double Lz, double *X) { initializeBox(L, Lz, N, X); }

/* initializeWalls reads one uninitialised malloc word per array when M*M is odd
 * (SMC.c:481-485); two zeroed chunks of the same size are freed just before so that the
 * reference's mallocs get them back (glibc tcache) and that word is 0.0, the value SURVEY 8a
 * row W recorded.  The wall file goes to /dev/null (the function closes it). */
int refw_initialize_walls(double x0m, double x0sigma, double ymm, double ymsigma, double *W)
{
    FILE *f = fopen("/dev/null", "w");
    if (!f) return -1;
    double *p = calloc(M*M, sizeof(double)), *q = calloc(M*M, sizeof(double));
    free(q); free(p);
    initializeWalls(x0m, x0sigma, ymm, ymsigma, W, f);
    return 0;
}

/* clusterAnalysis leaks its N x N table (SMC.c:973 vs 1044): a few calls per process only */
void refw_cluster_analysis(const double *r, double L, int *LCA) { clusterAnalysis(r, N, L, LCA); }

void refw_simple_acf(const double *H, size_t length, int k_max, double *acf)
{ simple_acf(H, length, k_max, acf); }
