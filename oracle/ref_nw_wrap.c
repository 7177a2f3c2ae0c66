/*
 * oracle/ref_nw_wrap.c -- tail of the translation unit of oracle/_ref/libref_nw_N<n>.so.
 * TEST INFRASTRUCTURE.  The unit is streamed into gcc by oracle/build_ref.sh from the REAL
 * /root/reference/SMC_noMPI_noWall.c where it lies (nothing of it is written to disk):
 *   lines 1-10   its own standard #includes (line 11, <fftw3.h>, is absent from this image);
 *   lines 12-72  macros (the one line `#define N 32`, :18, carrying the N of this build),
 *                prototypes, struct Sim;
 *   lines 258-723  oneParticleMoves, markovProbability, initializeBox, initializeCavity,
 *                initializeWalls, forces, force, wallsForce, energy, energySingle,
 *                wallsEnergy, wallsEnergySingle, pressure, shiftSystem*, vecBoxMuller;
 *   lines 787-896  sum, intsum, dot, elforel, mean, intmean, zeros, variance, variance2;
 * then this file.  Left out: main (:74-143), sMC (:146-257: calls fft_acf), fft_acf and
 * simple_acf (:725-785: FFTW allocation calls).  Entry points below only call the real
 * functions, with libc's own srand()/rand().
 */

int refnw_N(void) { return N; }
void refnw_srand(unsigned int seed) { srand(seed); }

void refnw_initialize_box(double L, double *X) { initializeBox(L, N, X); }

/* energySingle and force of particle i (SMC_noMPI_noWall.c:599-619, 501-529) */
void refnw_single(const double *r, double L, int i, double *out)
{
    double Fx, Fy, Fz;
    out[0] = energySingle(r, L, i);
    force(r, L, i, &Fx, &Fy, &Fz);
    out[1] = Fx; out[2] = Fy; out[3] = Fz;
}

double refnw_energy(const double *r, double L) { return energy(r, L); }
double refnw_pressure(const double *r, double L) { return pressure(r, L); }

/* nsweeps calls of the real oneParticleMoves as sMC makes them (:218): jj zeroed as by calloc
 * (:186); after every sweep energy(R) goes to Es[n] and the positions to Rs[n][3N] when the
 * pointers are given.  seed_or_neg < 0 keeps libc's rand() state. */
void refnw_sweeps(long seed_or_neg, double *R, double *Rn, double L, double A, double T,
                  int nsweeps, int *jj, double *Es, double *Rs)
{
    if (seed_or_neg >= 0) srand((unsigned int)seed_or_neg);
    for (int n = 0; n < nsweeps; n++) {
        jj[n] = 0;
        oneParticleMoves(R, Rn, L, A, T, &jj[n]);
        if (Es) Es[n] = energy(R, L);
        if (Rs) memcpy(Rs + (size_t)n*3*N, R, 3*N*sizeof(double));
    }
}

void refnw_vec_box_muller(double sigma, size_t length, double *A) { vecBoxMuller(sigma, length, A); }
