/*
 * oracle/ref_smc_prelude.c -- head of the translation unit that oracle/build_ref.sh
 * pipes into gcc to build oracle/_ref/libref_smc_N<n>.so.  TEST INFRASTRUCTURE.
 *
 * The unit is, in this order (nothing of the reference is written to disk; the
 * line ranges are read from /root/reference where they lie and streamed):
 *   1. this file: the standard headers SMC.h:4-17 includes, then the reference's whole
 *      matematicose.c, included BEFORE `N` becomes a macro exactly as SMC.h:19 does
 *      (matematicose.c:258 has a local `int N`);
 *   2. SMC.h:26-121  (the macros M, N, a0, b0, LJ_CUTOFF, LCA_*, Ncx, Ncz, ...; struct Sim;
 *      the prototypes) with the one line `#define N 108` (SMC.h:29) carrying the N of this
 *      build -- the reference fixes N at compile time, SURVEY 8c item 2;
 *   3. SMC.c:269-1049 (oneParticleMoves, initializeBox, initializeWalls, shiftSystem*,
 *      boundsCheck, energySingle, forceSingle, energy, forces, pressure, wallsEnergySingle,
 *      wallsForce, wallsEnergy, wallsPressure, localDensityAndMobility(_nonuniz),
 *      clusterAnalysis) and SMC.c:1094-1169 (simple_acf, variance_corr, createZRange);
 *   4. oracle/ref_smc_wrap.c: exported entry points around those functions.
 * What is NOT in the unit: SMC.h:18,20 (<fftw3.h>, misccose.c: absent from this image and from
 * the reference tree), sMC (SMC.c:21-267: needs both, seeds with time(NULL), writes six CSVs)
 * and fft_acf (SMC.c:1055-1093: FFTW).  No stand-in for any of them is written: the
 * sliced functions need only libm and libc.
 */
#include <stdlib.h>
#include <stdio.h>
#include <stdbool.h>
#include <errno.h>
#include <signal.h>
#include <stdint.h>
#include <math.h>
#include <time.h>
#include <string.h>
#include <sys/types.h>
#include <sys/stat.h>
#include <unistd.h>
#include <limits.h>

void vecBoxMuller(double sigma, size_t length, double *A);

#include REF_MATEMATICOSE_C
