/*
 * Build wrapper for oracle/_ref: compiles the REAL reference file
 * matematicose.c (vecBoxMuller, sum, mean, intmean, variance) from where it
 * lies under /root/reference -- nothing of it is copied into this repo.
 *
 * matematicose.h uses `bool`/`size_t`/`malloc`/`perror` without including the
 * standard headers that declare them (in the reference they arrive through
 * SMC.h:4-17), so the standard headers are included here first; vecBoxMuller's
 * prototype is the one SMC.h:93 gives it (it makes the C99 `inline` definition
 * an external one).  The hot-path functions of SMC.c are built separately:
 * ref_smc_prelude.c / ref_smc_wrap.c / build_ref.sh.
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>

void vecBoxMuller(double sigma, size_t length, double *A);

#include REF_MATEMATICOSE_C
