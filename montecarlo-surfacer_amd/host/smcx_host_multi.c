/*
 * smcx_host_multi.c -- the sMC driver for the GPUs of one node, in plain C.
 *
 * The reference's parallel model is independent replica chains, one per MPI rank, that share R0
 * and W, differ in their seed, write their own files and never talk (SMC.c:40 "different for each
 * process", :43, :66-95).  Here the replicas are dealt to the devices in contiguous blocks, one
 * handle and one host thread per device; seeds follow the GLOBAL replica index
 * (smcx_params.first_replica), so results do not depend on the device count.  Nothing is exchanged
 * while sampling.  The ONE exchange is the final observable gather: every device packs its
 * per-replica records (smcx_export_observables_device) and an RCCL all-gather over xGMI
 * (ncclCommInitAll + ncclAllGather inside one group call, one communicator rank per device) puts
 * every record on every device; device 0's copy is read back and reduced on the host.  (8 + Ncz)
 * doubles per replica, 1.3 MB for 4096 replicas: latency-bound, the xGMI links never bind.
 */
#define __HIP_PLATFORM_AMD__ 1
#include "../../include/smcx_host.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char t_err[256];
static char g_multi_err[256];

const char *smcx_host_multi_error(void) { return g_multi_err; }

typedef struct shard {
    /* in */
    smcx_params p;
    const double *W, *R0;
    int maxsteps, gather_lapse, eqsteps;
    size_t width;          /* doubles per replica in the packed block */
    int pad_nrep;          /* replicas per device in the gather (the largest shard) */
    double *Rfinal;        /* host: this shard's slice of the caller's [nrep_total][3N] */
    /* out */
    int rc;
    char err[256];
    smcx_handle *h;
    double *d_send;        /* device: pad_nrep * width doubles, this shard's packed records first */
    double kernel_ms;
    /* sums over this shard's replicas of the quantities smcx_host_sMC averages */
    double P, dP, tau, cv, l1, l2[16], l3[16];
    int lca_analyses;
} shard;

static void *run_shard(void *arg)
{
    shard *s = (shard *)arg;
    const smcx_params *p = &s->p;
    const int nrep = p->nrep, N = p->N;
    int rc = smcx_create(p, &s->h);
    do {
        if (rc != SMCX_OK) break;
        rc = smcx_upload(s->h, s->R0, 0, s->W, NULL);
        if (rc != SMCX_OK) break;
        rc = smcx_run(s->h, s->eqsteps, s->maxsteps, s->gather_lapse);
        if (rc != SMCX_OK) break;
        /* the packed observable block of this device, padded to the common size, in ITS memory */
        if (hipSetDevice(p->device) != hipSuccess ||
            hipMalloc((void **)&s->d_send, (size_t)s->pad_nrep * s->width * sizeof(double)) != hipSuccess ||
            hipMemset(s->d_send, 0, (size_t)s->pad_nrep * s->width * sizeof(double)) != hipSuccess) {
            rc = SMCX_ERR_HIP;
            snprintf(s->err, sizeof(s->err), "device %d: hipMalloc of the gather buffer failed", p->device);
            break;
        }
        if (smcx_obs_device_bytes(s->h) != (size_t)nrep * s->width * sizeof(double)) { rc = SMCX_ERR_STATE; break; }
        rc = smcx_export_observables_device(s->h, s->d_send, (size_t)nrep * s->width * sizeof(double));
        if (rc != SMCX_OK) break;
        rc = smcx_download_positions(s->h, s->Rfinal);
        if (rc != SMCX_OK) break;
        int launches = 0;
        smcx_last_kernel_ms(s->h, &s->kernel_ms, &launches);
        /* the optional rows of struct Sim: per-shard sums (the caller divides by the replica total) */
        if (p->flags & SMCX_FLAG_CLUSTERS) {
            uint64_t *c = (uint64_t *)calloc((size_t)nrep * 33, sizeof(uint64_t));
            if (!c) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_cluster_counts(s->h, c, c + nrep, c + (size_t)nrep * 17, NULL, &s->lca_analyses);
            for (int r = 0; rc == SMCX_OK && r < nrep; r++) {
                s->l1 += (double)c[r];
                for (int v = 0; v < 16; v++) {
                    s->l2[v] += (double)c[nrep + (size_t)r * 16 + v];
                    s->l3[v] += (double)c[(size_t)nrep * 17 + (size_t)r * 16 + v];
                }
            }
            free(c);
            if (rc != SMCX_OK) break;
        }
        if (p->flags & SMCX_FLAG_PRESSURE) { /* SMC.c:207-208, 246-247 with the reference's indexing */
            const int gather_steps = s->maxsteps / s->gather_lapse;
            int ng = 0;
            double *Ps = (double *)calloc((size_t)nrep * (gather_steps + 1), sizeof(double));
            if (!Ps) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_pressure_series(s->h, Ps, &ng);
            if (rc == SMCX_OK && gather_steps > 0) {
                const double rho = N / (p->L * p->L * p->Lz);
                for (int r = 0; r < nrep; r++) {
                    double sum = 0, sum2 = 0;
                    for (int k = 0; k < gather_steps; k++) {
                        const double v = (k >= 1 && k - 1 < ng ? Ps[(size_t)r * ng + (k - 1)] : 0.0) + rho * p->T;
                        sum += v; sum2 += v * v;
                    }
                    const double mean = sum / gather_steps, var = sum2 / gather_steps - mean * mean;
                    s->P += mean;
                    s->dP += sqrt(var > 0 ? var : 0);
                }
            }
            free(Ps);
            if (rc != SMCX_OK) break;
        }
        if (p->flags & SMCX_FLAG_SERIES) { /* SMC.c:234-235, 249-250 */
            double *tc = (double *)calloc(2 * (size_t)nrep, sizeof(double));
            int keff = 0;
            if (!tc) { rc = SMCX_ERR_NOMEM; break; }
            rc = smcx_acf(s->h, 2500000 /* KMAX, SMC.h:61 */, NULL, &keff, tc, tc + nrep);
            for (int r = 0; rc == SMCX_OK && r < nrep; r++) { s->tau += tc[r]; s->cv += tc[nrep + r]; }
            free(tc);
        }
    } while (0);
    if (rc != SMCX_OK && !s->err[0])
        snprintf(s->err, sizeof(s->err), "device %d: %s (%s)", p->device, smcx_strerror(rc), smcx_last_error_string(s->h));
    s->rc = rc;
    return NULL;
}

/* the final observable gather: one communicator rank per device of this process, one grouped all-gather */
static int rccl_all_gather(shard *sh, int ndev, size_t count, double **recv)
{
    int rc = SMCX_OK;
    ncclComm_t *comm = (ncclComm_t *)calloc(ndev, sizeof(ncclComm_t));
    hipStream_t *st = (hipStream_t *)calloc(ndev, sizeof(hipStream_t));
    int *devs = (int *)calloc(ndev, sizeof(int));
    int have_comm = 0;
    if (!comm || !st || !devs) { free(comm); free(st); free(devs); return SMCX_ERR_NOMEM; }
    for (int d = 0; d < ndev; d++) devs[d] = sh[d].p.device;
    do {
        ncclResult_t nr = ncclCommInitAll(comm, ndev, devs);
        if (nr != ncclSuccess) {
            snprintf(t_err, sizeof(t_err), "ncclCommInitAll(%d devices): %s", ndev, ncclGetErrorString(nr));
            rc = SMCX_ERR_RCCL;
            break;
        }
        have_comm = 1;
        for (int d = 0; d < ndev && rc == SMCX_OK; d++) {
            if (hipSetDevice(devs[d]) != hipSuccess || hipStreamCreate(&st[d]) != hipSuccess ||
                hipMalloc((void **)&recv[d], (size_t)ndev * count * sizeof(double)) != hipSuccess) {
                snprintf(t_err, sizeof(t_err), "device %d: stream / receive buffer of the gather", devs[d]);
                rc = SMCX_ERR_HIP;
            }
        }
        if (rc != SMCX_OK) break;
        nr = ncclGroupStart();
        for (int d = 0; d < ndev && nr == ncclSuccess; d++)
            nr = ncclAllGather(sh[d].d_send, recv[d], count, ncclDouble, comm[d], st[d]);
        if (nr == ncclSuccess) nr = ncclGroupEnd();
        else ncclGroupEnd();
        if (nr != ncclSuccess) {
            snprintf(t_err, sizeof(t_err), "ncclAllGather: %s", ncclGetErrorString(nr));
            rc = SMCX_ERR_RCCL;
            break;
        }
        for (int d = 0; d < ndev; d++) {
            if (hipSetDevice(devs[d]) != hipSuccess || hipStreamSynchronize(st[d]) != hipSuccess) {
                snprintf(t_err, sizeof(t_err), "device %d: the gather did not complete", devs[d]);
                rc = SMCX_ERR_RCCL;
                break;
            }
        }
    } while (0);
    for (int d = 0; d < ndev; d++) {
        if (st[d]) { hipSetDevice(devs[d]); hipStreamDestroy(st[d]); }
        if (have_comm && comm[d]) ncclCommDestroy(comm[d]);
    }
    free(comm); free(st); free(devs);
    return rc;
}

int smcx_host_sMC_multi(const smcx_params *p, int ndev, const int *devices, const double *W, const double *R0,
                        int maxsteps, int gather_lapse, int eqsteps, smcx_sim *out)
{
    if (!p || !R0 || !out || ndev < 1 || p->nrep < ndev) return SMCX_ERR_PARAM;
    memset(out, 0, sizeof(*out));
    g_multi_err[0] = 0;
    t_err[0] = 0;
    int have = 0;
    int rc = smcx_device_count(&have);
    if (rc != SMCX_OK) return rc;
    for (int d = 0; d < ndev; d++) {
        const int dev = devices ? devices[d] : d;
        if (dev < 0 || dev >= have) return SMCX_ERR_PARAM;
    }
    const int total = p->nrep, N = p->N, Ncz = p->Ncz;
    const size_t width = SMCX_OBS_RECORD_DOUBLES + (size_t)Ncz;
    const int base = total / ndev, extra = total % ndev, pad = base + (extra ? 1 : 0);
    const char *mode = getenv("SMCX_HOST_GATHER"); /* "host": concatenate through host memory instead of RCCL */
    const int use_rccl = !(mode && strcmp(mode, "host") == 0);

    shard *sh = (shard *)calloc(ndev, sizeof(shard));
    pthread_t *th = (pthread_t *)calloc(ndev, sizeof(pthread_t));
    double **recv = (double **)calloc(ndev, sizeof(double *));
    double *all = (double *)malloc((size_t)ndev * pad * width * sizeof(double));
    out->Rfinal = (double *)malloc((size_t)total * 3 * N * sizeof(double));
    out->rep_E = (double *)calloc(total, sizeof(double));
    out->rep_dE = (double *)calloc(total, sizeof(double));
    out->rep_acceptance = (double *)calloc(total, sizeof(double));
    out->zprofile = (double *)calloc(Ncz, sizeof(double));
    if (!sh || !th || !recv || !all || !out->Rfinal || !out->rep_E || !out->rep_dE || !out->rep_acceptance || !out->zprofile)
        rc = SMCX_ERR_NOMEM;
    int started = 0;
    if (rc == SMCX_OK) {
        int first = 0;
        for (int d = 0; d < ndev; d++) { /* contiguous blocks, the first `extra` one replica longer */
            shard *s = &sh[d];
            s->p = *p;
            s->p.device = devices ? devices[d] : d;
            s->p.nrep = base + (d < extra ? 1 : 0);
            s->p.first_replica = p->first_replica + (uint32_t)first;
            s->W = W; s->R0 = R0;
            s->maxsteps = maxsteps; s->gather_lapse = gather_lapse; s->eqsteps = eqsteps;
            s->width = width; s->pad_nrep = pad;
            s->Rfinal = out->Rfinal + (size_t)first * 3 * N;
            first += s->p.nrep;
        }
        for (int d = 0; d < ndev; d++) {
            if (pthread_create(&th[d], NULL, run_shard, &sh[d]) != 0) { rc = SMCX_ERR_STATE; break; }
            started++;
        }
        for (int d = 0; d < started; d++) pthread_join(th[d], NULL);
        for (int d = 0; d < started && rc == SMCX_OK; d++)
            if (sh[d].rc != SMCX_OK) { rc = sh[d].rc; snprintf(g_multi_err, sizeof(g_multi_err), "%s", sh[d].err); }
    }
    if (rc == SMCX_OK) {
        const size_t count = (size_t)pad * width;
        if (use_rccl) {
            rc = rccl_all_gather(sh, ndev, count, recv);
            if (rc == SMCX_OK && (hipSetDevice(sh[0].p.device) != hipSuccess ||
                                  hipMemcpy(all, recv[0], (size_t)ndev * count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) {
                snprintf(t_err, sizeof(t_err), "reading the gathered records back from device %d", sh[0].p.device);
                rc = SMCX_ERR_HIP;
            }
        } else {
            for (int d = 0; d < ndev && rc == SMCX_OK; d++)
                if (hipSetDevice(sh[d].p.device) != hipSuccess ||
                    hipMemcpy(all + (size_t)d * count, sh[d].d_send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
                    rc = SMCX_ERR_HIP;
        }
        if (rc != SMCX_OK && !g_multi_err[0]) snprintf(g_multi_err, sizeof(g_multi_err), "%s", t_err);
    }
    if (rc == SMCX_OK) {
        /* records {accepted, nsamp, sumE, sumE2, E_last, therm_accepted, gathers, oob} then zhist, per shard:
         * the reductions of smcx_observables / smcx_therm_acceptance (SMC.c:244-248, 124) */
        out->nrep = total; out->N = N; out->Ncz = Ncz;
        double gsum = 0;
        int g = 0;
        for (int d = 0; d < ndev; d++) {
            const double *blk = all + (size_t)d * pad * width;
            const int n = sh[d].p.nrep;
            const double *zh = blk + (size_t)n * SMCX_OBS_RECORD_DOUBLES;
            for (int r = 0; r < n; r++, g++) {
                const double *o = blk + (size_t)r * SMCX_OBS_RECORD_DOUBLES;
                const double len = o[1] > 0 ? o[1] : 1.0, mean = o[2] / len;
                out->rep_acceptance[g] = maxsteps > 0 ? (o[0] / maxsteps) / N : 0.0;
                out->rep_E[g] = mean;
                out->rep_dE[g] = sqrt(o[3] / len - mean * mean);
                out->therm_acceptance += (eqsteps > 0 ? (o[5] / eqsteps) / N : 0.0) / total;
                gsum += o[6];
                for (int k = 0; k < Ncz; k++) out->zprofile[k] += zh[(size_t)r * Ncz + k];
            }
        }
        for (int r = 0; r < total; r++) {
            out->E += out->rep_E[r] / total;
            out->dE += out->rep_dE[r] / total;
            out->acceptance_ratio += out->rep_acceptance[r] / total;
        }
        if (gsum > 0)
            for (int k = 0; k < Ncz; k++) out->zprofile[k] /= gsum;
        for (int d = 0; d < ndev; d++) {
            out->P += sh[d].P / total; out->dP += sh[d].dP / total;
            out->tau += sh[d].tau / total; out->cv += sh[d].cv / total;
            if (sh[d].kernel_ms > out->kernel_ms) out->kernel_ms = sh[d].kernel_ms; /* the devices run side by side */
            out->lca_analyses = sh[d].lca_analyses;
        }
        if (out->lca_analyses > 0) {
            const double w = 1.0 / ((double)total * out->lca_analyses);
            for (int d = 0; d < ndev; d++) {
                out->l1 += w * sh[d].l1;
                for (int v = 0; v < 16; v++) { out->l2[v] += w * sh[d].l2[v]; out->l3[v] += w * sh[d].l3[v]; }
            }
        }
        if (out->kernel_ms > 0)
            out->pair_evals_per_s = (double)total * (eqsteps + maxsteps) * 2.0 * N * (N - 1.0) / (out->kernel_ms * 1e-3);
    }
    for (int d = 0; sh && d < ndev; d++) {
        if (sh[d].d_send || (recv && recv[d])) hipSetDevice(sh[d].p.device);
        if (sh[d].d_send) hipFree(sh[d].d_send);
        if (recv && recv[d]) hipFree(recv[d]);
        if (sh[d].h) smcx_destroy(sh[d].h);
    }
    free(sh); free(th); free(recv); free(all);
    if (rc != SMCX_OK) smcx_host_sim_free(out);
    return rc;
}
