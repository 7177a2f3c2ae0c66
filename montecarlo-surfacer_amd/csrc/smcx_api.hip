// smcx_api.hip -- the C ABI of include/smcx.h on top of the gfx950 kernels.
// Host-side orchestration only: no physics is computed on the CPU here, and
// there is no CPU fallback -- without a HIP device every compute entry fails
// with SMCX_ERR_NODEVICE / SMCX_ERR_HIP.
#include "../../include/smcx.h"
#include "smcx_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <algorithm>
#include <vector>

// qwords per replica in the clock rows (the two-team stamps variant keeps 16 phase sums per team there)
#ifdef SMCX_TT_STAMPS
constexpr size_t CLK_COLS = 256;
#else
constexpr size_t CLK_COLS = 4;
#endif

using namespace smcx;

namespace {

thread_local std::string g_last_error;

struct Handle {
    smcx_params p;
    DevCtx c;
    int S = 0, WPR = 0;
    KernelPlan plan;        // the sweep kernel of this handle, fixed at smcx_create (plan_kernel)
    int chunk = 1;          // sweeps of random numbers per pre-pass launch
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *d_save = nullptr;   // [nrep] energy at smcx_run entry
    double *d_tmp = nullptr;    // [nrep] scratch (energies)
    bool uploaded = false;
    int last_maxsteps = 0;
    int last_eqsteps = 0;
    double last_ms = 0.0;       // whole run
    double last_sweep_ms = 0.0; // sweep kernels only
    int last_launches = 0;
    SweepTimer timer;            // start/stop events around every launch of the sweep kernel
    int last_gathers = 0;
    // cluster analysis (SMCX_FLAG_CLUSTERS or on demand)
    unsigned *lca_bits = nullptr;          // [lca_batch][lca_words]
    unsigned long long *lca_counts = nullptr; // [nrep + 1][LCA_COUNTS]; the last row is scratch
    long lca_words = 0;
    int lca_batch = 0;
    int lca_analyses = 0;
    std::string err;
};

} // namespace

struct smcx_handle {
    Handle h;
};

#define HIPCHK(hh, call)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(e_);               \
            g_last_error = m_;                                                                \
            if (hh) (hh)->err = m_;                                                           \
            return SMCX_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// ---- glibc srand(): the state after seeding and its 310 discarded outputs ----------
// (SURVEY.md 8a row R).  Output layout: 31 state words oldest first, then the
// count of generated-but-unconsumed outputs (0 after seeding).
static void seed_state(uint32_t seed, uint32_t out[32])
{
    uint32_t s[31];
    if (seed == 0) seed = 1;
    int32_t w = (int32_t)seed; // glibc keeps the running word in int32_t
    s[0] = seed;
    for (int i = 1; i < 31; i++) {
        const int64_t hi = w / 127773, lo = w % 127773;
        w = (int32_t)(16807 * lo - 2836 * hi);
        if (w < 0) w += 2147483647;
        s[i] = (uint32_t)w;
    }
    int f = 3, r = 0;
    for (int i = 0; i < 310; i++) {
        s[f] += s[r];
        f = (f + 1) % 31;
        r = (r + 1) % 31;
    }
    // the word at f is the oldest (it is the next to be overwritten)
    for (int j = 0; j < 31; j++) out[j] = s[(f + j) % 31];
    out[31] = 0;
}

extern "C" void smcx_rng_seed(uint32_t *rng, uint32_t seed) { seed_state(seed, rng); }

// the rand() state (31 words, oldest first) advanced by Q blocks of 31 outputs, as a matrix over Z/2^32 stored by
// columns: out[k * 31 + j] = coefficient of old word k in new word j.  One block: new[j] = old[j] + (j < 3 ?
// old[j + 28] : new[j - 3])  (r[i] = r[i-31] + r[i-3]); rng_prepass_kernel's waves start a multiple of Q apart.
static void rng_jump_matrix(int Q, std::vector<uint32_t> &out)
{
    out.assign(31 * 31, 0u);
    for (int k = 0; k < 31; k++) {
        uint32_t v[31] = {0};
        v[k] = 1u;
        for (int b = 0; b < Q; b++) {
            uint32_t n[31];
            for (int j = 0; j < 31; j++) n[j] = v[j] + (j < 3 ? v[j + 28] : n[j - 3]);
            std::memcpy(v, n, sizeof(v));
        }
        for (int j = 0; j < 31; j++) out[k * 31 + j] = v[j];
    }
}

extern "C" void smcx_default_params(smcx_params *p, int32_t N, int32_t nrep)
{
    std::memset(p, 0, sizeof(*p));
    p->N = N;
    p->M = 3;                        // SMC.h:26
    p->nrep = nrep;
    p->device = 0;
    p->L = 33.0;                     // main.c:41-44
    p->Lz = 240.0;
    p->T = 1.1;                      // main.c:18
    p->A = 1.1;                      // main.c:48-51, gamma = 1
    p->cutoff = 3.0;                 // SMC.h:38
    p->a0 = 5.960464477539063e-9;    // SMC.h:32
    p->b0 = 2.44140625e-5;           // SMC.h:33
    p->Ncx = 33;                     // SMC.h:53
    p->Ncz = 33;                     // SMC.h:55
    p->flags = SMCX_FLAGS_REFERENCE;
    p->base_seed = 12345;
    p->first_replica = 0;
    p->lca_time = 10;                // SMC.h:48
    p->lca_cutoff = 1.7;             // SMC.h:50
}

extern "C" const char *smcx_strerror(int status)
{
    switch (status) {
    case SMCX_OK: return "ok";
    case SMCX_ERR_PARAM: return "invalid parameter";
    case SMCX_ERR_HIP: return "HIP runtime error";
    case SMCX_ERR_STATE: return "call order violated";
    case SMCX_ERR_NOMEM: return "out of memory";
    case SMCX_ERR_UNSUPPORTED: return "unsupported configuration";
    case SMCX_ERR_NODEVICE: return "no HIP device";
    case SMCX_ERR_RCCL: return "RCCL collective failed";
    default: return "unknown status";
    }
}

extern "C" const char *smcx_last_error_string(const smcx_handle *h)
{
    if (h && !h->h.err.empty()) return h->h.err.c_str();
    return g_last_error.c_str();
}

extern "C" int smcx_device_count(int *count)
{
    if (!count) return SMCX_ERR_PARAM;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return SMCX_ERR_NODEVICE;
    }
    *count = n;
    return n > 0 ? SMCX_OK : SMCX_ERR_NODEVICE;
}

static int validate(const smcx_params *p)
{
    if (!p) return SMCX_ERR_PARAM;
    if (p->N < 2 || (p->N & 1)) return SMCX_ERR_PARAM; // odd 3N: vecBoxMuller leaves displ[3N-1] unset
    if (p->nrep < 1) return SMCX_ERR_PARAM;
    if (!(p->L > 0) || !(p->Lz > 0) || !(p->T > 0) || !(p->A > 0) || !(p->cutoff > 0))
        return SMCX_ERR_PARAM;
    if (p->Ncx < 1 || p->Ncz < 1 || p->Ncx > 255 || p->Ncz > 255) return SMCX_ERR_PARAM;
    if (p->tune_kernel < 0 || p->tune_kernel > SMCX_KERNEL_MT) return SMCX_ERR_PARAM;
    if (p->tune_resort < 0 || p->tune_resort > 1024) return SMCX_ERR_PARAM;
    if (p->flags & SMCX_FLAG_WALLS) {
        if (p->M < 1) return SMCX_ERR_PARAM;
        if (p->M * p->M + 1 > 30) return SMCX_ERR_UNSUPPORTED;
    }
    if ((p->flags & SMCX_FLAG_CLUSTERS) && (p->lca_time < 1 || !(p->lca_cutoff > 0)))
        return SMCX_ERR_PARAM;
    return SMCX_OK;
}

// particles per lane (S) and wavefronts per replica (WPR): the smallest capacity 64*WPR*S
// that holds N from a table of the geometries measured fastest on MI355X.
// The measurement switches of the sweep kernels are smcx_params fields (tune_kernel, tune_resort, tune_slots, tune_waves).
// The SMCX_* environment variables of rounds 1-2 exist only in VARIANT builds of this library (make VARIANT=x: -DSMCX_VARIANT,
// libsmcx_x.so, loaded by name for an A/B session) and there only under SMCX_ALLOW_ENV_TUNING=1.  The product library and the
// diagnostic build never read them: a set variable makes smcx_create fail instead of silently changing the kernel a caller gets.
static const char *const k_tune_env[] = {"SMCX_MX", "SMCX_MI", "SMCX_MA", "SMCX_MB", "SMCX_MC", "SMCX_MCW", "SMCX_MZ",
                                         "SMCX_RESORT", "SMCX_ZSORT_TPB", "SMCX_LEAD", "SMCX_NO_LEAD", "SMCX_NO_WINDOWS"};

static int make_tune(const smcx_params *p, Tune *t)
{
    *t = Tune();
    t->kernel = p->tune_kernel;
    // sweeps per z sort of the one-wavefront z-ordered kernels: 2 by default (the sort saved outweighs the wider group
    // ranges of the second sweep: 9.75 against 9.90 ms per step at config 3, profiles/r04_config2_forms.txt; 1 in round 3)
    t->resort = p->tune_resort > 0 ? p->tune_resort : 2;
    if (const char *e = getenv("SMCX_CHECK_MB")) t->check_mb = atoi(e); // read by the diagnostic build only (ma_cap)
    bool allowed = false;
#ifdef SMCX_VARIANT
    const char *allow = getenv("SMCX_ALLOW_ENV_TUNING");
    allowed = allow && allow[0] == '1';
#endif
    if (!allowed) {
        for (const char *name : k_tune_env)
            if (getenv(name)) {
                g_last_error = std::string("environment variable ") + name +
                               " is set: this library does not read measurement switches from the environment (use "
                               "smcx_params.tune_kernel / tune_resort; only VARIANT builds honour SMCX_ALLOW_ENV_TUNING=1)";
                return SMCX_ERR_PARAM;
            }
        return SMCX_OK;
    }
#ifdef SMCX_VARIANT
    auto off = [](const char *n) { const char *e = getenv(n); return e && e[0] == '0'; };
    if (t->kernel == 0) {
        if (const char *e = getenv("SMCX_MX")) t->kernel = (e[0] == '0') ? 1 : 2;
    }
    if (t->kernel == 0 || t->kernel == 2) { // each switch steps one rung down the ladder of plan_kernel
        int cap = FORM_MC;
        if (off("SMCX_MC")) cap = FORM_MB;
        if (off("SMCX_MB")) cap = FORM_MA;
        if (off("SMCX_MA")) cap = FORM_MI;
        if (off("SMCX_MI")) cap = FORM_MX;
        if (cap != FORM_MC) t->kernel = cap;
    }
    if (getenv("SMCX_NO_LEAD")) t->lead = 0;
    if (getenv("SMCX_LEAD")) t->lead = 1;
    if (const char *e = getenv("SMCX_MZ")) t->mz = (e[0] != '0');
    if (const char *e = getenv("SMCX_RESORT")) { const int v = atoi(e); if (v > 0) t->resort = v; }
    if (getenv("SMCX_NO_WINDOWS")) t->windows = 0;
    if (const char *e = getenv("SMCX_ZSORT_TPB")) { const int v = atoi(e); if (v == 128 || v == 256 || v == 512 || v == 1024) t->zsort_tpb = v; }
#endif
    return SMCX_OK;
}

static bool plan_for(const smcx_params *p, int s, int w, const Tune &t, KernelPlan *pl)
{
    return plan_kernel(p->N, (p->flags & SMCX_FLAG_WALLS) ? p->M * p->M : 0, p->L, p->Lz, p->cutoff * p->cutoff, s, w, t, pl);
}

static int choose_geometry(const smcx_params *p, const Tune &t, int *S, int *WPR)
{
    KernelPlan pl;
    if (p->tune_slots > 0 || p->tune_waves > 0) {
        int s = p->tune_slots > 0 ? p->tune_slots : 16;
        int w = p->tune_waves > 0 ? p->tune_waves : 1;
        if (plan_for(p, s, w, t, &pl) && pl.form == FORM_MT) { *S = s; *WPR = w; return SMCX_OK; } // (64 x 8: the one two-team form built)
        if (!geometry_supported(s, w) || (long)s * w * 64 < p->N) return SMCX_ERR_UNSUPPORTED;
        if (p->tune_kernel == 1 && !fp64_supported(s, w)) return SMCX_ERR_UNSUPPORTED;
        if (p->tune_kernel >= 2 && !mx_supported(s, w)) return SMCX_ERR_UNSUPPORTED;
        *S = s; *WPR = w;
        return SMCX_OK;
    }
    // Rule fitted to measurements on MI355X (profiles/r01_geometry_N*.log, r01_screened_kernel.log).
    // Screened kernel (2 VGPRs per particle, S >= 16): as few wavefronts per replica as possible --
    // 64 particles per lane, N/4096 wavefronts -- because every extra wavefront repeats the
    // sequential part of a move and adds a barrier.  fp64 kernels (S < 16, or asked for): 16
    // particles per lane (96 VGPRs of positions, 3 waves per SIMD) and N/1024 wavefronts, 32 per
    // lane beyond 8 wavefronts.  With few replicas, halve S and double the wavefronts until the
    // chip has about two waves per SIMD to work on.
    auto pow2_at_least = [](long v) { int r = 1; while (r < v) r *= 2; return r; };
    int s, w;
    if (p->tune_kernel == SMCX_KERNEL_MT) { // asked for by name: sweep_kernel_mt64x8 (8192 < N <= 16384), the one two-team form built
        if (plan_for(p, 64, 8, t, &pl) && pl.form == FORM_MT) { *S = 64; *WPR = 8; return SMCX_OK; }
        return SMCX_ERR_UNSUPPORTED;        // (smcx.h: any other N; round 3's 16 x 2 and 32 x 16 forms were retired in round 4)
    }
    const bool fp64_only = (p->tune_kernel == 1);
    if (!fp64_only && (p->N > 512 || p->tune_kernel >= 2)) {
        s = pow2_at_least((p->N + 63) / 64); w = 1;
        if (s < 16) s = 16;
        while (s > 64) { s /= 2; w *= 2; }
        // one wavefront per replica has hand-scheduled kernels (sweep_kernel_mc*/mb64/ma*) that beat any split over
        // several wavefronts even with few replicas (N = 2048: 3.1-3.6 ms per sweep against 5.2-9.0 for 16 x 2 at
        // 128..1024 replicas, tools/probes/geom_rule.py); the split stays for boxes those kernels do not serve
        bool one_wave = (w == 1) && plan_for(p, s, 1, t, &pl) && pl.form >= FORM_MI;
        if (w == 2 && s == 64 && plan_for(p, 32, 4, t, &pl) && pl.form == FORM_MC) { // 4096 < N <= 8192: 32 x 4, z-ordered
            s = 32; w = 4; one_wave = true;
        }
        if (w > 1 && s * w == 256 && plan_for(p, s, w, t, &pl) && pl.form == FORM_MC) { // its several-wave forms: 64 x 4, or
            one_wave = true;                                                     // 32 x 8 while the chip has room
            if ((long)p->nrep * 8 <= 2048) { s = 32; w = 8; }                    // (44.6 against 47.3 ms per sweep at 256)
        }
        // latency-bound shares (few replicas: at most two wavefronts per SIMD): the two-team form of the same kernel --
        // probe A and probe B on different wavefronts, one exchange per move (measured, tools/probes/tt_probe.py:
        // N = 16384 x 256 replicas 28.2 ms per sweep against 36.4 for 32 x 8; N = 1024 x 1024 1.59 against 1.68; with twice
        // the replicas the one-probe-after-the-other forms win: 50 against 57 ms, 1.89 against 2.24)
        if (p->tune_kernel == 0 || p->tune_kernel == SMCX_KERNEL_SCREENED) {
            if (p->N > 8192 && (long)p->nrep * 8 <= 2048 && plan_for(p, 64, 8, t, &pl) && pl.form == FORM_MT) { s = 64; w = 8; one_wave = true; }
            // (N <= 1024 with at most 1024 replicas ran the two-team form 16 x 2 in round 3; round 4's one-wavefront kernel
            // with both probes in one pass and the cells' positions in LDS, sweep_kernel_ml16, is faster: 1.27 against 1.31 ms
            // per sweep at 1024 replicas, 1.26 against 1.36 at 512 -- profiles/r04_config2_forms.txt; the 16 x 2 kernel was
            // retired with it: tune_kernel = SMCX_KERNEL_MT at N <= 8192 answers SMCX_ERR_UNSUPPORTED, and tune_slots = 16 with
            // tune_waves = 2 selects the compiled several-wavefront kernel sweep_kernel_mx<16, 2>)
        }
        while (!one_wave && (long)p->nrep * w < 2048 && s > 16 && w < 8 && geometry_supported(s / 2, w * 2)) { s /= 2; w *= 2; }
    } else {
        if (p->N <= 1024) { s = pow2_at_least((p->N + 63) / 64); w = 1; }
        else {
            s = 16; w = pow2_at_least((p->N + 1023) / 1024);
            if (w > 8) { s = 32; w = pow2_at_least((p->N + 2047) / 2048); }
        }
        while ((long)p->nrep * w < 2048 && s > 4 && w < 8 && geometry_supported(s / 2, w * 2)) { s /= 2; w *= 2; }
    }
    if (plan_for(p, s, w, t, &pl) && pl.form == FORM_MT) { *S = s; *WPR = w; return SMCX_OK; }
    if (geometry_supported(s, w) && (long)s * w * 64 >= p->N) { *S = s; *WPR = w; return SMCX_OK; }
    static const int cand[][2] = {{1, 1}, {2, 1}, {4, 1}, {8, 1}, {8, 2}, {16, 2}, {16, 4},
                                  {32, 4}, {32, 8}, {32, 16}};
    for (auto &g : cand) {
        if ((long)g[0] * g[1] * 64 >= p->N && geometry_supported(g[0], g[1])) {
            *S = g[0]; *WPR = g[1];
            return SMCX_OK;
        }
    }
    return SMCX_ERR_UNSUPPORTED;
}

static void fill_ctx(Handle &h)
{
    const smcx_params &p = h.p;
    DevCtx &c = h.c;
    c.N = p.N; c.M = p.M; c.M2 = p.M * p.M; c.nrep = p.nrep;
    c.Ncx = p.Ncx; c.Ncz = p.Ncz;
    c.flags = p.flags;
    c.L = p.L; c.invL = 1.0 / p.L; c.Lz = p.Lz; c.invLz = 1.0 / p.Lz; c.halfLz = p.Lz / 2;
    c.T = p.T; c.invT = 1.0 / p.T;
    c.cutoff2 = p.cutoff * p.cutoff;
    c.a0 = p.a0; c.b0 = p.b0;
    c.c3NT2 = 3 * p.N * p.T / 2; // SMC.c:211
    c.chunk = h.chunk;
    c.rawStride = 4L * p.N + 1 + 31 + 3; // one sweep of outputs + a partial block, padded
}

extern "C" int smcx_destroy(smcx_handle *hh)
{
    if (!hh) return SMCX_OK;
    Handle &h = hh->h;
    hipSetDevice(h.p.device);
    DevCtx &c = h.c;
    hipFree(c.R); hipFree((void *)c.W); hipFree(c.rng); hipFree(c.raw); hipFree((void *)c.rngJump); hipFree(c.displ);
    hipFree(c.uni); hipFree(c.offs); hipFree(c.obs); hipFree(c.zhist); hipFree(c.Eseries);
    hipFree(c.jjseries); hipFree(c.rec); hipFree(h.d_save); hipFree(h.d_tmp);
    hipFree(c.D); hipFree(c.Mu); hipFree(c.Rbin); hipFree(c.Pseries);
    hipFree(h.lca_bits); hipFree(h.lca_counts); hipFree(c.clk); hipFree((void *)c.wtab); hipFree(c.Rs); hipFree(c.loc); hipFree(c.prio);
#ifdef SMCX_CHECK
    hipFree(c.dbg);
#endif
    for (hipEvent_t e : h.timer.evs) hipEventDestroy(e);
    if (h.ev0) hipEventDestroy(h.ev0);
    if (h.ev1) hipEventDestroy(h.ev1);
    if (h.stream) hipStreamDestroy(h.stream);
    delete hh;
    return SMCX_OK;
}

extern "C" int smcx_create(const smcx_params *p, smcx_handle **out)
{
    if (!out) return SMCX_ERR_PARAM;
    *out = nullptr;
    int rc = validate(p);
    if (rc != SMCX_OK) return rc;
    Tune tune; // (before the device is looked for: a stale measurement switch is refused on any machine)
    rc = make_tune(p, &tune);
    if (rc != SMCX_OK) return rc;
    int ndev = 0;
    rc = smcx_device_count(&ndev);
    if (rc != SMCX_OK) return rc;
    if (p->device < 0 || p->device >= ndev) return SMCX_ERR_PARAM;

    smcx_handle *hh = new (std::nothrow) smcx_handle();
    if (!hh) return SMCX_ERR_NOMEM;
    Handle &h = hh->h;
    h.p = *p;
    std::memset(&h.c, 0, sizeof(h.c));
    rc = choose_geometry(p, tune, &h.S, &h.WPR);
    if (rc == SMCX_OK && !plan_for(p, h.S, h.WPR, tune, &h.plan)) rc = SMCX_ERR_UNSUPPORTED;
    if (rc != SMCX_OK) { delete hh; return rc; }
    // few replicas of N <= 1024 through the one-wavefront kernel (at most one wavefront per SIMD): the form that keeps the
    // fp64 positions of all cells in LDS, sweep_kernel_ml16 -- nothing else would hide a candidate fetch's trip to L2
    if (h.plan.form == FORM_MC && h.plan.S == 16 && h.plan.WPR == 1 && p->nrep <= 1024 &&
        (p->tune_kernel == 0 || p->tune_kernel == SMCX_KERNEL_SCREENED)) {
        h.plan.lpos = true;
        h.plan.name = ma_kernel_name(FORM_MC, 16, -1);
    }

    // sweeps of random numbers kept on the device at once: bounded by ~6 GB
    const double per_sweep = (double)p->nrep * (32.0 * p->N + 8);
    h.chunk = (int)std::fmax(1.0, std::fmin(16.0, std::floor(6.0e9 / per_sweep)));
    fill_ctx(h);

#define CRT(call)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            g_last_error = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            smcx_destroy(hh);                                                                 \
            return e_ == hipErrorOutOfMemory ? SMCX_ERR_NOMEM : SMCX_ERR_HIP;                 \
        }                                                                                     \
    } while (0)

    DevCtx &c = h.c;
    const size_t nrep = p->nrep, N = p->N;
    CRT(hipSetDevice(p->device));
    // a replica count that is no multiple of what the device holds at once: launch groups run as windows of units (MaArgs2)
    c.granule = h.plan.zordered() ? ma_resident_replicas(h.plan, p->device) : 0;
    c.windows = (h.plan.tune.windows && c.granule > 0 && p->nrep > c.granule && p->nrep % c.granule != 0) ? 1 : 0;
    CRT(hipStreamCreate(&h.stream));
    CRT(hipEventCreate(&h.ev0));
    CRT(hipEventCreate(&h.ev1));
    CRT(hipMalloc(&c.R, nrep * 3 * N * sizeof(double)));
    CRT(hipMalloc((void **)&c.W, (size_t)(2 * c.M2 + 2) * sizeof(double))); // + a0, b0 of the plane (sweep_kernel_mi)
    CRT(hipMalloc(&c.rng, nrep * 32 * sizeof(uint32_t)));
    CRT(hipMalloc(&c.raw, nrep * (size_t)c.rawStride * sizeof(uint32_t)));
    {   // quarter of the blocks of 31 rand() outputs a sweep consumes at most (4N + 1 outputs)
        c.rngQ = (int)(((4L * p->N + 1 + 30) / 31 + 3) / 4);
        std::vector<uint32_t> J;
        rng_jump_matrix(c.rngQ, J);
        CRT(hipMalloc((void **)&c.rngJump, J.size() * sizeof(uint32_t)));
        CRT(hipMemcpy((void *)c.rngJump, J.data(), J.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    CRT(hipMalloc(&c.displ, nrep * h.chunk * 3 * N * sizeof(double)));
    CRT(hipMalloc(&c.uni, nrep * h.chunk * N * sizeof(double)));
    CRT(hipMalloc(&c.offs, nrep * h.chunk * sizeof(int)));
    CRT(hipMalloc(&c.obs, nrep * sizeof(ObsRec)));
    CRT(hipMalloc(&c.rec, nrep * h.chunk * sizeof(SweepRec)));
    CRT(hipMalloc(&c.zhist, nrep * p->Ncz * sizeof(unsigned long long)));
    CRT(hipMalloc((void **)&c.wtab, (size_t)(c.M2 + 1) * 4 * sizeof(double)));
    CRT(hipMalloc(&c.clk, nrep * CLK_COLS * sizeof(unsigned long long)));
    CRT(hipMemset(c.clk, 0, nrep * CLK_COLS * sizeof(unsigned long long)));
    // progress table of the SIMDs' wavefronts (hand-scheduled kernels): 8 XCDs x 128 (SE, SH, CU) x 4 SIMDs rows of
    // 4 words; the kernels build the index from exact-width register fields (gen_sweep_ma.py, PRIO)
    CRT(hipMalloc(&c.prio, 16384 * sizeof(unsigned)));
    CRT(hipMemset(c.prio, 0, 16384 * sizeof(unsigned)));
    if (h.plan.zordered()) {
        // cell-ordered copy of the positions and the cell of each particle (sweep_kernel_mb64 / mc*)
        CRT(hipMalloc(&c.Rs, nrep * (size_t)h.S * h.WPR * 64 * 3 * sizeof(double)));
        CRT(hipMalloc(&c.loc, nrep * N * sizeof(unsigned short)));
    }
    CRT(hipMalloc(&h.d_save, nrep * sizeof(double)));
    CRT(hipMalloc(&h.d_tmp, nrep * sizeof(double)));
    if (p->flags & SMCX_FLAG_FULL_HIST) {
        const size_t Nc = (size_t)p->Ncx * p->Ncx * p->Ncz;
        CRT(hipMalloc(&c.D, nrep * Nc * sizeof(unsigned long long)));
        CRT(hipMalloc(&c.Mu, nrep * Nc * sizeof(unsigned long long)));
        CRT(hipMalloc(&c.Rbin, nrep * N * sizeof(int)));
    }
#ifdef SMCX_CHECK
    CRT(hipMalloc(&c.dbg, 8 * sizeof(unsigned long long)));
    CRT(hipMemset(c.dbg, 0, 8 * sizeof(unsigned long long)));
#endif
    CRT(hipMemset(c.obs, 0, nrep * sizeof(ObsRec)));
    CRT(hipMemset(c.zhist, 0, nrep * p->Ncz * sizeof(unsigned long long)));
    CRT(hipMemset((void *)c.W, 0, (size_t)(2 * c.M2 + 2) * sizeof(double)));
    {
        const double plane[2] = {p->a0, p->b0};
        CRT(hipMemcpy((void *)(c.W + 2 * c.M2), plane, sizeof(plane), hipMemcpyHostToDevice));
    }
#undef CRT
    *out = hh;
    return SMCX_OK;
}

extern "C" int smcx_geometry(const smcx_handle *hh, int *slots, int *waves, int *lds_bytes)
{
    if (!hh) return SMCX_ERR_PARAM;
    if (slots) *slots = hh->h.S;
    if (waves) *waves = hh->h.WPR;
    if (lds_bytes) *lds_bytes = (int)(sizeof(RoleTable) + 2 * hh->h.WPR * 8 * 8 + 2 * 2 * 4 * 8);
    return SMCX_OK;
}

extern "C" int smcx_screen_bound(const smcx_params *p, int lds_z, double *thr, double *u2, double *to_fixed,
                                 double *zsafe)
{
    if (!p || !thr || !u2 || !to_fixed || !zsafe) return SMCX_ERR_PARAM;
    if (!(p->L > 0) || !(p->Lz > 0) || !(p->cutoff > 0)) return SMCX_ERR_PARAM;
    mx_bound_values(p->L, p->Lz, p->cutoff * p->cutoff, lds_z != 0, thr, u2, to_fixed, zsafe);
    return SMCX_OK;
}

extern "C" int smcx_screen_bound_int(const smcx_params *p, double *thr, double *u2, double *to_fixed, double *zsafe,
                                     double *uz, int32_t *neg_c, int32_t *zshift)
{
    if (!p || !thr || !u2 || !to_fixed || !zsafe || !uz || !neg_c || !zshift) return SMCX_ERR_PARAM;
    if (!(p->L > 0) || !(p->Lz > 0) || !(p->cutoff > 0)) return SMCX_ERR_PARAM;
    int nc = 0, zs = 0;
    mi_bound_values(p->L, p->Lz, p->cutoff * p->cutoff, thr, u2, to_fixed, zsafe, uz, &nc, &zs);
    *neg_c = nc; *zshift = zs;
    return zs ? SMCX_OK : SMCX_ERR_UNSUPPORTED;
}

extern "C" int smcx_screen_bound_byte(const smcx_params *p, double *to_fixed, double *zsafe, int32_t *neg_t, int32_t *reach_z)
{
    if (!p || !to_fixed || !zsafe || !neg_t || !reach_z) return SMCX_ERR_PARAM;
    if (!(p->L > 0) || !(p->Lz > 0) || !(p->cutoff > 0)) return SMCX_ERR_PARAM;
    int nt = 0, rz = 0;
    mc_bound_values(p->L, p->cutoff * p->cutoff, to_fixed, zsafe, &nt, &rz);
    *neg_t = nt; *reach_z = rz;
    return mc_box_supported(p->L, p->Lz, p->cutoff * p->cutoff) ? SMCX_OK : SMCX_ERR_UNSUPPORTED;
}

extern "C" int smcx_kernel_form(const smcx_handle *hh, int *form, char *name, int len)
{
    if (!hh) return SMCX_ERR_PARAM;
    const Handle &h = hh->h;
    if (form) *form = h.plan.form == FORM_FP64 ? 1 : 2;
    if (name && len > 0) std::snprintf(name, (size_t)len, "%s", h.plan.name);
    return SMCX_OK;
}

extern "C" int smcx_replica_granule(smcx_handle *hh, int *granule, char *note, int len)
{
    if (!hh || !granule) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    HIPCHK(&h, hipSetDevice(h.p.device));
    const int g = ma_resident_replicas(h.plan, h.p.device);
    *granule = g;
    if (note && len > 0) {
        note[0] = 0;
        if (g > 0 && h.p.nrep % g != 0) {
            const int rounds = (h.p.nrep + g - 1) / g, last = h.p.nrep - (rounds - 1) * g;
            if (rounds == 1)
                std::snprintf(note, (size_t)len,
                              "%d replicas with %s: the device could run %d of them at once; a sweep is sequential inside a replica, "
                              "so fewer replicas do not make it proportionally shorter (half of them: about 80 %% of the time).  "
                              "%d replicas per GPU use the device fully.",
                              h.p.nrep, h.plan.name, g, g);
            else if (h.c.windows) {
                // (round 5) the sweeps between two gathers run as windows of g (replica, block) units: full launches
                const int every = h.plan.WPR == 1 ? (h.plan.tune.resort > 0 ? h.plan.tune.resort : 1) : 1;
                const int nb = (10 + every - 1) / every;   // a group of 10 sweeps, the throughput runs' gather_lapse
                const long launches = ((long)h.p.nrep * nb + g - 1) / g;
                std::snprintf(note, (size_t)len,
                              "%d replicas with %s: the device runs %d of them at once and a sweep is sequential inside a replica.  The "
                              "sweeps between two gathers are cut into blocks of %d sweep(s) and launched as windows of %d (replica, "
                              "block) units, so that every launch but the last of a group is full: 10 sweeps cost %ld launches "
                              "(%d rounds of all replicas would be %d); a multiple of %d replicas per GPU has no partial launch at all.",
                              h.p.nrep, h.plan.name, g, every, g, launches, nb, nb * rounds, g);
            } else
                std::snprintf(note, (size_t)len,
                              "%d replicas with %s: the device runs %d of them at once and a sweep is sequential inside a replica, so "
                              "a sweep takes %d rounds, the last with %d replica(s) on a nearly empty chip (a lone wavefront still "
                              "needs about half the time of a full round).  Use a multiple of %d replicas per GPU: %d cost about the "
                              "same time.",
                              h.p.nrep, h.plan.name, g, rounds, last, g, rounds * g);
        }
    }
    return SMCX_OK;
}

#include "smcx_kernel_ids.h"
extern "C" int smcx_kernel_source_id(const char *kernel, char *id, int len)
{
    if (!kernel || !id || len < 17) return SMCX_ERR_PARAM;
    for (const auto &e : smcx_kernel_ids)
        if (e.prefix ? std::strncmp(kernel, e.kernel, std::strlen(e.kernel)) == 0 : std::strcmp(kernel, e.kernel) == 0) {
            std::snprintf(id, (size_t)len, "%s", e.id);
            return SMCX_OK;
        }
    return SMCX_ERR_PARAM;
}

extern "C" int smcx_upload(smcx_handle *hh, const double *R0, int r0_per_replica, const double *W,
                           const uint32_t *seeds)
{
    if (!hh || !R0) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    const smcx_params &p = h.p;
    DevCtx &c = h.c;
    if ((p.flags & SMCX_FLAG_WALLS) && !W) return SMCX_ERR_PARAM;
    HIPCHK(&h, hipSetDevice(p.device));
    const size_t nrep = p.nrep, row = 3 * (size_t)p.N;
    {   // the pair test relies on wrapped x,y (|d| <= L); z is only sanity-bounded
        const size_t cnt = (r0_per_replica ? nrep : 1) * (size_t)p.N;
        const double hx = p.L / 2 * (1 + 1e-12), hz = 4 * p.Lz;
        for (size_t i = 0; i < cnt; i++) {
            const double *q = R0 + 3 * i;
            if (!(std::fabs(q[0]) <= hx) || !(std::fabs(q[1]) <= hx) || !(std::fabs(q[2]) <= hz)) {
                h.err = "R0: x,y must be wrapped into [-L/2, L/2] and |z| <= 4 Lz";
                g_last_error = h.err;
                return SMCX_ERR_PARAM;
            }
        }
    }
    if (r0_per_replica) {
        HIPCHK(&h, hipMemcpy(c.R, R0, nrep * row * sizeof(double), hipMemcpyHostToDevice));
    } else {
        // all replicas start from the common R0 (SMC.c:43)
        HIPCHK(&h, hipMemcpy(c.R, R0, row * sizeof(double), hipMemcpyHostToDevice));
        size_t have = 1;
        while (have < nrep) {
            const size_t n = (have < nrep - have) ? have : nrep - have;
            HIPCHK(&h, hipMemcpy(c.R + have * row, c.R, n * row * sizeof(double),
                                 hipMemcpyDeviceToDevice));
            have += n;
        }
    }
    if (W)
        HIPCHK(&h, hipMemcpy((void *)c.W, W, 2 * (size_t)c.M2 * sizeof(double), hipMemcpyHostToDevice));
    {   // wall table of sweep_kernel_ma: per site its position (i dw, j dw) (SMC.c:748-750) and strengths, the plane last
        std::vector<double> wt((size_t)(c.M2 + 1) * 4, 0.0);
        const double dw = p.L / p.M;
        for (int m = 0; m < c.M2; m++) {
            wt[4 * m] = (m / p.M) * dw; wt[4 * m + 1] = (m % p.M) * dw;
            wt[4 * m + 2] = W ? W[2 * m] : 0.0; wt[4 * m + 3] = W ? W[2 * m + 1] : 0.0;
        }
        wt[4 * c.M2 + 2] = p.a0; wt[4 * c.M2 + 3] = p.b0;
        HIPCHK(&h, hipMemcpy((void *)c.wtab, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> st(nrep * 32);
    for (size_t r = 0; r < nrep; r++) {
        const uint32_t seed = seeds ? seeds[r] : (uint32_t)(p.base_seed + p.first_replica + r);
        seed_state(seed, &st[r * 32]);
    }
    HIPCHK(&h, hipMemcpy(c.rng, st.data(), st.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(&h, hipMemset(c.obs, 0, nrep * sizeof(ObsRec)));
    HIPCHK(&h, hipMemset(c.zhist, 0, nrep * p.Ncz * sizeof(unsigned long long)));
    // E[0] = energy + wallsEnergy (SMC.c:48)
    HIPCHK(&h, launch_total_energy(c, h.d_tmp, h.stream));
    HIPCHK(&h, launch_obs_op(c, h.d_tmp, 1, h.stream));
    HIPCHK(&h, hipStreamSynchronize(h.stream));
    h.uploaded = true;
    h.last_maxsteps = 0;
    return SMCX_OK;
}

// buffers of the cluster analysis: the bit matrices of a batch of replicas (at most ~8 GB)
// and the per-replica counters
static int ensure_lca(Handle &h)
{
    if (h.lca_counts) return SMCX_OK;
    if (!(h.p.lca_cutoff > 0)) return SMCX_ERR_PARAM;
    const long np = (long)h.p.N * (h.p.N - 1) / 2;
    h.lca_words = (np + 31) / 32 + 1;
    long batch = (long)(8.0e9 / (4.0 * h.lca_words));
    if (batch < 1) batch = 1;
    if (batch > h.p.nrep) batch = h.p.nrep;
    if (batch > 65535) batch = 65535; // gridDim.y
    h.lca_batch = (int)batch;
    HIPCHK(&h, hipMalloc(&h.lca_bits, (size_t)batch * h.lca_words * sizeof(unsigned)));
    HIPCHK(&h, hipMalloc(&h.lca_counts, ((size_t)h.p.nrep + 1) * LCA_COUNTS * sizeof(unsigned long long)));
    HIPCHK(&h, hipMemsetAsync(h.lca_counts, 0, ((size_t)h.p.nrep + 1) * LCA_COUNTS * sizeof(unsigned long long), h.stream));
    return SMCX_OK;
}

static LcaArgs lca_args(const Handle &h)
{
    LcaArgs a;
    a.N = h.p.N; a.rep0 = 0;
    a.L = h.p.L; a.cut2 = h.p.lca_cutoff * h.p.lca_cutoff;
    a.R = h.c.R; a.bits = h.lca_bits; a.words = h.lca_words;
    a.counts = h.lca_counts; a.LCA = nullptr;
    return a;
}

// clusterAnalysis of every replica, batch by batch (SMC.c:145)
static int lca_all(Handle &h)
{
    LcaArgs a = lca_args(h);
    for (int r0 = 0; r0 < h.p.nrep; r0 += h.lca_batch) {
        a.rep0 = r0;
        const int nb = (h.p.nrep - r0 < h.lca_batch) ? h.p.nrep - r0 : h.lca_batch;
        HIPCHK(&h, launch_lca(a, nb, h.stream));
    }
    h.lca_analyses++;
    return SMCX_OK;
}

static int ensure_series(Handle &h, int maxsteps)
{
    DevCtx &c = h.c;
    if (!(h.p.flags & SMCX_FLAG_SERIES)) {
        c.series_stride = 0;
        return SMCX_OK;
    }
    if (c.series_stride < maxsteps + 1) {
        hipFree(c.Eseries); hipFree(c.jjseries);
        c.Eseries = nullptr; c.jjseries = nullptr;
        c.series_stride = maxsteps + 1;
        HIPCHK(&h, hipMalloc(&c.Eseries, (size_t)h.p.nrep * c.series_stride * sizeof(double)));
        HIPCHK(&h, hipMalloc(&c.jjseries, (size_t)h.p.nrep * c.series_stride * sizeof(int)));
    }
    HIPCHK(&h, hipMemsetAsync(c.Eseries, 0, (size_t)h.p.nrep * c.series_stride * sizeof(double), h.stream));
    HIPCHK(&h, hipMemsetAsync(c.jjseries, 0, (size_t)h.p.nrep * c.series_stride * sizeof(int), h.stream));
    return SMCX_OK;
}

// one phase of sMC: `steps` sweeps at step parameter A.  A launch group holds at most
// h.chunk sweeps (the random numbers kept on the device) and ends where the next density
// histogram is due, which is taken from the positions in memory before that sweep's moves.
static int run_phase(Handle &h, int steps, double A, int production, int gather_lapse)
{
    int done = 0;
    bool first = true;
    while (done < steps) {
        int k = (steps - done < h.chunk) ? steps - done : h.chunk;
        if (production) {
            if ((done + 1) % gather_lapse == 0) { // SMC.c:137-141
                if (h.c.Pseries) HIPCHK(&h, launch_pressure(h.c, h.last_gathers, h.stream));
                HIPCHK(&h, launch_hist(h.c, h.stream));
                h.last_gathers++;
                if ((h.p.flags & SMCX_FLAG_CLUSTERS) && h.last_gathers % h.p.lca_time == 0) { // SMC.c:143
                    const int rc = lca_all(h);
                    if (rc != SMCX_OK) return rc;
                }
            }
            // next sweep index n > done with (n+1) % gather_lapse == 0
            const int next = ((done + 1) / gather_lapse + 1) * gather_lapse - 1;
            if (next - done < k) k = next - done;
        }
        HIPCHK(&h, launch_rng_prepass(h.c, k, A, h.stream));
        HIPCHK(&h, launch_sweeps(h.c, h.plan, k, A, h.stream, &h.timer));
        HIPCHK(&h, launch_finalize(h.c, k, production, done, (production && first) ? 1 : 0, h.stream));
        h.last_launches++;
        first = false;
        done += k;
    }
    return SMCX_OK;
}

extern "C" int smcx_run(smcx_handle *hh, int eqsteps, int maxsteps, int gather_lapse)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    if (eqsteps < 0 || maxsteps < 0 || gather_lapse < 1) return SMCX_ERR_PARAM;
    HIPCHK(&h, hipSetDevice(h.p.device));
    int rc = ensure_series(h, maxsteps);
    if (rc != SMCX_OK) return rc;
    h.last_launches = 0;
    h.timer.reset();
    h.last_gathers = 0;
    if (h.c.D) { // D, Mu and Rbin start from zero in every sMC call (SMC.c:52-55)
        const size_t Nc = (size_t)h.p.Ncx * h.p.Ncx * h.p.Ncz;
        HIPCHK(&h, hipMemsetAsync(h.c.D, 0, (size_t)h.p.nrep * Nc * sizeof(unsigned long long), h.stream));
        HIPCHK(&h, hipMemsetAsync(h.c.Mu, 0, (size_t)h.p.nrep * Nc * sizeof(unsigned long long), h.stream));
        HIPCHK(&h, hipMemsetAsync(h.c.Rbin, 0, (size_t)h.p.nrep * h.p.N * sizeof(int), h.stream));
    }
    if (h.p.flags & SMCX_FLAG_PRESSURE) {
        const int need = maxsteps / gather_lapse + 1;
        if (h.c.pstride < need) {
            hipFree(h.c.Pseries);
            h.c.Pseries = nullptr;
            h.c.pstride = need;
            HIPCHK(&h, hipMalloc(&h.c.Pseries, (size_t)h.p.nrep * need * sizeof(double)));
        }
        HIPCHK(&h, hipMemsetAsync(h.c.Pseries, 0, (size_t)h.p.nrep * h.c.pstride * sizeof(double), h.stream));
    }
    if (h.p.flags & SMCX_FLAG_CLUSTERS) { // l1, l2, l3 start from zero in every sMC call (SMC.c:58-60)
        rc = ensure_lca(h);
        if (rc != SMCX_OK) return rc;
        HIPCHK(&h, hipMemsetAsync(h.lca_counts, 0, ((size_t)h.p.nrep + 1) * LCA_COUNTS * sizeof(unsigned long long), h.stream));
        h.lca_analyses = 0;
    }
    // zero the accumulators, remember E at entry (the reference's E[0])
    HIPCHK(&h, launch_obs_op(h.c, h.d_save, 0, h.stream));
    HIPCHK(&h, hipEventRecord(h.ev0, h.stream));
    // thermalisation at 2A (SMC.c:110-118)
    rc = run_phase(h, eqsteps, h.p.A * 2, 0, gather_lapse);
    if (rc != SMCX_OK) return rc;
    if (eqsteps > 0 && (h.p.flags & SMCX_FLAG_E0_RESTART)) // SMC.c:194 restarts from E[0]
        HIPCHK(&h, launch_obs_op(h.c, h.d_save, 1, h.stream));
    // production at A (SMC.c:134-196)
    rc = run_phase(h, maxsteps, h.p.A, 1, gather_lapse);
    if (rc != SMCX_OK) return rc;
    HIPCHK(&h, hipEventRecord(h.ev1, h.stream));
    HIPCHK(&h, hipStreamSynchronize(h.stream));
    float ms = 0.f;
    HIPCHK(&h, hipEventElapsedTime(&ms, h.ev0, h.ev1));
    h.last_ms = ms;
    HIPCHK(&h, h.timer.finish());
    h.last_sweep_ms = h.timer.sum_ms;
    h.last_maxsteps = maxsteps;
    h.last_eqsteps = eqsteps;
    return SMCX_OK;
}

extern "C" int smcx_last_kernel_ms(smcx_handle *hh, double *ms, int *launches)
{
    if (!hh) return SMCX_ERR_PARAM;
    if (ms) *ms = hh->h.last_sweep_ms;
    if (launches) *launches = hh->h.timer.launches(); // launches of the sweep kernel (sweep_kernel_mb64: one per sweep)
    return SMCX_OK;
}

extern "C" int smcx_last_clock(smcx_handle *hh, double *ghz, double *wave_cycles)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.c.clk) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    std::vector<unsigned long long> st((size_t)h.p.nrep * 4);
    HIPCHK(&h, hipMemcpy(st.data(), h.c.clk, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> f, cyc;
    for (int r = 0; r < h.p.nrep; r++) {
        const double dc = (double)(st[4 * r + 2] - st[4 * r]), dr = (double)(st[4 * r + 3] - st[4 * r + 1]);
        if (st[4 * r + 3] > st[4 * r + 1] && st[4 * r + 2] > st[4 * r]) { f.push_back(dc / dr * 0.1); cyc.push_back(dc); }
    }
    if (f.empty()) return SMCX_ERR_STATE; // no sweep kernel has stamped yet
    std::nth_element(f.begin(), f.begin() + f.size() / 2, f.end());
    std::nth_element(cyc.begin(), cyc.begin() + cyc.size() / 2, cyc.end());
    if (ghz) *ghz = f[f.size() / 2];
    if (wave_cycles) *wave_cycles = cyc[cyc.size() / 2];
    return SMCX_OK;
}

// spread of the wavefronts' lifetimes in the last sweep launch (100 MHz counter, microseconds): min, median, max,
// and the span from the first start to the last end
extern "C" int smcx_debug_wave_spread(smcx_handle *hh, double *out4)
{
    if (!hh || !out4) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.c.clk) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    std::vector<unsigned long long> st((size_t)h.p.nrep * 4);
    HIPCHK(&h, hipMemcpy(st.data(), h.c.clk, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> d;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int r = 0; r < h.p.nrep; r++)
        if (st[4 * r + 3] > st[4 * r + 1]) {
            d.push_back((double)(st[4 * r + 3] - st[4 * r + 1]) * 0.01);
            t0 = std::min(t0, st[4 * r + 1]); t1 = std::max(t1, st[4 * r + 3]);
        }
    if (d.empty()) return SMCX_ERR_STATE;
    std::sort(d.begin(), d.end());
    out4[0] = d.front(); out4[1] = d[d.size() / 2]; out4[2] = d.back(); out4[3] = (double)(t1 - t0) * 0.01;
    if (getenv("SMCX_SPREAD_DUMP")) { // lifetime percentiles, and the lifetimes in replica order (which replicas are slow?)
        for (int q : {1, 5, 10, 25, 50, 75, 90, 95, 99}) fprintf(stderr, "p%d %.0f  ", q, d[d.size() * q / 100]);
        fprintf(stderr, "\n");
        const int step = std::max(1, h.p.nrep / 64);
        for (int r = 0; r < h.p.nrep; r += step) fprintf(stderr, "%.0f ", (double)(st[4 * r + 3] - st[4 * r + 1]) * 0.01);
        fprintf(stderr, "\n");
        for (int r = 0; r < h.p.nrep; r += step) // shader clock seen by the same wavefronts, MHz
            fprintf(stderr, "%.0f ", (double)(st[4 * r + 2] - st[4 * r]) / (double)(st[4 * r + 3] - st[4 * r + 1]) * 100.0);
        fprintf(stderr, "\n");
    }
    return SMCX_OK;
}

// diagnostics: the raw clock rows of the last sweep launch, [nrep][4] (start s_memtime, s_memrealtime, end s_memtime,
// s_memrealtime); the two-team stamps variant of tools/probes/tt_phases.py: [nrep][256] = 16 phase sums of each of up to 16 wavefronts
extern "C" int smcx_debug_clk_rows(smcx_handle *hh, uint64_t *out)
{
    if (!hh || !out) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.c.clk) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(out, h.c.clk, (size_t)h.p.nrep * CLK_COLS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_last_run_ms(smcx_handle *hh, double *ms)
{
    if (!hh || !ms) return SMCX_ERR_PARAM;
    *ms = hh->h.last_ms;
    return SMCX_OK;
}

static int fetch_obs(Handle &h, std::vector<ObsRec> &obs)
{
    obs.resize(h.p.nrep);
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(obs.data(), h.c.obs, obs.size() * sizeof(ObsRec), hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_observables(smcx_handle *hh, double *acceptance_ratio, double *meanE, double *dE,
                                uint64_t *zhist, uint64_t *accepted, double *E_last)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    std::vector<ObsRec> obs;
    int rc = fetch_obs(h, obs);
    if (rc != SMCX_OK) return rc;
    const int N = h.p.N, maxsteps = h.last_maxsteps;
    for (int r = 0; r < h.p.nrep; r++) {
        const ObsRec &o = obs[r];
        if (acceptance_ratio) // intmean(jj, maxsteps)/N, SMC.c:248
            acceptance_ratio[r] = maxsteps > 0 ? (o.accepted / maxsteps) / N : 0.0;
        const double len = o.nsamp > 0 ? o.nsamp : 1.0;
        const double mean = o.sumE / len;
        if (meanE) meanE[r] = mean;                                       // SMC.c:244
        if (dE) dE[r] = std::sqrt(o.sumE2 / len - mean * mean);           // SMC.c:245, matematicose.c:96-103
        if (accepted) accepted[r] = (uint64_t)o.accepted;
        if (E_last) E_last[r] = o.Ecur;
    }
    if (zhist)
        HIPCHK(&h, hipMemcpy(zhist, h.c.zhist, (size_t)h.p.nrep * h.p.Ncz * sizeof(uint64_t),
                             hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_therm_acceptance(smcx_handle *hh, double *ratio)
{
    if (!hh || !ratio) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    std::vector<ObsRec> obs;
    int rc = fetch_obs(h, obs);
    if (rc != SMCX_OK) return rc;
    for (int r = 0; r < h.p.nrep; r++) // intmean(jt, eqsteps)/N, SMC.c:124
        ratio[r] = h.last_eqsteps > 0 ? (obs[r].therm_accepted / h.last_eqsteps) / h.p.N : 0.0;
    return SMCX_OK;
}

extern "C" int smcx_hist_info(smcx_handle *hh, uint64_t *gathers, uint64_t *oob)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    std::vector<ObsRec> obs;
    int rc = fetch_obs(h, obs);
    if (rc != SMCX_OK) return rc;
    for (int r = 0; r < h.p.nrep; r++) {
        if (gathers) gathers[r] = (uint64_t)obs[r].gathers;
        if (oob) oob[r] = (uint64_t)obs[r].oob;
    }
    return SMCX_OK;
}

extern "C" int smcx_series(smcx_handle *hh, double *E_series, int32_t *jj)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!(h.p.flags & SMCX_FLAG_SERIES) || !h.c.Eseries) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    const int ms = h.last_maxsteps, stride = h.c.series_stride;
    if (E_series)
        HIPCHK(&h, hipMemcpy2D(E_series, (size_t)(ms + 1) * sizeof(double), h.c.Eseries,
                               (size_t)stride * sizeof(double), (size_t)(ms + 1) * sizeof(double),
                               h.p.nrep, hipMemcpyDeviceToHost));
    if (jj && ms > 0)
        HIPCHK(&h, hipMemcpy2D(jj, (size_t)ms * sizeof(int), h.c.jjseries, (size_t)stride * sizeof(int),
                               (size_t)ms * sizeof(int), h.p.nrep, hipMemcpyDeviceToHost));
    return SMCX_OK;
}

#ifdef SMCX_CHECK
// diagnostic build only (libsmcx_check.so): what the fp64 test run beside the screen counted since
// smcx_create -- pairs inside the cutoff, candidates the screen flagged, pairs inside the cutoff
// that it did NOT flag (must stay 0)
extern "C" int smcx_debug_check_counts(smcx_handle *hh, uint64_t *out /*[3]*/)
{
    if (!hh || !out) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(out, h.c.dbg, 3 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SMCX_OK;
}
// the same plus the executed work of the z-ordered byte-screen kernels (SMCX_CHECK_MB=2): out[3] = 4-slot groups
// screened (256 cells per wavefront each), out[4] = screen passes (one per probe and wavefront), out[5] = rounds of the fp64
// body beyond the first, out[6] = probes with candidates on lanes l and l + 32, out[7] = (two-team kernel) candidates handed over
// to the list for which NO working lane was enabled -- must stay 0 (round 5: the s_bfm_b64 count of 64 wrapped to an empty mask)
extern "C" int smcx_debug_work_counts(smcx_handle *hh, uint64_t *out /*[8]*/)
{
    if (!hh || !out) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(out, h.c.dbg, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return SMCX_OK;
}
#endif

// for smcx_acf.hip: where the energy series of the last run lives
#ifdef SMCX_STAMPS
// diagnostic build only: per-phase cycle counts the instrumented sweep kernel left at the head of
// each replica's displacement block (tools/phase_stamps.py)
extern "C" int smcx_debug_stamps(smcx_handle *hh, double *out /*[nrep][8]*/)
{
    Handle &h = hh->h;
    const size_t stride = (size_t)h.chunk * 3 * h.p.N;
    for (int r = 0; r < h.p.nrep; r++)
        HIPCHK(&h, hipMemcpy(out + 8 * (size_t)r, h.c.displ + r * stride, 8 * sizeof(double), hipMemcpyDeviceToHost));
    return SMCX_OK;
}
#endif

extern "C" int smcx_internal_series_view(smcx_handle *hh, const double **E, int *stride, int *maxsteps,
                                          int *nrep, double *T, int *device, void **stream)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!(h.p.flags & SMCX_FLAG_SERIES) || !h.c.Eseries) return SMCX_ERR_STATE;
    *E = h.c.Eseries; *stride = h.c.series_stride; *maxsteps = h.last_maxsteps; *nrep = h.p.nrep;
    *T = h.p.T; *device = h.p.device; *stream = (void *)h.stream;
    return SMCX_OK;
}

extern "C" int smcx_density(smcx_handle *hh, uint64_t *D, uint64_t *Mu)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.c.D) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    const size_t bytes = (size_t)h.p.nrep * h.p.Ncx * h.p.Ncx * h.p.Ncz * sizeof(uint64_t);
    if (D) HIPCHK(&h, hipMemcpy(D, h.c.D, bytes, hipMemcpyDeviceToHost));
    if (Mu) HIPCHK(&h, hipMemcpy(Mu, h.c.Mu, bytes, hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_cluster_update(smcx_handle *hh)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    int rc = ensure_lca(h);
    if (rc != SMCX_OK) return rc;
    rc = lca_all(h);
    if (rc != SMCX_OK) return rc;
    HIPCHK(&h, hipStreamSynchronize(h.stream));
    return SMCX_OK;
}

extern "C" int smcx_cluster_counts(smcx_handle *hh, uint64_t *n1, uint64_t *h2, uint64_t *h3,
                                   uint64_t *overflow, int *analyses)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.lca_counts) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    const size_t nrep = h.p.nrep;
    std::vector<unsigned long long> tmp(nrep * LCA_COUNTS);
    HIPCHK(&h, hipMemcpy(tmp.data(), h.lca_counts, tmp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (size_t r = 0; r < nrep; r++) {
        const unsigned long long *c = &tmp[r * LCA_COUNTS];
        if (n1) n1[r] = c[0];
        for (int v = 0; v < 16; v++) {
            if (h2) h2[r * 16 + v] = c[1 + v];
            if (h3) h3[r * 16 + v] = c[17 + v];
        }
        if (overflow) overflow[r] = c[33];
    }
    if (analyses) *analyses = h.lca_analyses;
    return SMCX_OK;
}

extern "C" int smcx_cluster_analysis(smcx_handle *hh, int replica, int32_t *LCA, uint64_t *overflow)
{
    if (!hh || !LCA) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    if (replica < 0 || replica >= h.p.nrep) return SMCX_ERR_PARAM;
    HIPCHK(&h, hipSetDevice(h.p.device));
    int rc = ensure_lca(h);
    if (rc != SMCX_OK) return rc;
    const size_t np = (size_t)h.p.N * (h.p.N - 1) / 2;
    int *d_lca = nullptr;
    HIPCHK(&h, hipMalloc(&d_lca, 3 * np * sizeof(int)));
    hipError_t e = hipMemsetAsync(d_lca, 0, 3 * np * sizeof(int), h.stream);
    // counters of this call go to the scratch row: shift the base so that row `replica` is it
    unsigned long long *scratch = h.lca_counts + (size_t)h.p.nrep * LCA_COUNTS;
    if (e == hipSuccess) e = hipMemsetAsync(scratch, 0, LCA_COUNTS * sizeof(unsigned long long), h.stream);
    LcaArgs a = lca_args(h);
    a.rep0 = replica;
    a.counts = scratch - (size_t)replica * LCA_COUNTS;
    a.LCA = d_lca;
    if (e == hipSuccess) e = launch_lca(a, 1, h.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h.stream);
    if (e == hipSuccess) e = hipMemcpy(LCA, d_lca, 3 * np * sizeof(int), hipMemcpyDeviceToHost);
    unsigned long long ov = 0;
    if (e == hipSuccess) e = hipMemcpy(&ov, scratch + 33, sizeof(ov), hipMemcpyDeviceToHost);
    hipFree(d_lca);
    HIPCHK(&h, e);
    if (overflow) *overflow = ov;
    return SMCX_OK;
}

extern "C" int smcx_pressure_series(smcx_handle *hh, double *P, int *ngathers)
{
    if (!hh) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.c.Pseries) return SMCX_ERR_STATE;
    if (ngathers) *ngathers = h.last_gathers;
    if (P && h.last_gathers > 0) {
        HIPCHK(&h, hipSetDevice(h.p.device));
        HIPCHK(&h, hipMemcpy2D(P, (size_t)h.last_gathers * sizeof(double), h.c.Pseries,
                               (size_t)h.c.pstride * sizeof(double), (size_t)h.last_gathers * sizeof(double),
                               h.p.nrep, hipMemcpyDeviceToHost));
    }
    return SMCX_OK;
}

extern "C" int smcx_download_positions(smcx_handle *hh, double *R)
{
    if (!hh || !R) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(R, h.c.R, (size_t)h.p.nrep * 3 * h.p.N * sizeof(double), hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_total_energy(smcx_handle *hh, double *E)
{
    if (!hh || !E) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, launch_total_energy(h.c, h.d_tmp, h.stream));
    HIPCHK(&h, hipStreamSynchronize(h.stream));
    HIPCHK(&h, hipMemcpy(E, h.d_tmp, (size_t)h.p.nrep * sizeof(double), hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_rng_export(smcx_handle *hh, uint32_t *state)
{
    if (!hh || !state) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (!h.uploaded) return SMCX_ERR_STATE;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(state, h.c.rng, (size_t)h.p.nrep * 32 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SMCX_OK;
}

extern "C" int smcx_rng_import(smcx_handle *hh, const uint32_t *state)
{
    if (!hh || !state) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    for (int r = 0; r < h.p.nrep; r++)
        if (state[r * 32 + 31] > 30) return SMCX_ERR_PARAM;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, hipMemcpy(h.c.rng, state, (size_t)h.p.nrep * 32 * sizeof(uint32_t), hipMemcpyHostToDevice));
    return SMCX_OK;
}

extern "C" size_t smcx_obs_device_bytes(const smcx_handle *hh)
{
    if (!hh) return 0;
    return ((size_t)hh->h.p.nrep * SMCX_OBS_RECORD_DOUBLES + (size_t)hh->h.p.nrep * hh->h.p.Ncz) * sizeof(double);
}

extern "C" int smcx_export_observables_device(smcx_handle *hh, void *dst, size_t bytes)
{
    if (!hh || !dst) return SMCX_ERR_PARAM;
    Handle &h = hh->h;
    if (bytes != smcx_obs_device_bytes(hh)) return SMCX_ERR_PARAM;
    HIPCHK(&h, hipSetDevice(h.p.device));
    HIPCHK(&h, launch_pack_obs(h.c, (double *)dst, h.stream));
    HIPCHK(&h, hipStreamSynchronize(h.stream));
    return SMCX_OK;
}

// ---- teacher-forced evaluator -------------------------------------------------------
extern "C" int smcx_eval_moves(const smcx_params *p, const double *R, const double *W, const int32_t *n,
                               const double *prop, double *out)
{
    int rc = validate(p);
    if (rc != SMCX_OK) return rc;
    if (!R || !n || !prop || !out) return SMCX_ERR_PARAM;
    if ((p->flags & SMCX_FLAG_WALLS) && !W) return SMCX_ERR_PARAM;
    for (int r = 0; r < p->nrep; r++)
        if (n[r] < 0 || n[r] >= p->N) return SMCX_ERR_PARAM;
    int ndev = 0;
    rc = smcx_device_count(&ndev);
    if (rc != SMCX_OK) return rc;
    Handle h;
    h.p = *p;
    std::memset(&h.c, 0, sizeof(h.c));
    h.chunk = 1;
    fill_ctx(h);
    DevCtx &c = h.c;
    const size_t nrep = p->nrep, row = 3 * (size_t)p->N;
    int *d_n = nullptr;
    double *d_prop = nullptr, *d_out = nullptr;
    int status = SMCX_OK;
    auto fail = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && status == SMCX_OK) {
            g_last_error = std::string(what) + ": " + hipGetErrorString(e);
            status = SMCX_ERR_HIP;
        }
    };
    fail(hipSetDevice(p->device), "hipSetDevice");
    fail(hipMalloc(&c.R, nrep * row * sizeof(double)), "hipMalloc R");
    fail(hipMalloc((void **)&c.W, (size_t)(2 * c.M2 > 2 ? 2 * c.M2 : 2) * sizeof(double)), "hipMalloc W");
    fail(hipMalloc(&d_n, nrep * sizeof(int)), "hipMalloc n");
    fail(hipMalloc(&d_prop, nrep * 3 * sizeof(double)), "hipMalloc prop");
    fail(hipMalloc(&d_out, nrep * 8 * sizeof(double)), "hipMalloc out");
    if (status == SMCX_OK) {
        fail(hipMemcpy(c.R, R, nrep * row * sizeof(double), hipMemcpyHostToDevice), "copy R");
        if (W) fail(hipMemcpy((void *)c.W, W, 2 * (size_t)c.M2 * sizeof(double), hipMemcpyHostToDevice), "copy W");
        fail(hipMemcpy(d_n, n, nrep * sizeof(int), hipMemcpyHostToDevice), "copy n");
        fail(hipMemcpy(d_prop, prop, nrep * 3 * sizeof(double), hipMemcpyHostToDevice), "copy prop");
    }
    if (status == SMCX_OK) {
        fail(launch_eval_moves(c, d_n, d_prop, d_out, nullptr), "eval_moves_kernel");
        fail(hipDeviceSynchronize(), "sync");
        fail(hipMemcpy(out, d_out, nrep * 8 * sizeof(double), hipMemcpyDeviceToHost), "copy out");
    }
    hipFree(c.R); hipFree((void *)c.W); hipFree(d_n); hipFree(d_prop); hipFree(d_out);
    return status;
}

// ---- single-chain shim with the oneParticleMoves contract (SMC.h:102) ---------------
extern "C" int smcx_one_particle_moves(const smcx_params *p, uint32_t *rng, double *R, double *Rn,
                                       const double *W, double A, double T, int *j, double *U)
{
    if (!p || !rng || !R || !j || !U) return SMCX_ERR_PARAM;
    smcx_params q = *p;
    q.nrep = 1;
    q.A = A;
    q.T = T;
    q.flags &= ~SMCX_FLAG_SERIES;
    smcx_handle *hh = nullptr;
    int rc = smcx_create(&q, &hh);
    if (rc != SMCX_OK) return rc;
    Handle &h = hh->h;
    do {
        rc = smcx_upload(hh, R, 0, W, nullptr);
        if (rc != SMCX_OK) break;
        rc = smcx_rng_import(hh, rng);
        if (rc != SMCX_OK) break;
        // the caller's running energy goes in as the current energy (SMC.c:116-117)
        hipError_t e = hipMemcpy(h.d_tmp, U, sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { rc = SMCX_ERR_HIP; break; }
        if (launch_obs_op(h.c, h.d_tmp, 1, h.stream) != hipSuccess) { rc = SMCX_ERR_HIP; break; }
        rc = smcx_run(hh, 0, 1, 1 << 30);
        if (rc != SMCX_OK) break;
        uint64_t acc = 0;
        double Elast = 0.0;
        rc = smcx_observables(hh, nullptr, nullptr, nullptr, nullptr, &acc, &Elast);
        if (rc != SMCX_OK) break;
        rc = smcx_download_positions(hh, R);
        if (rc != SMCX_OK) break;
        if (Rn) std::memcpy(Rn, R, 3 * (size_t)q.N * sizeof(double)); // Rn == R on return, SMC.c:337-347
        rc = smcx_rng_export(hh, rng);
        if (rc != SMCX_OK) break;
        *j += (int)acc;
        *U = Elast;
    } while (0);
    smcx_destroy(hh);
    return rc;
}
