// smcx_kernels.hip -- gfx950 kernels of the SMC engine and their launchers.
//
// Kernels:
//   rng_prepass_kernel    per replica and sweep: the 4N+1 glibc rand() outputs one
//                         oneParticleMoves call consumes (SMC.c:284, 290, 335), turned
//                         into the 3N Box-Muller displacements (matematicose.c:183-193),
//                         the visiting offset and the N acceptance uniforms
//   sweep_kernel<S,WPR>   the hot path, all-fp64 form: the trial moves of SMC.c:292-348 for K
//   sweep_kernel_lead     sweeps, positions resident in registers as fp64, Metropolis step,
//                         incremental energy (SMC.c:340-341); _lead: sequential part of a move
//                         on one wavefront only.  The default form is the screened kernel of
//                         smcx_sweep_mx.hip; these run for fewer than 16 particles per lane
//                         and when smcx_params.tune_kernel asks for them
//   finalize_kernel       chain bookkeeping of sMC (SMC.c:194-195, 210-211, 244-250)
//   hist_kernel           localDensityAndMobility (SMC.c:912-927)
//   pressure_kernel       pressure + wallsPressure of a gather (SMC.c:696-720, 862-895)
//   total_energy_zk/_kernel  energy + wallsEnergy (SMC.c:626-646, 822-859)
//   eval_moves_kernel     teacher-forced Um,Fm,Un,Fn for one particle per replica
#include "smcx_device.hpp"
#include "smcx_kernels.h"

#include <cstdlib>

namespace smcx {

// ---------------------------------------------------------------------------------
// R: random numbers of `nsweeps` sweeps for every replica (256 threads per replica)
// ---------------------------------------------------------------------------------
// One block of 31 rand() outputs without the LDS crossbar.  new[j] = old[j] + (j < 3 ? old[j + 28] : new[j - 3]) is an
// inclusive prefix sum inside each residue class of j mod 3: class k lives in DPP row k (lane 16 k + p holds word 3 p + k,
// p = 0..10 for k = 0, 0..9 for k = 1, 2), so the sum is four row_shr adds of the vector ALU; the three words that cross
// classes (old[28], [29], [30] seed new[0], [1], [2]) travel through scalar registers.  The form with lane j = word j
// (rand_block, smcx_device.hpp: five ds_bpermute per block) was bound by the LDS crossbar its shuffles go through -- 0.26 ms
// of the pre-pass's 0.49 per sweep at 4096 x 4096 (profiles/r03_helpers_split.txt); build it again with
// make VARIANT=shfl EXTRA=-DSMCX_PREPASS_SHUFFLE.  Lanes above a class's last word hold sums nobody reads (a row_shr only
// moves words upwards).
__device__ __forceinline__ uint32_t rand_block_rows(uint32_t h, uint32_t m0, uint32_t m1, uint32_t m2)
{
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)h, 16 + 9);  // old[28]: class 1, p = 9
    const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)h, 32 + 9);  // old[29]: class 2, p = 9
    const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)h, 10);      // old[30]: class 0, p = 10
    uint32_t w = h + ((s0 & m0) | (s1 & m1) | (s2 & m2)); // m0, m1, m2: all ones in lane 0, 16, 32 (the classes' first words)
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, false); // row_shr:1 (lanes without a source add 0)
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x112, 0xf, 0xf, false); // row_shr:2
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x114, 0xf, 0xf, false); // row_shr:4
    w += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x118, 0xf, 0xf, false); // row_shr:8
    return w;
}

__global__ void __launch_bounds__(256)
rng_prepass_kernel(DevCtx c, int nsweeps, double A)
{
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int N = c.N;
    const int D = 4 * N + 1; // rand() calls per sweep: 3N normals, 1 offset, N uniforms
    uint32_t *raw = c.raw + (size_t)rep * c.rawStride;
    const double sigma = sqrt(2.0 * A); // SMC.c:284

    // rand() is r[i] = r[i-31] + r[i-3] mod 2^32 (a linear map of the 31-word state), consumed D at a time.  The
    // four waves generate a quarter of a sweep's blocks of 31 each: wave w starts from the state advanced by
    // w * rngQ blocks (rngJump applied w times, 31 x 31 multiply-adds each), wave 3 ends with the sweep's final
    // state and hands it to the others through LDS.
    // lane j < 31 holds state word r[i-31+j]; `left` outputs at the top of the last generated block of the
    // previous sweep have not been consumed yet
    __shared__ uint32_t sh_hist[32];
    uint32_t hist0 = (lane < 31) ? c.rng[rep * 32 + lane] : 0u;
    int left = uniform((int)c.rng[rep * 32 + 31]);
    const int Q = c.rngQ;

    for (int s = 0; s < nsweeps; s++) {
#ifndef SMCX_PREPASS_NOGEN // (measurement builds: make VARIANT=nogen EXTRA=-DSMCX_PREPASS_NOGEN, likewise _NOBM)
        if (wave == 0) {
            const uint32_t carry = __shfl(hist0, 31 - left + lane, 64);
            if (lane < left) raw[lane] = carry >> 1;
        }
        const int B = (D - left + 30) / 31;             // blocks this sweep
        const bool par = 3 * Q < B;                      // (a few particles only: wave 0 generates everything)
        uint32_t hist = hist0;
        for (int w = 0; par && w < wave; w++) {          // hist <- rngJump . hist
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 31; k++) {
#ifdef SMCX_PREPASS_SHUFFLE
                const uint32_t hk = __shfl(hist, k, 64);
#else
                const uint32_t hk = (uint32_t)__builtin_amdgcn_readlane((int)hist, k);
#endif
                acc += ((lane < 31) ? c.rngJump[k * 31 + lane] : 0u) * hk;
            }
            hist = acc;
        }
        const int b0 = par ? wave * Q : (wave == 0 ? 0 : B);
        const int b1 = par ? (wave == 3 ? B : (wave + 1) * Q) : B;
#ifdef SMCX_PREPASS_SHUFFLE
        for (int b = b0; b < b1; b++) {
            hist = rand_block(hist, lane);
            if (lane < 31) raw[left + 31 * b + lane] = hist >> 1;
        }
#else
        {   // the blocks run in the row layout of rand_block_rows: lane 16 k + p holds word 3 p + k
            const int word = 3 * (lane & 15) + (lane >> 4);
            const bool holds = lane < 48 && word < 31;
            const uint32_t m0 = lane == 0 ? ~0u : 0u, m1 = lane == 16 ? ~0u : 0u, m2 = lane == 32 ? ~0u : 0u;
            uint32_t hr = __shfl(hist, holds ? word : 0, 64);
            for (int b = b0; b < b1; b++) {
                hr = rand_block_rows(hr, m0, m1, m2);
                if (holds) raw[left + 31 * b + word] = hr >> 1;
            }
            hist = __shfl(hr, lane < 31 ? 16 * (lane % 3) + lane / 3 : 0, 64);
        }
#endif
        if (wave == (par ? 3 : 0) && lane < 31) sh_hist[lane] = hist;
        left = left + 31 * B - D;
        __syncthreads();
        hist0 = (lane < 31) ? sh_hist[lane] : 0u;
#endif
        __syncthreads();
        double *displ = c.displ + ((size_t)rep * c.chunk + s) * 3 * N;
        double *uni = c.uni + ((size_t)rep * c.chunk + s) * N;
        // Box-Muller pairs (matematicose.c:187-192; the second output swaps x1 and x2)
        for (int p = tid; p < (3 * N) / 2; p += 256) {
            const double x1 = (double)raw[2 * p] * (1.0 / 2147483648.0);
            const double x2 = (double)raw[2 * p + 1] * (1.0 / 2147483648.0);
#ifdef SMCX_PREPASS_NOBM
            displ[2 * p] = sigma * x1; displ[2 * p + 1] = sigma * x2;
#else
            displ[2 * p] = sigma * sqrt(-2.0 * log(1.0 - x1)) * cos(2.0 * M_PI * x2);
            displ[2 * p + 1] = sigma * sqrt(-2.0 * log(1.0 - x2)) * sin(2.0 * M_PI * x1);
#endif
        }
        // acceptance uniforms u = rand()/RAND_MAX (SMC.c:335), stored as log(u): the test
        // u < exp(-x/T) of SMC.c:329-335 is evaluated as log(u) < -x/T, which keeps the
        // exponential out of the sequential part of every move.  rand() == 0: the reference tests
        // 0 < exp(-x/T), true until exp underflows to zero at x/T >= 1075 ln 2; log(0) = -inf would
        // accept every finite x, so u = 0 is stored as that edge (LOG_U_ZERO)
        for (int i = tid; i < N; i += 256) {
            const uint32_t r = raw[3 * N + 1 + i];
            uni[i] = r ? log((double)r / 2147483647.0) : LOG_U_ZERO;
        }
        if (tid == 0) // SMC.c:290-294: the sweep starts at particle offset % N
            c.offs[(size_t)rep * c.chunk + s] = (int)(raw[3 * N] % (uint32_t)N);
        __syncthreads();
    }
    if (wave == 0) {
        if (lane < 31) c.rng[rep * 32 + lane] = hist0;
        if (lane == 31) c.rng[rep * 32 + 31] = (uint32_t)left;
    }
}

// ---------------------------------------------------------------------------------
// S1: the sweep
// ---------------------------------------------------------------------------------

#include "smcx_sweep_common.hpp"

// The hot kernel.  One workgroup = one replica chain; it keeps the chain's positions in
// registers across the `nsweeps` sweeps of the launch.  Per trial move: ONE pass over
// the register-resident neighbours evaluates the proposal of particle n (probe A) and
// the current position of particle n+1 (probe B); one 8-value reduction; the Metropolis
// decision in scalar registers.  Wave-uniform values live in SGPRs (uniform_d).
template <int S, int WPR, int MINW, int G>
__global__ void __launch_bounds__(64 * WPR, MINW)
sweep_kernel(SweepArgs a, DevCtx c, int nsweeps, double A)
{
    constexpr int T = 64 * WPR;
    __shared__ SweepShared<WPR> sh;

    const int rep = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int N = a.N;

    double *Rg = a.R + (size_t)rep * 3 * N;
    clock_stamp(a.clk, rep, 0);

    // ---- register-resident positions: particle l in lane l % T, slot l / T ------
    double x[S], y[S], z[S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int l = k * T + tid;
        if (l < N) { x[k] = Rg[3 * l]; y[k] = Rg[3 * l + 1]; z[k] = Rg[3 * l + 2]; }
        else { x[k] = 0.0; y[k] = 0.0; z[k] = FAR_PAD; }
    }
    int rot = 0; // register slot j holds logical slot (j + rot) % S

    if (wave == 0) fill_roles(c, sh.roles, lane); // the only use of the cold context
    __syncthreads();
    const int role = (wave == 0) ? sh.roles.role[lane] : -1;

    Geo g; g.L = a.L; g.invL = a.invL; g.cutoff2 = a.cutoff2;

    double E = uniform_d(a.obs[rep].Ecur);
    int par = 0;
    const double AoT = A * a.invT;         // SMC.c:307-309 (A/T)
    const double Ao4T = A * 0.25 * a.invT; // SMC.c:327 (A/(4T))

#pragma unroll 1
    for (int sw = 0; sw < nsweeps; sw++) {
        // positions written through by the owner wave in the last moves of the previous
        // sweep must have reached L2 before any wave fetches them again
        if constexpr (WPR > 1) __syncthreads();
        const double *displ = a.displ + ((size_t)rep * a.chunk + sw) * 3 * N;
        const double *uni = a.uni + ((size_t)rep * a.chunk + sw) * N;
        const int n0 = uniform(a.offs[(size_t)rep * a.chunk + sw]);

        int jacc = 0;
        // the visiting order n0..N-1, 0..n0-1 (SMC.c:292-294) is two ascending runs
#pragma unroll 1
        for (int run = 0; run < 2; run++) {
            const int first = run == 0 ? n0 : 0;
            const int len = run == 0 ? N - n0 : n0;
            if (len == 0) continue;
            const int vbase = run == 0 ? 0 : N - n0;
            const int ks = first / T;
            while (rot != ks) { rotate1<S>(x, y, z); rot = (rot + 1 == S) ? 0 : rot + 1; }
            // iteration i = -1 is the run's prologue: no particle moves, probe B alone
            // gives Um,Fm of the first particle.  tl = owner thread of particle n (slot 0).
            int tl = first - ks * T - 1;

            double Px = 0.0, Py = 0.0, Pz = FAR_PROBE;   // current position of particle n
            double Um = 0.0, Fmx = 0.0, Fmy = 0.0, Fmz = 0.0;
            double nBx = 0.0, nBy = 0.0, nBz = FAR_PROBE; // WPR > 1: next particle, fetched one
                                                          // move ahead of its use
            if constexpr (WPR > 1) {
                nBx = ld_coherent(Rg + 3 * first); nBy = ld_coherent(Rg + 3 * first + 1);
                nBz = ld_coherent(Rg + 3 * first + 2);
            }
            double bdx = 0.0, bdy = 0.0, bdz = 0.0, bu = 2.0;
#pragma unroll 1
            for (int i = -1; i < len; i++) {
                const int n = first + i;
                const bool hasA = (i >= 0);
                if (hasA && (i & 63) == 0) { // this wave's next 64 displacements / uniforms
                    if (i + lane < len) {
                        const int pn = n + lane;
                        bdx = displ[3 * pn]; bdy = displ[3 * pn + 1]; bdz = displ[3 * pn + 2];
                        bu = uni[vbase + i + lane];
                    }
                }
                const int j = i & 63;
                // proposal, SMC.c:307-316
                double Qx = 0.0, Qy = 0.0, Qz = FAR_PROBE;
                if (hasA) {
                    Qx = Px + (Fmx * AoT + rdlane(bdx, j));
                    Qy = Py + (Fmy * AoT + rdlane(bdy, j));
                    Qz = Pz + (Fmz * AoT + rdlane(bdz, j));
                    Qx = Qx - a.L * __builtin_rint(Qx * a.invL);
                    Qy = Qy - a.L * __builtin_rint(Qy * a.invL);
                    Qx = uniform_d(Qx); Qy = uniform_d(Qy); Qz = uniform_d(Qz);
                }

                const bool hasB = (i + 1 < len);
                const bool cross = hasB && (tl == T - 1);
                double Bx = 0.0, By = 0.0, Bz = FAR_PROBE;
                if (hasB) {
                    if constexpr (WPR == 1) {
                        if constexpr (S > 1) {
                            if (cross) { Bx = rdlane(x[1], 0); By = rdlane(y[1], 0); Bz = rdlane(z[1], 0); }
                            else { Bx = rdlane(x[0], tl + 1); By = rdlane(y[0], tl + 1); Bz = rdlane(z[0], tl + 1); }
                        } else {
                            Bx = rdlane(x[0], tl + 1); By = rdlane(y[0], tl + 1); Bz = rdlane(z[0], tl + 1);
                        }
                    } else {
                        Bx = uniform_d(nBx); By = uniform_d(nBy); Bz = uniform_d(nBz);
                    }
                }
                if constexpr (WPR > 1) {
                    // particle n+2 cannot change before its own move: fetch it now, use it next move
                    if (i + 2 < len) {
                        nBx = ld_coherent(Rg + 3 * (n + 2)); nBy = ld_coherent(Rg + 3 * (n + 2) + 1);
                        nBz = ld_coherent(Rg + 3 * (n + 2) + 2);
                    }
                }

                Acc8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                double side[4], tot[8], sOld[4], sNew[4];
                const bool exA0 = (tid == tl);
                const bool exB0 = (hasA && tid == tl) || (hasB && !cross && tid == tl + 1);
                const bool exB1 = cross && (tid == 0);
                // both probes at least one cutoff away from the periodic x,y edges?  (a disabled
                // probe sits at x = y = 0)  Then no pair needs the minimum-image wrap.
                const bool interior =
                    uniform((int)(fmax(fmax(fabs(Qx), fabs(Qy)), fmax(fabs(Bx), fabs(By))) <= a.edge)) != 0;
                fused_pass<S, G>(g, x, y, z, Qx, Qy, Qz, Bx, By, Bz, exA0, exB0, exB1, interior, v);
                if (wave == 0)
                    special_block(g, sh.roles, lane, role, hasA, hasB, hasA, Px, Py, Pz, Qx, Qy, Qz,
                                  Bx, By, Bz, v, side);
                else side[0] = side[1] = side[2] = side[3] = 0.0;
                combine<WPR>(sh, par, lane, wave, v, side, tot, sOld, sNew);

                bool acc = false;
                if (hasA) {
                    const double Un = 4.0 * tot[0], Fnx = tot[1], Fny = tot[2], Fnz = tot[3];
                    // SMC acceptance, SMC.c:326-335
                    const double dX = Fmx * AoT + rdlane(bdx, j);
                    const double dY = Fmy * AoT + rdlane(bdy, j);
                    const double dZ = Fmz * AoT + rdlane(bdz, j);
                    const double gx = Fnx - Fmx, gy = Fny - Fmy, gz = Fnz - Fmz;
                    const double deltaW = (gx * gx + gy * gy + gz * gz +
                                           2.0 * (gx * Fmx + gy * Fmy + gz * Fmz)) * Ao4T;
                    const double arg = Un - Um +
                                       (dX * (Fnx + Fmx) + dY * (Fny + Fmy) + dZ * (Fnz + Fmz)) * 0.5 + deltaW;
                    const double lu = rdlane(bu, j); // log(u), see rng_prepass_kernel
                    acc = (lu < -arg * a.invT);      // u < exp(-arg/T); NaN rejects (SMC.c:335)
                    acc = (uniform((int)acc) != 0);
                    const bool upd = acc && (tid == tl);
                    x[0] = upd ? Qx : x[0]; y[0] = upd ? Qy : y[0]; z[0] = upd ? Qz : z[0];
                    if constexpr (WPR > 1) {
                        if (upd) { Rg[3 * n] = Qx; Rg[3 * n + 1] = Qy; Rg[3 * n + 2] = Qz; }
                    }
                    if (acc) { E = uniform_d(E + (Un - Um)); jacc++; }
                }

                if (hasB) { // next particle's Um,Fm = B sums + the (n, n+1) pair term
                    double s0, s1, s2, s3;
                    if constexpr (WPR == 1) { // the term sits in lane 30 (old) or 31 (new) of this wave
                        const int src = acc ? SIDE_LANE_NEW : SIDE_LANE_OLD;
                        s0 = rdlane(sOld[0], src); s1 = rdlane(sOld[1], src);
                        s2 = rdlane(sOld[2], src); s3 = rdlane(sOld[3], src);
                    } else {
                        s0 = acc ? sNew[0] : sOld[0]; s1 = acc ? sNew[1] : sOld[1];
                        s2 = acc ? sNew[2] : sOld[2]; s3 = acc ? sNew[3] : sOld[3];
                    }
                    Um = uniform_d(4.0 * (tot[4] + s0));
                    Fmx = uniform_d(tot[5] + s1);
                    Fmy = uniform_d(tot[6] + s2);
                    Fmz = uniform_d(tot[7] + s3);
                    Px = Bx; Py = By; Pz = Bz;
                    if (cross) {
                        rotate1<S>(x, y, z);
                        rot = (rot + 1 == S) ? 0 : rot + 1;
                        tl = 0;
                    } else {
                        tl++;
                    }
                }
            }
        }
        // C: hand E[n+1] and jj[n] (SMC.c:194-195) to the bookkeeping kernel
        if (tid == 0) {
            SweepRec r; r.E = E; r.accepted = jacc; r.pad = 0;
            a.rec[(size_t)rep * a.chunk + sw] = r;
        }
    }

    clock_stamp(a.clk, rep, 1);
    // ---- positions back to memory (with several waves they were written through) ----
    if constexpr (WPR == 1) {
#pragma unroll
        for (int k = 0; k < S; k++) {
            int ls = k + rot; if (ls >= S) ls -= S;
            const int l = ls * T + tid;
            if (l < N) { Rg[3 * l] = x[k]; Rg[3 * l + 1] = y[k]; Rg[3 * l + 2] = z[k]; }
        }
    }
}

// ---------------------------------------------------------------------------------
// The same sweep for replicas spread over several wavefronts, with the sequential part
// of a move done ONCE: every wave runs the pass over its own slots and reduces its eight
// partial sums, then only wave 0 (the leader) combines them, takes the Metropolis
// decision, forms the next particle's Um,Fm and the next proposal, and publishes
// {accepted, next probe A, next probe B} through LDS.  The kernel is bound by VALU issue,
// so the ~150 instructions per move that the other WPR-1 waves no longer replicate are
// worth more than the second workgroup barrier they cost.
// ---------------------------------------------------------------------------------
template <int WPR> struct LeadShared {
    RoleTable roles;
    double red[WPR][8];
    double bc[8]; // accepted, Ax, Ay, Az, Bx, By, Bz, interior
};

template <int S, int WPR, int MINW, int G>
__global__ void __launch_bounds__(64 * WPR, MINW)
sweep_kernel_lead(SweepArgs a, DevCtx c, int nsweeps, double A)
{
    static_assert(WPR > 1, "leader/follower form needs several waves");
    constexpr int T = 64 * WPR;
    __shared__ LeadShared<WPR> sh;

    const int rep = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const bool leader = (wave == 0);
    const int N = a.N;
    double *Rg = a.R + (size_t)rep * 3 * N;

    double x[S], y[S], z[S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int l = k * T + tid;
        if (l < N) { x[k] = Rg[3 * l]; y[k] = Rg[3 * l + 1]; z[k] = Rg[3 * l + 2]; }
        else { x[k] = 0.0; y[k] = 0.0; z[k] = FAR_PAD; }
    }
    int rot = 0;

    if (leader) fill_roles(c, sh.roles, lane);
    __syncthreads();
    const int role = leader ? sh.roles.role[lane] : -1;

    Geo g; g.L = a.L; g.invL = a.invL; g.cutoff2 = a.cutoff2;
    double E = uniform_d(a.obs[rep].Ecur); // leader's
    const double AoT = A * a.invT;
    const double Ao4T = A * 0.25 * a.invT;

#pragma unroll 1
    for (int sw = 0; sw < nsweeps; sw++) {
        __syncthreads(); // write-through of the previous sweep's last moves has reached L2
        const double *displ = a.displ + ((size_t)rep * a.chunk + sw) * 3 * N;
        const double *uni = a.uni + ((size_t)rep * a.chunk + sw) * N;
        const int n0 = uniform(a.offs[(size_t)rep * a.chunk + sw]);
        int jacc = 0;
#pragma unroll 1
        for (int run = 0; run < 2; run++) {
            const int first = run == 0 ? n0 : 0;
            const int len = run == 0 ? N - n0 : n0;
            if (len == 0) continue;
            const int vbase = run == 0 ? 0 : N - n0;
            const int ks = first / T;
            while (rot != ks) { rotate1<S>(x, y, z); rot = (rot + 1 == S) ? 0 : rot + 1; }
            int tl = first - ks * T - 1;

            // leader state
            double Px = 0.0, Py = 0.0, Pz = FAR_PROBE, Um = 0.0, Fmx = 0.0, Fmy = 0.0, Fmz = 0.0;
            double nBx = 0.0, nBy = 0.0, nBz = FAR_PROBE, bdx = 0.0, bdy = 0.0, bdz = 0.0, bu = 2.0;
            // inputs of iteration -1: no proposal, probe B = the run's first particle
            __syncthreads(); // every wave has read the previous run's last broadcast
            if (leader) {
                const double b0 = ld_coherent(Rg + 3 * first), b1 = ld_coherent(Rg + 3 * first + 1),
                             b2 = ld_coherent(Rg + 3 * first + 2);
                if (len > 1) {
                    nBx = ld_coherent(Rg + 3 * (first + 1)); nBy = ld_coherent(Rg + 3 * (first + 1) + 1);
                    nBz = ld_coherent(Rg + 3 * (first + 1) + 2);
                }
                if (lane == 0) {
                    sh.bc[0] = 0.0; sh.bc[1] = 0.0; sh.bc[2] = 0.0; sh.bc[3] = FAR_PROBE;
                    sh.bc[4] = b0; sh.bc[5] = b1; sh.bc[6] = b2;
                    sh.bc[7] = (fmax(fabs(b0), fabs(b1)) <= a.edge) ? 1.0 : 0.0;
                }
            }
            __syncthreads();
            double Qx = 0.0, Qy = 0.0, Qz = FAR_PROBE;
            double Bx = uniform_d(sh.bc[4]), By = uniform_d(sh.bc[5]), Bz = uniform_d(sh.bc[6]);
            bool interior = uniform((int)(sh.bc[7] != 0.0)) != 0;

#pragma unroll 1
            for (int i = -1; i < len; i++) {
                const int n = first + i;
                const bool hasA = (i >= 0);
                const bool hasB = (i + 1 < len);
                const bool cross = hasB && (tl == T - 1);

                Acc8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                double side[4];
                const bool exA0 = (tid == tl);
                const bool exB0 = (hasA && tid == tl) || (hasB && !cross && tid == tl + 1);
                const bool exB1 = cross && (tid == 0);
                fused_pass<S, G>(g, x, y, z, Qx, Qy, Qz, Bx, By, Bz, exA0, exB0, exB1, interior, v);
                if (leader)
                    special_block(g, sh.roles, lane, role, hasA, hasB, hasA, Px, Py, Pz, Qx, Qy, Qz,
                                  Bx, By, Bz, v, side);
                const double r = reduce8(v.a0, v.a1, v.a2, v.a3, v.b0, v.b1, v.b2, v.b3, lane);
                if ((lane & 7) == 0) sh.red[wave][lane >> 3] = r;
                __syncthreads(); // #1: partial sums of all waves are in LDS

                if (leader) {
                    double t = 0.0;
                    if (lane < 8 * WPR) t = sh.red[lane >> 3][lane & 7];
                    if constexpr (WPR > 8) t += sh.red[(lane >> 3) + 8][lane & 7];
                    if constexpr (WPR >= 8) t = sum_x32(t);
                    if constexpr (WPR >= 4) t = sum_x16(t);
                    t = sum_x8(t);
                    double tot[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) tot[j] = rdlane(t, j);

                    bool acc = false;
                    if (hasA) { // SMC acceptance, SMC.c:326-335
                        const int j = i & 63;
                        const double Un = 4.0 * tot[0], Fnx = tot[1], Fny = tot[2], Fnz = tot[3];
                        const double dX = Fmx * AoT + rdlane(bdx, j);
                        const double dY = Fmy * AoT + rdlane(bdy, j);
                        const double dZ = Fmz * AoT + rdlane(bdz, j);
                        const double gx = Fnx - Fmx, gy = Fny - Fmy, gz = Fnz - Fmz;
                        const double deltaW = (gx * gx + gy * gy + gz * gz +
                                               2.0 * (gx * Fmx + gy * Fmy + gz * Fmz)) * Ao4T;
                        const double arg = Un - Um +
                                           (dX * (Fnx + Fmx) + dY * (Fny + Fmy) + dZ * (Fnz + Fmz)) * 0.5 + deltaW;
                        const double lu = rdlane(bu, j);
                        acc = uniform((int)(lu < -arg * a.invT)) != 0;
                        if (acc) { E = uniform_d(E + (Un - Um)); jacc++; }
                    }
                    // inputs of the next iteration: particle n+1 becomes the moving one
                    double qx = 0.0, qy = 0.0, qz = FAR_PROBE, bx = 0.0, by = 0.0, bz = FAR_PROBE;
                    if (hasB) {
                        const int src = acc ? SIDE_LANE_NEW : SIDE_LANE_OLD; // the (n, n+1) pair term
                        Um = uniform_d(4.0 * (tot[4] + rdlane(side[0], src)));
                        Fmx = uniform_d(tot[5] + rdlane(side[1], src));
                        Fmy = uniform_d(tot[6] + rdlane(side[2], src));
                        Fmz = uniform_d(tot[7] + rdlane(side[3], src));
                        Px = Bx; Py = By; Pz = Bz;
                        const int i1 = i + 1;
                        if ((i1 & 63) == 0) { // the next 64 displacements / log-uniforms
                            if (i1 + lane < len) {
                                const int pn = first + i1 + lane;
                                bdx = displ[3 * pn]; bdy = displ[3 * pn + 1]; bdz = displ[3 * pn + 2];
                                bu = uni[vbase + i1 + lane];
                            }
                        }
                        const int j1 = i1 & 63; // proposal, SMC.c:307-316
                        qx = Px + (Fmx * AoT + rdlane(bdx, j1));
                        qy = Py + (Fmy * AoT + rdlane(bdy, j1));
                        qz = Pz + (Fmz * AoT + rdlane(bdz, j1));
                        qx = qx - a.L * __builtin_rint(qx * a.invL);
                        qy = qy - a.L * __builtin_rint(qy * a.invL);
                        if (i1 + 1 < len) { bx = nBx; by = nBy; bz = nBz; }
                        if (i1 + 2 < len) { // fetched now, used one move later
                            nBx = ld_coherent(Rg + 3 * (first + i1 + 2));
                            nBy = ld_coherent(Rg + 3 * (first + i1 + 2) + 1);
                            nBz = ld_coherent(Rg + 3 * (first + i1 + 2) + 2);
                        }
                    }
                    if (lane == 0) {
                        sh.bc[0] = acc ? 1.0 : 0.0;
                        sh.bc[1] = qx; sh.bc[2] = qy; sh.bc[3] = qz;
                        sh.bc[4] = bx; sh.bc[5] = by; sh.bc[6] = bz;
                        sh.bc[7] = (fmax(fmax(fabs(qx), fabs(qy)), fmax(fabs(bx), fabs(by))) <= a.edge) ? 1.0 : 0.0;
                    }
                }
                __syncthreads(); // #2: the leader's verdict and the next probes are in LDS

                const bool acc = uniform((int)(sh.bc[0] != 0.0)) != 0;
                const bool upd = acc && (tid == tl);
                x[0] = upd ? Qx : x[0]; y[0] = upd ? Qy : y[0]; z[0] = upd ? Qz : z[0];
                if (upd) { Rg[3 * n] = Qx; Rg[3 * n + 1] = Qy; Rg[3 * n + 2] = Qz; }
                Qx = uniform_d(sh.bc[1]); Qy = uniform_d(sh.bc[2]); Qz = uniform_d(sh.bc[3]);
                Bx = uniform_d(sh.bc[4]); By = uniform_d(sh.bc[5]); Bz = uniform_d(sh.bc[6]);
                interior = uniform((int)(sh.bc[7] != 0.0)) != 0;
                if (hasB) {
                    if (cross) { rotate1<S>(x, y, z); rot = (rot + 1 == S) ? 0 : rot + 1; tl = 0; }
                    else tl++;
                }
            }
        }
        if (tid == 0) {
            SweepRec r; r.E = E; r.accepted = jacc; r.pad = 0;
            a.rec[(size_t)rep * a.chunk + sw] = r;
        }
    }
}

// ---------------------------------------------------------------------------------
// C: chain bookkeeping of sMC for the sweeps of one launch, in sweep order
// (SMC.c:116-117, 194-195, 210-211, 244-250); one thread per replica
// ---------------------------------------------------------------------------------
__global__ void finalize_kernel(DevCtx c, int nsweeps, int production, int sweep_base, int first_production)
{
    const int rep = blockIdx.x * blockDim.x + threadIdx.x;
    if (rep >= c.nrep) return;
    ObsRec ob = c.obs[rep];
    if (production && first_production) { // entry 0 of the energy series, SMC.c:48 / 194
        const double e0 = ob.Ecur + c.c3NT2;
        ob.sumE = e0; ob.sumE2 = e0 * e0; ob.nsamp = 1.0;
        if (c.Eseries) c.Eseries[(size_t)rep * c.series_stride] = ob.Ecur;
    }
    for (int sw = 0; sw < nsweeps; sw++) {
        const SweepRec r = c.rec[(size_t)rep * c.chunk + sw];
        ob.Ecur = r.E;
        if (production) {
            const double e = r.E + c.c3NT2;
            ob.sumE += e; ob.sumE2 += e * e; ob.nsamp += 1.0;
            ob.accepted += (double)r.accepted;
            if (c.Eseries) {
                const size_t o = (size_t)rep * c.series_stride + sweep_base + sw;
                c.Eseries[o + 1] = r.E;
                c.jjseries[o] = r.accepted;
            }
        } else {
            ob.therm_accepted += (double)r.accepted;
        }
    }
    c.obs[rep] = ob;
}

// ---------------------------------------------------------------------------------
// H: localDensityAndMobility's cell counts, summed over x,y (SMC.c:912-927), taken from
// the positions in memory between two sweep launches; 256 threads per replica
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hist_kernel(DevCtx c)
{
    __shared__ unsigned zh[256];
    __shared__ unsigned oob;
    const int rep = blockIdx.x, tid = threadIdx.x;
    zh[tid] = 0u;
    if (tid == 0) oob = 0u;
    __syncthreads();
    const double *Rg = c.R + (size_t)rep * 3 * c.N;
    const int Nc = c.Ncx * c.Ncx * c.Ncz;
    for (int l = tid; l < c.N; l += 256) {
        // floor() through uint8_t as the reference does (SMC.c:914-919)
        const int ci = (int)floor((Rg[3 * l] / c.L + .5) * c.Ncx) & 0xff;
        const int cj = (int)floor((Rg[3 * l + 1] / c.L + .5) * c.Ncx) & 0xff;
        const int ck = (int)floor((Rg[3 * l + 2] / c.Lz + .5) * c.Ncz) & 0xff;
        const int cell = ci * c.Ncx * c.Ncz + cj * c.Ncz + ck;
        if (cell < Nc) {
            atomicAdd(&zh[cell % c.Ncz], 1u);
            if (c.D) { // the full Ncx x Ncx x Ncz occupancy and mobility counters
                atomicAdd(&c.D[(size_t)rep * Nc + cell], 1ull);
                int *rb = c.Rbin + (size_t)rep * c.N + l;
                if (*rb != cell) { atomicAdd(&c.Mu[(size_t)rep * Nc + cell], 1ull); *rb = cell; }
            }
        } else {
            atomicAdd(&oob, 1u); // the reference writes D, Mu out of bounds here; its Rbin[n] = v is in bounds
            if (c.D) c.Rbin[(size_t)rep * c.N + l] = cell;
        }
    }
    __syncthreads();
    if (tid < c.Ncz && zh[tid]) c.zhist[(size_t)rep * c.Ncz + tid] += zh[tid];
    if (tid == 0) {
        c.obs[rep].gathers += 1.0;
        c.obs[rep].oob += (double)oob;
    }
}

// ---------------------------------------------------------------------------------
// virial pressure of one gather: pressure() SMC.c:696-720 + wallsPressure() SMC.c:862-895,
// the latter with the reference's own geometry (wall distance from z + L/2, no clamp,
// plane term once per site inside that site's cutoff); 256 threads per replica
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pressure_kernel(DevCtx c, int gather)
{
    __shared__ double part[4];
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = c.N;
    const double *Rg = c.R + (size_t)rep * 3 * N;
    const int half = N / 2;
    double acc = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double xi = Rg[3 * i], yi = Rg[3 * i + 1], zi = Rg[3 * i + 2];
        for (int d = 1; d <= half; d++) {
            if (2 * d == N && i >= half) break;
            int l = i + d; if (l >= N) l -= N;
            double dx = Rg[3 * l] - xi; dx = dx - c.L * __builtin_rint(dx * c.invL);
            double dy = Rg[3 * l + 1] - yi; dy = dy - c.L * __builtin_rint(dy * c.invL);
            const double dz = Rg[3 * l + 2] - zi;
            const double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c.cutoff2) {
                const double ir2 = 1.0 / dr2, ir6 = ir2 * ir2 * ir2;
                acc += 24.0 * ir6 - 48.0 * ir6 * ir6;
            }
        }
        if (c.flags & 0x1u) {
            double dz = zi + c.L / 2; // sic, SMC.c:880
            dz = dz - c.Lz * __builtin_rint(dz * c.invLz);
            const double iz2 = 1.0 / (dz * dz), iz6 = iz2 * iz2 * iz2;
            const double plane = 24.0 * c.b0 * iz6 - 48.0 * c.a0 * iz6 * iz6;
            const double dw = c.L / c.M;
            for (int m = 0; m < c.M2; m++) {
                double dx = xi - (m / c.M) * dw; dx = dx - c.L * __builtin_rint(dx * c.invL);
                double dy = yi - (m % c.M) * dw; dy = dy - c.L * __builtin_rint(dy * c.invL);
                const double dr2 = dx * dx + dy * dy + dz * dz;
                if (dr2 < c.cutoff2) {
                    const double ir2 = 1.0 / dr2, ir6 = ir2 * ir2 * ir2;
                    acc += 24.0 * c.W[2 * m + 1] * ir6 - 48.0 * c.W[2 * m] * ir6 * ir6 + plane;
                }
            }
        }
    }
    for (int m = 32; m >= 1; m >>= 1) acc += xchg(acc, m);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (tid == 0 && gather < c.pstride)
        c.Pseries[(size_t)rep * c.pstride + gather] = -(part[0] + part[1] + part[2] + part[3]) / (3 * c.L * c.L * c.Lz);
}

// ---------------------------------------------------------------------------------
// K5: energy + wallsEnergy per replica (SMC.c:626-646, 822-859).  Each pair is visited once.
// N <= 16384 (total_energy_zk): the replica's particles are ranked by z in LDS (18-bit z of the replica's own z range | index,
// bitonic sort) and rank a tests only the ranks after it whose z key lies within the cutoff -- in the slab geometries of the
// configurations 1/11 .. 1/40 of the N/2 partners of the plain walk (every smcx_upload and every energy check runs this kernel:
// 33 ms at 4096 x 2048 with the plain walk, 3.5 sweeps' worth).  Larger N (total_energy_kernel): thread i walks the N/2
// neighbours ahead of it on the ring of indices.  The sum is the same set of terms in another order (1e-13 relative).
// ---------------------------------------------------------------------------------
__device__ inline double wall_terms(const DevCtx &c, double xi, double yi, double zi)
{
    const double dz = wall_dz(c, zi);
    const double iz2 = 1.0 / (dz * dz), iz6 = iz2 * iz2 * iz2;
    double acc = c.a0 * iz6 * iz6 - c.b0 * iz6;
    const double dw = c.L / c.M;
    for (int m = 0; m < c.M2; m++) {
        double dx = xi - (m / c.M) * dw; dx = dx - c.L * __builtin_rint(dx * c.invL);
        double dy = yi - (m % c.M) * dw; dy = dy - c.L * __builtin_rint(dy * c.invL);
        const double dr2 = dx * dx + dy * dy + dz * dz;
        if (dr2 < c.cutoff2) {
            const double ir2 = 1.0 / dr2, ir6 = ir2 * ir2 * ir2;
            acc += c.W[2 * m] * ir6 * ir6 - c.W[2 * m + 1] * ir6;
        }
    }
    return acc;
}

__global__ void __launch_bounds__(256) total_energy_kernel(DevCtx c, double *out)
{
    __shared__ double part[4];
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = c.N;
    const double *Rg = c.R + (size_t)rep * 3 * N;
    const bool walls = (c.flags & 0x1u) != 0;
    const int half = N / 2;
    double acc = 0.0;
    for (int i = tid; i < N; i += 256) {
        const double xi = Rg[3 * i], yi = Rg[3 * i + 1], zi = Rg[3 * i + 2];
        for (int d = 1; d <= half; d++) {
            if (2 * d == N && i >= half) break; // antipodal pairs once
            int l = i + d; if (l >= N) l -= N;
            double dx = Rg[3 * l] - xi; dx = dx - c.L * __builtin_rint(dx * c.invL);
            double dy = Rg[3 * l + 1] - yi; dy = dy - c.L * __builtin_rint(dy * c.invL);
            const double dz = Rg[3 * l + 2] - zi;
            const double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c.cutoff2) {
                const double ir2 = 1.0 / dr2, ir6 = ir2 * ir2 * ir2;
                acc += ir6 * ir6 - ir6;
            }
        }
        if (walls) acc += wall_terms(c, xi, yi, zi);
    }
    for (int m = 32; m >= 1; m >>= 1) acc += xchg(acc, m);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (tid == 0) out[rep] = 4.0 * (part[0] + part[1] + part[2] + part[3]);
}

// dynamic LDS: npad keys (npad = the power of two >= N, at least 256); the reductions borrow the first 64 bytes
__global__ void __launch_bounds__(256) total_energy_zk(DevCtx c, double *out, int npad)
{
    extern __shared__ unsigned zkey[];
    constexpr int IB = 14, ZQ = (1 << 18) - 2; // index bits; the highest z key (a particle's key is never the padding's ~0u)
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = c.N;
    const double *Rg = c.R + (size_t)rep * 3 * N;
    const bool walls = (c.flags & 0x1u) != 0;
    double *red = reinterpret_cast<double *>(zkey);

    // the replica's z range (without walls nothing confines z)
    double lo = 1e300, hi = -1e300;
    for (int i = tid; i < N; i += 256) { const double z = Rg[3 * i + 2]; lo = fmin(lo, z); hi = fmax(hi, z); }
    for (int m = 32; m >= 1; m >>= 1) { lo = fmin(lo, xchg(lo, m)); hi = fmax(hi, xchg(hi, m)); }
    if (lane == 0) { red[wave] = lo; red[4 + wave] = hi; }
    __syncthreads();
    lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
    hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    __syncthreads();
    const double scale = (double)ZQ / fmax(hi - lo, 1e-300);
    // key differences above `reach` mean dz > cutoff: floor() moves a difference by less than one unit
    const double reachd = sqrt(c.cutoff2) * scale + 1.0;
    const unsigned reach = reachd < (double)ZQ ? (unsigned)reachd : (unsigned)ZQ;
    for (int i = tid; i < npad; i += 256) {
        unsigned k = ~0u;
        if (i < N) {
            const double q = (Rg[3 * i + 2] - lo) * scale;
            const unsigned zq = q > 0.0 ? (q < (double)ZQ ? (unsigned)q : (unsigned)ZQ) : 0u; // (a NaN ranks first: the sum is NaN anyway)
            k = (zq << IB) | (unsigned)i;
        }
        zkey[i] = k;
    }
    __syncthreads();
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < npad / 2; t += 256) {
                const int i = 2 * t - (t & (j - 1)), q = i + j;
                const unsigned a = zkey[i], b = zkey[q];
                if ((a > b) == ((i & k) == 0)) { zkey[i] = b; zkey[q] = a; }
            }
            __syncthreads();
        }

    double acc = 0.0;
    for (int a = tid; a < N; a += 256) { // the N particles hold the ranks below N
        const unsigned ka = zkey[a];
        const int i = (int)(ka & ((1u << IB) - 1u));
        const unsigned za = ka >> IB;
        const double xi = Rg[3 * i], yi = Rg[3 * i + 1], zi = Rg[3 * i + 2];
        for (int b = a + 1; b < N; b++) {
            const unsigned kb = zkey[b];
            if ((kb >> IB) - za > reach) break;
            const int l = (int)(kb & ((1u << IB) - 1u));
            double dx = Rg[3 * l] - xi; dx = dx - c.L * __builtin_rint(dx * c.invL);
            double dy = Rg[3 * l + 1] - yi; dy = dy - c.L * __builtin_rint(dy * c.invL);
            const double dz = Rg[3 * l + 2] - zi;
            const double dr2 = dx * dx + dy * dy + dz * dz;
            if (dr2 < c.cutoff2) {
                const double ir2 = 1.0 / dr2, ir6 = ir2 * ir2 * ir2;
                acc += ir6 * ir6 - ir6;
            }
        }
        if (walls) acc += wall_terms(c, xi, yi, zi);
    }
    __syncthreads(); // every rank has been read: the keys' first bytes take the partial sums
    for (int m = 32; m >= 1; m >>= 1) acc += xchg(acc, m);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) out[rep] = 4.0 * (red[0] + red[1] + red[2] + red[3]);
}

// ---------------------------------------------------------------------------------
// teacher-forced evaluator: one wavefront per replica, neighbours streamed from
// global memory through the same pair/wall/reduction code as the sweep
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
eval_moves_kernel(DevCtx c, const int *nsel, const double *prop, double *out)
{
    __shared__ RoleTable rt;
    const int rep = blockIdx.x, lane = threadIdx.x;
    const int N = c.N;
    const double *Rg = c.R + (size_t)rep * 3 * N;
    const int n = nsel[rep];
    const double Px = Rg[3 * n], Py = Rg[3 * n + 1], Pz = Rg[3 * n + 2];
    const double Qx = prop[3 * rep], Qy = prop[3 * rep + 1], Qz = prop[3 * rep + 2];
    Geo g; g.L = c.L; g.invL = c.invL; g.cutoff2 = c.cutoff2;
    fill_roles(c, rt, lane);
    __syncthreads();
    Acc8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    double side[4];
    for (int l = lane; l < N; l += 64) {
        const double xl = Rg[3 * l], yl = Rg[3 * l + 1], zl = Rg[3 * l + 2];
        pair_eval(g, Qx, Qy, Qz, xl, yl, zl, l != n, v.a0, v.a1, v.a2, v.a3);
        pair_eval(g, Px, Py, Pz, xl, yl, zl, l != n, v.b0, v.b1, v.b2, v.b3);
    }
    special_block(g, rt, lane, rt.role[lane], true, true, false, Px, Py, Pz, Qx, Qy, Qz, Px, Py, Pz,
                  v, side);
    const double r = reduce8(v.a0, v.a1, v.a2, v.a3, v.b0, v.b1, v.b2, v.b3, lane);
    double tot[8];
#pragma unroll
    for (int j = 0; j < 8; j++) tot[j] = rdlane(r, 8 * j);
    if (lane == 0) {
        double *o = out + 8 * rep;
        o[0] = 4.0 * tot[4]; o[1] = tot[5]; o[2] = tot[6]; o[3] = tot[7];
        o[4] = 4.0 * tot[0]; o[5] = tot[1]; o[6] = tot[2]; o[7] = tot[3];
    }
}

__global__ void obs_op_kernel(DevCtx c, double *save, int op)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= c.nrep) return;
    ObsRec &o = c.obs[r];
    if (op == 0) {
        save[r] = o.Ecur;
        o.accepted = 0.0; o.nsamp = 0.0; o.sumE = 0.0; o.sumE2 = 0.0;
        o.therm_accepted = 0.0; o.gathers = 0.0; o.oob = 0.0;
        for (int k = 0; k < c.Ncz; k++) c.zhist[(size_t)r * c.Ncz + k] = 0ull;
    } else {
        o.Ecur = save[r];
    }
}

__global__ void pack_obs_kernel(DevCtx c, double *dst)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nrec = c.nrep * 8;
    if (i < nrec) dst[i] = reinterpret_cast<const double *>(c.obs)[i];
    else if (i < nrec + c.nrep * c.Ncz) dst[i] = (double)c.zhist[i - nrec];
}

// ---------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------
typedef void (*sweep_fn)(SweepArgs, DevCtx, int, double);

// MINW (second __launch_bounds__ argument, waves per SIMD) caps the register
// allocation: 96 VGPRs of positions at S=16 fit 3 waves/SIMD, S=32 fits 2, S=64 one.
// G = slots per cutoff-test group (more independent chains for the single-wave S=64).
// (S, MINW, G) for one wavefront per replica; (S, WPR, MINW, G) for several
#define SMCX_FP64_TABLE1(X) X(1, 4, 1) X(2, 4, 2) X(4, 4, 2) X(8, 4, 2) X(16, 3, 2) X(32, 2, 2) X(64, 1, 2)
#define SMCX_FP64_TABLE(X) X(8, 2, 4, 2) X(16, 2, 3, 2) X(32, 2, 2, 2) X(4, 4, 4, 2) X(8, 4, 4, 2) X(16, 4, 3, 2) \
                           X(32, 4, 2, 2) X(8, 8, 4, 2) X(16, 8, 3, 2) X(32, 8, 2, 2) X(16, 16, 4, 2) X(32, 16, 2, 2)

static sweep_fn lookup(int S, int WPR, bool lead = true)
{
#define SMCX_CASE1(s, m, gg) if (S == s && WPR == 1) return sweep_kernel<s, 1, m, gg>;
#define SMCX_CASE(s, w, m, gg) if (S == s && WPR == w) return lead ? sweep_kernel_lead<s, w, m, gg> : sweep_kernel<s, w, m, gg>;
    SMCX_FP64_TABLE1(SMCX_CASE1)
    SMCX_FP64_TABLE(SMCX_CASE)
#undef SMCX_CASE
#undef SMCX_CASE1
    return nullptr;
}

// the launched instantiation as rocprofv3 prints it
const char *fp64_kernel_name(int S, int WPR, bool lead)
{
#define SMCX_CASE1(s, m, gg) if (S == s && WPR == 1) return "smcx::sweep_kernel<" #s ", 1, " #m ", " #gg ">";
#define SMCX_CASE(s, w, m, gg) if (S == s && WPR == w) return lead ? "smcx::sweep_kernel_lead<" #s ", " #w ", " #m ", " #gg ">" \
                                                                  : "smcx::sweep_kernel<" #s ", " #w ", " #m ", " #gg ">";
    SMCX_FP64_TABLE1(SMCX_CASE1)
    SMCX_FP64_TABLE(SMCX_CASE)
#undef SMCX_CASE
#undef SMCX_CASE1
    return "";
}

bool geometry_supported(int S, int WPR) { return lookup(S, WPR) != nullptr || mx_supported(S, WPR); }
bool fp64_supported(int S, int WPR) { return lookup(S, WPR) != nullptr; }

hipError_t launch_rng_prepass(const DevCtx &c, int nsweeps, double A, hipStream_t st)
{
    hipLaunchKernelGGL(rng_prepass_kernel, dim3(c.nrep), dim3(256), 0, st, c, nsweeps, A);
    return hipGetLastError();
}

// The one place that decides which sweep kernel serves a handle (see KernelPlan, smcx_kernels.h).
//  * fp64 kernels when asked for (tune.kernel == 1) or when no screened kernel is built for (S, WPR); with several
//    wavefronts their leader/follower form where it measured faster (profiles/r01_leader_follower.log: S <= 16, >= 4 waves);
//  * else the ladder mx -> mi -> ma -> mb -> mc, each step needing the one below: one wavefront per replica and a z
//    unit that covers the box (mi), 32 S < N <= 64 S with the standard unit (ma), 64 cells per lane (mb), a box the byte
//    screen resolves (mc); several wavefronts per replica go from mx straight to the several-wave mc forms.
bool plan_kernel(int N, int M2, double L, double Lz, double cutoff2, int S, int WPR, const Tune &t, KernelPlan *out)
{
    KernelPlan p;
    p.S = S; p.WPR = WPR; p.tune = t;
    if (mt_built(S, WPR, N, M2, L, Lz, cutoff2) && (t.kernel == FORM_MT || t.kernel == 0 || t.kernel == 2) &&
        ma_cap(t, S) >= FORM_MC) { // the two-team geometries (16 x 2, 64 x 8) exist for this form only
        p.form = FORM_MT;
        p.name = ma_kernel_name(FORM_MT, S, WPR);
        *out = p;
        return true;
    }
    const bool have64 = lookup(S, WPR) != nullptr, havemx = mx_supported(S, WPR);
    if (!have64 && !havemx) return false;
    if ((t.kernel == 1 && have64) || !havemx) {
        p.form = FORM_FP64;
        p.lead = WPR > 1 && (t.lead == 1 || (t.lead < 0 && S <= 16 && WPR >= 4));
        p.name = fp64_kernel_name(S, WPR, p.lead);
        *out = p;
        return true;
    }
    if (t.kernel == FORM_MT) return false; // asked for the two-team form in a geometry that has none
    int cap = t.kernel >= FORM_MX ? t.kernel : FORM_MC;
    p.form = FORM_MX;
    if (WPR == 1 && cap >= FORM_MI) {
        const int zs = mi_built(S, L, Lz, cutoff2);
        if (zs) { p.form = FORM_MI; p.zs = zs; }
    }
    if (p.form == FORM_MI && p.zs == 4 && ma_built(S, N, M2)) {
        const int top = cap < ma_cap(t, S) ? cap : ma_cap(t, S);
        if (top >= FORM_MA) p.form = FORM_MA;
        if (top >= FORM_MB && S == 64) p.form = FORM_MB;
        if (top >= FORM_MC && mc_box_supported(L, Lz, cutoff2)) p.form = FORM_MC;
    }
    if (p.form == FORM_MX && WPR > 1 && cap >= FORM_MC && ma_cap(t, S) >= FORM_MC && mcw_built(S, WPR, N, M2, L, Lz, cutoff2))
        p.form = FORM_MC;
    if (p.form == FORM_MX) {
        p.mz = mx_lds_z(S, WPR, Lz, t.mz);
        p.name = mx_kernel_name(S, WPR, p.mz);
    } else if (p.form == FORM_MI) {
        p.name = mi_kernel_name(S, p.zs);
    } else {
        p.name = ma_kernel_name(p.form, S, WPR);
    }
    *out = p;
    return true;
}

hipError_t launch_sweeps(const DevCtx &c, const KernelPlan &pl, int nsweeps, double A, hipStream_t st, SweepTimer *tm)
{
    const int S = pl.S, WPR = pl.WPR;
    SweepArgs a;
    a.N = c.N; a.chunk = c.chunk;
    a.L = c.L; a.invL = c.invL; a.cutoff2 = c.cutoff2; a.invT = c.invT;
    a.R = c.R; a.displ = c.displ; a.uni = c.uni; a.offs = c.offs; a.obs = c.obs; a.rec = c.rec;
    a.edge = c.L / 2 - sqrt(c.cutoff2); // |x|,|y| up to here: no pair needs the periodic image
    a.clk = c.clk;
#ifdef SMCX_CHECK
    a.dbg = c.dbg;
#endif
    if (pl.form == FORM_MT) return launch_sweeps_mt(a, c, pl, nsweeps, A, st, tm);
    if (pl.form >= FORM_MI && WPR == 1) return launch_sweeps_mi(a, c, pl, nsweeps, A, st, tm);
    if (pl.form == FORM_MC) {
        if (!c.Rs || !c.loc) return hipErrorInvalidValue;
        return launch_sweeps_mcw(a, c, WPR, nsweeps, A, st, tm);
    }
    hipError_t rc = tm ? tm->mark(st) : hipSuccess;
    if (rc != hipSuccess) return rc;
    if (pl.form == FORM_MX) {
        rc = launch_sweeps_mx(a, c, pl, nsweeps, A, st);
    } else {
        sweep_fn f = lookup(S, WPR, pl.lead);
        if (!f) return hipErrorInvalidValue;
        hipLaunchKernelGGL(f, dim3(c.nrep), dim3(64 * WPR), 0, st, a, c, nsweeps, A);
        rc = hipGetLastError();
    }
    if (rc == hipSuccess && tm) rc = tm->mark(st);
    return rc;
}

hipError_t launch_finalize(const DevCtx &c, int nsweeps, int production, int sweep_base,
                           int first_production, hipStream_t st)
{
    hipLaunchKernelGGL(finalize_kernel, dim3((c.nrep + 127) / 128), dim3(128), 0, st, c, nsweeps,
                       production, sweep_base, first_production);
    return hipGetLastError();
}

hipError_t launch_hist(const DevCtx &c, hipStream_t st)
{
    hipLaunchKernelGGL(hist_kernel, dim3(c.nrep), dim3(256), 0, st, c);
    return hipGetLastError();
}

hipError_t launch_pressure(const DevCtx &c, int gather, hipStream_t st)
{
    hipLaunchKernelGGL(pressure_kernel, dim3(c.nrep), dim3(256), 0, st, c, gather);
    return hipGetLastError();
}

hipError_t launch_total_energy(const DevCtx &c, double *out, hipStream_t st)
{
    if (c.N <= 16384) {
        int npad = 256;
        while (npad < c.N) npad <<= 1;
        hipLaunchKernelGGL(total_energy_zk, dim3(c.nrep), dim3(256), npad * sizeof(unsigned), st, c, out, npad);
    } else
        hipLaunchKernelGGL(total_energy_kernel, dim3(c.nrep), dim3(256), 0, st, c, out);
    return hipGetLastError();
}

hipError_t launch_eval_moves(const DevCtx &c, const int *nsel, const double *prop, double *out,
                             hipStream_t st)
{
    hipLaunchKernelGGL(eval_moves_kernel, dim3(c.nrep), dim3(64), 0, st, c, nsel, prop, out);
    return hipGetLastError();
}

hipError_t launch_obs_op(const DevCtx &c, double *save, int op, hipStream_t st)
{
    hipLaunchKernelGGL(obs_op_kernel, dim3((c.nrep + 255) / 256), dim3(256), 0, st, c, save, op);
    return hipGetLastError();
}

hipError_t launch_pack_obs(const DevCtx &c, double *dst, hipStream_t st)
{
    const int total = c.nrep * 8 + c.nrep * c.Ncz;
    hipLaunchKernelGGL(pack_obs_kernel, dim3((total + 255) / 256), dim3(256), 0, st, c, dst);
    return hipGetLastError();
}

} // namespace smcx
