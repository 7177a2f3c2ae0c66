// smcx_kernels.hip -- gfx950 kernels of the SMC engine and their launchers.
//
// Kernels:
//   rng_prepass_kernel    per replica and sweep: the 4N+1 glibc rand() outputs one
//                         oneParticleMoves call consumes (SMC.c:284, 290, 335), turned
//                         into the 3N Box-Muller displacements (matematicose.c:183-193),
//                         the visiting offset and the N acceptance uniforms
//   sweep_kernel<S,WPR>   the hot path: the trial moves of SMC.c:292-348 for K sweeps,
//                         positions resident in registers, Metropolis step, incremental
//                         energy (SMC.c:340-341), histogram (SMC.c:912-927)
//   total_energy_kernel   energy + wallsEnergy (SMC.c:626-646, 822-859)
//   eval_moves_kernel     teacher-forced Um,Fm,Un,Fn for one particle per replica
#include "smcx_device.hpp"
#include "smcx_kernels.h"

namespace smcx {

// ---------------------------------------------------------------------------------
// R: random numbers of `nsweeps` sweeps for every replica (256 threads per replica)
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rng_prepass_kernel(DevCtx c, int nsweeps, double A)
{
    const int rep = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int N = c.N;
    const int D = 4 * N + 1; // rand() calls per sweep: 3N normals, 1 offset, N uniforms
    uint32_t *raw = c.raw + (size_t)rep * c.rawStride;
    const double sigma = sqrt(2.0 * A); // SMC.c:284

    // lane j < 31 of wave 0 holds state word r[i-31+j]; `left` outputs at the top of
    // the last generated block of 31 have not been consumed yet
    uint32_t hist = (lane < 31) ? c.rng[rep * 32 + lane] : 0u;
    int left = uniform((int)c.rng[rep * 32 + 31]);

    for (int s = 0; s < nsweeps; s++) {
        if (wave == 0) {
            const uint32_t carry = __shfl(hist, 31 - left + lane, 64);
            if (lane < left) raw[lane] = carry >> 1;
            int have = left;
            while (have < D) {
                hist = rand_block(hist, lane);
                if (lane < 31) raw[have + lane] = hist >> 1;
                have += 31;
            }
            left = have - D;
        }
        __syncthreads();
        double *displ = c.displ + ((size_t)rep * c.chunk + s) * 3 * N;
        double *uni = c.uni + ((size_t)rep * c.chunk + s) * N;
        // Box-Muller pairs (matematicose.c:187-192; the second output swaps x1 and x2)
        for (int p = tid; p < (3 * N) / 2; p += 256) {
            const double x1 = (double)raw[2 * p] * (1.0 / 2147483648.0);
            const double x2 = (double)raw[2 * p + 1] * (1.0 / 2147483648.0);
            displ[2 * p] = sigma * sqrt(-2.0 * log(1.0 - x1)) * cos(2.0 * M_PI * x2);
            displ[2 * p + 1] = sigma * sqrt(-2.0 * log(1.0 - x2)) * sin(2.0 * M_PI * x1);
        }
        for (int i = tid; i < N; i += 256) // acceptance uniforms, SMC.c:335
            uni[i] = (double)raw[3 * N + 1 + i] / 2147483647.0;
        if (tid == 0) // SMC.c:290-294: the sweep starts at particle offset % N
            c.offs[(size_t)rep * c.chunk + s] = (int)(raw[3 * N] % (uint32_t)N);
        __syncthreads();
    }
    if (wave == 0) {
        if (lane < 31) c.rng[rep * 32 + lane] = hist;
        if (lane == 31) c.rng[rep * 32 + 31] = (uint32_t)left;
    }
}

// ---------------------------------------------------------------------------------
// S1: the sweep
// ---------------------------------------------------------------------------------

// rotate the register-resident particle slots by one: slot j <- slot j+1
template <int S>
__device__ __forceinline__ void rotate1(double (&x)[S], double (&y)[S], double (&z)[S])
{
    if constexpr (S > 1) {
        const double tx = x[0], ty = y[0], tz = z[0];
#pragma unroll
        for (int k = 0; k + 1 < S; k++) { x[k] = x[k + 1]; y[k] = y[k + 1]; z[k] = z[k + 1]; }
        x[S - 1] = tx; y[S - 1] = ty; z[S - 1] = tz;
    }
}

__device__ __forceinline__ double ld_coherent(const double *p)
{
    // L1-bypassing load (global_load ... sc1): positions written through by another wave
    unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)b);
}

template <int WPR> struct SweepShared {
    RoleTable roles;
    double red[2][WPR][8];
    double side[2][2][4];
    unsigned zh[256];
    unsigned oob;
};

// combine the eight wave totals (and, with several waves, the waves) into
// tot[8], identical in every lane of the workgroup; also fetches the two side terms
template <int WPR>
__device__ __forceinline__ void combine(SweepShared<WPR> &sh, int &par, int lane, int wave,
                                        const double (&v)[8], const double (&side)[4],
                                        double (&tot)[8], double (&sOld)[4], double (&sNew)[4])
{
    const double r = reduce8(v, lane);
    if constexpr (WPR == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) tot[j] = rdlane(r, 8 * j);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sOld[j] = rdlane(side[j], SIDE_LANE_OLD);
            sNew[j] = rdlane(side[j], SIDE_LANE_NEW);
        }
    } else {
        if ((lane & 7) == 0) sh.red[par][wave][lane >> 3] = r;
        if (wave == 0 && (lane == SIDE_LANE_OLD || lane == SIDE_LANE_NEW)) {
#pragma unroll
            for (int j = 0; j < 4; j++) sh.side[par][lane - SIDE_LANE_OLD][j] = side[j];
        }
        __syncthreads();
        // lane i reads the partial of wave i>>3 for value i&7, then the groups are summed
        double t = 0.0;
        if (lane < 8 * WPR) t = sh.red[par][lane >> 3][lane & 7];
        if constexpr (WPR > 8) t += sh.red[par][(lane >> 3) + 8][lane & 7];
        if constexpr (WPR >= 8) t += xchg(t, 32);
        if constexpr (WPR >= 4) t += xchg(t, 16);
        t += xchg(t, 8);
#pragma unroll
        for (int j = 0; j < 8; j++) tot[j] = rdlane(t, j);
#pragma unroll
        for (int j = 0; j < 4; j++) { sOld[j] = sh.side[par][0][j]; sNew[j] = sh.side[par][1][j]; }
        par ^= 1;
    }
}

template <int S, int WPR, int MINW>
__global__ void __launch_bounds__(64 * WPR, MINW)
sweep_kernel(DevCtx c, int nsweeps, double A, int production, int gather_lapse, int sweep_base,
             int first_production)
{
    constexpr int T = 64 * WPR;
    __shared__ SweepShared<WPR> sh;

    const int rep = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int N = c.N;

    double *Rg = c.R + (size_t)rep * 3 * N;

    // ---- register-resident positions: particle l in lane l % T, slot l / T ------
    double x[S], y[S], z[S];
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int l = k * T + tid;
        if (l < N) { x[k] = Rg[3 * l]; y[k] = Rg[3 * l + 1]; z[k] = Rg[3 * l + 2]; }
        else { x[k] = 0.0; y[k] = 0.0; z[k] = FAR_PAD; }
    }
    int rot = 0; // register slot j holds logical slot (j + rot) % S

    for (int i = tid; i < 256; i += T) sh.zh[i] = 0u;
    if (tid == 0) sh.oob = 0u;
    if (wave == 0) fill_roles(c, sh.roles, lane);
    __syncthreads();
    const int role = (wave == 0) ? sh.roles.role[lane] : -1;

    Geo g; g.L = c.L; g.invL = c.invL; g.cutoff2 = c.cutoff2;

    double E = c.obs[rep].Ecur; // identical in every lane
    if (production && first_production && tid == 0) { // entry 0 of the energy series, SMC.c:48/194
        const double e0 = E + c.c3NT2;
        c.obs[rep].sumE = e0; c.obs[rep].sumE2 = e0 * e0; c.obs[rep].nsamp = 1.0;
        if (c.Eseries) c.Eseries[(size_t)rep * c.series_stride] = E;
    }
    int gathers = 0;
    int par = 0;

    const double AoT = A * c.invT;         // SMC.c:307-309 (A/T)
    const double Ao4T = A * 0.25 * c.invT; // SMC.c:327 (A/(4T))

    for (int sw = 0; sw < nsweeps; sw++) {
        // ---- H: density histogram before this sweep's moves (SMC.c:137-141) --------
        if (production && ((sweep_base + sw + 1) % gather_lapse == 0)) {
#pragma unroll
            for (int k = 0; k < S; k++) {
                int ls = k + rot; if (ls >= S) ls -= S;
                if (ls * T + tid < N) {
                    const int ci = (int)floor((x[k] / c.L + .5) * c.Ncx) & 0xff;
                    const int cj = (int)floor((y[k] / c.L + .5) * c.Ncx) & 0xff;
                    const int ck = (int)floor((z[k] / c.Lz + .5) * c.Ncz) & 0xff;
                    const int cell = ci * c.Ncx * c.Ncz + cj * c.Ncz + ck;
                    if (cell < c.Ncx * c.Ncx * c.Ncz) atomicAdd(&sh.zh[cell % c.Ncz], 1u);
                    else atomicAdd(&sh.oob, 1u);
                }
            }
            gathers++;
        }

        const double *displ = c.displ + ((size_t)rep * c.chunk + sw) * 3 * N;
        const double *uni = c.uni + ((size_t)rep * c.chunk + sw) * N;
        const int n0 = uniform(c.offs[(size_t)rep * c.chunk + sw]);

        int jacc = 0;
        // the visiting order n0..N-1, 0..n0-1 (SMC.c:292-294) is two ascending runs
        for (int run = 0; run < 2; run++) {
            const int first = run == 0 ? n0 : 0;
            const int len = run == 0 ? N - n0 : n0;
            if (len == 0) continue;
            const int vbase = run == 0 ? 0 : N - n0;
            int ks = first / T;
            while (rot != ks) { rotate1<S>(x, y, z); rot = (rot + 1 == S) ? 0 : rot + 1; }
            int tl = first - ks * T; // owner thread of the current particle (always slot 0)

            // prologue: Um,Fm of the run's first particle (B sums only)
            double Px, Py, Pz;
            if constexpr (WPR == 1) {
                Px = rdlane(x[0], tl); Py = rdlane(y[0], tl); Pz = rdlane(z[0], tl);
            } else {
                Px = ld_coherent(Rg + 3 * first); Py = ld_coherent(Rg + 3 * first + 1);
                Pz = ld_coherent(Rg + 3 * first + 2);
            }
            double Um, Fmx, Fmy, Fmz;
            {
                double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, side[4], tot[8], s0[4], s1[4];
                fused_pass<S>(g, x, y, z, 0.0, 0.0, FAR_PROBE, Px, Py, Pz, true, tid == tl, false, v);
                if (wave == 0)
                    special_block(c, g, sh.roles, lane, role, false, true, false, Px, Py, Pz, 0.0, 0.0,
                                  FAR_PROBE, Px, Py, Pz, v, side);
                else side[0] = side[1] = side[2] = side[3] = 0.0;
                combine<WPR>(sh, par, lane, wave, v, side, tot, s0, s1);
                Um = 4.0 * tot[4]; Fmx = tot[5]; Fmy = tot[6]; Fmz = tot[7];
            }

            double bdx = 0.0, bdy = 0.0, bdz = 0.0, bu = 2.0;
            for (int i = 0; i < len; i++) {
                const int n = first + i;
                if ((i & 63) == 0) { // this wave's next 64 displacements / uniforms
                    if (i + lane < len) {
                        const int pn = n + lane;
                        bdx = displ[3 * pn]; bdy = displ[3 * pn + 1]; bdz = displ[3 * pn + 2];
                        bu = uni[vbase + i + lane];
                    }
                }
                const int j = i & 63;
                // proposal, SMC.c:307-316
                const double dX = Fmx * AoT + rdlane(bdx, j);
                const double dY = Fmy * AoT + rdlane(bdy, j);
                const double dZ = Fmz * AoT + rdlane(bdz, j);
                double Qx = Px + dX, Qy = Py + dY, Qz = Pz + dZ;
                Qx = Qx - c.L * __builtin_rint(Qx * c.invL);
                Qy = Qy - c.L * __builtin_rint(Qy * c.invL);
                Qx = uniform_d(Qx); Qy = uniform_d(Qy); Qz = uniform_d(Qz);

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
                        Bx = uniform_d(ld_coherent(Rg + 3 * (n + 1)));
                        By = uniform_d(ld_coherent(Rg + 3 * (n + 1) + 1));
                        Bz = uniform_d(ld_coherent(Rg + 3 * (n + 1) + 2));
                    }
                }

                double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, side[4], tot[8], sOld[4], sNew[4];
                const bool exA0 = (tid == tl);
                const bool exB0 = (tid == tl) || (hasB && !cross && tid == tl + 1);
                const bool exB1 = cross && (tid == 0);
                fused_pass<S>(g, x, y, z, Qx, Qy, Qz, Bx, By, Bz, exA0, exB0, exB1, v);
                if (wave == 0)
                    special_block(c, g, sh.roles, lane, role, true, hasB, true, Px, Py, Pz, Qx, Qy, Qz,
                                  Bx, By, Bz, v, side);
                else side[0] = side[1] = side[2] = side[3] = 0.0;
                combine<WPR>(sh, par, lane, wave, v, side, tot, sOld, sNew);

                const double Un = 4.0 * tot[0], Fnx = tot[1], Fny = tot[2], Fnz = tot[3];
                // SMC acceptance, SMC.c:326-335
                const double gx = Fnx - Fmx, gy = Fny - Fmy, gz = Fnz - Fmz;
                const double deltaW = (gx * gx + gy * gy + gz * gz +
                                       2.0 * (gx * Fmx + gy * Fmy + gz * Fmz)) * Ao4T;
                const double arg = Un - Um +
                                   (dX * (Fnx + Fmx) + dY * (Fny + Fmy) + dZ * (Fnz + Fmz)) * 0.5 + deltaW;
                const double ap = exp(-arg * c.invT);
                const double u = rdlane(bu, j);
                const bool acc = (u < ap);
                const bool upd = acc && (tid == tl);
                x[0] = upd ? Qx : x[0]; y[0] = upd ? Qy : y[0]; z[0] = upd ? Qz : z[0];
                if constexpr (WPR > 1) {
                    if (upd) { Rg[3 * n] = Qx; Rg[3 * n + 1] = Qy; Rg[3 * n + 2] = Qz; }
                }
                if (acc) { E += Un - Um; jacc++; }

                if (hasB) { // next particle's Um,Fm = B sums + the (n, n+1) pair term
                    Um = 4.0 * (tot[4] + (acc ? sNew[0] : sOld[0]));
                    Fmx = tot[5] + (acc ? sNew[1] : sOld[1]);
                    Fmy = tot[6] + (acc ? sNew[2] : sOld[2]);
                    Fmz = tot[7] + (acc ? sNew[3] : sOld[3]);
                    Px = Bx; Py = By; Pz = Bz;
                    if (cross) {
                        rotate1<S>(x, y, z);
                        rot = (rot + 1 == S) ? 0 : rot + 1;
                        ks++; tl = 0;
                    } else {
                        tl++;
                    }
                }
            }
        }

        // ---- C: chain bookkeeping, SMC.c:116-117 / 194-195 / 210-211 ---------------
        if (tid == 0) {
            ObsRec &ob = c.obs[rep];
            if (production) {
                const double e = E + c.c3NT2;
                ob.sumE += e; ob.sumE2 += e * e; ob.nsamp += 1.0;
                ob.accepted += (double)jacc;
                if (c.Eseries) {
                    const size_t o = (size_t)rep * c.series_stride + sweep_base + sw;
                    c.Eseries[o + 1] = E;
                    c.jjseries[o] = jacc;
                }
            } else {
                ob.therm_accepted += (double)jacc;
            }
        }
    }

    // ---- write state back ---------------------------------------------------------
    if constexpr (WPR == 1) {
#pragma unroll
        for (int k = 0; k < S; k++) {
            int ls = k + rot; if (ls >= S) ls -= S;
            const int l = ls * T + tid;
            if (l < N) { Rg[3 * l] = x[k]; Rg[3 * l + 1] = y[k]; Rg[3 * l + 2] = z[k]; }
        }
    }
    __syncthreads();
    for (int i = tid; i < c.Ncz; i += T)
        if (sh.zh[i]) c.zhist[(size_t)rep * c.Ncz + i] += sh.zh[i];
    if (tid == 0) {
        ObsRec &ob = c.obs[rep];
        ob.Ecur = E;
        ob.gathers += (double)gathers;
        ob.oob += (double)sh.oob;
    }
}

// ---------------------------------------------------------------------------------
// K5: energy + wallsEnergy per replica.  Each pair is visited once: thread i
// walks the N/2 neighbours ahead of it on the ring of indices.
// ---------------------------------------------------------------------------------
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
        if (walls) {
            const double dz = wall_dz(c, zi);
            const double iz2 = 1.0 / (dz * dz), iz6 = iz2 * iz2 * iz2;
            acc += c.a0 * iz6 * iz6 - c.b0 * iz6;
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
        }
    }
    for (int m = 32; m >= 1; m >>= 1) acc += xchg(acc, m);
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (tid == 0) out[rep] = 4.0 * (part[0] + part[1] + part[2] + part[3]);
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
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0}, side[4];
    for (int l = lane; l < N; l += 64) {
        const double xl = Rg[3 * l], yl = Rg[3 * l + 1], zl = Rg[3 * l + 2];
        pair_eval(g, Qx, Qy, Qz, xl, yl, zl, l != n, v[0], v[1], v[2], v[3]);
        pair_eval(g, Px, Py, Pz, xl, yl, zl, l != n, v[4], v[5], v[6], v[7]);
    }
    special_block(c, g, rt, lane, rt.role[lane], true, true, false, Px, Py, Pz, Qx, Qy, Qz, Px, Py, Pz,
                  v, side);
    const double r = reduce8(v, lane);
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
typedef void (*sweep_fn)(DevCtx, int, double, int, int, int, int);

// MINW (second __launch_bounds__ argument, waves per SIMD) caps the register
// allocation: 96 VGPRs of positions at S=16 fit 3 waves/SIMD, S=32 fits 2, S=64 one.
static sweep_fn lookup(int S, int WPR)
{
#define SMCX_CASE(s, w, m) if (S == s && WPR == w) return sweep_kernel<s, w, m>;
    SMCX_CASE(1, 1, 4) SMCX_CASE(2, 1, 4) SMCX_CASE(4, 1, 4) SMCX_CASE(8, 1, 4)
    SMCX_CASE(16, 1, 3) SMCX_CASE(32, 1, 2) SMCX_CASE(64, 1, 1)
    SMCX_CASE(16, 2, 3) SMCX_CASE(32, 2, 2)
    SMCX_CASE(16, 4, 3) SMCX_CASE(32, 4, 2)
    SMCX_CASE(16, 8, 3) SMCX_CASE(32, 8, 2)
    SMCX_CASE(16, 16, 4) SMCX_CASE(32, 16, 2)
#undef SMCX_CASE
    return nullptr;
}

bool geometry_supported(int S, int WPR) { return lookup(S, WPR) != nullptr; }

hipError_t launch_rng_prepass(const DevCtx &c, int nsweeps, double A, hipStream_t st)
{
    hipLaunchKernelGGL(rng_prepass_kernel, dim3(c.nrep), dim3(256), 0, st, c, nsweeps, A);
    return hipGetLastError();
}

hipError_t launch_sweeps(const DevCtx &c, int S, int WPR, int nsweeps, double A, int production,
                         int gather_lapse, int sweep_base, int first_production, hipStream_t st)
{
    sweep_fn f = lookup(S, WPR);
    if (!f) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f, dim3(c.nrep), dim3(64 * WPR), 0, st, c, nsweeps, A, production,
                       gather_lapse, sweep_base, first_production);
    return hipGetLastError();
}

hipError_t launch_total_energy(const DevCtx &c, double *out, hipStream_t st)
{
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
