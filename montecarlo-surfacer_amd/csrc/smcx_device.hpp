// smcx_device.hpp -- device-side building blocks of the SMC sweep for gfx950.
//
// Design (DESIGN.md has the long version):
//  * one replica chain = one workgroup of WPR wavefronts (64 lanes each);
//    every lane owns S particles IN REGISTERS (x[S],y[S],z[S], fp64), particle
//    l lives in lane l % (64*WPR), register slot l / (64*WPR);
//  * a trial move needs sum_{l != n} pair(R[l], probe) for two probes (the
//    proposed position of particle n and the current position of the NEXT
//    particle n+1): one fused pass over the S register slots does both, so
//    positions are touched once per move and one 8-value wavefront reduction
//    serves two of the reference's four O(N) loops each (SMC.c:300-304,319-321);
//  * wall sites, the featureless plane and the (n, n+1) pair correction are
//    extra pseudo-neighbours evaluated by otherwise idle lanes of wave 0;
//  * the register file is rotated by one slot whenever the visiting order
//    crosses a slot boundary, so the moving particle is always in slot 0 and
//    no register is ever indexed at run time.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smcx {

constexpr double FAR_PAD = 1.0e150;    // z of padding particles (never inside any cutoff)
constexpr double FAR_PROBE = -1.0e150; // z of a disabled probe
// log-uniform stored for rand() == 0 (SMC.c:335 tests u < exp(-x/T)): exp(-y) > 0 exactly for
// y < 1075 ln 2 = 745.1332191019412, so "log u < -x/T" with this value accepts the same moves
constexpr double LOG_U_ZERO = -745.1332191019412;

struct ObsRec {                   // per-replica accumulators, 64 B
    double accepted;              // production accepted moves (exact in fp64 up to 2^53)
    double nsamp;                 // entries of the energy series seen so far
    double sumE, sumE2;           // running sums of (E + 3NT/2) and its square
    double Ecur;                  // current incremental energy (SMC.c:116-117,194-195)
    double therm_accepted;
    double gathers;
    double oob;
};

// what one sweep hands to the bookkeeping kernel: the chain's running energy after
// the sweep and its number of accepted moves (E[n+1] and jj[n] of SMC.c:194-195)
struct SweepRec {
    double E;
    int accepted;
    int pad;
};

struct DevCtx {
    int N, M, M2, nrep;
    int Ncx, Ncz;
    unsigned flags;
    int series_stride;            // entries per replica in Eseries (0 = off)
    double L, invL, Lz, invLz, halfLz, T, invT, cutoff2, a0, b0;
    double c3NT2;                 // 3*N*T/2 (SMC.c:211)
    double *R;                    // [nrep][3N] AoS, the reference layout
    const double *W;              // [2*M2]
    uint32_t *rng;                // [nrep][32]
    uint32_t *raw;                // [nrep][rawStride] raw rand() outputs of one sweep (pre-pass scratch)
    double *displ;                // [nrep][chunk][3N] Gaussian displacements, one block per sweep
    double *uni;                  // [nrep][chunk][N] log of the acceptance uniforms
    int *offs;                    // [nrep][chunk] first particle of each sweep (offset % N)
    int chunk;                    // sweeps of random numbers held by the scratch buffers
    ObsRec *obs;                  // [nrep]
    unsigned long long *zhist;    // [nrep][Ncz]
    double *Eseries;              // [nrep][series_stride] or null
    int *jjseries;                // [nrep][series_stride] or null
    long rawStride;
    const uint32_t *rngJump;      // [31][31] column-major: the rand() state advanced by rngQ blocks of 31 outputs, as a
    int rngQ;                     //          linear map over Z/2^32 (the four waves of the pre-pass start a quarter apart)
    SweepRec *rec;                // [nrep][chunk] per-sweep records of the last sweep launch
    // optional observables (SMCX_FLAG_FULL_HIST / SMCX_FLAG_PRESSURE), null when off
    unsigned long long *D;        // [nrep][Ncx*Ncx*Ncz] cell occupancy   (SMC.c:921)
    unsigned long long *Mu;       // [nrep][Ncx*Ncx*Ncz] cell changes     (SMC.c:922-925)
    int *Rbin;                    // [nrep][N] cell of each particle at the previous gather
    double *Pseries;              // [nrep][pstride] pressure + wallsPressure per gather (SMC.c:140)
    int pstride;
    unsigned long long *clk;      // [nrep][4] shader-clock and 100 MHz stamps of the last sweep launch (see SweepArgs)
    const double *wtab;           // [M2 + 1][4] wall table of sweep_kernel_ma: site x, y, W[2m], W[2m+1]; plane last
    double *Rs;                   // [nrep][4096][3] sweep_kernel_mb: positions in cell (z-sorted) order, or null
    unsigned short *loc;          // [nrep][N] sweep_kernel_mb: cell of each particle, or null
    unsigned *prio;               // [16384] sweep_kernel_mb/mc: progress of the wavefronts of each SIMD (issue priorities)
    int granule;                  // replicas the device runs at once with the plan's z-ordered kernel (0: unknown / not one of them)
    int windows;                  // 1: nrep is no multiple of granule -- launch groups run as windows of `granule` units (MaArgs2)
#ifdef SMCX_CHECK
    unsigned long long *dbg;      // [4] diagnostic build only (see SweepArgs)
#endif
};

// cluster analysis (smcx_lca.hip): per-replica counters n1, h2[16], h3[16], dropped stores
constexpr int LCA_COUNTS = 34;
struct LcaArgs {
    int N, rep0;                  // rep0: first replica of this batch
    double L, cut2;               // LCA_cutoff^2 (SMC.h:50)
    const double *R;              // [nrep][3N]
    unsigned *bits;               // [batch][words] num1 as a bit matrix
    long words;                   // 32-bit words per replica
    unsigned long long *counts;   // [nrep][LCA_COUNTS]
    int *LCA;                     // optional [3*N(N-1)/2] output of a single replica
};

// the (few) things the hot kernel needs; everything cold stays in DevCtx
struct SweepArgs {
    int N, chunk;
    double L, invL, cutoff2, invT;
    double *R;               // [nrep][3N]
    const double *displ;     // [nrep][chunk][3N]
    const double *uni;       // [nrep][chunk][N]
    const int *offs;         // [nrep][chunk]
    const ObsRec *obs;       // [nrep] (Ecur at entry)
    SweepRec *rec;           // [nrep][chunk]
    double edge;             // L/2 - cutoff: probes with |x|,|y| <= edge need no minimum image
    unsigned long long *clk; // [nrep][4] s_memtime, s_memrealtime at the start and at the end of the launch
#ifdef SMCX_CHECK
    unsigned long long *dbg; // diagnostic build: pairs inside the cutoff, candidates, misses of the screen
#endif
};

// start / end stamps of a sweep launch: shader clock (s_memtime) and the constant 100 MHz counter
// (s_memrealtime); their ratio is the clock the kernel actually ran at (smcx_last_clock).  Two scalar
// instructions and one store per launch, outside the move loop.
__device__ __forceinline__ void clock_stamp(unsigned long long *clk, int rep, int end)
{
    if (clk && threadIdx.x == 0) {
        clk[4 * (size_t)rep + 2 * end] = (unsigned long long)__builtin_amdgcn_s_memtime();
        clk[4 * (size_t)rep + 2 * end + 1] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
    }
}

// ---- cross-lane helpers ------------------------------------------------------
__device__ __forceinline__ double rdlane(double v, int lane)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// move a wave-uniform double into scalar registers
__device__ __forceinline__ double uniform_d(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double xchg(double v, int mask) { return __shfl_xor(v, mask, 64); }

// ---- cross-lane sums without LDS ------------------------------------------------
// gfx950 has v_permlane32_swap / v_permlane16_swap (swap the upper half / the odd
// 16-lane rows of one register with the lower half / even rows of another) and the
// DPP row controls; together they give every xor-butterfly level in the VALU.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// lanes 0-31: a(lane) + a(lane+32)      lanes 32-63: b(lane-32) + b(lane)
__device__ __forceinline__ double sum_swap32(double a, double b)
{
    const u32x2 l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2 h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
// rows 0,2: a(row) + a(row+1)           rows 1,3: b(row-1) + b(row)
__device__ __forceinline__ double sum_swap16(double a, double b)
{
    const u32x2 l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const u32x2 h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)h[0], (int)l[0]) + __hiloint2double((int)h[1], (int)l[1]);
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_ROR8 = 0x128;        // row_ror:8        lane ^ 8 within a row of 16
constexpr int DPP_HALF_MIRROR = 0x141; // row_half_mirror  lane i <-> 7-i within 8 lanes
constexpr int DPP_QUAD_X2 = 0x4E;      // quad_perm:[2,3,0,1]  lane ^ 2
constexpr int DPP_QUAD_X1 = 0xB1;      // quad_perm:[1,0,3,2]  lane ^ 1

// Reduce eight per-lane values over the 64 lanes of a wavefront.  Each level halves
// the number of live values instead of carrying all eight through all six levels:
// 4 + 2 + 1 exchanges for the top three levels, 3 for the rest, 34 VALU instructions
// in all and no LDS.  On return every lane of lane-group g = lane>>3 holds the wave
// total of v[g].
__device__ __forceinline__ double reduce8(double v0, double v1, double v2, double v3, double v4,
                                          double v5, double v6, double v7, int lane)
{
    const double w0 = sum_swap32(v0, v4), w1 = sum_swap32(v1, v5); // low half: v[i], high: v[4+i]
    const double w2 = sum_swap32(v2, v6), w3 = sum_swap32(v3, v7);
    const double u0 = sum_swap16(w0, w2), u1 = sum_swap16(w1, w3); // rows: v[i], v[2+i], v[4+i], v[6+i]
    const bool b3 = lane & 8;
    const double keep = b3 ? u1 : u0;
    const double send = b3 ? u0 : u1;
    double r = keep + dpp_mov<DPP_ROR8>(send);                     // 8-lane groups: v[lane>>3]
    r += dpp_mov<DPP_HALF_MIRROR>(r);
    r += dpp_mov<DPP_QUAD_X2>(r);
    r += dpp_mov<DPP_QUAD_X1>(r);
    return r;
}

// sum of t over lanes differing in bit 5 / 4 / 3, result in every lane
__device__ __forceinline__ double sum_x32(double t) { return sum_swap32(t, t); }
__device__ __forceinline__ double sum_x16(double t) { return sum_swap16(t, t); }
__device__ __forceinline__ double sum_x8(double t) { return t + dpp_mov<DPP_ROR8>(t); }

// ---- Lennard-Jones term shared by pairs, wall sites and the plane -------------
// e += ca/r^12 - cb/r^6 ; F += (48 ca/r^14 - 24 cb/r^8) d      (K1-K4, SMC.c:577-578,
// 610-614, 741, 758, 788-789, 806-809; the common factor 4 of the energies is
// applied once after the reduction)
__device__ __forceinline__ void lj_acc(double dx, double dy, double dz, double dr2, double ca,
                                       double cb, double &e, double &fx, double &fy, double &fz)
{
    // 1/dr2: hardware reciprocal seed + two Newton steps (<= 2 ulp; the IEEE division
    // sequence costs three times as many instructions for the last half ulp)
    double ir2 = __builtin_amdgcn_rcp(dr2);
    ir2 = fma(fma(-dr2, ir2, 1.0), ir2, ir2);
    ir2 = fma(fma(-dr2, ir2, 1.0), ir2, ir2);
    const double ir6 = ir2 * ir2 * ir2;
    const double t = ca * ir6 * ir6;
    const double s = cb * ir6;
    e += t - s;
    const double f = (48.0 * t - 24.0 * s) * ir2;
    fx += f * dx;
    fy += f * dy;
    fz += f * dz;
}

// the eight per-lane partial sums of a move: probe A (e, fx, fy, fz) and probe B.
// Named scalars, not an array: an array is kept as one 16-register tuple and copied
// wholesale around every conditional update.
struct Acc8 {
    double a0, a1, a2, a3, b0, b1, b2, b3;
};

struct Geo {                      // wave-uniform constants of the pair loop
    double L, invL, cutoff2;
};

// one pair evaluation: probe p against neighbour (x,y,z); d = probe - neighbour,
// minimum image in x,y only (SMC.c:567-573, 601-607).
// Both coordinates are kept wrapped into [-L/2, L/2] (SMC.c:315-316, 461), so
// |d - L rint(d/L)| = min(|d|, L - |d|): two instructions per axis for the cutoff
// test, which almost every pair fails; the signed minimum image is formed only
// inside the cutoff.
__device__ __forceinline__ void pair_eval(const Geo &g, double px, double py, double pz, double x,
                                          double y, double z, bool ok, double &e, double &fx,
                                          double &fy, double &fz)
{
    const double dx = px - x;
    const double dy = py - y;
    const double dz = pz - z;
    const double mx = fmin(fabs(dx), g.L - fabs(dx));
    const double my = fmin(fabs(dy), g.L - fabs(dy));
    const double q = mx * mx + my * my + dz * dz;
    if (q < g.cutoff2 && ok) {
        const double sx = dx - g.L * __builtin_rint(dx * g.invL);
        const double sy = dy - g.L * __builtin_rint(dy * g.invL);
        const double dr2 = sx * sx + sy * sy + dz * dz;
        lj_acc(sx, sy, dz, dr2, 1.0, 1.0, e, fx, fy, fz);
    }
}

// squared minimum-image distance for the cutoff test (see pair_eval).  WRAP=false is
// for a probe at least one cutoff away from the x and y box edges: no neighbour can
// then be inside the cutoff through the periodic image (|d| <= L - rc), so |d| itself
// decides and the test costs 7 instead of 11 instructions.
template <bool WRAP>
__device__ __forceinline__ double pair_q(const Geo &g, double px, double py, double pz, double x,
                                         double y, double z)
{
    const double dx = px - x, dy = py - y, dz = pz - z;
    if constexpr (WRAP) {
        const double mx = fmin(fabs(dx), g.L - fabs(dx));
        const double my = fmin(fabs(dy), g.L - fabs(dy));
        return mx * mx + my * my + dz * dz;
    } else {
        return dx * dx + dy * dy + dz * dz;
    }
}

// the rare part of a pair evaluation: the pair is inside the cutoff
__device__ __forceinline__ void pair_hit(const Geo &g, double px, double py, double pz, double x,
                                         double y, double z, double &e, double &fx, double &fy,
                                         double &fz)
{
    const double dx = px - x, dy = py - y, dz = pz - z;
    const double sx = dx - g.L * __builtin_rint(dx * g.invL);
    const double sy = dy - g.L * __builtin_rint(dy * g.invL);
    const double dr2 = sx * sx + sy * sy + dz * dz;
    lj_acc(sx, sy, dz, dr2, 1.0, 1.0, e, fx, fy, fz);
}

// Fused pass over the register-resident neighbours: probe A (proposed position
// of the moving particle n) accumulates into v.a*, probe B (current position
// of the next particle) into v.b*.  exA0/exB0 mask this lane's slot-0
// particle, exB1 its slot-1 particle (n and n+1 are always there, see kernel).
// The cutoff test is done for G slots x 2 probes at a time and followed by ONE
// branch: straight-line blocks of fp64 instructions with 2*G independent
// dependency chains instead of a branch after every 11 instructions.  `interior`
// (wave-uniform) selects the cheaper test when both probes are away from the x,y edges.
template <int S, int G>
__device__ __forceinline__ void fused_pass(const Geo &g, const double (&x)[S], const double (&y)[S],
                                           const double (&z)[S], double ax, double ay, double az,
                                           double bx, double by, double bz, bool exA0, bool exB0,
                                           bool exB1, bool interior, Acc8 &v)
{
    constexpr int GG = (S >= G) ? G : S;
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += GG) {
        double qa[GG], qb[GG];
        if (interior) { // wave-uniform: a scalar branch around straight-line code
#pragma unroll
            for (int j = 0; j < GG; j++) {
                qa[j] = pair_q<false>(g, ax, ay, az, x[k0 + j], y[k0 + j], z[k0 + j]);
                qb[j] = pair_q<false>(g, bx, by, bz, x[k0 + j], y[k0 + j], z[k0 + j]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < GG; j++) {
                qa[j] = pair_q<true>(g, ax, ay, az, x[k0 + j], y[k0 + j], z[k0 + j]);
                qb[j] = pair_q<true>(g, bx, by, bz, x[k0 + j], y[k0 + j], z[k0 + j]);
            }
        }
        bool ha[GG], hb[GG];
        unsigned long long any = 0; // lane masks stay in scalar registers (v_cmp -> s_or)
#pragma unroll
        for (int j = 0; j < GG; j++) {
            const int k = k0 + j;
            const bool okA = (k == 0) ? !exA0 : true;
            const bool okB = (k == 0) ? !exB0 : ((k == 1) ? !exB1 : true);
            ha[j] = (qa[j] < g.cutoff2) && okA;
            hb[j] = (qb[j] < g.cutoff2) && okB;
            any |= __builtin_amdgcn_ballot_w64(ha[j]) | __builtin_amdgcn_ballot_w64(hb[j]);
        }
        if (any) {
#pragma unroll
            for (int j = 0; j < GG; j++) {
                const int k = k0 + j;
                if (ha[j]) pair_hit(g, ax, ay, az, x[k], y[k], z[k], v.a0, v.a1, v.a2, v.a3);
                if (hb[j]) pair_hit(g, bx, by, bz, x[k], y[k], z[k], v.b0, v.b1, v.b2, v.b3);
            }
        }
    }
}

// ---- pseudo-neighbours handled by spare lanes of wave 0 ------------------------
// role 0/1: wall site acting on probe A / B          -> v[0..3] / v[4..7]
// role 4/5: featureless plane acting on probe A / B  (no x,y offset, no cutoff)
// role 2  : pair (old position of n) -> B, kept aside   (lane 30)
// role 3  : pair (new position of n) -> B, kept aside   (lane 31)
// After the accept/reject decision the matching side term completes B's sums.
// The per-lane constants live in LDS and are read once per move by wave 0.
struct RoleTable {         // lanes l and l+32 stand for the same site (acting on probe A resp. B)
    double sx[32], sy[32]; // site position i*dw, j*dw (SMC.c:748-750)
    double ca[32], cb[32]; // W[2m], W[2m+1] or a0, b0 or 1,1
    int role[64];          // -1 = none
    double Lz, invLz, halfLz; // wall geometry, read by the few wall lanes only
};

constexpr int SIDE_LANE_OLD = 30;
constexpr int SIDE_LANE_NEW = 31;

// filled by the 64 lanes of one wave
__device__ __forceinline__ void fill_roles(const DevCtx &c, RoleTable &rt, int lane)
{
    int role = -1;
    double sx = 0.0, sy = 0.0, ca = 1.0, cb = 1.0;
    const bool walls = (c.flags & 0x1u) != 0;
    const int half = lane & 31;
    if (walls && half <= c.M2 && half < 30) {
        role = lane >> 5;
        if (half == c.M2) {
            role |= 4; ca = c.a0; cb = c.b0;
        } else {
            const double dw = c.L / c.M;
            sx = (half / c.M) * dw;
            sy = (half % c.M) * dw;
            ca = c.W[2 * half];
            cb = c.W[2 * half + 1];
        }
    }
    if (lane == SIDE_LANE_OLD) role = 2;
    if (lane == SIDE_LANE_NEW) role = 3;
    if (lane < 32) { rt.sx[lane] = sx; rt.sy[lane] = sy; rt.ca[lane] = ca; rt.cb[lane] = cb; }
    rt.role[lane] = role;
    if (lane == 0) { rt.Lz = c.Lz; rt.invLz = c.invLz; rt.halfLz = c.halfLz; }
}

// signed distance to the nearer wall with the reference's clamp (SMC.c:736-739)
template <class Ctx>
__device__ __forceinline__ double wall_dz(const Ctx &c, double rz)
{
    double dz = rz + c.halfLz;
    dz = dz - c.Lz * __builtin_rint(dz * c.invLz);
    if (rz <= -c.halfLz) dz = 0.0001;
    else if (rz >= c.halfLz) dz = -0.0001;
    return dz;
}

// P: current position of n, A: its proposal, B: current position of the next particle.
// Executed by wave 0 only; `role` is this lane's rt.role[lane].
__device__ __forceinline__ void special_block(const Geo &g, const RoleTable &rt,
                                              int lane, int role, bool hasA, bool hasB, bool sides,
                                              double Px, double Py, double Pz, double Ax, double Ay,
                                              double Az, double Bx, double By, double Bz,
                                              Acc8 &v, double (&side)[4])
{
    side[0] = side[1] = side[2] = side[3] = 0.0;
    bool active = false;
    const bool wallrole = (role >= 0) && ((role & 2) == 0);
    const bool onA = wallrole && ((role & 1) == 0);
    if (wallrole) active = onA ? hasA : hasB;
    else if (role >= 2) active = hasA && hasB && sides;
    if (active) {
        const bool plane = wallrole && (role & 4);
        const double tx = onA ? Ax : Bx, ty = onA ? Ay : By, tz = onA ? Az : Bz;
        double dx, dy, dz;
        if (wallrole) {
            dx = tx - rt.sx[lane & 31]; dy = ty - rt.sy[lane & 31];
            dz = wall_dz(rt, tz);
        } else {
            const bool old = (role == 2);
            dx = tx - (old ? Px : Ax); dy = ty - (old ? Py : Ay);
            dz = tz - (old ? Pz : Az);
        }
        dx = dx - g.L * __builtin_rint(dx * g.invL);
        dy = dy - g.L * __builtin_rint(dy * g.invL);
        if (plane) { dx = 0.0; dy = 0.0; }
        const double dr2 = dx * dx + dy * dy + dz * dz;
        double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
        if (plane || dr2 < g.cutoff2)
            lj_acc(dx, dy, dz, dr2, rt.ca[lane & 31], rt.cb[lane & 31], e, fx, fy, fz);
        if (wallrole && onA) { v.a0 += e; v.a1 += fx; v.a2 += fy; v.a3 += fz; }
        else if (wallrole) { v.b0 += e; v.b1 += fx; v.b2 += fy; v.b3 += fz; }
        else { side[0] = e; side[1] = fx; side[2] = fy; side[3] = fz; }
    }
}

// ---- glibc rand(): 31 new outputs per step -------------------------------------
// r[i] = r[i-31] + r[i-3] (mod 2^32), output r[i] >> 1.  Lane j < 31 holds
// r[i-31+j].  new[j] = old[j] + (j < 3 ? old[j+28] : new[j-3]) is an inclusive
// prefix sum over the three residue classes of j mod 3: four shifted adds.
__device__ __forceinline__ uint32_t rand_block(uint32_t h, int lane)
{
    uint32_t t = __shfl_down(h, 28, 64);
    uint32_t w = h + ((lane < 3) ? t : 0u);
#pragma unroll
    for (int d = 3; d <= 24; d *= 2) {
        t = __shfl_up(w, d, 64);
        if (lane >= d) w += t;
    }
    return w;
}

} // namespace smcx
