// smcx_sweep_mx.hip -- the screened form of the sweep kernel.  Its own translation unit because
// it is built with -fno-slp-vectorize: the SLP vectoriser otherwise repacks the screening
// arithmetic onto shuffled register pairs, moves the probes into VGPRs and spills the positions.
#include "smcx_device.hpp"
#include "smcx_kernels.h"

#include <cmath>
#include <cstdlib>

namespace smcx {

#include "smcx_sweep_common.hpp"

// ---------------------------------------------------------------------------------
// Screened form of the same sweep.  The cutoff test -- which 99.96 % of the pair
// evaluations fail at the benchmark density -- is screened on compact copies of the
// positions, with a threshold widened by a proven error bound; every candidate is then
// decided and evaluated in fp64, with the arithmetic of the fp64 kernels, from the fp64
// positions in memory.  Same pairs inside the cutoff, same values, same order of summation
// per lane: the results differ from those of the fp64 kernels by rounding only.
//
// Compact copy of a particle:
//   x,y  two 16-bit fixed-point numbers in one register, the box [-L/2, L/2) mapped onto the
//        whole int16 range: the subtraction of two of them (v_pk_sub_i16) wraps modulo 2^16,
//        which IS the minimum image -- no rint, no branch on where the probe sits;
//        dx^2 + dy^2 is one v_dot2_i32_i16 (saturating);
//   z    (not periodic, unbounded) either fp32 in a second register (ZL = false), or fp16 in
//        LDS, two bytes per particle (ZL = true): dz of two slots is then one v_pk_add_f16 and
//        dz^2 enters the fp32 sum through v_fma_mix_f32.
// With one or two VGPRs per particle 64 particles per lane fit beside everything else, so
// N = 4096 runs as ONE wavefront per replica -- no workgroup barrier, no cross-wave
// reduction -- at two (ZL = false) or four (ZL = true) waves per SIMD.
//
// Screening bound (host side: launch_sweeps_mx).  With u = L/65536 the stored x differs
// from the true one by <= u/2, so an integer difference is within 1 unit of the true
// (minimum-image) one and dxi^2 + dyi^2 <= (true, in units) + 2*sqrt(2)*R + 2 for a pair
// inside the cutoff R = rc/u.  z: |fl32(z) - z| <= 2^-24 |z|, resp. |fl16(z) - z| <= zsafe *
// 2^-12 for |z| < zsafe (a power of two).  The sum of both, plus the rounding of the final
// expression, is `margin`; the test q - thr < 0 with thr = rc^2 + margin never rejects a
// pair the fp64 test accepts.  Particles whose z is outside the safe range are kept in a
// per-lane mask and are always candidates; a probe outside it makes every slot one.
// Padding slots hold z = +inf, a disabled probe z = -inf: q = +inf, never candidates.
// ---------------------------------------------------------------------------------
struct MxArgs {
    float thr;      // screening threshold rc^2 + margin
    float u2;       // (L/65536)^2: integer units^2 -> length^2
    double toFix;   // 65536/L
    float zsafe;    // |z| up to which the error bound holds
};

typedef short mx_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short mx_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 mx_h2 __attribute__((ext_vector_type(2)));

// x,y -> two int16 in one register (x low, y high); x = +L/2 wraps onto -L/2, the same point
__device__ __forceinline__ unsigned mx_pack_xy(double x, double y, double toFix)
{
    const int xi = (int)__builtin_rint(x * toFix), yi = (int)__builtin_rint(y * toFix);
    return ((unsigned)xi & 0xffffu) | ((unsigned)yi << 16);
}
__device__ __forceinline__ float mx_z32(double z, float zsafe)
{
    return (fabs(z) < (double)zsafe) ? (float)z : __builtin_nanf(""); // unsafe: see the kernel's mask
}
__device__ __forceinline__ unsigned mx_z16(double z, float zsafe) // fp16 bits
{
    const unsigned short bits = __builtin_bit_cast(unsigned short, (_Float16)(float)z);
    return (fabs(z) < (double)zsafe) ? (unsigned)bits : 0x7E00u; // NaN; unsafe: see the kernel's mask
}

// rotate the register-resident slots by one: slot j <- slot j+1
template <int S, int SZ>
__device__ __forceinline__ void mx_rotate(unsigned (&xy)[S], float (&z)[SZ])
{
    const unsigned t = xy[0];
#pragma unroll
    for (int k = 0; k + 1 < S; k++) xy[k] = xy[k + 1];
    xy[S - 1] = t;
    if constexpr (SZ == S) {
        const float tz = z[0];
#pragma unroll
        for (int k = 0; k + 1 < S; k++) z[k] = z[k + 1];
        z[S - 1] = tz;
    }
}
// the same for this wavefront's block of fp16 z in LDS (slots 2j, 2j+1 of a lane in one dword)
template <int S>
__device__ __forceinline__ void mx_rotate_lds(unsigned (&zw)[S / 2][64], int lane)
{
    const unsigned first = zw[0][lane];
    unsigned cur = first;
#pragma unroll
    for (int j = 0; j + 1 < S / 2; j++) {
        const unsigned nxt = zw[j + 1][lane];
        zw[j][lane] = __builtin_amdgcn_alignbit(nxt, cur, 16); // (cur.hi, nxt.lo)
        cur = nxt;
    }
    zw[S / 2 - 1][lane] = __builtin_amdgcn_alignbit(first, cur, 16);
}

// ---- screening: candidate bits, slot k in bit k%32 of word k/32 ------------------------
// The screen computes q - thr; its sign bit is the flag and is shifted into the lane's candidate
// word with one v_alignbit_b32 ({word, q} >> 31): no compare, no carry through VCC, no branch.
// Measured (A/B in one session, profiles/r01_screened_kernel.log): 6-9 % faster than
// v_cmp + v_addc_co through the carry, which in turn beat scalar-register masks OR-ed per group of
// four slots with a branch to a slow path.  A NaN has no defined sign: particles and probes
// outside the safe z range are flagged by hand (kernel, `unsafe`).  The shift builds the word in
// reverse; the last group turns it round.
template <int S, int NW>
__device__ __forceinline__ void mx_flag4(int k0, const float (&qa)[4], const float (&qb)[4], unsigned (&ca)[NW],
                                         unsigned (&cb)[NW])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        ca[(k0 + j) >> 5] = __builtin_amdgcn_alignbit(ca[(k0 + j) >> 5], __builtin_bit_cast(unsigned, qa[j]), 31);
        cb[(k0 + j) >> 5] = __builtin_amdgcn_alignbit(cb[(k0 + j) >> 5], __builtin_bit_cast(unsigned, qb[j]), 31);
    }
    if (k0 + 4 == S) {
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const int width = (w == NW - 1) ? S - 32 * w : 32;
            ca[w] = __builtin_bitreverse32(ca[w]) >> (32 - width);
            cb[w] = __builtin_bitreverse32(cb[w]) >> (32 - width);
        }
    }
}

// u2 * (dxi^2 + dyi^2) - thr of one particle and one probe
__device__ __forceinline__ float mx_qxy(unsigned pxy, unsigned xy, float u2, float thr)
{
    // the difference must wrap modulo 2^16 (that is the minimum image): unsigned lanes, where
    // wrapping is defined, read back as signed
    const mx_s2 d = __builtin_bit_cast(mx_s2, __builtin_bit_cast(mx_u2, pxy) - __builtin_bit_cast(mx_u2, xy));
    return __builtin_fmaf(u2, (float)__builtin_amdgcn_sdot2(d, d, 0, true), -thr);
}

// z in registers (fp32): az, bz are the probes' z
template <int S>
__device__ __forceinline__ void mx_screen(const unsigned (&xy)[S], const float (&z)[S], unsigned axy, float az,
                                          unsigned bxy, float bz, float u2, float thr,
                                          unsigned (&ca)[(S + 31) / 32], unsigned (&cb)[(S + 31) / 32])
{
    static_assert(S % 4 == 0, "groups of four slots");
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += 4) {
        float qa[4], qb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float da = az - z[k0 + j], db = bz - z[k0 + j];
            qa[j] = __builtin_fmaf(da, da, mx_qxy(axy, xy[k0 + j], u2, thr));
            qb[j] = __builtin_fmaf(db, db, mx_qxy(bxy, xy[k0 + j], u2, thr));
        }
        mx_flag4<S, (S + 31) / 32>(k0, qa, qb, ca, cb);
    }
}
// z in LDS (fp16): azz, bzz hold the probes' z twice
template <int S>
__device__ __forceinline__ void mx_screen_lds(const unsigned (&xy)[S], const unsigned (&zw)[S / 2][64], int lane,
                                              unsigned axy, unsigned azz, unsigned bxy, unsigned bzz, float u2,
                                              float thr, unsigned (&ca)[(S + 31) / 32],
                                              unsigned (&cb)[(S + 31) / 32])
{
    static_assert(S % 4 == 0, "groups of four slots");
    // the LDS reads run one group ahead of their use
    unsigned zn[2] = {zw[0][lane], zw[1][lane]};
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += 4) {
        float qa[4], qb[4];
        const unsigned zc[2] = {zn[0], zn[1]};
        if (k0 + 4 < S) { zn[0] = zw[k0 / 2 + 2][lane]; zn[1] = zw[k0 / 2 + 3][lane]; }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const mx_h2 zz = __builtin_bit_cast(mx_h2, zc[p]);
            const mx_h2 da = __builtin_bit_cast(mx_h2, azz) - zz; // two slots at once
            const mx_h2 db = __builtin_bit_cast(mx_h2, bzz) - zz;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int j = 2 * p + h;
                const float fa = (float)(h ? da.y : da.x), fb = (float)(h ? db.y : db.x);
                qa[j] = __builtin_fmaf(fa, fa, mx_qxy(axy, xy[k0 + j], u2, thr)); // v_fma_mix_f32
                qb[j] = __builtin_fmaf(fb, fb, mx_qxy(bxy, xy[k0 + j], u2, thr));
            }
        }
        mx_flag4<S, (S + 31) / 32>(k0, qa, qb, ca, cb);
    }
}

// take this lane's lowest-slot candidate out of the words: its register slot, or -1
template <int NW>
__device__ __forceinline__ int mx_pick(unsigned (&cw)[NW])
{
    int k = -1;
#pragma unroll
    for (int w = NW - 1; w >= 0; w--)
        if (cw[w] != 0u) k = 32 * w + __builtin_ctz(cw[w]);
    if (k >= 0) cw[k >> 5] &= ~(1u << (k & 31));
    return k;
}
template <int NW>
__device__ __forceinline__ bool mx_any(const unsigned (&cw)[NW])
{
    unsigned u = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) u |= cw[w];
    return u != 0u;
}

// fp64 position of the particle in register slot k of this thread, from memory.  With one
// wavefront per replica every position is written and read by the same lane, so a plain
// (L1-cached) load is coherent; several wavefronts need the L1-bypassing load.
template <int S, int T, bool COHERENT>
__device__ __forceinline__ void mx_fetch(const double *Rg, int N, int tid, int rot, int k, double &X,
                                         double &Y, double &Z)
{
    int ls = k + rot; // register slot -> logical slot
    if (ls >= S) ls -= S;
    const int l = ls * T + tid;
    if (l >= N) { X = 0.0; Y = 0.0; Z = FAR_PAD; return; } // a padding slot (flagged only through an unsafe probe)
    const double *q = Rg + 3 * l;
    if constexpr (COHERENT) { X = ld_coherent(q); Y = ld_coherent(q + 1); Z = ld_coherent(q + 2); }
    else { X = q[0]; Y = q[1]; Z = q[2]; }
}

// the fp64 decision and evaluation of one candidate: pair_hit's arithmetic (signed minimum
// image, SMC.c:567-578, 601-614), the cutoff test on its dr2
__device__ __forceinline__ void mx_exact(const Geo &g, double px, double py, double pz, double x, double y,
                                         double z, double &e, double &fx, double &fy, double &fz)
{
    const double dx = px - x, dy = py - y, dz = pz - z;
    const double sx = dx - g.L * __builtin_rint(dx * g.invL);
    const double sy = dy - g.L * __builtin_rint(dy * g.invL);
    const double dr2 = sx * sx + sy * sy + dz * dz;
    if (dr2 < g.cutoff2) lj_acc(sx, sy, dz, dr2, 1.0, 1.0, e, fx, fy, fz);
}

#ifdef SMCX_STAMPS // diagnostic build only (tools/phase_stamps.py): cycles per phase of a move
#define STAMP(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); ph[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

template <int S, int WPR, int MINW, bool ZL>
__global__ void __launch_bounds__(64 * WPR, MINW)
sweep_kernel_mx(SweepArgs a, DevCtx c, int nsweeps, double A, MxArgs m)
{
    constexpr int T = 64 * WPR;
    constexpr int NW = (S + 31) / 32;
    constexpr int SZ = ZL ? 1 : S; // z slots kept in registers
    static_assert(S % 4 == 0, "slot groups");
    __shared__ SweepShared<WPR> sh;
    __shared__ unsigned zl[ZL ? WPR : 1][ZL ? S / 2 : 1][64]; // fp16 z, slot pairs x lanes

    const int rep = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = uniform(tid >> 6);
    const int N = a.N;
    double *Rg = a.R + (size_t)rep * 3 * N;
    clock_stamp(a.clk, rep, 0);

    // ---- compact copies: particle l in lane l % T, slot l / T --------------------------
    unsigned xy[S];
    float z[SZ];
    unsigned (&zw)[ZL ? S / 2 : 1][64] = zl[ZL ? wave : 0];
#pragma unroll
    for (int j = 0; j < S / 2; j++) {
        unsigned pair = 0u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 2 * j + h;
            const int l = k * T + tid;
            const bool real = (l < N);
            xy[k] = real ? mx_pack_xy(Rg[3 * l], Rg[3 * l + 1], m.toFix) : 0u;
            if constexpr (ZL) pair |= (real ? mx_z16(Rg[3 * l + 2], m.zsafe) : 0x7C00u) << (16 * h); // pad: +inf
            else z[k] = real ? mx_z32(Rg[3 * l + 2], m.zsafe) : __builtin_inff();
        }
        if constexpr (ZL) zw[j][lane] = pair;
    }
    int rot = 0; // register slot j holds logical slot (j + rot) % S
    // slots of this lane whose z is outside the safe range (always candidates), bit = register slot
    unsigned long long unsafe = 0ull;
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int l = k * T + tid;
        if (l < N && !(fabs(Rg[3 * l + 2]) < (double)m.zsafe)) unsafe |= 1ull << k;
    }
    auto rotate = [&]() {
        mx_rotate<S, SZ>(xy, z);
        if constexpr (ZL) mx_rotate_lds<S>(zw, lane);
        unsafe = (unsafe >> 1) | ((unsafe & 1ull) << (S - 1));
        rot = (rot + 1 == S) ? 0 : rot + 1;
    };

    if (wave == 0) fill_roles(c, sh.roles, lane);
    __syncthreads();
    const int role = (wave == 0) ? sh.roles.role[lane] : -1;

    Geo g; g.L = a.L; g.invL = a.invL; g.cutoff2 = a.cutoff2;
    double E = uniform_d(a.obs[rep].Ecur);
    int par = 0;
#ifdef SMCX_STAMPS
    long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_amdgcn_s_memtime();
#endif
#ifdef SMCX_CHECK // diagnostic build only (libsmcx_check.so): the fp64 test beside the screen, every slot
    unsigned long long chk_in = 0, chk_cand = 0, chk_miss = 0;
#endif
    const double AoT = A * a.invT;         // SMC.c:307-309 (A/T)
    const double Ao4T = A * 0.25 * a.invT; // SMC.c:327 (A/(4T))

#pragma unroll 1
    for (int sw = 0; sw < nsweeps; sw++) {
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
            while (rot != ks) rotate();
            int tl = first - ks * T - 1; // owner thread of particle n (slot 0); -1 in the prologue

            double Px = 0.0, Py = 0.0, Pz = FAR_PROBE;
            double Um = 0.0, Fmx = 0.0, Fmy = 0.0, Fmz = 0.0;
            // the next particle's fp64 position comes from memory, one move ahead of its use
            double nBx = ld_coherent(Rg + 3 * first), nBy = ld_coherent(Rg + 3 * first + 1),
                   nBz = ld_coherent(Rg + 3 * first + 2);
            // this move's displacement and log-uniform: wave-uniform, read-only for the whole
            // kernel -> scalar loads through the constant address space, one move ahead
            typedef const __attribute__((address_space(4))) double *kptr;
            const kptr dK = (kptr)(unsigned long long)displ + 3 * (size_t)first;
            const kptr uK = (kptr)(unsigned long long)uni + vbase;
            double ndx = dK[0], ndy = dK[1], ndz = dK[2], nlu = uK[0];
#pragma unroll 1
            for (int i = -1; i < len; i++) { // i = -1: the run's prologue, probe B alone
                const int n = first + i;
                const bool hasA = (i >= 0);
                const double ddx = ndx, ddy = ndy, ddz = ndz, lu = nlu; // of move i (unused for i = -1)
                if (i + 1 < len) {
                    ndx = dK[3 * (i + 1)]; ndy = dK[3 * (i + 1) + 1]; ndz = dK[3 * (i + 1) + 2];
                    nlu = uK[i + 1];
                }
                double Qx = 0.0, Qy = 0.0, Qz = FAR_PROBE; // proposal, SMC.c:307-316
                if (hasA) {
                    Qx = Px + (Fmx * AoT + ddx);
                    Qy = Py + (Fmy * AoT + ddy);
                    Qz = Pz + (Fmz * AoT + ddz);
                    Qx = Qx - a.L * __builtin_rint(Qx * a.invL);
                    Qy = Qy - a.L * __builtin_rint(Qy * a.invL);
                    Qx = uniform_d(Qx); Qy = uniform_d(Qy); Qz = uniform_d(Qz);
                }
                const bool hasB = (i + 1 < len);
                const bool cross = hasB && (tl == T - 1);
                double Bx = 0.0, By = 0.0, Bz = FAR_PROBE;
                if (hasB) { Bx = uniform_d(nBx); By = uniform_d(nBy); Bz = uniform_d(nBz); }
                if (i + 2 < len) {
                    nBx = ld_coherent(Rg + 3 * (n + 2)); nBy = ld_coherent(Rg + 3 * (n + 2) + 1);
                    nBz = ld_coherent(Rg + 3 * (n + 2) + 2);
                }
                STAMP(0); // proposal, probe fetch

                // ---- screening ------------------------------------------------------
                const unsigned axy = (unsigned)uniform((int)mx_pack_xy(Qx, Qy, m.toFix));
                const unsigned bxy = (unsigned)uniform((int)mx_pack_xy(Bx, By, m.toFix));
                unsigned ca[NW], cb[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) { ca[w] = 0u; cb[w] = 0u; }
                float azf = 0.f;   // the proposal's z as stored on acceptance
                unsigned azh = 0u;
                if constexpr (ZL) {
                    azh = hasA ? mx_z16(Qz, m.zsafe) : 0xFC00u; // -inf: a disabled probe
                    const unsigned bzh = hasB ? mx_z16(Bz, m.zsafe) : 0xFC00u;
                    const unsigned azz = (unsigned)uniform((int)(azh | (azh << 16)));
                    const unsigned bzz = (unsigned)uniform((int)(bzh | (bzh << 16)));
                    mx_screen_lds<S>(xy, zw, lane, axy, azz, bxy, bzz, m.u2, m.thr, ca, cb);
                } else {
                    azf = hasA ? mx_z32(Qz, m.zsafe) : -__builtin_inff();
                    const float bzf = hasB ? mx_z32(Bz, m.zsafe) : -__builtin_inff();
                    mx_screen<S>(xy, z, axy, azf, bxy, bzf, m.u2, m.thr, ca, cb);
                }
                {   // the sign-bit screen does not see NaN: unsafe particles and unsafe probes by hand
                    const bool ua = hasA && !(fabs(Qz) < (double)m.zsafe), ub = hasB && !(fabs(Bz) < (double)m.zsafe);
#pragma unroll
                    for (int w = 0; w < NW; w++) {
                        const unsigned uw = (unsigned)(unsafe >> (32 * w));
                        const unsigned all = (w == NW - 1 && (S & 31)) ? ((1u << (S & 31)) - 1u) : ~0u;
                        ca[w] = ua ? all : (ca[w] | uw);
                        cb[w] = ub ? all : (cb[w] | uw);
                    }
                }
                // the moving particle itself and the particle probe B stands for are not neighbours
                {
                    const bool exA0 = (tid == tl);
                    const bool exB0 = (hasA && tid == tl) || (hasB && !cross && tid == tl + 1);
                    const bool exB1 = cross && (tid == 0);
                    if (exA0) ca[0] &= ~1u;
                    if (exB0) cb[0] &= ~1u;
                    if (exB1) cb[0] &= ~2u;
                }
                STAMP(1); // screening
#ifdef SMCX_CHECK
                {   // every slot of this lane against both probes in fp64 (pair_hit's arithmetic, positions from
                    // memory): a pair inside the cutoff that the screen did not flag is a miss
                    unsigned long long fa = ca[0], fb = cb[0];
                    if constexpr (NW > 1) { fa |= (unsigned long long)ca[1] << 32; fb |= (unsigned long long)cb[1] << 32; }
                    chk_cand += __builtin_popcountll(fa) + __builtin_popcountll(fb);
                    const bool exA0 = (tid == tl);
                    const bool exB0 = (hasA && tid == tl) || (hasB && !cross && tid == tl + 1);
                    const bool exB1 = cross && (tid == 0);
#pragma unroll 1
                    for (int k = 0; k < S; k++) {
                        double X, Y, Z;
                        mx_fetch<S, T, true>(Rg, N, tid, rot, k, X, Y, Z);
                        for (int pr = 0; pr < 2; pr++) {
                            if (pr == 0 ? !hasA : !hasB) continue;
                            if (pr == 0 ? (k == 0 && exA0) : ((k == 0 && exB0) || (k == 1 && exB1))) continue;
                            const double dx = (pr ? Bx : Qx) - X, dy = (pr ? By : Qy) - Y, dz = (pr ? Bz : Qz) - Z;
                            const double sx = dx - g.L * __builtin_rint(dx * g.invL);
                            const double sy = dy - g.L * __builtin_rint(dy * g.invL);
                            if (sx * sx + sy * sy + dz * dz < g.cutoff2) {
                                chk_in++;
                                if (!(((pr ? fb : fa) >> k) & 1ull)) chk_miss++;
                            }
                        }
                    }
                }
#endif

                // ---- exact evaluation of the candidates ------------------------------------
                // every lane fetches its candidates' fp64 positions (one of each probe per round)
                // and decides and evaluates them in fp64, in ascending slot order per probe
                Acc8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                double side[4], tot[8], sOld[4], sNew[4];
                if (__builtin_amdgcn_ballot_w64(mx_any<NW>(ca) || mx_any<NW>(cb))) {
                    int kA = mx_pick<NW>(ca), kB = mx_pick<NW>(cb);
                    do {
                        double XA = 0, YA = 0, ZA = 0, XB = 0, YB = 0, ZB = 0;
                        if (kA >= 0) mx_fetch<S, T, (WPR > 1)>(Rg, N, tid, rot, kA, XA, YA, ZA);
                        if (kB >= 0) mx_fetch<S, T, (WPR > 1)>(Rg, N, tid, rot, kB, XB, YB, ZB);
                        if (kA >= 0) mx_exact(g, Qx, Qy, Qz, XA, YA, ZA, v.a0, v.a1, v.a2, v.a3);
                        if (kB >= 0) mx_exact(g, Bx, By, Bz, XB, YB, ZB, v.b0, v.b1, v.b2, v.b3);
                        kA = mx_pick<NW>(ca); kB = mx_pick<NW>(cb);
                    } while (__builtin_amdgcn_ballot_w64(kA >= 0 || kB >= 0));
                }
                STAMP(2); // candidates
                if (wave == 0)
                    special_block(g, sh.roles, lane, role, hasA, hasB, hasA, Px, Py, Pz, Qx, Qy, Qz,
                                  Bx, By, Bz, v, side);
                else side[0] = side[1] = side[2] = side[3] = 0.0;
                STAMP(3); // walls, plane, side pair
                combine<WPR>(sh, par, lane, wave, v, side, tot, sOld, sNew);

                bool acc = false;
                if (hasA) { // SMC acceptance, SMC.c:326-348
                    const double Un = 4.0 * tot[0], Fnx = tot[1], Fny = tot[2], Fnz = tot[3];
                    const double dX = Fmx * AoT + ddx;
                    const double dY = Fmy * AoT + ddy;
                    const double dZ = Fmz * AoT + ddz;
                    const double gx = Fnx - Fmx, gy = Fny - Fmy, gz = Fnz - Fmz;
                    const double deltaW = (gx * gx + gy * gy + gz * gz +
                                           2.0 * (gx * Fmx + gy * Fmy + gz * Fmz)) * Ao4T;
                    const double arg = Un - Um +
                                       (dX * (Fnx + Fmx) + dY * (Fny + Fmy) + dZ * (Fnz + Fmz)) * 0.5 + deltaW;
                    acc = (lu < -arg * a.invT); // u < exp(-arg/T); NaN rejects (SMC.c:335)
                    acc = (uniform((int)acc) != 0);
                    if (acc && tid == tl) {
                        xy[0] = axy;
                        unsafe = (unsafe & ~1ull) | (!(fabs(Qz) < (double)m.zsafe) ? 1ull : 0ull);
                        if constexpr (ZL) reinterpret_cast<unsigned short *>(&zw[0][lane])[0] = (unsigned short)azh;
                        else z[0] = azf;
                        Rg[3 * n] = Qx; Rg[3 * n + 1] = Qy; Rg[3 * n + 2] = Qz; // the fp64 state
                    }
                    if (acc) { E = uniform_d(E + (Un - Um)); jacc++; }
                }
                STAMP(4); // reduction, Metropolis step

                if (hasB) { // next particle's Um,Fm = B sums + the (n, n+1) pair term
                    double s0, s1, s2, s3;
                    if constexpr (WPR == 1) {
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
                    if (cross) { rotate(); tl = 0; }
                    else tl++;
                }
                STAMP(5); // next particle's Um/Fm, slot rotation
            }
        }
        // C: hand E[n+1] and jj[n] (SMC.c:194-195) to the bookkeeping kernel
        if (tid == 0) {
            SweepRec r; r.E = E; r.accepted = jacc; r.pad = 0;
            a.rec[(size_t)rep * a.chunk + sw] = r;
        }
    }
    clock_stamp(a.clk, rep, 1);
#ifdef SMCX_CHECK
    atomicAdd(&a.dbg[0], chk_in); atomicAdd(&a.dbg[1], chk_cand); atomicAdd(&a.dbg[2], chk_miss);
#endif
#ifdef SMCX_STAMPS
    if (tid == 0) { // diagnostic: overwrite the head of this replica's (consumed) displacement block
        double *dbg = const_cast<double *>(a.displ) + (size_t)rep * a.chunk * 3 * N;
        for (int k = 0; k < 6; k++) dbg[k] = (double)ph[k];
    }
#endif
}

typedef void (*sweep_mx_fn)(SweepArgs, DevCtx, int, double, MxArgs);

// fp32 z in registers: two VGPRs per particle
// (particles per lane, wavefronts per replica, waves per SIMD the register budget is set for)
#define SMCX_MX_TABLE(X) X(16, 1, 3) X(32, 1, 3) X(64, 1, 2) X(16, 2, 3) X(32, 2, 3) X(64, 2, 2) \
                         X(16, 4, 4) X(32, 4, 3) X(64, 4, 2) X(32, 8, 3) X(64, 8, 2)
#define SMCX_MZ_TABLE(X) X(64, 1, 4) X(32, 2, 4) X(64, 4, 4) X(32, 1, 4) X(64, 2, 4)

static sweep_mx_fn lookup_mx(int S, int WPR)
{
#define SMCX_MX(s, w, m) if (S == s && WPR == w) return sweep_kernel_mx<s, w, m, false>;
    SMCX_MX_TABLE(SMCX_MX)
#undef SMCX_MX
    return nullptr;
}

// fp16 z in LDS: one VGPR per particle
static sweep_mx_fn lookup_mz(int S, int WPR)
{
#define SMCX_MZ(s, w, m) if (S == s && WPR == w) return sweep_kernel_mx<s, w, m, true>;
    SMCX_MZ_TABLE(SMCX_MZ)
#undef SMCX_MZ
    return nullptr;
}

bool mx_supported(int S, int WPR) { return lookup_mx(S, WPR) != nullptr; }

// the launched instantiation as rocprofv3 prints it
const char *mx_kernel_name(int S, int WPR, bool mz)
{
#define SMCX_MX(s, w, m) if (!mz && S == s && WPR == w) return "smcx::sweep_kernel_mx<" #s ", " #w ", " #m ", false>";
    SMCX_MX_TABLE(SMCX_MX)
#undef SMCX_MX
#define SMCX_MZ(s, w, m) if (mz && S == s && WPR == w) return "smcx::sweep_kernel_mx<" #s ", " #w ", " #m ", true>";
    SMCX_MZ_TABLE(SMCX_MZ)
#undef SMCX_MZ
    return "";
}

bool mx_lds_z(int S, int WPR, double Lz, int force)
{
    bool mz = (S == 64 && WPR == 1 && Lz <= 480.0);
    if (force >= 0) mz = (force != 0);
    if (Lz > 32768.0) mz = false; // zsafe (the power of two above Lz/2) must stay a finite fp16
    return mz && lookup_mz(S, WPR) != nullptr;
}

// screening threshold and scales: the bound in the comment of sweep_kernel_mx
static MxArgs mx_bound(double L, double Lz, double cutoff2, bool mz)
{
    const double rc = sqrt(cutoff2), eps = 5.9604644775390625e-8; // 2^-24
    const double zsafe = 2.0 * Lz; // the walls keep particles within Lz/2; beyond zsafe the screen passes everything on
    const double u = L / 65536.0, R = rc / u;
    const double m_xy = (2.0 * sqrt(2.0) * R + 2.0) * u * u;           // fixed-point x,y
    const double m_z = 2.0 * rc * (3.0 * eps * zsafe) + 1e-9;           // fp32 z of particle and probe
    const double margin = (m_xy + m_z) * 1.01 + 8.0 * eps * (cutoff2 + m_xy + m_z) + 1e-6 * cutoff2;
    MxArgs m;
    m.thr = nextafterf((float)(cutoff2 + margin), INFINITY);
    m.u2 = (float)(u * u);
    m.toFix = 65536.0 / L;
    m.zsafe = (float)zsafe;
    if (mz) { // fp16 z: |z| < zsafe_h (power of two) is kept to zsafe_h * 2^-12, particle and probe
        double zs = 1.0;
        while (zs < 0.51 * Lz) zs *= 2.0;
        const double dz = 2.0 * zs / 4096.0 + (rc + 1.0) / 1024.0;
        const double mzh = 2.0 * rc * dz + dz * dz;
        const double margin_h = (m_xy + mzh) * 1.01 + 8.0 * eps * (cutoff2 + m_xy + mzh) + 1e-6 * cutoff2;
        m.thr = nextafterf((float)(cutoff2 + margin_h), INFINITY);
        m.zsafe = (float)zs;
    }
    return m;
}

// host-visible copy of the numbers above (smcx_screen_bound: the CPU test of the bound uses them)
void mx_bound_values(double L, double Lz, double cutoff2, bool lds_z, double *thr, double *u2, double *toFix,
                     double *zsafe)
{
    const MxArgs m = mx_bound(L, Lz, cutoff2, lds_z);
    *thr = m.thr; *u2 = m.u2; *toFix = m.toFix; *zsafe = m.zsafe;
}

hipError_t launch_sweeps_mx(const SweepArgs &a, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A,
                            hipStream_t st)
{
    const int S = pl.S, WPR = pl.WPR;
    sweep_mx_fn fm = lookup_mx(S, WPR);
    // z as fp16 in LDS (ZL = true) where it measured faster: one wavefront per replica with 64
    // particles per lane, which then fits four waves per SIMD -- unless the box is so tall that fp16
    // would widen the screen noticeably (decided in plan_kernel; Tune.mz forces the choice for A/B measurements).
    const bool mz = pl.mz;
    if (mz) fm = lookup_mz(S, WPR);
    if (!fm) return hipErrorInvalidValue;
    const MxArgs m = mx_bound(c.L, c.Lz, c.cutoff2, mz);
    hipLaunchKernelGGL(fm, dim3(c.nrep), dim3(64 * WPR), 0, st, a, c, nsweeps, A, m);
    return hipGetLastError();
}

} // namespace smcx
