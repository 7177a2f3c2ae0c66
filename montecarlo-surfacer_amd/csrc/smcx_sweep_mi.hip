// smcx_sweep_mi.hip -- the screened sweep kernel for ONE wavefront per replica, all-integer screen.
//
// Same algorithm as sweep_kernel_mx (smcx_sweep_mx.hip): the cutoff test of the O(N^2) pair loops
// (SMC.c:563-575, 597-609) is screened on compact copies of the positions with a threshold widened
// by a proven error bound, and every candidate is decided and evaluated in fp64 from the fp64
// positions in memory (mx_exact / lj_acc: the arithmetic of the fp64 kernels).  What differs is the
// instruction stream, rebuilt from the issue costs measured on MI355X (profiles/r02_issue_costs.txt:
// "fast" VALU forms 1.9 SIMD cycles per wave-instruction at four waves per SIMD -- v_sub_u32,
// v_ashrrev, v_mov, v_and/or, VGPR-only fp32 -- everything packed, 3-operand integer, converting,
// fp64 or with an SGPR source 3.2-3.45):
//
//   compact copy   x,y  two int16 in one VGPR, box [-L/2, L/2) on the int16 range (as before);
//                  z    int16 in LDS, unit uz = 2^ZS * (L/65536), slot pairs x lanes;
//   per slot and probe, 5.5 instructions, 15.9 cycles (was 6.5 and 20.7):
//                  v_sub_u32        d  = probe.xy - xy         the packed difference; both halves wrap
//                                                               modulo 2^16 = the minimum image; the
//                                                               borrow into y costs one unit (margin)
//                  v_dot2_i32_i16   I  = dx^2 + dy^2 - C       C = T << 2ZS in the accumulator
//                  v_ashrrev_i32    I >>= 2ZS                   xy units^2 -> z units^2
//                  (1/2 v_pk_sub_i16 clamp: dz of two slots)
//                  v_mad_i32_i16    q  = dz^2 + I               < 0 <=> candidate
//                  v_alignbit_b32   word = {word, q} >> 31      the sign bit is the flag
//   no conversion to floating point, no overflow (|I >> 2ZS| < 2^(31-2ZS), dz^2 <= 2^30), and a
//   threshold 0.6 % above rc^2 at the reference box (fp16 z needed 4.4 %).
//   Probe B (the current position of the next particle) takes its compact copy from the registers
//   (v_readlane of the owner lane) and its fp64 position from a load issued before the screen and
//   awaited after it; the proposal's compact copy is formed on the VALU from the vector registers.
//
// Screening bound (launch_sweeps_mi).  u = L/65536, uz = 2^ZS u, R = rc/u.  Stored x, y are within u/2
// of the true ones, so the integer differences are within 1 (x) and 2 (y: the borrow) of the true
// minimum-image ones; stored z within uz/2, so dz (in uz) within 1.  For a pair inside the cutoff,
// in u^2:  dxi^2 + dyi^2 + 4^ZS dzi^2  <=  R^2 + 2 sqrt(5) R + 5 + 4^ZS (2 R / 2^ZS + 1), and the
// shift drops less than 4^ZS.  T = floor((that + slack) / 4^ZS) + 1 never rejects a pair the fp64
// test accepts.  Particles and probes with |z| >= zsafe = 32767 uz are handled as in
// sweep_kernel_mx: always candidates (per-lane `unsafe` mask; a probe outside flags every slot).
#include "smcx_device.hpp"
#include "smcx_kernels.h"

#include <cmath>
#include <cstdlib>

namespace smcx {

#include "smcx_sweep_common.hpp"

struct MiArgs {
    double toFix; // 65536 / L
    double zFix;  // 1 / uz
    double zsafe; // 32767 uz
    int negC;     // -(T << 2ZS)
};

typedef short mi_s2 __attribute__((ext_vector_type(2)));

// x,y -> two int16 in one register (x low, y high); x = +L/2 wraps onto -L/2, the same point
__device__ __forceinline__ unsigned mi_pack_xy(double x, double y, double toFix)
{
    const int xi = (int)__builtin_rint(x * toFix), yi = (int)__builtin_rint(y * toFix);
    return ((unsigned)xi & 0xffffu) | ((unsigned)yi << 16);
}
// z -> int16 in units uz (the caller keeps |z| >= zsafe out of the screen)
__device__ __forceinline__ unsigned mi_z16(double z, double zFix)
{
    int zi = (int)__builtin_rint(z * zFix); // v_cvt_i32_f64 saturates
    zi = zi < -32767 ? -32767 : (zi > 32767 ? 32767 : zi);
    return (unsigned)zi & 0xffffu;
}

// dz^2 + acc with dz the low / high int16 of `pair`: one v_mad_i32_i16 (hipcc would unpack the
// halves with two more instructions)
template <bool HI> __device__ __forceinline__ int mi_mad16(unsigned pair, int acc)
{
    int q;
    if constexpr (HI) asm("v_mad_i32_i16 %0, %1, %1, %2 op_sel:[1,1,0,0]" : "=v"(q) : "v"(pair), "v"(acc));
    else asm("v_mad_i32_i16 %0, %1, %1, %2" : "=v"(q) : "v"(pair), "v"(acc));
    return q;
}
// dx^2 + dy^2 + negC of the packed difference d (hipcc would copy negC to a VGPR for v_dot2c)
__device__ __forceinline__ int mi_dot2(unsigned d, int negC)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %1, %2" : "=v"(r) : "v"(d), "s"(negC));
    return r;
}

// {word, q} >> 31: shifts the sign of q into the word.  As an opaque statement: written with the
// builtin, the optimiser rewrites the 64-step shift chain into an OR tree over all the q's, keeps
// them live together and spills the positions
__device__ __forceinline__ unsigned mi_flag(unsigned word, int q)
{
    unsigned r;
    asm("v_alignbit_b32 %0, %1, %2, 31" : "=v"(r) : "v"(word), "v"(q));
    return r;
}

// One group of the screen: four slots (k0 .. k0+3, positions xy0..xy3, z pairs z01 = slots k0, k0+1 and
// z23) against both probes, as ONE statement of 44 instructions.  The eight chains are interleaved so
// that every result is consumed at least seven instructions after it was issued: no wait states are
// needed anywhere (a v_dot2 result wants three before a dependent VALU read, which hipcc cannot see
// inside a statement and therefore cannot pad), and no instruction waits on its predecessor.
// Slots are taken in descending order, so the shifted-in sign bits land in place without a reversal.
template <int ZS>
__device__ __forceinline__ void mi_screen4(unsigned xy0, unsigned xy1, unsigned xy2, unsigned xy3, unsigned z01,
                                           unsigned z23, unsigned axy, unsigned azz, unsigned bxy, unsigned bzz,
                                           int negC, unsigned &wa, unsigned &wb)
{
    unsigned a0, a1, a2, a3, b0, b1, b2, b3, za1, zb1, za0, zb0;
    asm("v_sub_u32 %2, %14, %21\n\t"  "v_sub_u32 %6, %16, %21\n\t"
        "v_sub_u32 %3, %14, %20\n\t"  "v_sub_u32 %7, %16, %20\n\t"
        "v_sub_u32 %4, %14, %19\n\t"  "v_sub_u32 %8, %16, %19\n\t"
        "v_sub_u32 %5, %14, %18\n\t"  "v_sub_u32 %9, %16, %18\n\t"
        "v_dot2_i32_i16 %2, %2, %2, %24\n\t"  "v_dot2_i32_i16 %6, %6, %6, %24\n\t"
        "v_dot2_i32_i16 %3, %3, %3, %24\n\t"  "v_dot2_i32_i16 %7, %7, %7, %24\n\t"
        "v_dot2_i32_i16 %4, %4, %4, %24\n\t"  "v_dot2_i32_i16 %8, %8, %8, %24\n\t"
        "v_dot2_i32_i16 %5, %5, %5, %24\n\t"  "v_dot2_i32_i16 %9, %9, %9, %24\n\t"
        "v_pk_sub_i16 %10, %15, %23 clamp\n\t"  "v_pk_sub_i16 %11, %17, %23 clamp\n\t"
        "v_pk_sub_i16 %12, %15, %22 clamp\n\t"  "v_pk_sub_i16 %13, %17, %22 clamp\n\t"
        "v_ashrrev_i32 %2, %25, %2\n\t"  "v_ashrrev_i32 %6, %25, %6\n\t"
        "v_ashrrev_i32 %3, %25, %3\n\t"  "v_ashrrev_i32 %7, %25, %7\n\t"
        "v_ashrrev_i32 %4, %25, %4\n\t"  "v_ashrrev_i32 %8, %25, %8\n\t"
        "v_ashrrev_i32 %5, %25, %5\n\t"  "v_ashrrev_i32 %9, %25, %9\n\t"
        "v_mad_i32_i16 %2, %10, %10, %2 op_sel:[1,1,0,0]\n\t"  "v_mad_i32_i16 %6, %11, %11, %6 op_sel:[1,1,0,0]\n\t"
        "v_mad_i32_i16 %3, %10, %10, %3\n\t"                   "v_mad_i32_i16 %7, %11, %11, %7\n\t"
        "v_mad_i32_i16 %4, %12, %12, %4 op_sel:[1,1,0,0]\n\t"  "v_mad_i32_i16 %8, %13, %13, %8 op_sel:[1,1,0,0]\n\t"
        "v_mad_i32_i16 %5, %12, %12, %5\n\t"                   "v_mad_i32_i16 %9, %13, %13, %9\n\t"
        "v_alignbit_b32 %0, %0, %2, 31\n\t"  "v_alignbit_b32 %1, %1, %6, 31\n\t"
        "v_alignbit_b32 %0, %0, %3, 31\n\t"  "v_alignbit_b32 %1, %1, %7, 31\n\t"
        "v_alignbit_b32 %0, %0, %4, 31\n\t"  "v_alignbit_b32 %1, %1, %8, 31\n\t"
        "v_alignbit_b32 %0, %0, %5, 31\n\t"  "v_alignbit_b32 %1, %1, %9, 31"
        : "+v"(wa), "+v"(wb),                                               // 0, 1
          "=&v"(a3), "=&v"(a2), "=&v"(a1), "=&v"(a0),                       // 2..5: probe A, slots k0+3 .. k0
          "=&v"(b3), "=&v"(b2), "=&v"(b1), "=&v"(b0),                       // 6..9: probe B
          "=&v"(za1), "=&v"(zb1), "=&v"(za0), "=&v"(zb0)                    // 10..13: dz of (k0+2, k0+3), (k0, k0+1)
        : "v"(axy), "s"(azz), "v"(bxy), "s"(bzz),                           // 14..17
          "v"(xy0), "v"(xy1), "v"(xy2), "v"(xy3), "v"(z01), "v"(z23),       // 18..23
          "s"(negC), "n"(2 * ZS));                                          // 24, 25
}

// the screen: candidate bits of probes A and B, slot k in bit k % 32 of word k / 32
template <int S, int ZS>
__device__ __forceinline__ void mi_screen(const unsigned (&xy)[S], const unsigned (&zw)[S / 2][64], int lane,
                                          unsigned axy, unsigned azz, unsigned bxy, unsigned bzz, int negC,
                                          unsigned (&ca)[(S + 31) / 32], unsigned (&cb)[(S + 31) / 32])
{
    static_assert(S % 4 == 0, "groups of four slots");
    unsigned zn[2] = {zw[S / 2 - 2][lane], zw[S / 2 - 1][lane]}; // the LDS reads run one group ahead
#pragma unroll
    for (int k0 = S - 4; k0 >= 0; k0 -= 4) {
        const unsigned zc[2] = {zn[0], zn[1]};
        if (k0 >= 4) { zn[0] = zw[k0 / 2 - 2][lane]; zn[1] = zw[k0 / 2 - 1][lane]; }
        mi_screen4<ZS>(xy[k0], xy[k0 + 1], xy[k0 + 2], xy[k0 + 3], zc[0], zc[1], axy, azz, bxy, bzz, negC,
                       ca[k0 >> 5], cb[k0 >> 5]);
    }
}

// rotate this wavefront's int16 z in LDS by one slot (slots 2j, 2j+1 of a lane share a dword)
template <int S>
__device__ __forceinline__ void mi_rotate_lds(unsigned (&zw)[S / 2][64], int lane)
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

// sum of four per-lane values over the 64 lanes: on return the lanes of 16-lane row r hold the
// wave total of v_r.  21 VALU instructions, no LDS (v_permlane32/16_swap, DPP row operations)
__device__ __forceinline__ double reduce4(double v0, double v1, double v2, double v3)
{
    const double w0 = sum_swap32(v0, v2), w1 = sum_swap32(v1, v3); // low half: v0 | v1, high half: v2 | v3
    double r = sum_swap16(w0, w1);                                  // rows: v0, v1, v2, v3
    r += dpp_mov<DPP_ROR8>(r);
    r += dpp_mov<DPP_HALF_MIRROR>(r);
    r += dpp_mov<DPP_QUAD_X2>(r);
    r += dpp_mov<DPP_QUAD_X1>(r);
    return r;
}

// wave-uniform constants of the wall terms (K3/K4, SMC.c:729-813)
struct MiWalls {
    int on, M, M2;
    double dw, Lz, invLz, halfLz;
    const double *Wx; // [2 (M2 + 1)]: W[2m], W[2m+1] of the sites, then a0, b0 of the plane
};

constexpr int MI_SIDE_LANE = 30; // the lane that evaluates the pair (particle n, probe B)

// take this lane's lowest candidate out of `w` and start the load of its fp64 position
template <int S>
__device__ __forceinline__ void mi_fetch(const double *Rg, int N, int lane, int rot, bool skip, unsigned long long &w,
                                         double &X, double &Y, double &Z, bool &have)
{
    have = false; X = 0.0; Y = 0.0; Z = 0.0;
    if (!skip && w != 0ull) {
        int ls = __builtin_ctzll(w) + rot; if (ls >= S) ls -= S;
        w &= w - 1ull;
        const int l = ls * 64 + lane;
        if (l < N) { // a padding slot is flagged only through an unsafe probe
            const double *q = Rg + 3 * l;
            X = q[0]; Y = q[1]; Z = q[2];
            have = true;
        }
    }
}

// One probe against everything it interacts with (K1-K4 for one configuration, SMC.c:300-304 or
// 319-321): the wall sites and the plane on lanes 0..M2 (round 0), optionally the pair with one
// given position on lane 30 (round 0), and this lane's candidates from the screen, one per round,
// their fp64 positions fetched from memory (the first one by the caller, ahead of time: X, Y, Z,
// have).  ONE body of the fp64 arithmetic (signed minimum image, cutoff test, lj_acc's sequence)
// serves all of them; the items differ only in where dx, dy, dz and the two coefficients come
// from.  Returns e, fx, fy, fz summed over the wavefront.
template <int S>
__device__ __forceinline__ void mi_probe(const Geo &g, const MiWalls &wl, const double *Rg, int N, int lane, int rot,
                                         int site, double px, double py, double pz, unsigned long long w,
                                         double X, double Y, double Z, bool have,
                                         bool withSide, double sx, double sy, double sz,
                                         double &E4, double &Fx, double &Fy, double &Fz)
{
    double e = 0.0, fx = 0.0, fy = 0.0, fz = 0.0;
    bool first = true;
    for (;;) {
        const bool wall = first && wl.on && lane <= wl.M2;
        const bool plane = wall && lane == wl.M2;
        const bool side = first && withSide && lane == MI_SIDE_LANE;
        double dx = 0.0, dy = 0.0, dz = 0.0, ca = 1.0, cb = 1.0;
        if (wall) {   // K3/K4: site (i dw, j dw) or the featureless plane, distance to the nearer wall
            const double2 cc = *reinterpret_cast<const double2 *>(wl.Wx + 2 * lane);
            ca = cc.x; cb = cc.y;
            dx = px - (double)(site & 0xff) * wl.dw;   // SMC.c:748-750
            dy = py - (double)(site >> 8) * wl.dw;
            dz = wall_dz(wl, pz);
            have = true;
        } else if (side) {
            dx = px - sx; dy = py - sy; dz = pz - sz;
            have = true;
        } else if (have) {
            dx = px - X; dy = py - Y; dz = pz - Z;
        }
        if (have) {
            double mx = dx - g.L * __builtin_rint(dx * g.invL); // SMC.c:571-572, 605-606, 751-752
            double my = dy - g.L * __builtin_rint(dy * g.invL);
            if (plane) { mx = 0.0; my = 0.0; }                   // SMC.c:740-741, 787-789: no x,y, no cutoff
            const double dr2 = mx * mx + my * my + dz * dz;
            if (plane || dr2 < g.cutoff2) lj_acc(mx, my, dz, dr2, ca, cb, e, fx, fy, fz);
        }
        if (!__builtin_amdgcn_ballot_w64(w != 0ull)) break;
        first = false;
        mi_fetch<S>(Rg, N, lane, rot, false, w, X, Y, Z, have);
    }
    const double r = reduce4(e, fx, fy, fz);
    E4 = rdlane(r, 0); Fx = rdlane(r, 16); Fy = rdlane(r, 32); Fz = rdlane(r, 48);
}

#ifdef SMCX_STAMPS // diagnostic build only (tools/phase_stamps.py): cycles per phase of a move, summed in LDS by lane 0
#define MI_STAMP(k) do { const long long t_ = __builtin_amdgcn_s_memtime(); \
                         if (lane == 0) phs[k] += (unsigned long long)(t_ - tlast); tlast = t_; } while (0)
#else
#define MI_STAMP(k) do { } while (0)
#endif

template <int S, int ZS, int MINW>
__global__ void __launch_bounds__(64, MINW)
sweep_kernel_mi(SweepArgs a, MiWalls wl, int nsweeps, double A, MiArgs m)
{
    constexpr int NW = (S + 31) / 32;
    typedef unsigned long long u64;
    __shared__ unsigned zw[S / 2][64]; // int16 z, slot pairs x lanes
    __shared__ double p0[65][3];       // fp64 position of every lane's slot-0 particle (the next probes B);
                                       // row 64: lane 0's slot-1 particle, probe B when the order crosses slots
#ifdef SMCX_STAMPS
    __shared__ unsigned long long phs[8];
    if (threadIdx.x < 8) phs[threadIdx.x] = 0ull;
    long long tlast = __builtin_amdgcn_s_memtime();
#endif

    const int rep = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = a.N;
    double *Rg = a.R + (size_t)rep * 3 * N;
    clock_stamp(a.clk, rep, 0);

    // ---- compact copies: particle l in lane l % 64, slot l / 64 --------------------------
    unsigned xy[S];
    u64 unsafe = 0ull; // slots of this lane with |z| >= zsafe (always candidates), bit = register slot
#pragma unroll
    for (int j = 0; j < S / 2; j++) {
        unsigned pair = 0u;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int k = 2 * j + h;
            const int l = k * 64 + lane;
            const bool real = (l < N);
            const double zr = real ? Rg[3 * l + 2] : 0.0;
            xy[k] = real ? mi_pack_xy(Rg[3 * l], Rg[3 * l + 1], m.toFix) : 0u;
            pair |= (real ? mi_z16(zr, m.zFix) : 0x7fffu) << (16 * h);
            if (real && !(fabs(zr) < m.zsafe)) unsafe |= 1ull << k;
        }
        zw[j][lane] = pair;
    }
    int rot = 0; // register slot j holds logical slot (j + rot) % S
    auto fill_p0 = [&]() { // the fp64 positions of the particles now in slot 0, and of the first one of slot 1
        const int l = rot * 64 + lane;
        if (l < N) { p0[lane][0] = Rg[3 * l]; p0[lane][1] = Rg[3 * l + 1]; p0[lane][2] = Rg[3 * l + 2]; }
        const int l1 = (rot + 1) * 64; // no move order crosses from the last slot to the first
        if (lane == 0 && l1 < N) { p0[64][0] = Rg[3 * l1]; p0[64][1] = Rg[3 * l1 + 1]; p0[64][2] = Rg[3 * l1 + 2]; }
    };
    auto rotate = [&]() {
        const unsigned t = xy[0];
#pragma unroll
        for (int k = 0; k + 1 < S; k++) xy[k] = xy[k + 1];
        xy[S - 1] = t;
        mi_rotate_lds<S>(zw, lane);
        unsafe = (unsafe >> 1) | ((unsafe & 1ull) << (S - 1));
        rot = (rot + 1 == S) ? 0 : rot + 1;
        fill_p0();
    };
    fill_p0();
    const int site = wl.on && lane < wl.M2 ? ((lane / wl.M) | ((lane % wl.M) << 8)) : 0; // SMC.c:748-750

    Geo g; g.L = a.L; g.invL = a.invL; g.cutoff2 = a.cutoff2;
    double E = uniform_d(a.obs[rep].Ecur);
#ifdef SMCX_CHECK // diagnostic build only (libsmcx_check.so): the fp64 test beside the screen, every slot
    u64 chk_in = 0, chk_cand = 0, chk_miss = 0;
#endif
    const double AoT = A * a.invT;         // SMC.c:307-309 (A/T)
    const double Ao4T = A * 0.25 * a.invT; // SMC.c:327 (A/(4T))

#pragma unroll 1
    for (int sw = 0; sw < nsweeps; sw++) {
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
            const int ks = first / 64;
            while (rot != ks) rotate();
            int tl = first - ks * 64 - 1; // owner lane of particle n (slot 0); -1 in the prologue

            // wave-uniform state of the chain between two moves
            double Px = 0.0, Py = 0.0, Pz = FAR_PROBE;            // current position of particle n
            double Um = 0.0, Fmx = 0.0, Fmy = 0.0, Fmz = 0.0;     // its energy and force (SMC.c:300-304)
            double Qx = 0.0, Qy = 0.0, Qz = FAR_PROBE;            // its proposal (SMC.c:307-316)
            double ddx = 0.0, ddy = 0.0, ddz = 0.0, lu = 0.0;     // displacement and log-uniform of the move
            unsigned axy = 0u, azz = 0u, az16 = 0u;               // compact copy of the proposal
            bool ua = false;
            // displacements and log-uniforms are wave-uniform and read-only for the whole kernel:
            // scalar loads through the constant address space
            typedef const __attribute__((address_space(4))) double *kptr;
            const kptr dK = (kptr)(unsigned long long)displ + 3 * (size_t)first;
            const kptr uK = (kptr)(unsigned long long)uni + vbase;
#pragma unroll 1
            for (int i = -1; i < len; i++) { // i = -1: the run's prologue, probe B alone
                const int n = first + i;
                const bool hasA = (i >= 0);
                const bool hasB = (i + 1 < len);
                const bool cross = hasB && (tl == 63);
                MI_STAMP(0); // loop control, scalar loads
                // ---- probe B = current position of particle n+1: the compact copy its owner lane holds ----
                unsigned bxy = 0u, bzz = 0u;
                bool ub = false;
                double Bxv = 0.0, Byv = 0.0, Bzv = FAR_PROBE; // its fp64 position: asked for here, used after probe A
                if (hasB) {
                    const unsigned zp0 = zw[0][lane]; // slots 0 and 1 of every lane
                    unsigned bz16;
                    {
                        const int row = cross ? 64 : tl + 1;
                        Bxv = p0[row][0]; Byv = p0[row][1]; Bzv = p0[row][2];
                    }
                    if (cross) {
                        bxy = (unsigned)__builtin_amdgcn_readlane((int)xy[S > 1 ? 1 : 0], 0);
                        bz16 = (unsigned)__builtin_amdgcn_readlane((int)zp0, 0) >> 16;
                        ub = ((unsigned)__builtin_amdgcn_readlane((int)(unsigned)unsafe, 0) >> 1) & 1u;
                    } else {
                        bxy = (unsigned)__builtin_amdgcn_readlane((int)xy[0], tl + 1);
                        bz16 = (unsigned)__builtin_amdgcn_readlane((int)zp0, tl + 1) & 0xffffu;
                        ub = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)unsafe, tl + 1) & 1u;
                    }
                    bzz = bz16 * 0x10001u;
                }
                // the x,y probes go through vector registers: v_sub_u32 with two VGPR sources issues in
                // 1.9 cycles, with an SGPR source in 3.2
                unsigned axyv, bxyv;
                asm("v_mov_b32 %0, %1" : "=v"(axyv) : "s"(axy));
                asm("v_mov_b32 %0, %1" : "=v"(bxyv) : "s"(bxy));

                MI_STAMP(1); // probe B's compact copy
                // ---- screening ------------------------------------------------------
                unsigned ca[NW], cb[NW];
#pragma unroll
                for (int w = 0; w < NW; w++) { ca[w] = 0u; cb[w] = 0u; }
                mi_screen<S, ZS>(xy, zw, lane, axyv, azz, bxyv, bzz, m.negC, ca, cb);
#ifdef SMCX_SCREEN_TWICE // timing experiment only: the screen's cost in place = the time this build adds
                {
                    unsigned ca2[NW], cb2[NW];
#pragma unroll
                    for (int w = 0; w < NW; w++) { ca2[w] = 0u; cb2[w] = 0u; }
                    asm volatile("" : "+v"(axyv), "+v"(bxyv));
                    mi_screen<S, ZS>(xy, zw, lane, axyv, azz, bxyv, bzz, m.negC, ca2, cb2);
#pragma unroll
                    for (int w = 0; w < NW; w++) { ca[w] |= ca2[w]; cb[w] |= cb2[w]; }
                }
#endif
                MI_STAMP(2); // screen
                // displacement and log-uniform of move i + 1: scalar loads, asked for now and used at the end of
                // the move.  NOT before the screen: a scalar load in flight shares the LDS counter and returns
                // out of order, which turns every counted wait of the screen's LDS read-ahead into a full drain
                double ndx = 0.0, ndy = 0.0, ndz = 0.0, nlu = 0.0;
                if (hasB) {
                    ndx = dK[3 * (i + 1)]; ndy = dK[3 * (i + 1) + 1]; ndz = dK[3 * (i + 1) + 2];
                    nlu = uK[i + 1];
                }
                u64 wa = ca[0], wb = cb[0];
                if constexpr (NW > 1) { wa |= (u64)ca[1] << 32; wb |= (u64)cb[1] << 32; }
                {   // particles outside the safe z range are always candidates, a probe outside it flags
                    // every slot; a disabled probe flags none
                    constexpr u64 all = (S == 64) ? ~0ull : ((1ull << (S & 63)) - 1ull);
                    wa |= unsafe; wb |= unsafe;
                    if (ua) wa = all;
                    if (ub) wb = all;
                    if (!hasA) wa = 0ull;
                    if (!hasB) wb = 0ull;
                }
                // the moving particle itself and the particle probe B stands for are not neighbours;
                // particle n reaches probe B through the side pair, at the position the move leaves it
                const bool exB0 = (hasA && lane == tl) || (hasB && !cross && lane == tl + 1);
                if (lane == tl) wa &= ~1ull;
                if (exB0) wb &= ~1ull;
                if (cross && lane == 0) wb &= ~2ull;
#ifdef SMCX_CHECK
                {   // every slot of this lane against both probes in fp64 (pair_hit's arithmetic, positions from
                    // memory): a pair inside the cutoff that the screen did not flag is a miss
                    double cBx = 0.0, cBy = 0.0, cBz = FAR_PROBE;
                    if (hasB) { cBx = ld_coherent(Rg + 3 * (n + 1)); cBy = ld_coherent(Rg + 3 * (n + 1) + 1); cBz = ld_coherent(Rg + 3 * (n + 1) + 2); }
                    chk_cand += __builtin_popcountll(wa) + __builtin_popcountll(wb);
#pragma unroll 1
                    for (int k = 0; k < S; k++) {
                        int ls = k + rot; if (ls >= S) ls -= S;
                        const int l = ls * 64 + lane;
                        if (l >= N) continue;
                        const double X = ld_coherent(Rg + 3 * l), Y = ld_coherent(Rg + 3 * l + 1), Z = ld_coherent(Rg + 3 * l + 2);
                        for (int pr = 0; pr < 2; pr++) {
                            if (pr == 0 ? !hasA : !hasB) continue;
                            if (pr == 0 ? (k == 0 && lane == tl) : ((k == 0 && exB0) || (k == 1 && cross && lane == 0))) continue;
                            const double dx = (pr ? cBx : Qx) - X, dy = (pr ? cBy : Qy) - Y, dz = (pr ? cBz : Qz) - Z;
                            const double sx = dx - g.L * __builtin_rint(dx * g.invL);
                            const double sy = dy - g.L * __builtin_rint(dy * g.invL);
                            if (sx * sx + sy * sy + dz * dz < g.cutoff2) {
                                chk_in++;
                                if (!(((pr ? wb : wa) >> k) & 1ull)) chk_miss++;
                            }
                        }
                    }
                }
#endif

                // the first candidate of either probe: both loads are in flight while probe A is evaluated;
                // the lanes that stand for a wall site, the plane or the side pair take theirs a round later
                double XA, YA, ZA, XB, YB, ZB;
                bool gotA, gotB;
                const bool wallLane = wl.on && lane <= wl.M2;
                mi_fetch<S>(Rg, N, lane, rot, wallLane, wa, XA, YA, ZA, gotA);
#ifndef SMCX_MI_LATE_B
                mi_fetch<S>(Rg, N, lane, rot, wallLane || (hasA && lane == MI_SIDE_LANE), wb, XB, YB, ZB, gotB);
#endif
                __builtin_amdgcn_sched_barrier(0);
                MI_STAMP(3); // unsafe / exclusion bits, first fetches
                // ---- the proposal of particle n: Un, Fn, the Metropolis step (SMC.c:319-348) --------
                bool acc = false;
                if (hasA) {
                    double Un, Fnx, Fny, Fnz;
                    mi_probe<S>(g, wl, Rg, N, lane, rot, site, Qx, Qy, Qz, wa, XA, YA, ZA, gotA, false, 0.0, 0.0, 0.0,
                                Un, Fnx, Fny, Fnz);
                    MI_STAMP(4); // probe A: walls, candidates, reduction
                    Un *= 4.0;
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
                    if (acc) {
                        if (lane == tl) {
                            xy[0] = axyv;
                            unsafe = (unsafe & ~1ull) | (ua ? 1ull : 0ull);
                            reinterpret_cast<unsigned short *>(&zw[0][lane])[0] = (unsigned short)az16;
                            Rg[3 * n] = Qx; Rg[3 * n + 1] = Qy; Rg[3 * n + 2] = Qz; // the fp64 state
                            p0[lane][0] = Qx; p0[lane][1] = Qy; p0[lane][2] = Qz;
                        }
                        E = uniform_d(E + (Un - Um)); jacc++;
                        Px = Qx; Py = Qy; Pz = Qz; // where the move leaves particle n
                    }
                }
                MI_STAMP(5); // Metropolis step, update
                // ---- particle n+1 at its current position: its Um, Fm (SMC.c:300-304), then its proposal ----
                if (hasB) {
                    const double Bx = uniform_d(Bxv), By = uniform_d(Byv), Bz = uniform_d(Bzv);
#ifdef SMCX_MI_LATE_B
                    mi_fetch<S>(Rg, N, lane, rot, wallLane || (hasA && lane == MI_SIDE_LANE), wb, XB, YB, ZB, gotB);
#endif
                    double e4;
                    mi_probe<S>(g, wl, Rg, N, lane, rot, site, Bx, By, Bz, wb, XB, YB, ZB, gotB, hasA, Px, Py, Pz,
                                e4, Fmx, Fmy, Fmz);
                    Um = 4.0 * e4;
                    MI_STAMP(6); // probe B: walls, side pair, candidates, reduction
                    Px = Bx; Py = By; Pz = Bz;
                    ddx = ndx; ddy = ndy; ddz = ndz; lu = nlu;
                    {   // proposal of particle n+1 (SMC.c:307-316); its compact copy from the vector registers
                        double qx = Px + (Fmx * AoT + ddx);
                        double qy = Py + (Fmy * AoT + ddy);
                        const double qz = Pz + (Fmz * AoT + ddz);
                        qx = qx - a.L * __builtin_rint(qx * a.invL);
                        qy = qy - a.L * __builtin_rint(qy * a.invL);
                        axy = (unsigned)uniform((int)mi_pack_xy(qx, qy, m.toFix));
                        az16 = (unsigned)uniform((int)mi_z16(qz, m.zFix));
                        azz = az16 * 0x10001u;
                        ua = uniform((int)!(fabs(qz) < m.zsafe)) != 0;
                        Qx = uniform_d(qx); Qy = uniform_d(qy); Qz = uniform_d(qz);
                    }
                    if (cross) { rotate(); tl = 0; }
                    else tl++;
                }
                MI_STAMP(7); // next proposal, slot rotation
            }
        }
        // C: hand E[n+1] and jj[n] (SMC.c:194-195) to the bookkeeping kernel
        if (lane == 0) {
            SweepRec r; r.E = E; r.accepted = jacc; r.pad = 0;
            a.rec[(size_t)rep * a.chunk + sw] = r;
        }
    }
    clock_stamp(a.clk, rep, 1);
#ifdef SMCX_CHECK
    atomicAdd(&a.dbg[0], chk_in); atomicAdd(&a.dbg[1], chk_cand); atomicAdd(&a.dbg[2], chk_miss);
#endif
#ifdef SMCX_STAMPS
    if (lane < 8) { // diagnostic: overwrite the head of this replica's (consumed) displacement block
        double *dbg = const_cast<double *>(a.displ) + (size_t)rep * a.chunk * 3 * N;
        dbg[lane] = (double)phs[lane];
    }
#endif
}

typedef void (*sweep_mi_fn)(SweepArgs, MiWalls, int, double, MiArgs);

// (particles per lane, z unit shift, waves per SIMD the register budget is set for)
#define SMCX_MI_TABLE(X) X(16, 4, 4) X(32, 4, 4) X(64, 4, 4) X(16, 6, 4) X(32, 6, 4) X(64, 6, 4)

static sweep_mi_fn lookup_mi(int S, int ZS)
{
#define SMCX_MI(s, z, w) if (S == s && ZS == z) return sweep_kernel_mi<s, z, w>;
    SMCX_MI_TABLE(SMCX_MI)
#undef SMCX_MI
    return nullptr;
}

// z unit shift for this box: the smallest built one whose range 32767 uz covers the walls with room
// to spare (particles beyond it are still handled, one by one, as always-candidates); 0 = none fits
static int mi_zshift(double L, double Lz)
{
    for (int zs : {4, 6})
        if (32767.0 * std::ldexp(L / 65536.0, zs) >= 0.55 * Lz) return zs;
    return 0;
}

static bool mi_bound(double L, double Lz, double cutoff2, int ZS, MiArgs *out)
{
    const double u = L / 65536.0, rc = std::sqrt(cutoff2), R = rc / u, k = std::ldexp(1.0, 2 * ZS);
    if (!(R < 32000.0)) return false; // L > 2.05 rc: dx^2 + dy^2 - C must stay inside int32
    const double margin = 2.0 * std::sqrt(5.0) * R + 5.0 + k * (2.0 * R / std::ldexp(1.0, ZS) + 1.0) + k;
    const double total = (R * R + margin) * (1.0 + 1e-9) + 2.0;
    const double T = std::floor(total / k) + 1.0;
    if (!(T * k < 2.0e9)) return false;
    out->toFix = 65536.0 / L;
    out->zFix = 1.0 / std::ldexp(u, ZS);
    out->zsafe = 32767.0 * std::ldexp(u, ZS);
    out->negC = -(int)(T * k);
    (void)Lz;
    return true;
}

int mi_built(int S, double L, double Lz, double cutoff2)
{
    const int zs = mi_zshift(L, Lz);
    MiArgs m;
    return (zs != 0 && lookup_mi(S, zs) != nullptr && mi_bound(L, Lz, cutoff2, zs, &m)) ? zs : 0;
}

// the launched instantiation as rocprofv3 prints it
const char *mi_kernel_name(int S, int zs)
{
#define SMCX_MI(s, z, w) if (S == s && zs == z) return "smcx::sweep_kernel_mi<" #s ", " #z ", " #w ">";
    SMCX_MI_TABLE(SMCX_MI)
#undef SMCX_MI
    return "";
}

// host-visible copy of the screen's numbers (smcx_screen_bound: the CPU test of the bound uses them):
// threshold in length^2, xy unit^2, 65536/L, zsafe, z unit
void mi_bound_values(double L, double Lz, double cutoff2, double *thr, double *u2, double *toFix, double *zsafe,
                     double *uz, int *negC, int *zshift)
{
    const int zs = mi_zshift(L, Lz);
    MiArgs m = {0, 0, 0, 0};
    *zshift = zs;
    if (zs == 0 || !mi_bound(L, Lz, cutoff2, zs, &m)) { *zshift = 0; return; }
    const double u = L / 65536.0;
    *thr = -(double)m.negC * u * u; *u2 = u * u; *toFix = m.toFix; *zsafe = m.zsafe; *uz = 1.0 / m.zFix; *negC = m.negC;
}

hipError_t launch_sweeps_mi(const SweepArgs &a, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A,
                            hipStream_t st, SweepTimer *tm)
{
    const int S = pl.S, zs = pl.zs;
    sweep_mi_fn f = zs ? lookup_mi(S, zs) : nullptr;
    MiArgs m;
    if (!f || !mi_bound(c.L, c.Lz, c.cutoff2, zs, &m)) return hipErrorInvalidValue;
    // 64 particles per lane with the standard z unit: the hand-scheduled form of this kernel
    if (pl.form >= FORM_MA) {
        if (!c.wtab) return hipErrorInvalidValue;
        return launch_sweeps_ma(a, c, pl, c.wtab, nsweeps, A, m.toFix, m.zFix, m.zsafe, m.negC, st, tm);
    }
    MiWalls wl;
    wl.on = (c.flags & 0x1u) ? 1 : 0; wl.M = c.M; wl.M2 = c.M2;
    wl.dw = c.L / c.M; wl.Lz = c.Lz; wl.invLz = c.invLz; wl.halfLz = c.halfLz;
    wl.Wx = c.W; // smcx_create / smcx_upload keep a0, b0 behind the 2 M2 site strengths
    hipError_t rc = tm ? tm->mark(st) : hipSuccess;
    if (rc != hipSuccess) return rc;
    hipLaunchKernelGGL(f, dim3(c.nrep), dim3(64), 0, st, a, wl, nsweeps, A, m);
    rc = hipGetLastError();
    if (rc == hipSuccess && tm) rc = tm->mark(st);
    return rc;
}

} // namespace smcx
