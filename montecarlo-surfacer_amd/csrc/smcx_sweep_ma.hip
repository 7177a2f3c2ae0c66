// smcx_sweep_ma.hip -- sweep_kernel_ma: the screened sweep kernel for one wavefront per replica and 64
// particles per lane (2048 < N <= 4096), with its body written instruction by instruction
// (gen_sweep_ma.py -> smcx_sweep_ma_body.inc).  Same algorithm and per-pair arithmetic as
// sweep_kernel_mi (smcx_sweep_mi.hip); see the generator for the layout of a move.
#include "smcx_device.hpp"
#include "smcx_kernels.h"

#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <vector>

namespace smcx {

// kernel arguments, read by the body with scalar loads at these offsets (gen_sweep_ma.py: K_*)
struct MaArgs {
    double *R;                   // 0x00
    const double *displ;         // 0x08
    const double *uni;           // 0x10
    const int *offs;             // 0x18
    const ObsRec *obs;           // 0x20
    SweepRec *rec;               // 0x28
    const double *wtab;          // 0x30  [M2 + 1][4]: site x, y, then the two coefficients (the plane last)
    unsigned long long *clk;     // 0x38
    double L, invL, cutoff2, invT, AoT, Ao4T, toFix, zFix;   // 0x40 .. 0x78
    double zsafe, halfLz, Lz, invLz;                          // 0x80 .. 0x98
    int N, chunk, nsweeps, negC;                              // 0xa0 .. 0xac
    int M2, RZ;                                               // 0xb0  (M2 < 0: no walls) ; mb: reach of the screen in z units
    double *Rs;                                               // 0xb8  mb: positions in cell order [nrep][S*64][3]
    const unsigned short *loc;                                // 0xc0  mb: cell of each particle [nrep][N]
    int sw0, pad0;                                            // 0xc8  mb: first sweep of this launch within the chunk
    unsigned long long *dbg;                                  // 0xd0  diagnostic build: the screen's counters
    unsigned *prio;                                           // 0xd8  mb/mc: progress table of the SIMDs' wavefronts [16384]
    double znear;                                             // 0xe0  mg: |z| from which a wall's sites can be inside the cutoff
};
static_assert(sizeof(MaArgs) == 0xe8, "offsets are hard-wired in gen_sweep_ma.py");

// What a launch of a z-ordered sweep kernel really takes (round 5).  A plain launch: workgroup b runs replica b with the
// arguments `a` (use = 0).  A WINDOW of units (use = 1): the `nsweeps` of a launch group are cut into blocks of `every` sweeps
// (one z sort each) and unit u = block * nmod + replica; workgroup b of the launch that starts at unit u0 runs replica
// (u0 + b) % nmod in block (u0 + b) / nmod, with the arguments `a` if that is block blk0 and `b` if it is blk0 + 1 (a launch of
// at most nmod units spans two blocks at most; a and b differ in sw0 and nsweeps only).  The body reads its arguments through
// the kernarg pointer it is handed, so the wrapper hands it the address of `a` or of `b` inside the kernarg segment.
// Why: a sweep is sequential inside a replica and the device holds G replicas at once (smcx_replica_granule), so a launch of
// nrep = q G + r replicas costs q + 1 rounds, the last one nearly empty (4097 replicas of N = 4096: 1.55 x the time of 4096,
// 6144: the price of 8192, profiles/r04_replica_cliff.txt).  Launches of G consecutive UNITS keep every round full: the
// replicas of a launch sit in different blocks of the group, which is legal because replicas are independent chains
// (SMC.c:40, 66-95: the reference's ranks) and the random numbers of the whole group are generated before it starts; unit
// (replica, block + 1) lies nmod >= G units behind (replica, block), i.e. in a LATER launch of the same stream.
struct MaArgs2 {
    MaArgs a, b;
    unsigned u0, nmod, blk0, use;
};

// |z| below which no wall SITE can be within the cutoff of a probe and the clamp of SMC.c:736-739 cannot apply: the wall is
// Lz/2 - |z| away and a site at least that far; a relative margin covers the roundings of the kernel's own wall distance.
// Probes beyond it take the path that evaluates the sites (always right; this bound only has to be conservative).
static double wall_sites_reach(double halfLz, double cutoff2, int M2)
{
    if (M2 < 0) return 1e300;                                 // no walls: nothing to reach
    const double rc = std::sqrt(cutoff2);
    return halfLz - rc - 1e-6 * (halfLz + rc);
}

// LDS (dynamic, at launch; the body addresses it by fixed offsets): unsigned zw[S/2][64] (int16 z, slot pairs x
// lanes) at 0, double p0[65][3] (fp64 positions of the slot-0 particles + lane 0's slot-1 particle) behind it
constexpr unsigned ma_lds_bytes(int S) { return (unsigned)(S / 2) * 256u + 65u * 24u; }
// sweep_kernel_mb: the same 9752 bytes (16 wavefronts per CU fit in 160 KB); int gb[64][2], the (lowest, highest) z
// of each 4-slot group, overlays p0 while the compact copies are built.  The diagnostic build adds per-lane counters.
#ifdef SMCX_CHECK
constexpr unsigned mb_lds_bytes(int S) { return ma_lds_bytes(S) + 8u + 512u; }
#else
constexpr unsigned mb_lds_bytes(int S) { return ma_lds_bytes(S); }
#endif

#define SMCX_MA_SGPRS "s0", "s1", "s2", "s3", "s4", "s5", "s6", "s7", "s8", "s9", "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17", "s18", "s19", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95"
#define SMCX_MA_V63 "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
#define SMCX_MA_V79 SMCX_MA_V63, "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79"
#define SMCX_MA_V95 SMCX_MA_V79, "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95"
#define SMCX_MA_V127 SMCX_MA_V95, "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// kernarg pointer and replica of this workgroup (see MaArgs2); all uniform: scalar registers.  (kp is left as the compiler has
// it: rebuilding it from two v_readfirstlane halves SIGN-EXTENDED the low word -- the builtin returns int -- and faulted
// whenever the kernarg buffer's address had bit 31 set)
#define SMCX_UNIT(p)                                                                                          \
    unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();                       \
    unsigned rep = blockIdx.x;                                                                                \
    if (p.use) {                                                                                              \
        const unsigned u = p.u0 + blockIdx.x;                                                                 \
        rep = u % p.nmod;                                                                                     \
        if (u / p.nmod != p.blk0) kp += (unsigned long long)offsetof(MaArgs2, b);                             \
    }                                                                                                         \
    rep = (unsigned)__builtin_amdgcn_readfirstlane((int)rep);

// one kernel per particles-per-lane count; launch bounds = the waves per SIMD 64 + S VGPRs allow
__global__ void __launch_bounds__(64, 4) sweep_kernel_ma64(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#include "smcx_sweep_ma_body64.inc"
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", SMCX_MA_SGPRS, SMCX_MA_V127);
}

__global__ void __launch_bounds__(64, 5) sweep_kernel_ma32(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#include "smcx_sweep_ma_body32.inc"
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", SMCX_MA_SGPRS, SMCX_MA_V95);
}

__global__ void __launch_bounds__(64, 6) sweep_kernel_ma16(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#include "smcx_sweep_ma_body16.inc"
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", SMCX_MA_SGPRS, SMCX_MA_V79);
}

// z-binned form (gen_sweep_ma.py ... zb): the cells hold the particles in z order (zsort_kernel below), a probe
// screens only the 4-slot groups whose z range can reach it
__global__ void __launch_bounds__(64, 4) sweep_kernel_mb64(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mbc_body64.inc" // + the full screen beside every ranged pass, counting what the latter lacks
#else
#include "smcx_sweep_mb_body64.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V127);
}

// byte form (gen_sweep_ma.py ... z8): one word per cell, three instructions per slot and probe, no z words in LDS
__global__ void __launch_bounds__(64, 4) sweep_kernel_mc64(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcc_body64.inc" // + the fp64 cutoff test of every cell beside every pass, counting unflagged pairs
#else
#include "smcx_sweep_mc_body64.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V127);
}
// 32 and 16 particles per lane (1024 < N <= 2048, 512 < N <= 1024): 8 and 4 groups
__global__ void __launch_bounds__(64, 5) sweep_kernel_mc32(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcc_body32.inc"
#else
#include "smcx_sweep_mc_body32.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V95);
}
__global__ void __launch_bounds__(64, 6) sweep_kernel_mc16(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcc_body16.inc"
#else
#include "smcx_sweep_mc_body16.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V79);
}
// sweep_kernel_mc16 with the fp64 positions of all 1024 cells in LDS (gen_sweep_ma.py ... z8l): the few-replica form of
// N <= 1024 -- with one wavefront per SIMD nothing hides a candidate fetch's round trip to L2, so the candidates come from LDS
__global__ void __launch_bounds__(64, 2) sweep_kernel_ml16(MaArgs2 p)
{
    unsigned lane = threadIdx.x;
    SMCX_UNIT(p)
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mlc_body16.inc"
#else
#include "smcx_sweep_ml_body16.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V79);
}
// four wavefronts per replica (8192 < N <= 16384; gen_sweep_ma.py ... z8w): wave w owns the cells 4096 w .. of the
// z order; LDS = four copies of the row cache (2048 B apart) + the exchange area of the reductions (2 x 2048 B)
__global__ void __launch_bounds__(256, 1) sweep_kernel_mc64x4(MaArgs2 p)
{
    unsigned lane = threadIdx.x & 63;
    SMCX_UNIT(p)
    unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcwc_body64.inc" // + the fp64 test of every cell of this wave beside every pass
#else
#include "smcx_sweep_mcw_body64.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep), "+s"(wv)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V127);
}
// eight wavefronts per replica with 32 cells per lane each (the same generator output with NS = 32)
__global__ void __launch_bounds__(512, 1) sweep_kernel_mc32x8(MaArgs2 p)
{
    unsigned lane = threadIdx.x & 63;
    SMCX_UNIT(p)
    unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcwc_body32.inc"
#else
#include "smcx_sweep_mcw_body32.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep), "+s"(wv)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V95);
}
// four wavefronts per replica with 32 cells per lane each: 4096 < N <= 8192 (NS = 32 with WPR = 4)
__global__ void __launch_bounds__(256, 1) sweep_kernel_mc32x4(MaArgs2 p)
{
    unsigned lane = threadIdx.x & 63;
    SMCX_UNIT(p)
    unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mcw4c_body32.inc"
#else
#include "smcx_sweep_mcw4_body32.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep), "+s"(wv)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V95);
}
// TWO TEAMS of wavefronts per replica (gen_sweep_ma.py ... z8t): team A (waves 0..K-1) evaluates probe A while team B
// (waves K..2K-1) evaluates probe B; both hold all cells, wave w owns slab w mod K of the z order; one exchange per move.
// Built: 64 cells per lane x 8 wavefronts (8192 < N <= 16384 with at most 256 replicas per GPU: BASELINE config 5).  Round 3
// also built 16 x 2 (N <= 1024; slower than sweep_kernel_ml16 since round 4) and 32 x 16 (never the fastest): retired; the
// generator still writes them (gen_sweep_ma.py ... 16 z8t 2 / 32 z8t 16).
__global__ void __launch_bounds__(512, 1) sweep_kernel_mt64x8(MaArgs2 p)
{
    unsigned lane = threadIdx.x & 63;
    SMCX_UNIT(p)
    unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    asm volatile(
#ifdef SMCX_CHECK
#include "smcx_sweep_mtc_body64.inc"
#else
#include "smcx_sweep_mt_body64.inc"
#endif
        : "+v"(lane), "+s"(kp), "+s"(rep), "+s"(wv)
        :
        : "memory", "vcc", "scc", "m0", SMCX_MA_SGPRS, SMCX_MA_V127, "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148");
}
// LDS of the two-team kernels: a row cache per wave, the exchange area [2][waves][64] doubles, the side area
#ifdef SMCX_CHECK
constexpr unsigned mt_lds_bytes(int wpr) { return (unsigned)wpr * 2048u + 2u * (unsigned)wpr * 512u + 128u + (unsigned)wpr * 32u; }
#elif defined(SMCX_TT_STAMPS) // measurement variant (SMCX_GEN_TT_STAMPS=1 make VARIANT=stamps EXTRA=-DSMCX_TT_STAMPS): + 16 phase words per wave
constexpr unsigned mt_lds_bytes(int wpr) { return (unsigned)wpr * 2048u + 2u * (unsigned)wpr * 512u + 128u + (unsigned)wpr * 64u; }
#else
constexpr unsigned mt_lds_bytes(int wpr) { return (unsigned)wpr * 2048u + 2u * (unsigned)wpr * 512u + 128u; }
#endif
#ifdef SMCX_CHECK
constexpr unsigned mcw_lds_bytes(int wpr) { return (unsigned)wpr * 2048u + 2u * (unsigned)wpr * 512u + 128u + (unsigned)wpr * 32u; }
#else
constexpr unsigned mcw_lds_bytes(int wpr) { return (unsigned)wpr * 2048u + 2u * (unsigned)wpr * 512u + 128u; } // + the side pair's results (2 buffers)
#endif
// sweep_kernel_mc16/32/64: the row cache (65 x 24 B; the diagnostic build's counters behind it), then at 2048 the hand-over
// list of the merged pass (64 words) and at 2304 the side pair's results (64 B): gen_sweep_ma.py LDS_LIST, LDS_SIDEM
constexpr unsigned mc_lds_bytes() { return 2304u + 64u; }
// sweep_kernel_ml16: the positions of the 1024 cells (24 KB) and the wall table's room (1 KB) in front of that
constexpr unsigned ml_lds_bytes() { return 16u * 64u * 24u + 1024u + mc_lds_bytes(); }

// Order of the cells for sweep_kernel_mb.  Cell = slot * 64 + lane; a group = 4 slots = 256 cells.
//  1. the particles of a replica sorted by z (bitonic sort of (float z, particle) keys in LDS): group g holds
//     ranks 256 g .. 256 g + 255, so its z range is as narrow as the configuration allows;
//  2. inside every full group, sorted along a Morton curve in (x, y) and dealt to the lanes round-robin, the deal
//     of group g starting at lane 25 g mod 64: the candidates of one probe are close in (x, y) and spread over
//     three or four consecutive groups, so they land in different lanes and the kernel needs one evaluation
//     round per probe instead of one per candidate of the fullest lane.
// Written as Rs[cell] = position and loc[particle] = cell; a partial last group keeps z order, so the cells beyond N
// are the empty ones.  One workgroup per replica.  The sweep kernel keeps R and Rs both current: this reads R only.
__device__ inline unsigned spread8(unsigned v)
{
    v &= 0xffu;
    v = (v | (v << 4)) & 0x0f0fu;
    v = (v | (v << 2)) & 0x3333u;
    v = (v | (v << 1)) & 0x5555u;
    return v;
}

// bitonic network over CELLS 32-bit keys, TPB threads x CELLS/TPB keys: the compare-exchanges whose partner is
// inside a thread's own consecutive keys run in registers, the others through LDS
template <int CELLS, int TPB, bool ASCENDING_AT>
__device__ inline void bitonic_lds(unsigned *key, int kmax)
{
    constexpr int PT = CELLS / TPB; // keys per thread (consecutive)
    const int base = threadIdx.x * PT;
    for (int k = 2; k <= kmax; k <<= 1) {
        int j = k >> 1;
        for (; j >= PT; j >>= 1) { // partner in another thread's keys
            for (int t = threadIdx.x; t < CELLS / 2; t += TPB) {
                const int i = 2 * t - (t & (j - 1));
                const int q = i + j;
                const unsigned a = key[i], b = key[q];
                const bool up = ((i & k) == 0) || (ASCENDING_AT && k == kmax);
                if ((a > b) == up) { key[i] = b; key[q] = a; }
            }
            __syncthreads();
        }
        // j < PT: all partners inside this thread's PT keys
        unsigned r[PT];
#pragma unroll
        for (int e = 0; e < PT; e++) r[e] = key[base + e];
#pragma unroll
        for (int jj = PT / 2; jj > 0; jj >>= 1) { // compile-time strides: r[] stays in registers
            if (jj < k) {
#pragma unroll
                for (int e = 0; e < PT; e++) {
                    if ((e & jj) == 0) {
                        const int i = base + e;
                        const bool up = ((i & k) == 0) || (ASCENDING_AT && k == kmax);
                        const unsigned a = r[e], b = r[e | jj];
                        const bool sw = (a > b) == up;
                        r[e] = sw ? b : a; r[e | jj] = sw ? a : b;
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < PT; e++) key[base + e] = r[e];
        __syncthreads();
    }
}

// first keys: z in units of L/256 (what sweep_kernel_mc64 keeps the groups' z ranges in), biased to unsigned, above
// the 12-bit particle index; second keys: group, Morton code of (x, y), particle
#ifndef ZSORT_DEAL
#define ZSORT_DEAL 25 // lane where the deal of group g starts = 25 g mod 64 (64 / golden ratio^2: the starts of any few consecutive
                      // groups stay far apart); measured 9..39: 25..29 best, 16 (g mod 4) +1.7 %, 13 +2.5 %, 9 +4 %
#endif
template <int CELLS, int TPB>
__global__ void __launch_bounds__(TPB) zsort_kernel(const double *__restrict__ R, double *__restrict__ Rs,
                                                    unsigned short *__restrict__ loc, int N, double toFix, int K, unsigned u0,
                                                    unsigned nmod)
{
    const unsigned rpl = nmod ? (u0 + blockIdx.x) % nmod : blockIdx.x; // the replica of this workgroup (a window of units: MaArgs2)
    // bits of the particle index in the keys; second keys: 32 - NB - (bits of the group) are left for the Morton code
    constexpr int NB = CELLS <= 4096 ? 12 : 14, GB = CELLS <= 4096 ? 4 : 6, MB = 32 - NB - GB;
    static_assert(CELLS <= 16384 && MB >= 12, "key layout");
    __shared__ unsigned key[CELLS];
    const double *Rr = R + (size_t)rpl * 3 * N;
    const double zFix = toFix * (1.0 / 256.0); // 256 / L
    for (int n = threadIdx.x; n < CELLS; n += TPB) {
        unsigned k = ~0u;
        if (n < N) {
            int zq = (int)rint(Rr[3 * n + 2] * zFix);
            zq = zq < -32767 ? -32767 : zq > 32766 ? 32766 : zq;   // (32766: a particle's key is never the all-ones key of an empty cell)
            k = ((unsigned)(zq + 32768) << NB) | (unsigned)n;
        }
        key[n] = k;
    }
    __syncthreads();
#ifndef SMCX_ZSORT_NOSORT
    bitonic_lds<CELLS, TPB, false>(key, CELLS);
#endif
    // After the sort the N particles occupy the ranks p < N (an empty cell's key ~0u is larger than any particle's).  From here on
    // a rank is a particle IF p < N -- never "its key is not ~0u": the second key of the LAST particle (n = CELLS - 1) in the
    // last group with the highest Morton code IS all ones.  Rounds 2-4 tested the key: at N = 4096 and N = 16384 that particle,
    // whenever it sat in the top z group and in the (+L/2, +L/2) corner cell of the x,y grid, was taken for an empty cell -- its
    // Rs entry zeroed (a phantom particle at the origin for every probe nearby, the real one invisible) and its `loc` left
    // stale until the next sort.  Found in round 5 by comparing the energy carried along 500 sweeps with the recomputed one
    // (one replica in a thousand off by 0.2); the fcc start puts the last particle next to that corner
    // (tests/test_gpu_configs.py::test_last_particle_in_the_top_corner_cell, profiles/r05_zsort_sentinel_bug.txt).
    const int full = N >> 8; // groups with 256 particles
    for (int p = threadIdx.x; p < CELLS; p += TPB) {
        const unsigned k = key[p];
        if (p >= N) continue;
        const unsigned n = k & ((1u << NB) - 1u);
        unsigned sub = (unsigned)(p & 255) << (MB - 8); // a partial group keeps z order
#ifndef SMCX_ZSORT_NOMORTON // (measurement builds: make VARIANT=x EXTRA=-DSMCX_ZSORT_NOMORTON, likewise _NOGATHER, _NOSORT)
        if ((p >> 8) < full) {
            const unsigned ix = ((unsigned)(int)rint(Rr[3 * n] * toFix) + 0x8000u) >> 8;     // 8 bits of the wrapped x
            const unsigned iy = ((unsigned)(int)rint(Rr[3 * n + 1] * toFix) + 0x8000u) >> 8;
            sub = (spread8(ix) | (spread8(iy) << 1)) >> (16 - MB);                            // the top MB bits of the code
        }
#endif
        key[p] = ((unsigned)(p >> 8) << (32 - GB)) | (sub << NB) | n;
    }
    __syncthreads();
#ifndef SMCX_ZSORT_NOSORT
    bitonic_lds<CELLS, TPB, true>(key, 256);
#endif
    for (int p = threadIdx.x; p < CELLS; p += TPB) {
        const unsigned k = key[p];
        const int g = p >> 8, r = p & 255;
        int c = g < full ? (4 * g + (r >> 6)) * 64 + ((r + ZSORT_DEAL * g) & 63) : p;
        if (K > 1) // K wavefronts share the cells: group g is local group g / K of wave g % K (its cells start at
            c = (g % K) * (CELLS / K) + (g / K) * 256 + (c & 255); // (g % K) CELLS / K) -- a probe's 3-4 groups in reach
                                                                    // then lie on different wavefronts
        double *d = Rs + ((size_t)rpl * CELLS + c) * 3;
        if (p < N) { // (a partial last group keeps its particles in front: their keys are smaller than ~0u, see above)
            const unsigned n = k & ((1u << NB) - 1u);
            loc[(size_t)rpl * N + n] = (unsigned short)c;
#ifdef SMCX_ZSORT_NOGATHER
            d[0] = d[1] = d[2] = 1.0;
#else
            d[0] = Rr[3 * n]; d[1] = Rr[3 * n + 1]; d[2] = Rr[3 * n + 2];
#endif
        } else {
            d[0] = d[1] = d[2] = 0.0;
        }
    }
}

// hand-scheduled kernels are generated for 16, 32 and 64 cells per lane, one wavefront per replica
bool ma_built(int S, int N, int M2)
{
    if (S != 16 && S != 32 && S != 64) return false;
    return N > 32 * S && N <= 64 * S && M2 + 1 <= 30;
}

// highest form the build offers for S.  The product build: all of them (mb needs 64 cells per lane, plan_kernel).
// The diagnostic build (-DSMCX_CHECK) counts the screen's misses in sweep_kernel_mi (same screen, hipcc-scheduled)
// unless check_mb asks for sweep_kernel_mb64 (1: ranged passes checked against full ones; 64 cells per lane) or the
// sweep_kernel_mc* kernels (2: the fp64 test of every cell beside every pass, and the executed-work counters).
int ma_cap(const Tune &t, int S)
{
#ifdef SMCX_CHECK
    if (t.check_mb == 2) return FORM_MC;
    return (t.check_mb == 1 && S == 64) ? FORM_MB : FORM_MI;
#else
    (void)t; (void)S;
    return FORM_MC;
#endif
}

// several wavefronts per replica with the cells in z order: 64 x 4 or 32 x 8 for 8192 < N <= 16384, 32 x 4 for
// 4096 < N <= 8192, in a box the byte screen serves
bool mcw_built(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2)
{
    const bool big = ((S == 64 && WPR == 4) || (S == 32 && WPR == 8)) && N > 8192 && N <= 16384;
    const bool mid = S == 32 && WPR == 4 && N > 4096 && N <= 8192;
    return (big || mid) && M2 + 1 <= 30 && mc_box_supported(L, Lz, cutoff2);
}

// two teams of wavefronts per replica: 16 cells per lane x 2 waves (512 < N <= 1024) or 64 x 8 (8192 < N <= 16384)
bool mt_built(int S, int WPR, int N, int M2, double L, double Lz, double cutoff2)
{
    const bool big = S == 64 && WPR == 8 && N > 8192 && N <= 16384;
    return big && M2 + 2 <= 30 && mc_box_supported(L, Lz, cutoff2);
}

// The launches of one launch group of `nsweeps` sweeps, cut into blocks of `every` sweeps (one z sort + one kernel launch
// each): plain -- one launch per block over all replicas -- or, when the replica count is not a multiple of what the device
// holds at once (c.windows, MaArgs2), windows of c.granule consecutive units.
struct UnitLaunch {
    unsigned grid, u0, nmod, blk0, use;
    int sw0[2], nsw[2];
};
static std::vector<UnitLaunch> unit_launches(const DevCtx &c, int nsweeps, int every)
{
    std::vector<UnitLaunch> v;
    const int nb = (nsweeps + every - 1) / every;
    auto block = [&](UnitLaunch &l, int k, int blk) { l.sw0[k] = blk * every; l.nsw[k] = nsweeps - blk * every < every ? nsweeps - blk * every : every; };
    if (!c.windows) {
        for (int b = 0; b < nb; b++) {
            UnitLaunch l{(unsigned)c.nrep, 0u, 0u, (unsigned)b, 0u, {0, 0}, {0, 0}};
            block(l, 0, b); block(l, 1, b);
            v.push_back(l);
        }
        return v;
    }
    const long total = (long)c.nrep * nb, G = c.granule;
    for (long u0 = 0; u0 < total; u0 += G) {
        const long cnt = total - u0 < G ? total - u0 : G;
        const int b0 = (int)(u0 / c.nrep), b1 = (int)((u0 + cnt - 1) / c.nrep);
        UnitLaunch l{(unsigned)cnt, (unsigned)u0, (unsigned)c.nrep, (unsigned)b0, 1u, {0, 0}, {0, 0}};
        block(l, 0, b0); block(l, 1, b1);
        v.push_back(l);
    }
    return v;
}
static MaArgs2 unit_args(const MaArgs &a, const UnitLaunch &l)
{
    MaArgs2 p;
    p.a = a; p.b = a;
    p.a.sw0 = l.sw0[0]; p.a.nsweeps = l.nsw[0];
    p.b.sw0 = l.sw0[1]; p.b.nsweeps = l.nsw[1];
    p.u0 = l.u0; p.nmod = l.nmod; p.blk0 = l.blk0; p.use = l.use;
    return p;
}

hipError_t launch_sweeps_mt(const SweepArgs &s, const DevCtx &c, const KernelPlan &pl, int nsweeps, double A, hipStream_t st,
                            SweepTimer *tm)
{
    if (!c.Rs || !c.loc) return hipErrorInvalidValue;
    MaArgs a;
    a.R = s.R; a.displ = s.displ; a.uni = s.uni; a.offs = s.offs; a.obs = s.obs; a.rec = s.rec;
    a.wtab = c.wtab; a.clk = s.clk;
    a.L = s.L; a.invL = s.invL; a.cutoff2 = s.cutoff2; a.invT = s.invT;
    a.AoT = A * s.invT; a.Ao4T = A * 0.25 * s.invT;
    a.halfLz = c.halfLz; a.Lz = c.Lz; a.invLz = c.invLz;
    a.N = s.N; a.chunk = s.chunk; a.nsweeps = 1;
    a.M2 = (c.flags & 0x1u) ? c.M2 : -1;
    mc_bound_values(c.L, c.cutoff2, &a.toFix, &a.zsafe, &a.negC, &a.RZ);
    a.zFix = a.toFix;
    a.Rs = c.Rs; a.loc = c.loc; a.pad0 = 0; a.dbg = nullptr; a.prio = c.prio;
    a.znear = wall_sites_reach(c.halfLz, c.cutoff2, a.M2);
#ifdef SMCX_CHECK
    a.dbg = s.dbg;
#endif
    const double toFix16 = 65536.0 / c.L;
    for (const UnitLaunch &l : unit_launches(c, nsweeps, 1)) {
        hipLaunchKernelGGL((zsort_kernel<4 * 64 * 64, 1024>), dim3(l.grid), dim3(1024), 0, st, (const double *)s.R, c.Rs, c.loc, s.N, toFix16,
                           pl.WPR / 2, l.u0, l.nmod);
        hipError_t rc = tm ? tm->mark(st) : hipSuccess;
        if (rc != hipSuccess) return rc;
        hipLaunchKernelGGL(sweep_kernel_mt64x8, dim3(l.grid), dim3(512), mt_lds_bytes(8), st, unit_args(a, l));
        rc = hipGetLastError();
        if (rc == hipSuccess && tm) rc = tm->mark(st);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

// the multi-wave form has its own launcher: no int16-screen numbers are needed (the byte screen's come from mc_bound)
hipError_t launch_sweeps_mcw(const SweepArgs &s, const DevCtx &c, int WPR, int nsweeps, double A, hipStream_t st,
                             SweepTimer *tm)
{
    MaArgs a;
    a.R = s.R; a.displ = s.displ; a.uni = s.uni; a.offs = s.offs; a.obs = s.obs; a.rec = s.rec;
    a.wtab = c.wtab; a.clk = s.clk;
    a.L = s.L; a.invL = s.invL; a.cutoff2 = s.cutoff2; a.invT = s.invT;
    a.AoT = A * s.invT; a.Ao4T = A * 0.25 * s.invT;
    a.halfLz = c.halfLz; a.Lz = c.Lz; a.invLz = c.invLz;
    a.N = s.N; a.chunk = s.chunk; a.nsweeps = 1;
    a.M2 = (c.flags & 0x1u) ? c.M2 : -1;
    mc_bound_values(c.L, c.cutoff2, &a.toFix, &a.zsafe, &a.negC, &a.RZ);
    a.zFix = a.toFix;
    a.Rs = c.Rs; a.loc = c.loc; a.pad0 = 0; a.dbg = nullptr; a.prio = c.prio;
    a.znear = wall_sites_reach(c.halfLz, c.cutoff2, a.M2);
#ifdef SMCX_CHECK
    a.dbg = s.dbg;
#endif
    const double toFix16 = 65536.0 / c.L; // the Morton code of the z sort takes x, y in units of L/65536
    const bool mid = s.N <= 8192; // 32 cells per lane x 4 wavefronts: 8192 cells
    for (const UnitLaunch &l : unit_launches(c, nsweeps, 1)) {
        if (mid)
            hipLaunchKernelGGL((zsort_kernel<2 * 64 * 64, 1024>), dim3(l.grid), dim3(1024), 0, st, (const double *)s.R, c.Rs, c.loc,
                               s.N, toFix16, WPR, l.u0, l.nmod);
        else
            hipLaunchKernelGGL((zsort_kernel<4 * 64 * 64, 1024>), dim3(l.grid), dim3(1024), 0, st, (const double *)s.R, c.Rs, c.loc,
                               s.N, toFix16, WPR, l.u0, l.nmod);
        const MaArgs2 p2 = unit_args(a, l);
        hipError_t rc = tm ? tm->mark(st) : hipSuccess;
        if (rc != hipSuccess) return rc;
        if (mid)
            hipLaunchKernelGGL(sweep_kernel_mc32x4, dim3(l.grid), dim3(256), mcw_lds_bytes(4), st, p2);
        else if (WPR == 4)
            hipLaunchKernelGGL(sweep_kernel_mc64x4, dim3(l.grid), dim3(256), mcw_lds_bytes(4), st, p2);
        else
            hipLaunchKernelGGL(sweep_kernel_mc32x8, dim3(l.grid), dim3(512), mcw_lds_bytes(8), st, p2);
        rc = hipGetLastError();
        if (rc == hipSuccess && tm) rc = tm->mark(st);
        if (rc != hipSuccess) return rc;
    }
    return hipSuccess;
}

void mc_bound(double L, double cutoff2, double *toFix, double *zsafe, int *negC, int *RZ);

bool mc_box_supported(double L, double Lz, double cutoff2)
{
    const double u = L / 256.0, rc = std::sqrt(cutoff2);
    return rc >= 16.0 * u && 0.5 * Lz < 32000.0 * u && L >= 2.0 * rc;
}

void mc_bound_values(double L, double cutoff2, double *toFix, double *zsafe, int *negT, int *RZ) { mc_bound(L, cutoff2, toFix, zsafe, negT, RZ); }

// conservative threshold of the byte screen.  With d = packed difference of the two words (borrows leak one unit
// from z into x and from x into y) and q = rc/u: for a pair inside the cutoff |dx|,|dy| <= |true|/u + 2 (two
// roundings + borrow), |dz| <= |true|/u + 1, the high z byte contributes <= 1, so the sum of the four squared
// bytes is < (q + 3)^2 + 1 (Minkowski with (2,2,1)); all bytes stay below 128 since q + 3 < 120.
void mc_bound(double L, double cutoff2, double *toFix, double *zsafe, int *negC, int *RZ)
{
    const double u = L / 256.0, q = std::sqrt(cutoff2) / u;
    *toFix = 256.0 / L;
    *zsafe = 32766.0 * u;
    *negC = -((int)std::floor((q + 3.0) * (q + 3.0)) + 3);
    *RZ = (int)std::floor(q) + 3; // |dz| in units of a pair inside the cutoff, with the two roundings
}

const char *ma_kernel_name(int form, int S, int WPR)
{
    if (form == FORM_MT) return "smcx::sweep_kernel_mt64x8";
    if (form == FORM_MC && WPR == 4 && S == 32) return "smcx::sweep_kernel_mc32x4";
    if (form == FORM_MC && WPR == 4) return "smcx::sweep_kernel_mc64x4";
    if (form == FORM_MC && WPR == 8) return "smcx::sweep_kernel_mc32x8";
    if (form == FORM_MC && WPR == -1) return "smcx::sweep_kernel_ml16";   // (asked for with WPR = -1: positions in LDS)
    if (form == FORM_MC) return S == 64 ? "smcx::sweep_kernel_mc64" : S == 32 ? "smcx::sweep_kernel_mc32" : "smcx::sweep_kernel_mc16";
    if (form == FORM_MB) return "smcx::sweep_kernel_mb64";
    return S == 64 ? "smcx::sweep_kernel_ma64" : S == 32 ? "smcx::sweep_kernel_ma32" : "smcx::sweep_kernel_ma16";
}

// wtab: [M2 + 1][4] doubles on the device, built by the caller (smcx_api.hip)
hipError_t launch_sweeps_ma(const SweepArgs &s, const DevCtx &c, const KernelPlan &pl, const double *wtab, int nsweeps,
                            double A, double toFix, double zFix, double zsafe, int negC, hipStream_t st, SweepTimer *tm)
{
    const int S = pl.S;
    MaArgs a;
    a.R = s.R; a.displ = s.displ; a.uni = s.uni; a.offs = s.offs; a.obs = s.obs; a.rec = s.rec;
    a.wtab = wtab; a.clk = s.clk;
    a.L = s.L; a.invL = s.invL; a.cutoff2 = s.cutoff2; a.invT = s.invT;
    a.AoT = A * s.invT; a.Ao4T = A * 0.25 * s.invT; a.toFix = toFix; a.zFix = zFix;
    a.zsafe = zsafe; a.halfLz = c.halfLz; a.Lz = c.Lz; a.invLz = c.invLz;
    a.N = s.N; a.chunk = s.chunk; a.nsweeps = nsweeps; a.negC = negC;
    a.M2 = (c.flags & 0x1u) ? c.M2 : -1;
    a.RZ = 0; a.Rs = nullptr; a.loc = nullptr; a.sw0 = 0; a.pad0 = 0; a.dbg = nullptr; a.prio = c.prio;
    a.znear = wall_sites_reach(c.halfLz, c.cutoff2, a.M2);
#ifdef SMCX_CHECK
    a.dbg = s.dbg;
#endif
    if (pl.zordered()) {
        if (!c.Rs || !c.loc) return hipErrorInvalidValue;
        // a slot is flagged only if dz^2 < ceil(C / 4^ZS) (dz in z units, C = -negC): the screen's reach in z
        const long T = ((long)(-negC) + 255) >> 8;
        a.RZ = (int)std::floor(std::sqrt((double)T)) + 1;
        a.Rs = c.Rs; a.loc = c.loc;
        // the cells are re-sorted by z every `every` sweeps (smcx_params.tune_resort, default 2 since round 4): the groups' z ranges
        // only widen inside a launch
        const bool mc = pl.form == FORM_MC;
        if (mc) {
            mc_bound(c.L, c.cutoff2, &a.toFix, &a.zsafe, &a.negC, &a.RZ);
            a.zFix = a.toFix;
        }
        const int every = pl.tune.resort > 0 ? pl.tune.resort : 1;
        for (const UnitLaunch &l : unit_launches(c, nsweeps, every)) {
            const int tpb = pl.tune.zsort_tpb; // 512 threads (8 keys each) measured best of 128..1024 for 4096 cells
#define SMCX_ZSORT(C, T) hipLaunchKernelGGL((zsort_kernel<C, T>), dim3(l.grid), dim3(T), 0, st, (const double *)s.R, c.Rs, c.loc, s.N, toFix, 1, l.u0, l.nmod)
            if (S == 64 && tpb == 128) SMCX_ZSORT(64 * 64, 128);
            else if (S == 64 && tpb == 256) SMCX_ZSORT(64 * 64, 256);
            else if (S == 64 && tpb == 1024) SMCX_ZSORT(64 * 64, 1024);
            else if (S == 64) SMCX_ZSORT(64 * 64, 512);
            else if (S == 32) SMCX_ZSORT(32 * 64, 256);
            else SMCX_ZSORT(16 * 64, 256);
#undef SMCX_ZSORT
            const MaArgs2 p2 = unit_args(a, l);
            hipError_t rc = tm ? tm->mark(st) : hipSuccess;
            if (rc != hipSuccess) return rc;
            if (mc && S == 64)
                hipLaunchKernelGGL(sweep_kernel_mc64, dim3(l.grid), dim3(64), mc_lds_bytes(), st, p2);
            else if (mc && S == 32)
                hipLaunchKernelGGL(sweep_kernel_mc32, dim3(l.grid), dim3(64), mc_lds_bytes(), st, p2);
            else if (mc)
                if (pl.lpos)
                    hipLaunchKernelGGL(sweep_kernel_ml16, dim3(l.grid), dim3(64), ml_lds_bytes(), st, p2);
                else
                    hipLaunchKernelGGL(sweep_kernel_mc16, dim3(l.grid), dim3(64), mc_lds_bytes(), st, p2);
            else
                hipLaunchKernelGGL(sweep_kernel_mb64, dim3(l.grid), dim3(64), mb_lds_bytes(64), st, p2);
            rc = hipGetLastError();
            if (rc == hipSuccess && tm) rc = tm->mark(st);
            if (rc != hipSuccess) return rc;
        }
        return hipSuccess;
    }
    void (*f)(MaArgs2) = S == 64 ? sweep_kernel_ma64 : S == 32 ? sweep_kernel_ma32 : sweep_kernel_ma16;
    hipError_t rc = tm ? tm->mark(st) : hipSuccess;
    if (rc != hipSuccess) return rc;
    MaArgs2 p2;
    p2.a = a; p2.b = a; p2.u0 = p2.nmod = p2.blk0 = p2.use = 0;
    hipLaunchKernelGGL(f, dim3(c.nrep), dim3(64), ma_lds_bytes(S), st, p2);
    rc = hipGetLastError();
    if (rc == hipSuccess && tm) rc = tm->mark(st);
    return rc;
}

// how many replicas the device runs AT ONCE with this plan's kernel (workgroups resident per CU x CUs; 0 = not one of the
// z-ordered hand-scheduled kernels).  A sweep is sequential inside a replica and one launch runs one sweep of all replicas, so
// the time of a sweep is a step function of nrep / this number: smcx_replica_granule reports it.
int ma_resident_replicas(const KernelPlan &pl, int device)
{
    const void *f = nullptr;
    int threads = 64;
    unsigned lds = 0;
    if (pl.form == FORM_MC && pl.WPR == 1) {
        f = pl.S == 64 ? (const void *)sweep_kernel_mc64 : pl.S == 32 ? (const void *)sweep_kernel_mc32 : (const void *)sweep_kernel_mc16;
        lds = mc_lds_bytes();
        if (pl.lpos) { f = (const void *)sweep_kernel_ml16; lds = ml_lds_bytes(); }
    } else if (pl.form == FORM_MB) {
        f = (const void *)sweep_kernel_mb64; lds = mb_lds_bytes(64);
    } else if (pl.form == FORM_MC) {
        threads = 64 * pl.WPR; lds = mcw_lds_bytes(pl.WPR);
        f = (pl.WPR == 4 && pl.S == 32) ? (const void *)sweep_kernel_mc32x4 : pl.WPR == 4 ? (const void *)sweep_kernel_mc64x4
                                                                                         : (const void *)sweep_kernel_mc32x8;
    } else if (pl.form == FORM_MT) {
        threads = 64 * pl.WPR;
        lds = mt_lds_bytes(pl.WPR);
        f = (const void *)sweep_kernel_mt64x8;
    }
    if (!f) return 0;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, f, threads, lds) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu * cus;
}

} // namespace smcx
