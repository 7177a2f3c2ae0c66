// smcx_lca.hip -- common-neighbour cluster analysis of every replica (clusterAnalysis,
// SMC.c:971-1045, called from sMC at SMC.c:143-155).  Integer / bit work on top of one
// O(N^2) distance pass; results are bit-identical to the reference's arithmetic, including
// its overlapping pair index (see below).
//
// The reference keeps three N(N-1)/2 arrays per call (bool num1, int num2, int num3).  Here
// num1 is a bit matrix in HBM (N(N-1)/2 bits per replica, 1 MiB at N = 4096), built by one
// thread per pair, and num2/num3 are produced per set entry by one wavefront that scans the
// entry's matrix row 32 pairs per lane-word; they go straight into per-replica counters
// (and, for the single-replica entry point, into the LCA array of the reference's layout).
//
// Pair index, as written in the reference (SMC.c:987, 1007): idx(l,i) = (l*l-3*l+2)/2 + i,
// i < l.  That advances by l-1 per row while row l has l entries, so idx(l,l-1) ==
// idx(l+1,0): the two pairs share one entry -- num1 is the OR of the two, num2 and num3
// accumulate over both, (l,l-1) first, and because the two are also consecutive in the
// reference's loop its scratch list common_nn simply keeps growing across them.  The
// look-up of the (i,i2) entry at (i2*i2-3*i2+2)/2 + i is used for i > i2 too (SMC.c:1017),
// and only consecutive common neighbours are tested for a bond (SMC.c:1028-1033).  All of
// this is reproduced; stores past the reference's common_nn[8] (undefined behaviour there)
// are dropped and counted in the overflow counter.
#include "smcx_kernels.h"

namespace smcx {

namespace {

__device__ __forceinline__ int tri(int k) { return (k * k - 3 * k + 2) / 2; } // SMC.c:987

__device__ __forceinline__ unsigned bit_at(const unsigned *bits, int j)
{
    return (bits[j >> 5] >> (j & 31)) & 1u;
}

// num1: one block per row l (blockIdx.x + 1), threads stride over i < l (SMC.c:984-1000)
__global__ void __launch_bounds__(256) lca_bonds_kernel(LcaArgs a)
{
#pragma clang fp contract(off)
    const int l = blockIdx.x + 1;
    const int rep = a.rep0 + blockIdx.y;
    const double *r = a.R + (size_t)rep * 3 * a.N;
    unsigned *bits = a.bits + (size_t)blockIdx.y * a.words;
    const double xl = r[3 * l], yl = r[3 * l + 1], zl = r[3 * l + 2];
    const int row = tri(l);
    for (int i = threadIdx.x; i < l; i += 256) {
        double dx = xl - r[3 * i];
        dx = dx - a.L * __builtin_rint(dx / a.L);
        double dy = yl - r[3 * i + 1];
        dy = dy - a.L * __builtin_rint(dy / a.L);
        const double dz = zl - r[3 * i + 2];
        const double d2 = dx * dx + dy * dy + dz * dz;
        if (d2 < a.cut2) {
            const int idx = row + i;
            atomicOr(&bits[idx >> 5], 1u << (idx & 31));
        }
    }
}

// state of one pair entry while its (one or two) pairs are processed by a wavefront;
// everything here is wave-uniform
struct Entry {
    int c;       // num2[idx]
    int n3;      // num3[idx]
    int over;    // dropped stores
    int cn0, cn1, cn2, cn3, cn4, cn5, cn6, cn7; // common_nn
};

__device__ __forceinline__ void put(Entry &e, int v)
{
    switch (e.c) {
    case 0: e.cn0 = v; break;
    case 1: e.cn1 = v; break;
    case 2: e.cn2 = v; break;
    case 3: e.cn3 = v; break;
    case 4: e.cn4 = v; break;
    case 5: e.cn5 = v; break;
    case 6: e.cn6 = v; break;
    case 7: e.cn7 = v; break;
    default: e.over++; break;
    }
    e.c++;
}

// one (l,i) of the second loop, SMC.c:1008-1035
__device__ __forceinline__ void lca_pair(const unsigned *bits, int l, int i, int lane, Entry &e)
{
    const int row = tri(l);
    const int wlo = row >> 5, whi = (row + l - 1) >> 5;
    for (int wb = wlo; wb <= whi; wb += 64) {
        const int wi = wb + lane;
        unsigned word = (wi <= whi) ? bits[wi] : 0u;
        const int lo = wi << 5;
        // keep the bits of entries (l,i2), 0 <= i2 < l, i2 != i
        if (lo < row) word &= (row - lo >= 32) ? 0u : (~0u << (row - lo));
        if (lo + 32 > row + l) word &= (row + l - lo <= 0) ? 0u : (~0u >> (lo + 32 - row - l));
        const int pi = row + i - lo;
        if (pi >= 0 && pi < 32) word &= ~(1u << pi);
        // ... that are also bonded to i (SMC.c:1016-1018)
        unsigned cm = 0;
        for (unsigned t = word; t; t &= t - 1) {
            const int b = __builtin_ctz(t);
            const int i2 = lo + b - row;
            if (bit_at(bits, tri(i2) + i)) cm |= 1u << b;
        }
        // append them in ascending i2 (SMC.c:1020-1021)
        unsigned long long nz = __builtin_amdgcn_ballot_w64(cm != 0);
        while (nz) {
            const int src = __builtin_ctzll(nz);
            nz &= nz - 1;
            unsigned u = (unsigned)__builtin_amdgcn_readlane((int)cm, src);
            for (; u; u &= u - 1) put(e, ((wb + src) << 5) + __builtin_ctz(u) - row);
        }
    }
    if (e.c > 1) { // SMC.c:1026-1034, over the whole list every time
        const int n = e.c < 8 ? e.c : 8;
        if (n > 1 && bit_at(bits, tri(e.cn1) + e.cn0)) e.n3++;
        if (n > 2 && bit_at(bits, tri(e.cn2) + e.cn1)) e.n3++;
        if (n > 3 && bit_at(bits, tri(e.cn3) + e.cn2)) e.n3++;
        if (n > 4 && bit_at(bits, tri(e.cn4) + e.cn3)) e.n3++;
        if (n > 5 && bit_at(bits, tri(e.cn5) + e.cn4)) e.n3++;
        if (n > 6 && bit_at(bits, tri(e.cn6) + e.cn5)) e.n3++;
        if (n > 7 && bit_at(bits, tri(e.cn7) + e.cn6)) e.n3++;
    }
}

// num2, num3: every wavefront scans 64 words of the bit matrix at a time and handles the
// set entries one after the other
__global__ void __launch_bounds__(256) lca_types_kernel(LcaArgs a)
{
    const int lane = threadIdx.x & 63;
    const int wave = uniform((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    const int nwaves = gridDim.x * 4;
    const int rep = a.rep0 + blockIdx.y;
    const unsigned *bits = a.bits + (size_t)blockIdx.y * a.words;
    unsigned long long *cnt = a.counts + (size_t)rep * LCA_COUNTS;
    const int N = a.N;
    unsigned long long n1 = 0, over = 0;

    for (long w0 = (long)wave * 64; w0 < a.words; w0 += (long)nwaves * 64) {
        const long wi = w0 + lane;
        const unsigned word = (wi < a.words) ? bits[wi] : 0u;
        unsigned long long nz = __builtin_amdgcn_ballot_w64(word != 0);
        while (nz) {
            const int src = __builtin_ctzll(nz);
            nz &= nz - 1;
            unsigned u = (unsigned)__builtin_amdgcn_readlane((int)word, src);
            for (; u; u &= u - 1) {
                const int idx = (int)((w0 + src) << 5) + __builtin_ctz(u);
                // smallest l >= 1 whose row reaches idx: l(l-1)/2 >= idx
                int l = (int)__builtin_ceil((1.0 + __builtin_sqrt(1.0 + 8.0 * (double)idx)) * 0.5);
                if (l < 1) l = 1;
                while (l > 1 && (long)(l - 1) * (l - 2) / 2 >= idx) l--;
                while ((long)l * (l - 1) / 2 < idx) l++;
                Entry e = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                lca_pair(bits, l, idx - tri(l), lane, e);
                if ((long)l * (l - 1) / 2 == idx && l + 1 < N) lca_pair(bits, l + 1, 0, lane, e);
                n1++;
                over += e.over;
                if (lane == 0) {
                    atomicAdd(&cnt[1 + (e.c > 15 ? 15 : e.c)], 1ull);
                    atomicAdd(&cnt[17 + (e.n3 > 15 ? 15 : e.n3)], 1ull);
                    if (a.LCA) { // SMC.c:1038-1044
                        a.LCA[3 * (size_t)idx + 0] = 1;
                        a.LCA[3 * (size_t)idx + 1] = e.c;
                        a.LCA[3 * (size_t)idx + 2] = e.n3;
                    }
                }
            }
        }
    }
    if (lane == 0) {
        if (n1) atomicAdd(&cnt[0], n1);
        if (over) atomicAdd(&cnt[33], over);
    }
}

} // namespace

hipError_t launch_lca(const LcaArgs &a, int nbatch, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(a.bits, 0, (size_t)nbatch * a.words * sizeof(unsigned), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(lca_bonds_kernel, dim3(a.N - 1, nbatch), dim3(256), 0, st, a);
    long blocks = (a.words + 255) / 256; // one word per thread, at most
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(lca_types_kernel, dim3((unsigned)blocks, nbatch), dim3(256), 0, st, a);
    return hipGetLastError();
}

} // namespace smcx
