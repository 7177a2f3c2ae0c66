// smcx_sweep_common.hpp -- pieces shared by the sweep kernels of smcx_kernels.hip (fp64) and
// smcx_sweep_mx.hip (fp32-screened): slot rotation, coherent loads, the LDS exchange block
// and the 8-value combine.  Included inside namespace smcx.
#pragma once

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
};

// combine the eight wave totals (and, with several waves, the waves) into
// tot[8], identical in every lane of the workgroup; also fetches the two side terms
template <int WPR>
__device__ __forceinline__ void combine(SweepShared<WPR> &sh, int &par, int lane, int wave,
                                        const Acc8 &v, const double (&side)[4],
                                        double (&tot)[8], double (&sOld)[4], double (&sNew)[4])
{
    const double r = reduce8(v.a0, v.a1, v.a2, v.a3, v.b0, v.b1, v.b2, v.b3, lane);
    if constexpr (WPR == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) tot[j] = rdlane(r, 8 * j);
#pragma unroll
        for (int j = 0; j < 4; j++) { sOld[j] = side[j]; sNew[j] = side[j]; } // read by lane later
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
        if constexpr (WPR >= 8) t = sum_x32(t);
        if constexpr (WPR >= 4) t = sum_x16(t);
        t = sum_x8(t);
#pragma unroll
        for (int j = 0; j < 8; j++) tot[j] = rdlane(t, j);
#pragma unroll
        for (int j = 0; j < 4; j++) { sOld[j] = sh.side[par][0][j]; sNew[j] = sh.side[par][1][j]; }
        par ^= 1;
    }
}

