// smcx_acf.hip -- energy autocorrelation of the last run: the reference's
// fft_acf (SMC.c:1051-1089) for every replica, with hipFFT where the reference uses FFTW.
//
//   Z = H - mean(H)                                  (SMC.c:1069-1071)
//   F = r2c FFT of Z, length n                       (:1073-1074)
//   T[k] = |F[k]|^2 for k < lfft = n/2 + n%2         (:1076-1077)
//   C = backward complex FFT of T, length lfft       (:1079-1080)   <- the reference's own choice:
//   acf[i] = Re C[i] / Re C[0], i < k_max            (:1082-1083)      a half-length transform
//   tau = sum(acf)  (SMC.c:235),  cv = variance(E)/T^2  (SMC.c:250)
// H is the production energy series E[0..maxsteps] (the constant 3NT/2 of SMC.c:210-211
// cancels in Z).  Needs SMCX_FLAG_SERIES.
#include "../../include/smcx.h"
#include "smcx_kernels.h"

#include <hipfft/hipfft.h>
#include <vector>

namespace smcx {

__global__ void acf_center_kernel(const double *E, int stride, int n, double *Z)
{
    __shared__ double part[4];
    const int rep = blockIdx.x, tid = threadIdx.x;
    const double *e = E + (size_t)rep * stride;
    double s = 0.0;
    for (int i = tid; i < n; i += 256) s += e[i];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    const double mean = (part[0] + part[1] + part[2] + part[3]) / n;
    for (int i = tid; i < n; i += 256) Z[(size_t)rep * n + i] = e[i] - mean;
}

__global__ void acf_psd_kernel(const hipfftDoubleComplex *F, int fstride, int lfft, hipfftDoubleComplex *T)
{
    const int rep = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= lfft) return;
    const hipfftDoubleComplex f = F[(size_t)rep * fstride + k];
    hipfftDoubleComplex t;
    t.x = f.x * f.x + f.y * f.y;
    t.y = 0.0;
    T[(size_t)rep * lfft + k] = t;
}

__global__ void acf_norm_kernel(const hipfftDoubleComplex *C, int lfft, int kmax, double *acf, double *tau)
{
    const int rep = blockIdx.x;
    const hipfftDoubleComplex *c = C + (size_t)rep * lfft;
    const double c0 = c[0].x;
    for (int i = threadIdx.x; i < kmax; i += blockDim.x) acf[(size_t)rep * kmax + i] = c[i].x / c0;
    __syncthreads();
    if (threadIdx.x == 0) { // sum() of matematicose.c:15-23 adds in index order
        double s = 0.0;
        for (int i = 0; i < kmax; i++) s += c[i].x / c0;
        tau[rep] = s;
    }
}

} // namespace smcx

using namespace smcx;

// implemented in smcx_api.hip
extern "C" int smcx_internal_series_view(smcx_handle *h, const double **E, int *stride, int *maxsteps,
                                          int *nrep, double *T, int *device, void **stream);

extern "C" int smcx_acf(smcx_handle *h, int k_max, double *acf, int *k_eff, double *tau, double *cv)
{
    const double *E = nullptr;
    int stride = 0, maxsteps = 0, nrep = 0, device = 0;
    double T = 0.0;
    void *st = nullptr;
    int rc = smcx_internal_series_view(h, &E, &stride, &maxsteps, &nrep, &T, &device, &st);
    if (rc != SMCX_OK) return rc;
    hipStream_t stream = (hipStream_t)st;
    const int n = maxsteps + 1;                         // fft_acf(E, maxsteps+1, KMAX), SMC.c:234
    if (n < 8 || k_max < 1) return SMCX_ERR_PARAM;
    if (n < k_max * 2 + 1) k_max = n / 2 - 2;           // SMC.c:1054-1057
    if (k_eff) *k_eff = k_max;
    const int lfft = n / 2 + n % 2;                     // SMC.c:1063
    const int fstride = n / 2 + 1;                      // what a real-to-complex transform returns
    if (hipSetDevice(device) != hipSuccess) return SMCX_ERR_HIP;

    double *Z = nullptr, *d_acf = nullptr, *d_tau = nullptr;
    hipfftDoubleComplex *F = nullptr, *Tm = nullptr, *C = nullptr;
    hipfftHandle p1 = 0, p2 = 0;
    bool ok = true;
    auto hip_ok = [&](hipError_t e) { if (e != hipSuccess) ok = false; return ok; };
    auto fft_ok = [&](hipfftResult r) { if (r != HIPFFT_SUCCESS) ok = false; return ok; };
    hip_ok(hipMalloc(&Z, (size_t)nrep * n * sizeof(double)));
    hip_ok(hipMalloc(&F, (size_t)nrep * fstride * sizeof(hipfftDoubleComplex)));
    hip_ok(hipMalloc(&Tm, (size_t)nrep * lfft * sizeof(hipfftDoubleComplex)));
    hip_ok(hipMalloc(&C, (size_t)nrep * lfft * sizeof(hipfftDoubleComplex)));
    hip_ok(hipMalloc(&d_acf, (size_t)nrep * k_max * sizeof(double)));
    hip_ok(hipMalloc(&d_tau, (size_t)nrep * sizeof(double)));
    if (ok) {
        int nn[1] = {n}, ll[1] = {lfft};
        fft_ok(hipfftPlanMany(&p1, 1, nn, nullptr, 1, n, nullptr, 1, fstride, HIPFFT_D2Z, nrep));
        if (ok) fft_ok(hipfftPlanMany(&p2, 1, ll, nullptr, 1, lfft, nullptr, 1, lfft, HIPFFT_Z2Z, nrep));
        if (ok) { fft_ok(hipfftSetStream(p1, stream)); fft_ok(hipfftSetStream(p2, stream)); }
    }
    if (ok) {
        hipLaunchKernelGGL(acf_center_kernel, dim3(nrep), dim3(256), 0, stream, E, stride, n, Z);
        fft_ok(hipfftExecD2Z(p1, Z, F));
        if (ok) {
            hipLaunchKernelGGL(acf_psd_kernel, dim3((lfft + 255) / 256, nrep), dim3(256), 0, stream, F, fstride,
                               lfft, Tm);
            fft_ok(hipfftExecZ2Z(p2, Tm, C, HIPFFT_BACKWARD));
        }
        if (ok) {
            hipLaunchKernelGGL(acf_norm_kernel, dim3(nrep), dim3(256), 0, stream, C, lfft, k_max, d_acf, d_tau);
            hip_ok(hipStreamSynchronize(stream));
        }
        if (ok && acf) hip_ok(hipMemcpy(acf, d_acf, (size_t)nrep * k_max * sizeof(double), hipMemcpyDeviceToHost));
        if (ok && tau) hip_ok(hipMemcpy(tau, d_tau, (size_t)nrep * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (p1) hipfftDestroy(p1);
    if (p2) hipfftDestroy(p2);
    hipFree(Z); hipFree(F); hipFree(Tm); hipFree(C); hipFree(d_acf); hipFree(d_tau);
    if (!ok) return SMCX_ERR_HIP;
    if (cv) { // results.cv = variance(E)/T^2, SMC.c:250
        std::vector<double> dE(nrep);
        rc = smcx_observables(h, nullptr, nullptr, dE.data(), nullptr, nullptr, nullptr);
        if (rc != SMCX_OK) return rc;
        for (int r = 0; r < nrep; r++) cv[r] = dE[r] * dE[r] / (T * T);
    }
    return SMCX_OK;
}
