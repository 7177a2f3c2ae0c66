// gather24.hip -- what does FETCH_SIZE report for the sweep kernels' candidate fetches?  
//
// The z-ordered sweep kernels fetch candidates as 24-byte records (one fp64 position: global_load_dwordx4 + global_load_dwordx2
// at cell * 24) scattered over `Rs`.  MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced reads (16 B per lane: the
// counter reports HALF the bytes -- 128-byte requests tallied at 64) and says "other access widths are uncalibrated: calibrate on
// a known byte count in your own access pattern".  This is that calibration: kernels with a KNOWN number of record fetches at
// random record indices, from a table far larger than the 256 MiB Infinity Cache (3 GiB) and from one that fits it (96 MiB), beside
// the guide's own case (a streamed 16 B per lane read of the same 3 GiB) as the control.
//
// A 24-byte record at offset 24 i touches ONE 64-byte sector unless (24 i mod 64) > 40, i.e. for 2 of every 8 consecutive
// records: 1.25 sectors of 64 B = 80 B per record, or 1.125 lines of 128 B = 144 B per record.  So per record fetched
//     FETCH_SIZE = 80 B   <=> the counter tallies 64-byte requests of this pattern at their size   (factor x1, 80 B true)
//     FETCH_SIZE = 72 B   <=> 128-byte requests tallied at 64 as for the streamed read            (factor x2, 144 B true)
// and the time per record against the achievable HBM bandwidth says which is physically moved.
//
//   hipcc -O2 --offload-arch=gfx950 gather24.hip -o gather24
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- ./gather24        (tools/ubench/gather24_report.py OUT)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)   // splitmix64
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

// every lane fetches `per_lane` records of 24 B (dwordx4 + dwordx2, as the sweep kernels do) at random record indices
__global__ void __launch_bounds__(256) gather24(const double *tab, uint64_t nrec, int per_lane, double *sink)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < per_lane; k++) {
        const uint64_t rec = mix(tid * 1315423911ull + k) % nrec;
        const double *p = tab + 3 * rec;
        double2 a; double b;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(p) : "memory");   // early clobber: the address is read again by the second load
        acc += a.x + a.y + b;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// the same records, but the 64 lanes of a wavefront fetch 64 CONSECUTIVE records (a row fill of the row cache: coalesced 24-byte records)
__global__ void __launch_bounds__(256) rows24(const double *tab, uint64_t nrec, int per_lane, double *sink)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int k = 0; k < per_lane; k++) {
        const uint64_t rec = (mix(wave * 2654435761ull + k) % (nrec / 64)) * 64 + lane;
        const double *p = tab + 3 * rec;
        double2 a; double b;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx2 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(p) : "memory");   // early clobber: the address is read again by the second load
        acc += a.x + a.y + b;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

// the guide's calibrated case: a streamed read, 16 B per lane, every byte of the table once
__global__ void __launch_bounds__(256) stream16(const double2 *tab, uint64_t n16, double *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        double2 a;
        asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(a) : "v"(tab + i) : "memory");
        acc += a.x + a.y;
    }
    if (acc == 1.2345e300) sink[0] = acc;
}

static float timed(hipEvent_t e0, hipEvent_t e1)
{
    float ms = 0.f;
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

int main()
{
    const uint64_t big = 3ull << 30, small = 96ull << 20;           // bytes: 12 x and 0.375 x the Infinity Cache
    double *tab, *sink;
    CHECK(hipMalloc(&tab, big));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(tab, 0, big));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int blocks = 256 * 32, per_lane = 64;                      // 2^21 lanes x 64 records = 1.34e8 records = 3.2 GB of records
    const double nfetch = (double)blocks * 256 * per_lane;
    for (int rep = 0; rep < 2; rep++) {                              // the second round is the one to read (first touch, page tables)
        CHECK(hipEventRecord(e0));
        stream16<<<blocks, 256>>>((const double2 *)tab, big / 16, sink);
        CHECK(hipEventRecord(e1));
        float ms = timed(e0, e1);
        printf("stream16  table 3072 MiB: %.0f bytes read, %.3f ms, %.1f GB/s\n", (double)big, ms, big / ms / 1e6);
        struct { const char *name; uint64_t bytes; } T[2] = {{"3072 MiB", big}, {"96 MiB", small}};
        for (int t = 0; t < 2; t++) {
            const uint64_t nrec = T[t].bytes / 24;
            CHECK(hipEventRecord(e0));
            gather24<<<blocks, 256>>>(tab, nrec, per_lane, sink);
            CHECK(hipEventRecord(e1));
            ms = timed(e0, e1);
            printf("gather24  table %s: %.0f records of 24 B (%.0f record bytes; 1.25 sectors of 64 B = %.0f B; 1.125 lines of 128 B = %.0f B), "
                   "%.3f ms, %.2f G records/s\n", T[t].name, nfetch, nfetch * 24, nfetch * 80, nfetch * 144, ms, nfetch / ms / 1e6);
            CHECK(hipEventRecord(e0));
            rows24<<<blocks, 256>>>(tab, nrec, per_lane, sink);
            CHECK(hipEventRecord(e1));
            ms = timed(e0, e1);
            printf("rows24    table %s: %.0f records of 24 B in rows of 64 consecutive records (%.0f record bytes), %.3f ms, %.1f GB/s of records\n",
                   T[t].name, nfetch, nfetch * 24, ms, nfetch * 24 / ms / 1e6);
        }
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
